// What bounds the k-loop of the full-row GEMM + LayerNorm kernel (rowgemm.hip)?  One 512-thread workgroup per CU pulls
// 48 KB per k-step - a 16 KB activation slab that streams from HBM (private per workgroup) and a 32 KB weight slab that
// every workgroup fetches from the same 512 KB buffer (L2) - for 16 k-steps, and only lands it in LDS:
//   V0  LDS-DMA (global_load_lds_dwordx4), 3-stage ring, counted vmcnt            - what rowgemm_dma_kernel does
//   V1  registers -> ds_write_b128, prefetch distance 1 (6 x 16 B per thread), one LDS buffer, 2 barriers per step
//   V2  registers -> ds_write_b128, prefetch distance 2 (12 x 16 B per thread), two LDS buffers, 1 barrier per step
//   V0r LDS-DMA as V0, but both slabs addressed the way the kernel sees them: rows of row-major [M][1024] / [256][1024]
//       bf16 matrices, 128 bytes (one k-step) from each row per step, instead of contiguous 16 KB / 32 KB pieces
//   hipcc --offload-arch=gfx950 -O3 tools/probes/fill_probe.hip -o /tmp/fill && /tmp/fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define STEPS 16
#define SLAB 49152            // bytes per k-step
#define AB 16384              // activation part

__device__ __forceinline__ void dma16(const void* g, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_off) : "memory", "m0");
}
__device__ __forceinline__ unsigned lds_off(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p; }
__device__ __forceinline__ void bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// source address of 16-byte item `i` (0 .. 3071) of k-step `ks`
__device__ __forceinline__ const char* item(const char* a, const char* b, int ks, int i) {
    return i < AB / 16 ? a + (long)ks * AB + i * 16 : b + (long)ks * (SLAB - AB) + (i - AB / 16) * 16;
}

// the same item when both slabs are what the kernel reads: 128 activation rows / 256 weight rows of row-major matrices
// with 2 KB rows (K = 1024 bf16), 128 bytes of each row per k-step
__device__ __forceinline__ const char* item_rows(const char* a, const char* b, int ks, int i) {
    const int j = i < AB / 16 ? i : i - AB / 16;          // 8 items = the 128 bytes of one row in this k-step
    return (i < AB / 16 ? a : b) + (long)(j >> 3) * (STEPS * 128) + ks * 128 + (j & 7) * 16;
}

template <bool ROWS>
__global__ __launch_bounds__(512, 1) void v0(const char* A, const char* B, unsigned* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * SLAB];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* a = A + (long)blockIdx.x * STEPS * AB;
    auto issue = [&](int ks, int st) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {                      // 6 instructions x 1 KB per wave = 48 KB per workgroup
            const int q = wave * 6 + i;
            dma16(ROWS ? item_rows(a, B, ks, q * 64 + lane) : item(a, B, ks, q * 64 + lane), lds_off(smem + st * SLAB + q * 1024));
        }
    };
    issue(0, 0); issue(1, 1);
    unsigned acc = 0;
    for (int ks = 0; ks < STEPS; ++ks) {
        if (ks + 1 < STEPS) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bar();
        if (ks + 2 < STEPS) issue(ks + 2, (ks + 2) % 3);
        acc ^= *reinterpret_cast<const unsigned*>(smem + (ks % 3) * SLAB + ((tid * 52 + ks * 4) % SLAB & ~3));
    }
    if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}

template <int DIST>
__global__ __launch_bounds__(512, 1) void vreg(const char* A, const char* B, unsigned* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * SLAB];
    const int tid = threadIdx.x;
    const char* a = A + (long)blockIdx.x * STEPS * AB;
    uint4 r[DIST][6];
    auto fetch = [&](int ks, uint4 (&v)[6]) {
#pragma unroll
        for (int i = 0; i < 6; ++i) v[i] = *reinterpret_cast<const uint4*>(item(a, B, ks, i * 512 + tid));
    };
    auto stash = [&](int buf, const uint4 (&v)[6]) {
#pragma unroll
        for (int i = 0; i < 6; ++i) *reinterpret_cast<uint4*>(smem + buf * SLAB + (i * 512 + tid) * 16) = v[i];
    };
    unsigned acc = 0;
    if (DIST == 1) {
        fetch(0, r[0]);
        for (int ks = 0; ks < STEPS; ++ks) {
            stash(0, r[0]);
            __syncthreads();
            if (ks + 1 < STEPS) fetch(ks + 1, r[0]);
            acc ^= *reinterpret_cast<const unsigned*>(smem + ((tid * 52 + ks * 4) % SLAB & ~3));
            __syncthreads();
        }
    } else {
        fetch(0, r[0]); fetch(1, r[1 % DIST]);
#pragma unroll
        for (int ks = 0; ks < STEPS; ++ks) {               // fully unrolled: the register sets alternate statically
            stash(ks & 1, r[ks % DIST]);
            if (ks + 2 < STEPS) fetch(ks + 2, r[ks % DIST]);
            __syncthreads();                                // buffer ks & 1 published; buffer (ks + 1) & 1 free next step
            acc ^= *reinterpret_cast<const unsigned*>(smem + (ks & 1) * SLAB + ((tid * 52 + ks * 4) % SLAB & ~3));
        }
    }
    if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}

template <typename K>
static double run(K k, const char* A, const char* B, unsigned* sink, int grid) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, A, B, sink);
    hipEventRecord(e0);
    const int reps = 20;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, A, B, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3 / reps;
}

int main() {
    const int grid = 225;
    char *A, *B; unsigned* sink;
    hipMalloc(&A, (size_t)grid * STEPS * AB); hipMalloc(&B, (size_t)STEPS * (SLAB - AB)); hipMalloc(&sink, 4096);
    hipMemset(A, 1, (size_t)grid * STEPS * AB); hipMemset(B, 2, (size_t)STEPS * (SLAB - AB));
    const double t0 = run(v0<false>, A, B, sink, grid), t0r = run(v0<true>, A, B, sink, grid), t1 = run(vreg<1>, A, B, sink, grid), t2 = run(vreg<2>, A, B, sink, grid);
    const double kb = STEPS * SLAB / 1024.0;
    printf("225 workgroups x %d steps x 48 KB:\n", STEPS);
    printf("  V0 LDS-DMA 3-stage ring      %6.1f us  (%.2f us/step, %.1f GB/s per CU)\n", t0, t0 / STEPS, kb * 1024 / t0 / 1e3);
    printf("  V0r same, row-major A slab   %6.1f us  (%.2f us/step, %.1f GB/s per CU)\n", t0r, t0r / STEPS, kb * 1024 / t0r / 1e3);
    printf("  V1 registers, distance 1     %6.1f us  (%.2f us/step, %.1f GB/s per CU)\n", t1, t1 / STEPS, kb * 1024 / t1 / 1e3);
    printf("  V2 registers, distance 2     %6.1f us  (%.2f us/step, %.1f GB/s per CU)\n", t2, t2 / STEPS, kb * 1024 / t2 / 1e3);
    return 0;
}
