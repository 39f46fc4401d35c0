// Can ONE wave per SIMD overlap v_mfma_f32_32x32x16_bf16 with the ds_read_b128 that feeds it?  (the row-local chain kernel,
// csrc/chain.hip: 1 KB fragment read per MFMA and wave, 4 waves per CU, stages measured at 2 200 cycles for 1 024 cycles of
// multiplies whatever the order of reads and multiplies.)  One 256-thread workgroup per CU, 32 KB of LDS "stage", loops of
// 32-MFMA "stages":
//   M   multiplies only (fragments constant)                       R   reads only (32 x ds_read_b128 per stage and wave)
//   MR  multiply i consumes the fragment read 8 multiplies earlier, its register refilled right behind it (the chain's order)
//   MRx as MR, but the multiply does NOT consume the read fragments (constant operands; the reads only have to land)
//   MR2 as MR with HALF the reads (every second multiply reuses the previous fragment): is it LDS bandwidth?
//   MRD  MR + the chain's weight stream: 8 LDS-DMA pieces (1 KB) per wave and stage into a 4-slot ring, counted vmcnt wait and
//        one barrier per stage (in its middle)                     MRB  MR + the barrier only (no DMA)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_lds_overlap_probe.hip -o /tmp/mlo && /tmp/mlo
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define STAGES 64

__device__ __forceinline__ void dma16(const void* g, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_off) : "memory", "m0");
}
__device__ __forceinline__ unsigned lds_off(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p; }

template <int MODE>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void probe(float* sink, unsigned long long* cyc, const unsigned char* W) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[4 * 32768];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto issue = [&](int s, int g) {            // two of the wave's eight pieces of stage s
        const unsigned char* src = W + (size_t)(s % 52) * 32768 + (size_t)(wave * 8 + 2 * g) * 1024 + lane * 16;
        const unsigned dst = lds_off(smem) + (unsigned)(s & 3) * 32768 + (unsigned)(wave * 8 + 2 * g) * 1024;
        dma16(src, dst);
        dma16(src + 1024, dst + 1024);
    };
    for (int i = tid; i < 4 * 32768 / 4; i += 256) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 255);
    __syncthreads();
    f32x16 Z[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) Z[j][r] = 0.f;
    bf16x8 X[4], A[8];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) X[q][e] = (__bf16)(0.01f * (lane + q + e));
    const unsigned char* F = smem + lane * 16;
#pragma unroll
    for (int i = 0; i < 8; ++i) A[i] = *reinterpret_cast<const bf16x8*>(F + i * 1024);
    if (MODE == 5) {
        for (int s = 0; s < 3; ++s)
            for (int g = 0; g < 4; ++g) issue(s, g);
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < STAGES; ++s) {
        const unsigned char* Fs = smem + (s & 3) * 32768 + lane * 16;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) Z[i] = MFMA(A[i], X[g], Z[i]);                                        // M
                if (MODE == 1) A[i] = *reinterpret_cast<const bf16x8*>(Fs + (g * 8 + i) * 1024);     // R
                if (MODE == 2 || MODE >= 5) { Z[i] = MFMA(A[i], X[g], Z[i]); A[i] = *reinterpret_cast<const bf16x8*>(Fs + (g * 8 + i) * 1024); }   // MR
                if (MODE == 3) { Z[i] = MFMA(X[(g + 1) & 3], X[g], Z[i]); A[i] = *reinterpret_cast<const bf16x8*>(Fs + (g * 8 + i) * 1024); }   // MRx
                if (MODE == 4) { Z[i] = MFMA(A[i & 6], X[g], Z[i]); if ((i & 1) == 0) A[i] = *reinterpret_cast<const bf16x8*>(Fs + (g * 8 + i) * 1024); }   // MR2
            }
            if (MODE >= 2) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (MODE != 4 || (i & 1) == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (MODE >= 5 && g == 1) {
                if (MODE == 5) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            if (MODE == 5 && g >= 2) { issue(s + 3, 2 * (g - 2)); issue(s + 3, 2 * (g - 2) + 1); }
        }
        if (MODE == 1 || MODE == 3) {                      // keep the fragments alive without multiplying them
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(A[i]));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc += Z[j][r];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += (float)A[i][0];
    if (acc == 12345.678f) sink[blockIdx.x] = acc;
    if (MODE == 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

template <int MODE>
static void run(const char* name, float* sink, unsigned long long* cyc, const unsigned char* W) {
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(256), 0, 0, sink, cyc, W);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(256), 0, 0, sink, cyc, W);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[1024];
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < 1024; ++i) mean += (double)h[i];
    mean /= 1024;
    printf("%-4s %8.0f cycles per 32-MFMA stage (wave mean over 256 CUs x 4 waves)   launch %.1f us\n", name, mean / STAGES, 1e3 * ms / 10);
}

int main() {
    float* sink; unsigned long long* cyc;
    unsigned char* W;
    hipMalloc(&sink, 4096); hipMalloc(&cyc, 1024 * 8); hipMalloc(&W, 52 * 32768); hipMemset(W, 0x11, 52 * 32768);
    run<0>("M", sink, cyc, W);
    run<1>("R", sink, cyc, W);
    run<2>("MR", sink, cyc, W);
    run<3>("MRx", sink, cyc, W);
    run<4>("MR2", sink, cyc, W);
    run<6>("MRB", sink, cyc, W);
    run<5>("MRD", sink, cyc, W);
    return 0;
}
