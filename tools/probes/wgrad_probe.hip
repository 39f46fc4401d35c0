// Where do the ~1.5 us per 32-row step of wgrad_dma_kernel (csrc/gemm_bf16.hip) go?  tools/probes/stream_tile_probe.hip shows
// that its data movement ALONE sustains 0.52 us per step (15 us per launch, 74 MB) at the same ring depth and occupancy.
// This probe carries a copy of the kernel's loop with stage switches, and candidate restructurings, against a naive
// reference:
//   V 0  the library kernel's loop (3-stage ring of 32 rows, 2 x 2 MFMA tiles per wave, 8 transposing reads per 4 MFMAs)
//   ablations of V 0:  1 = no MFMAs / no fragment reads (one plain read per image)   2 = no DMA (stale LDS)
//                      4 = no epilogue stores
//   V 1  64-row steps: half the barriers (2-stage... 3 x 32 KB ring), fragments of the next 16 rows read while the
//        MFMAs of the current 16 run (explicit software pipeline over the 4 sub-steps)
// Rows per split, tile order and swizzle as in the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/wgrad_probe.hip -o tools/probes/wgrad_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

struct P {
    const uint16_t* X; long ldx;
    const uint16_t* Y; long ldy;
    float* ws;              // [splits][K][N]
    int M, K, N, tiles_k, tiles_n, rows_per_split, ablate;
};

__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_off) : "memory", "m0");
}
__device__ __forceinline__ unsigned lds_offset(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    int xcd = bid % 8, slot = bid / 8, q = nwg / 8, r = nwg % 8;
    int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}
__device__ __forceinline__ bf16x8 tr8(const unsigned char* img, int row, int unit) {
    typedef __attribute__((address_space(3))) s16x4* lds_ptr;
    const int swz = (row & 3) << 3;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(img + row * 256 + ((unit ^ swz) << 3)));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(img + (row + 4) * 256 + ((unit ^ swz) << 3)));
    s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// ---- V 0: the library loop, with stage switches ------------------------------------------------------------------------
template <int ABL>
__global__ __launch_bounds__(256, 2) void wgrad_v0(P p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * 16384];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, kg = lane >> 5;
    const int ntiles = p.tiles_k * p.tiles_n;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int bid = lid % ntiles, z = lid / ntiles;
    const int tn = bid % p.tiles_n, tk = bid / p.tiles_n;
    const int k0 = tk * 128, n0 = tn * 128;
    const int mbeg = z * p.rows_per_split;
    const int mend = min(p.M, mbeg + p.rows_per_split);
    const int nsteps = (mend - mbeg) / 32;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int drow = lane >> 4, dpos = lane & 15;
    auto issue = [&](int step, int stage) {
        if (ABL & 2) return;
        unsigned char* Xi = smem + stage * 16384;
        unsigned char* Yi = Xi + 8192;
        const long m0 = mbeg + (long)step * 32;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wave * 8 + i * 4 + drow;
            const int c = dpos ^ ((row & 3) << 2);
            lds_dma16(p.X + (m0 + row) * p.ldx + k0 + c * 8, lds_offset(Xi + (wave * 8 + i * 4) * 256));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wave * 8 + i * 4 + drow;
            const int c = dpos ^ ((row & 3) << 2);
            lds_dma16(p.Y + (m0 + row) * p.ldy + n0 + c * 8, lds_offset(Yi + (wave * 8 + i * 4) * 256));
        }
    };
    const int trow = (lane >> 5) * 8 + ((lane & 15) >> 2);
    const int tunit = ((lane >> 4) & 1) * 4 + (lane & 3);
    if (nsteps > 0) issue(0, 0);
    if (nsteps > 1) issue(1, 1);
    for (int s_ = 0; s_ < nsteps; ++s_) {
        if (s_ + 1 < nsteps) __builtin_amdgcn_s_waitcnt(0xF74);
        else __builtin_amdgcn_s_waitcnt(0xF70);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (s_ + 2 < nsteps) issue(s_ + 2, (s_ + 2) % 3);
        const unsigned char* Xi = smem + (s_ % 3) * 16384;
        const unsigned char* Yi = Xi + 8192;
        if (ABL & 1) {
            acc[0][0][0] += (float)(((const unsigned*)Xi)[tid] ^ ((const unsigned*)Yi)[tid]);
            continue;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int row = ks * 16 + trow;
            bf16x8 a0 = tr8(Xi, row, wr * 16 + tunit);
            bf16x8 a1 = tr8(Xi, row, wr * 16 + 8 + tunit);
            bf16x8 b0 = tr8(Yi, row, wc * 16 + tunit);
            bf16x8 b1 = tr8(Yi, row, wc * 16 + 8 + tunit);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    float* Cb = p.ws + (long)z * p.K * p.N;
    if ((ABL & 4) && acc[0][0][0] != 12345.f) return;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wc * 64 + j * 32 + l31;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = k0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                Cb[(long)row * p.N + col] = acc[i][j][r];
            }
    }
}

// ---- V 1: 64-row steps, explicit software pipeline of the fragment reads ----------------------------------------------------
// Stage = 64 rows x 256 B of X and of Y = 32 KB; ring of RING stages; 8 DMA instructions per wave per step.
template <int RING>
__global__ __launch_bounds__(256, 1) void wgrad_v1(P p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, kg = lane >> 5;
    const int ntiles = p.tiles_k * p.tiles_n;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int bid = lid % ntiles, z = lid / ntiles;
    const int tn = bid % p.tiles_n, tk = bid / p.tiles_n;
    const int k0 = tk * 128, n0 = tn * 128;
    const int mbeg = z * p.rows_per_split;
    const int mend = min(p.M, mbeg + p.rows_per_split);
    const int nsteps = (mend - mbeg) / 64;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int drow = lane >> 4, dpos = lane & 15;
    auto issue = [&](int step, int stage) {
        unsigned char* Xi = smem + stage * 32768;
        unsigned char* Yi = Xi + 16384;
        const long m0 = mbeg + (long)step * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wave * 16 + i * 4 + drow;
            const int c = dpos ^ ((row & 3) << 2);
            lds_dma16(p.X + (m0 + row) * p.ldx + k0 + c * 8, lds_offset(Xi + (wave * 16 + i * 4) * 256));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wave * 16 + i * 4 + drow;
            const int c = dpos ^ ((row & 3) << 2);
            lds_dma16(p.Y + (m0 + row) * p.ldy + n0 + c * 8, lds_offset(Yi + (wave * 16 + i * 4) * 256));
        }
    };
    const int trow = (lane >> 5) * 8 + ((lane & 15) >> 2);
    const int tunit = ((lane >> 4) & 1) * 4 + (lane & 3);
    for (int s = 0; s < RING - 1 && s < nsteps; ++s) issue(s, s);
    for (int s_ = 0; s_ < nsteps; ++s_) {
        const int ahead = min(RING - 2, nsteps - 1 - s_);
        if (ahead <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (s_ + RING - 1 < nsteps) issue(s_ + RING - 1, (s_ + RING - 1) % RING);
        const unsigned char* Xi = smem + (s_ % RING) * 32768;
        const unsigned char* Yi = Xi + 16384;
        bf16x8 a0 = tr8(Xi, trow, wr * 16 + tunit), a1 = tr8(Xi, trow, wr * 16 + 8 + tunit);
        bf16x8 b0 = tr8(Yi, trow, wc * 16 + tunit), b1 = tr8(Yi, trow, wc * 16 + 8 + tunit);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 na0, na1, nb0, nb1;
            if (ks < 3) {
                const int row = (ks + 1) * 16 + trow;
                na0 = tr8(Xi, row, wr * 16 + tunit); na1 = tr8(Xi, row, wr * 16 + 8 + tunit);
                nb0 = tr8(Yi, row, wc * 16 + tunit); nb1 = tr8(Yi, row, wc * 16 + 8 + tunit);
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
            if (ks < 3) { a0 = na0; a1 = na1; b0 = nb0; b1 = nb1; }
        }
    }
    float* Cb = p.ws + (long)z * p.K * p.N;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wc * 64 + j * 32 + l31;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = k0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                Cb[(long)row * p.N + col] = acc[i][j][r];
            }
    }
}


// ---- V 2: V 0's ring and steps, the MFMAs of a step's second half carried ACROSS the barrier ---------------------------------
// Per step: [wait, barrier, DMA of step s + 2] read(s, rows 0-15) | MFMA(fragments of step s - 1, rows 16-31, read before the
// barrier) | read(s, rows 16-31) | MFMA(s, rows 0-15).  Every group of 4 MFMAs runs while the next 8 transposing reads are in
// flight; the barrier's lgkmcnt(0) only waits for reads whose MFMAs come after it.
template <int ABL>
__global__ __launch_bounds__(256, 2) void wgrad_v2(P p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * 16384];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, kg = lane >> 5;
    const int ntiles = p.tiles_k * p.tiles_n;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int bid = lid % ntiles, z = lid / ntiles;
    const int tn = bid % p.tiles_n, tk = bid / p.tiles_n;
    const int k0 = tk * 128, n0 = tn * 128;
    const int mbeg = z * p.rows_per_split;
    const int mend = min(p.M, mbeg + p.rows_per_split);
    const int nsteps = (mend - mbeg) / 32;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int drow = lane >> 4, dpos = lane & 15;
    auto issue = [&](int step, int stage) {
        if (ABL & 2) return;
        unsigned char* Xi = smem + stage * 16384;
        unsigned char* Yi = Xi + 8192;
        const long m0 = mbeg + (long)step * 32;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wave * 8 + i * 4 + drow;
            const int c = dpos ^ ((row & 3) << 2);
            lds_dma16(p.X + (m0 + row) * p.ldx + k0 + c * 8, lds_offset(Xi + (wave * 8 + i * 4) * 256));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wave * 8 + i * 4 + drow;
            const int c = dpos ^ ((row & 3) << 2);
            lds_dma16(p.Y + (m0 + row) * p.ldy + n0 + c * 8, lds_offset(Yi + (wave * 8 + i * 4) * 256));
        }
    };
    const int trow = (lane >> 5) * 8 + ((lane & 15) >> 2);
    const int tunit = ((lane >> 4) & 1) * 4 + (lane & 3);
    if (nsteps > 0) issue(0, 0);
    if (nsteps > 1) issue(1, 1);
    bf16x8 pa0, pa1, pb0, pb1;                       // the carried fragments (rows 16-31 of the previous step)
#pragma unroll
    for (int e = 0; e < 8; ++e) { pa0[e] = (__bf16)0.f; pa1[e] = (__bf16)0.f; pb0[e] = (__bf16)0.f; pb1[e] = (__bf16)0.f; }
    for (int s_ = 0; s_ < nsteps; ++s_) {
        if (s_ + 1 < nsteps) __builtin_amdgcn_s_waitcnt(0xF74);
        else __builtin_amdgcn_s_waitcnt(0xF70);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (s_ + 2 < nsteps) issue(s_ + 2, (s_ + 2) % 3);
        const unsigned char* Xi = smem + (s_ % 3) * 16384;
        const unsigned char* Yi = Xi + 8192;
        bf16x8 a0 = tr8(Xi, trow, wr * 16 + tunit), a1 = tr8(Xi, trow, wr * 16 + 8 + tunit);
        bf16x8 b0 = tr8(Yi, trow, wc * 16 + tunit), b1 = tr8(Yi, trow, wc * 16 + 8 + tunit);
        __builtin_amdgcn_sched_barrier(0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa0, pb0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa0, pb1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa1, pb0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa1, pb1, acc[1][1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        pa0 = tr8(Xi, 16 + trow, wr * 16 + tunit); pa1 = tr8(Xi, 16 + trow, wr * 16 + 8 + tunit);
        pb0 = tr8(Yi, 16 + trow, wc * 16 + tunit); pb1 = tr8(Yi, 16 + trow, wc * 16 + 8 + tunit);
        __builtin_amdgcn_sched_barrier(0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa0, pb0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa0, pb1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa1, pb0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa1, pb1, acc[1][1], 0, 0, 0);
    float* Cb = p.ws + (long)z * p.K * p.N;
    if ((ABL & 4) && acc[0][0][0] != 12345.f) return;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wc * 64 + j * 32 + l31;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = k0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                Cb[(long)row * p.N + col] = acc[i][j][r];
            }
    }
}

__global__ void fill(uint16_t* x, long n, unsigned seed) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const int v = (int)(h % 17u) - 8;                         // small integers: exact in bf16, exact sums in fp32
        const float f = (float)v;
        x[i] = (uint16_t)(__builtin_bit_cast(unsigned, f) >> 16);
    }
}
// sum of the split slabs of one element, and the naive answer, for a sample of elements
__global__ void check(P p, int splits, int* bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 4096) return;
    const int k = (i * 37) % p.K, n = (i * 101) % p.N;
    float got = 0.f;
    for (int z = 0; z < splits; ++z) got += p.ws[((long)z * p.K + k) * p.N + n];
    float want = 0.f;
    for (int m = 0; m < p.M; ++m) {
        const unsigned a = (unsigned)p.X[(long)m * p.ldx + k] << 16, b = (unsigned)p.Y[(long)m * p.ldy + n] << 16;
        want += __builtin_bit_cast(float, a) * __builtin_bit_cast(float, b);
    }
    if (got != want) atomicAdd(bad, 1);
}

template <typename F>
static float timeit(F launch, int reps) {
    hipEvent_t t0, t1;
    (void)hipEventCreate(&t0); (void)hipEventCreate(&t1);
    float best = 1e9f;
    for (int r = 0; r < reps; ++r) {
        (void)hipEventRecord(t0, 0);
        launch();
        (void)hipEventRecord(t1, 0);
        (void)hipDeviceSynchronize();
        float ms;
        (void)hipEventElapsedTime(&ms, t0, t1);
        if (ms < best) best = ms;
    }
    return best * 1e3f;
}

int main() {
    const int Ms[3] = {28800, 6400, 57600};
    const int shapes[2][2] = {{1024, 256}, {256, 256}};
    uint16_t *X, *Y;
    float* ws;
    int* bad;
    const long maxM = 57600;
    CK(hipMalloc(&X, (size_t)maxM * 1024 * 2)); CK(hipMalloc(&Y, (size_t)maxM * 1024 * 2));
    CK(hipMalloc(&ws, (size_t)64 * 1024 * 256 * 4)); CK(hipMalloc(&bad, 4));
    hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, X, maxM * 1024, 1u);
    hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, Y, maxM * 1024, 7u);
    CK(hipDeviceSynchronize());
    for (int mi = 0; mi < 3; ++mi)
        for (int sh = 0; sh < 2; ++sh) {
            const int M = Ms[mi], K = shapes[sh][0], N = shapes[sh][1];
            P p;
            p.X = X; p.ldx = K; p.Y = Y; p.ldy = N; p.ws = ws; p.M = M; p.K = K; p.N = N; p.tiles_k = K / 128; p.tiles_n = N / 128;
            p.ablate = 0;
            const int ntiles = p.tiles_k * p.tiles_n;
            for (int wgcu = 1; wgcu <= 2; ++wgcu) {
                int splits = wgcu * 256 / ntiles;
                int rps = ((M + splits - 1) / splits + 63) / 64 * 64;
                splits = (M + rps - 1) / rps;
                if (M % 64) continue;
                p.rows_per_split = rps;
                const int grid = ntiles * splits;
                printf("M %5d K %4d N %4d grid %3d (%d/CU) rows/split %4d |", M, K, N, grid, wgcu, rps);
                auto verify = [&]() -> int {
                    (void)hipMemset(bad, 0, 4);
                    hipLaunchKernelGGL(check, dim3(16), dim3(256), 0, 0, p, splits, bad);
                    int h = -1;
                    (void)hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
                    return h;
                };
                if (wgcu == 2) {
                    float t = timeit([&] { hipLaunchKernelGGL(wgrad_v0<0>, dim3(grid), dim3(256), 0, 0, p); }, 5);
                    const int b = verify();
                    printf(" v0 %5.1fus (bad %d)", t, b);
                    printf(" noMFMA %5.1f", timeit([&] { hipLaunchKernelGGL(wgrad_v0<1>, dim3(grid), dim3(256), 0, 0, p); }, 5));
                    printf(" noDMA %5.1f", timeit([&] { hipLaunchKernelGGL(wgrad_v0<2>, dim3(grid), dim3(256), 0, 0, p); }, 5));
                    printf(" noEPI %5.1f", timeit([&] { hipLaunchKernelGGL(wgrad_v0<4>, dim3(grid), dim3(256), 0, 0, p); }, 5));
                    printf(" noMFMA+noEPI %5.1f", timeit([&] { hipLaunchKernelGGL(wgrad_v0<5>, dim3(grid), dim3(256), 0, 0, p); }, 5));
                    printf(" noDMA+noEPI %5.1f", timeit([&] { hipLaunchKernelGGL(wgrad_v0<6>, dim3(grid), dim3(256), 0, 0, p); }, 5));
                    (void)hipMemset(ws, 0, (size_t)64 * 1024 * 256 * 4);
                    float t2 = timeit([&] { hipLaunchKernelGGL(wgrad_v2<0>, dim3(grid), dim3(256), 0, 0, p); }, 5);
                    const int b2 = verify();
                    printf(" | v2 %5.1fus (bad %d)", t2, b2);
                    printf(" noDMA %5.1f", timeit([&] { hipLaunchKernelGGL(wgrad_v2<2>, dim3(grid), dim3(256), 0, 0, p); }, 5));
                    printf(" noEPI %5.1f", timeit([&] { hipLaunchKernelGGL(wgrad_v2<4>, dim3(grid), dim3(256), 0, 0, p); }, 5));
                    printf(" noDMA+noEPI %5.1f", timeit([&] { hipLaunchKernelGGL(wgrad_v2<6>, dim3(grid), dim3(256), 0, 0, p); }, 5));
                }
                {
                    (void)hipFuncSetAttribute((const void*)wgrad_v1<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 32768);
                    (void)hipFuncSetAttribute((const void*)wgrad_v1<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 32768);
                    (void)hipFuncSetAttribute((const void*)wgrad_v1<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768);
                    (void)hipMemset(ws, 0, (size_t)64 * 1024 * 256 * 4);
                    float t2 = timeit([&] { hipLaunchKernelGGL(wgrad_v1<2>, dim3(grid), dim3(256), 2 * 32768, 0, p); }, 5);
                    const int b2 = verify();
                    float t3 = timeit([&] { hipLaunchKernelGGL(wgrad_v1<3>, dim3(grid), dim3(256), 3 * 32768, 0, p); }, 5);
                    const int b3 = verify();
                    float t4 = timeit([&] { hipLaunchKernelGGL(wgrad_v1<4>, dim3(grid), dim3(256), 4 * 32768, 0, p); }, 5);
                    printf(" | v1 ring2 %5.1fus (bad %d) ring3 %5.1fus (bad %d) ring4 %5.1fus", t2, b2, t3, b3, t4);
                }
                printf("\n");
            }
        }
    return 0;
}
