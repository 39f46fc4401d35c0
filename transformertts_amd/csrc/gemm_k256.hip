// Weight-stationary bf16 GEMM for the K = 256 projections of the dense block (TTSMI_BF16 path):
//     C[M, N] = act(A[M, 256] . Bt[N, 256]^T + bias)          qkv projection (N = 768), FFN1 (N = 1024, ReLU), d(ctx)
//
// With K = 256 a 128 x 128 output tile loads 128 KB of operands for 32-64 KB of output and runs 4 k-steps: the
// general kernels (gemm_bf16.hip) spend these launches in per-tile prologues / epilogues and in L2 -> LDS traffic
// (236 MB for the FFN1 shape, 32-42 us per launch at M = 28 800).  Here a workgroup owns a 128-column chunk of W^T for
// its whole life - the chunk's MFMA fragments live in REGISTERS (each of the 8 waves keeps its 32 columns x 256 k =
// 64 VGPRs) - and walks down the rows of A: 64-row tiles (32 KB) arrive by `global_load_lds_dwordx4` into a 3-deep LDS
// ring, two tiles in flight while one is multiplied (16 MFMAs per wave), so A is the only operand that moves.
// L2 -> LDS traffic: (N / 128) x |A| (118 MB for FFN1).  Measured at M = 28 800: qkv 32 -> 25 us, FFN1 41 -> 29 us,
// d(ctx) 15.1 -> 13.5 us; the rate is that of the LDS-DMA fill path (~20 GB/s per CU here; MI355X_MICROARCH.md lists
// ~25 GB/s per CU), not of L2.  A variant that read the activation fragments straight from global memory into
// registers (lane = row, 16 bytes per k-step, no LDS) was 2x SLOWER (57 / 68 us): 32 rows x 32 B per wave instruction
// with a 4x re-read across the column waves is bound by the address path, not by bytes.
//
// Grid: (N / 128) column chunks x G row groups (G a multiple of 8); the chunks of one row group run on the same XCD
// (blocks b, b + 8, ... share an XCD), so a row tile is fetched from HBM once and served to the other chunks from that
// XCD's L2.  Accumulators are transposed (C^T = W . A^T: lane = row, 4 consecutive columns per register quad) so the
// output tile is staged through LDS with vector writes and leaves as full-row 16-byte stores.
#include <stdlib.h>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define KW_K 256
#define KW_BM 64
#define KW_BN 128
#define KW_STAGE (KW_BM * KW_K * 2)          // 32 768 bytes
#define KW_STAGES 3
#define KW_DMA_PER_WAVE 8                    // wave instructions (1 KB each) per tile and LOADER wave (waves 0-3)

struct K256P {
    const uint16_t* A; long lda;
    const uint16_t* Bt; long ldb;
    void* C; long ldc;
    const float* bias;
    int relu;
    int M, N;
    int nchunks, ngroups, ntiles;
};

__device__ __forceinline__ void kw_dma16(const void* gsrc, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_off) : "memory", "m0");
}
__device__ __forceinline__ void kw_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ unsigned kw_lds_offset(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ uint2 kw_pack4(float a, float b, float c, float d) {
    bf16x4 h;
    h[0] = (__bf16)a; h[1] = (__bf16)b; h[2] = (__bf16)c; h[3] = (__bf16)d;
    return *reinterpret_cast<uint2*>(&h);
}

template <bool OUT_H>
__global__ __launch_bounds__(512, 1) void gemm_k256_kernel(K256P p) {
    constexpr int SLD = OUT_H ? (KW_BN + 8) : (KW_BN + 4);                 // staging row stride (elements)
    constexpr int STAGING = KW_BM * SLD * (OUT_H ? 2 : 4);
    __shared__ __attribute__((aligned(16))) unsigned char smem[KW_STAGES * KW_STAGE + STAGING];
    unsigned char* stg = smem + KW_STAGES * KW_STAGE;

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 3, wmh = wave >> 2;
    // blocks b, b + 8, ... share an XCD: give every XCD whole row groups (all column chunks of a group together)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int group = xcd + 8 * (slot / p.nchunks), chunk = slot % p.nchunks;
    const int n0 = chunk * KW_BN + wn * 32;
    const int my_tiles = group < p.ntiles ? (p.ntiles - group + p.ngroups - 1) / p.ngroups : 0;

    // ---- this wave's 32 columns of W^T as MFMA A-operand fragments: bfrag[s] = Bt[n0 + l31][16 s + 8 hh .. + 8]
    bf16x8 bfrag[KW_K / 16];
    {
        const int n = n0 + l31;
        const uint16_t* src = p.Bt + (long)min(n, p.N - 1) * p.ldb + 8 * hh;
#pragma unroll
        for (int s = 0; s < KW_K / 16; ++s) {
            uint4 v = *reinterpret_cast<const uint4*>(src + 16 * s);
            if (n >= p.N) v = make_uint4(0, 0, 0, 0);
            bfrag[s] = *reinterpret_cast<bf16x8*>(&v);
        }
    }
    // bias of this lane's 16 columns n0 + (r & 3) + 8 (r >> 2) + 4 hh
    float bias[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = n0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        bias[r] = (p.bias != nullptr && n < p.N) ? p.bias[n] : 0.f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // Roles (vmcnt is per wave, and it counts stores as well as loads): waves 0-3 issue every tile DMA and wait for them
    // with exact counts; waves 4-7 issue every output store and never wait on vmcnt inside the loop.
    const bool loader = wave < 4;

    // ---- A tiles by LDS-DMA.  A wave instruction moves 2 rows (2 x 512 B): lane -> row 2 q + (lane >> 5), 16-byte
    // position lane & 31; position pos of row r holds source chunk pos ^ (r & 15) (conflict-free ds_read_b128 below).
    auto issue = [&](int it, int stage) {
        const int m0 = (group + it * p.ngroups) * KW_BM;
        unsigned char* base = smem + stage * KW_STAGE;
#pragma unroll
        for (int i = 0; i < KW_DMA_PER_WAVE; ++i) {
            const int q = wave * KW_DMA_PER_WAVE + i;
            const int row = 2 * q + hh;
            const int c = l31 ^ (row & 15);
            const int gm = min(m0 + row, p.M - 1);
            kw_dma16(p.A + (long)gm * p.lda + c * 8, kw_lds_offset(base + q * 1024));
        }
    };
    if (loader) {
        if (my_tiles > 0) issue(0, 0);
        if (my_tiles > 1) issue(1, 1);
    }

    const int arow = wmh * 32 + l31;
    for (int it = 0; it < my_tiles; ++it) {
        // tile `it` has landed once at most the 8 DMAs of the tile issued after it are outstanding (loader waves)
        if (loader) {
            if (it + 1 < my_tiles) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        kw_barrier();                    // everybody's pieces landed; stage (it-1)%3 and the staging tile are retired
        if (loader && it + 2 < my_tiles) issue(it + 2, (it + 2) % KW_STAGES);
        const unsigned char* As = smem + (it % KW_STAGES) * KW_STAGE;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = bias[r];
#pragma unroll
        for (int s = 0; s < KW_K / 16; ++s) {
            const int pos = (2 * s + hh) ^ (arow & 15);
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(As + arow * 512 + pos * 16);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfrag[s], a, acc, 0, 0, 0);          // C^T: rows = n, cols = m
        }
        if (p.relu) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = fmaxf(acc[r], 0.f);
        }
        // ---- C^T -> row-major staging tile [64][128]
        if constexpr (OUT_H) {
            uint16_t* st = reinterpret_cast<uint16_t*>(stg) + arow * SLD + wn * 32 + 4 * hh;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<uint2*>(st + 8 * g) = kw_pack4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
        } else {
            float* st = reinterpret_cast<float*>(stg) + arow * SLD + wn * 32 + 4 * hh;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(st + 8 * g) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
        }
        kw_barrier();
        // ---- full-row stores (the storing half of the workgroup)
        if (loader) continue;
        const int st_tid = tid - 256;
        const int m0 = (group + it * p.ngroups) * KW_BM;
        const int ncol0 = chunk * KW_BN;
        if constexpr (OUT_H) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = (st_tid >> 4) + 16 * j, c8 = (st_tid & 15) * 8;
                if (m0 + row < p.M && ncol0 + c8 < p.N)
                    *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(p.C) + (long)(m0 + row) * p.ldc + ncol0 + c8) =
                        *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(stg) + row * SLD + c8);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int row = (st_tid >> 5) + 8 * j, c4 = (st_tid & 31) * 4;
                if (m0 + row < p.M && ncol0 + c4 < p.N)
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + (long)(m0 + row) * p.ldc + ncol0 + c4) =
                        *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(stg) + row * SLD + c4);
            }
        }
    }
}

extern "C" {

// 1 if ttsmi_hgemm_tn would route this launch to the weight-stationary kernel (exposed for tests / benchmarks)
int ttsmi_hgemm_k256_eligible(int M, int N, int K) {
    static int mode = -1;      // TTSMI_HGEMM_K256: 0 = never, 1 = every eligible launch, default = by size
    if (mode < 0) { const char* e = getenv("TTSMI_HGEMM_K256"); mode = e ? atoi(e) : 2; }
    if (mode == 0 || K != KW_K || N % 8 != 0) return 0;
    if (mode == 1) return 1;
    return M >= 4096 && N >= 256;
}

}  // extern "C"

int ttsmi_hgemm_k256_launch(const uint16_t* a, long lda, const uint16_t* bt, long ldb, const float* bias, void* c, long ldc,
                            int M, int N, int relu, int out_bf16, hipStream_t st) {
    K256P p;
    memset(&p, 0, sizeof(p));
    p.A = a; p.lda = lda; p.Bt = bt; p.ldb = ldb; p.C = c; p.ldc = ldc; p.bias = bias; p.relu = relu; p.M = M; p.N = N;
    p.nchunks = ttsmi_cdiv(N, KW_BN);
    p.ntiles = ttsmi_cdiv(M, KW_BM);
    // one workgroup per CU: groups = a multiple of 8 with chunks x groups <= 256 (at least 8)
    int groups = (256 / p.nchunks) / 8 * 8;
    if (groups < 8) groups = 8;
    const int need = (p.ntiles + 7) / 8 * 8;
    if (groups > need) groups = need;
    p.ngroups = groups;
    dim3 grid(p.nchunks * p.ngroups);
    if (out_bf16) hipLaunchKernelGGL((gemm_k256_kernel<true>), grid, dim3(512), 0, st, p);
    else hipLaunchKernelGGL((gemm_k256_kernel<false>), grid, dim3(512), 0, st, p);
    return 0;
}
