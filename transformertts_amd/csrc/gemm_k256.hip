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
    const uint16_t* mask; long ldmask;      // MODE 2: bf16 [M, N], output kept where mask > 0 (ReLU' of the saved activation)
    // (output > 0) as one bit per element, in the order the STORING threads of the kernel hold the tile: block (row tile,
    // column chunk) = 256 threads x one word (4 groups of 8 columns; wide kernel: two words, 8 groups) - written by MODE 1
    // (optional) and read back by MODE 4 of the same kernel variant with one coalesced load per thread and tile
    uint32_t* bits_out;
    const uint32_t* bits_in;                // MODE 4: the mask (59 MB of bf16 activation -> 3.7 MB at M = 28 800, N = 1 024)
    int ablate;                             // measurement only (TTSMI_K256_ABLATE, wide kernel): 1 no stores, 2 no MFMA, 4 no DMA
    void* C2; long ldc2; int n_acc;         // MODE 3: columns < n_acc (a multiple of 128) are ADDED to the fp32 C, the
                                            // rest leave as bf16 into C2 (column n -> C2[., n - n_acc])
};

__device__ __forceinline__ void kw_dma16(const void* gsrc, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_off) : "memory", "m0");
}
__device__ __forceinline__ void kw_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ unsigned kw_lds_offset(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
// bit e of the result = bf16 element e of the 8-element group is > 0 (a half moved to the top of an int32 is > 0)
__device__ __forceinline__ uint32_t kw_pos_bits(const uint4& v) {
    auto two = [](uint32_t w) { return (((int32_t)(w << 16) > 0) ? 1u : 0u) | (((int32_t)(w & 0xFFFF0000u) > 0) ? 2u : 0u); };
    return two(v.x) | (two(v.y) << 2) | (two(v.z) << 4) | (two(v.w) << 6);
}
__device__ __forceinline__ void kw_and_bits(uint4& v, uint32_t b) {
    auto sel = [](uint32_t b2) { return ((b2 & 1u) ? 0x0000FFFFu : 0u) | ((b2 & 2u) ? 0xFFFF0000u : 0u); };
    v.x &= sel(b); v.y &= sel(b >> 2); v.z &= sel(b >> 4); v.w &= sel(b >> 6);
}
__device__ __forceinline__ uint2 kw_pack4(float a, float b, float c, float d) {
    bf16x4 h;
    h[0] = (__bf16)a; h[1] = (__bf16)b; h[2] = (__bf16)c; h[3] = (__bf16)d;
    return *reinterpret_cast<uint2*>(&h);
}

// MODE 0: fp32 C; 1: bf16 C; 2: bf16 C with a bf16 ReLU' mask (the masked FFN1 dgrad); 3: fp32 accumulate for the first
// n_acc columns, bf16 C2 for the rest (one launch for the two halves of the output-projection dgrad: d(h) += and d(ctx))
template <int MODE>
__global__ __launch_bounds__(512, 1) void gemm_k256_kernel(K256P p) {
    constexpr bool OUT_H = MODE == 1 || MODE == 2 || MODE == 4;
    constexpr int SLD_H = KW_BN + 8, SLD_F = KW_BN + 4;                    // staging row strides (elements)
    constexpr int SLD = OUT_H ? SLD_H : SLD_F;
    constexpr int STAGING = KW_BM * (OUT_H ? SLD_H * 2 : SLD_F * 4);        // MODE 3 stages either type in the fp32-sized tile
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
    // the first two tiles are requested BEFORE the weight fragments: one memory round trip at the head of the kernel instead
    // of two (the fragments, then the tiles) - these launches are 12-15 us at the encoder's 6 400 rows, mostly such latencies
    if (loader) {
        if (my_tiles > 0) issue(0, 0);
        if (my_tiles > 1) issue(1, 1);
    }

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
    const int arow = wmh * 32 + l31;
    for (int it = 0; it < my_tiles; ++it) {
        // tile `it` has landed once at most the 8 DMAs of the tile issued after it are outstanding (loader waves)
        if (loader) {
            if (it + 1 < my_tiles) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        kw_barrier();                    // everybody's pieces landed; stage (it-1)%3 and the staging tile are retired
        if (loader && it + 2 < my_tiles) issue(it + 2, (it + 2) % KW_STAGES);
        // epilogue operands from global memory (the mask / the accumulate target) are requested by the storing waves NOW,
        // a whole multiply ahead of their use; everything older on their vmcnt is the previous tile's stores
        uint4 pre_m[4];
        uint32_t pre_b = 0u, out_b = 0u;
        float4 pre_c[8];
        const bool acc_chunk = MODE == 3 && chunk * KW_BN < p.n_acc;       // workgroup-uniform
        if (MODE >= 2 && !loader) {
            const int st_tid = tid - 256;
            const int m0 = (group + it * p.ngroups) * KW_BM, ncol0 = chunk * KW_BN;
            if (MODE == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = (st_tid >> 4) + 16 * j, c8 = (st_tid & 15) * 8;
                    pre_m[j] = (m0 + row < p.M && ncol0 + c8 < p.N)
                                   ? *reinterpret_cast<const uint4*>(p.mask + (long)(m0 + row) * p.ldmask + ncol0 + c8)
                                   : make_uint4(0u, 0u, 0u, 0u);
                }
            } else if (MODE == 4) {
                pre_b = p.bits_in[((long)(group + it * p.ngroups) * p.nchunks + chunk) * 256 + st_tid];
            } else if (acc_chunk) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int row = (st_tid >> 5) + 8 * j, c4 = (st_tid & 31) * 4;
                    pre_c[j] = (m0 + row < p.M && ncol0 + c4 < p.N)
                                   ? *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.C) + (long)(m0 + row) * p.ldc + ncol0 + c4)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
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
        if (OUT_H || (MODE == 3 && !acc_chunk)) {
            uint16_t* st = reinterpret_cast<uint16_t*>(stg) + arow * SLD_H + wn * 32 + 4 * hh;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<uint2*>(st + 8 * g) = kw_pack4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
        } else {
            float* st = reinterpret_cast<float*>(stg) + arow * SLD_F + wn * 32 + 4 * hh;
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
        if (MODE >= 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the prefetched epilogue operands
        if (OUT_H || (MODE == 3 && !acc_chunk)) {
            uint16_t* dstb = MODE == 3 ? reinterpret_cast<uint16_t*>(p.C2) : reinterpret_cast<uint16_t*>(p.C);
            const long ldd = MODE == 3 ? p.ldc2 : p.ldc;
            const int cd0 = MODE == 3 ? ncol0 - p.n_acc : ncol0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = (st_tid >> 4) + 16 * j, c8 = (st_tid & 15) * 8;
                if (m0 + row < p.M && ncol0 + c8 < p.N) {
                    uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(stg) + row * SLD_H + c8);
                    if (MODE == 2) {
                        // bf16 halves of a word: > 0  <=>  the half, moved to the top of an int32, is > 0
                        auto sel = [](uint32_t m) {
                            return (((int32_t)(m << 16) > 0) ? 0x0000FFFFu : 0u) | (((int32_t)(m & 0xFFFF0000u) > 0) ? 0xFFFF0000u : 0u);
                        };
                        v.x &= sel(pre_m[j].x); v.y &= sel(pre_m[j].y); v.z &= sel(pre_m[j].z); v.w &= sel(pre_m[j].w);
                    }
                    if (MODE == 4) kw_and_bits(v, pre_b >> (8 * j));
                    if (MODE == 1) out_b |= kw_pos_bits(v) << (8 * j);
                    *reinterpret_cast<uint4*>(dstb + (long)(m0 + row) * ldd + cd0 + c8) = v;
                }
            }
            if (MODE == 1 && p.bits_out != nullptr)
                p.bits_out[((long)(group + it * p.ngroups) * p.nchunks + chunk) * 256 + st_tid] = out_b;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int row = (st_tid >> 5) + 8 * j, c4 = (st_tid & 31) * 4;
                if (m0 + row < p.M && ncol0 + c4 < p.N) {
                    float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(stg) + row * SLD_F + c4);
                    if (MODE == 3) { v.x += pre_c[j].x; v.y += pre_c[j].y; v.z += pre_c[j].z; v.w += pre_c[j].w; }
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + (long)(m0 + row) * p.ldc + ncol0 + c4) = v;
                }
            }
        }
    }
}

// ---- 256-column variant with separated roles (bf16 output, MODE 1 / 2) ----------------------------------------------
// Stage ablation of the kernel above on the FFN1 shape (TTSMI_K256_ABLATE, 28.7 us): without the stores 22.7, without
// the multiply 19.3, without the DMA 27.2, with none of them 11.5 - the phases ADD UP.  Every wave multiplies, then every
// workgroup of the launch reaches its store phase at the same moment: 59 MB of stores are issued in bursts that cover a
// third of the run (HBM alone writes them in 10 us).  Here the roles are separated:
//   waves 0-3 ("compute"): issue the tile DMAs, own 64 columns of W^T each (128 registers of MFMA fragments: a workgroup
//              covers 256 columns, so the activation tile passes the LDS-DMA path N / 256 times instead of N / 128),
//              multiply the whole 64-row tile and leave it in the bf16 staging tile;
//   waves 4-7 ("store"):   copy the staging tile of step t - 1 into registers right after it is published, and issue
//              its global stores (under the prefetched ReLU' mask in MODE 2) WHILE the compute waves multiply tile t.
// Two barriers per tile as before, but the store stream now runs underneath the multiply, and the LDS reads of the
// activation fragments are halved (4 reading waves instead of 8).
#define KWW_BN 256
template <int MODE>
__global__ __launch_bounds__(512, 1) void gemm_k256_wide_kernel(K256P p) {
    static_assert(MODE == 1 || MODE == 2 || MODE == 4, "bf16 output only");
    constexpr int SLD = KWW_BN + 8;                                        // staging row stride (bf16 elements)
    __shared__ __attribute__((aligned(16))) unsigned char smem[KW_STAGES * KW_STAGE + KW_BM * SLD * 2];
    __shared__ float biasS[KWW_BN];
    uint16_t* stg = reinterpret_cast<uint16_t*>(smem + KW_STAGES * KW_STAGE);

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool compute = wave < 4;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int group = xcd + 8 * (slot / p.nchunks), chunk = slot % p.nchunks;
    const int ncol0 = chunk * KWW_BN;
    const int my_tiles = group < p.ntiles ? (p.ntiles - group + p.ngroups - 1) / p.ngroups : 0;
    const int ab = TTSMI_ABLATE_BITS(p.ablate);

    auto issue = [&](int it, int stage) {                 // compute waves only: 8 DMA instructions per wave and tile
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
    const bool dma_on = !(ab & 4);
    // (requested before the weight fragments: one round trip at the head of the kernel instead of two)
    if (compute && dma_on) {
        if (my_tiles > 0) issue(0, 0);
        if (my_tiles > 1) issue(1, 1);
    }
    if (tid < KWW_BN) biasS[tid] = (p.bias != nullptr && ncol0 + tid < p.N) ? p.bias[ncol0 + tid] : 0.f;
    // ---- compute waves: 64 columns of W^T as MFMA A-operand fragments, bfrag[cb][s] = Bt[n0 + 32 cb + l31][16 s + 8 hh ..]
    bf16x8 bfrag[2][KW_K / 16];
    if (compute) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int n = ncol0 + wave * 64 + cb * 32 + l31;
            const uint16_t* src = p.Bt + (long)min(n, p.N - 1) * p.ldb + 8 * hh;
#pragma unroll
            for (int s = 0; s < KW_K / 16; ++s) {
                uint4 v = *reinterpret_cast<const uint4*>(src + 16 * s);
                if (n >= p.N) v = make_uint4(0, 0, 0, 0);
                bfrag[cb][s] = *reinterpret_cast<bf16x8*>(&v);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                      // bias table
    // Two loops with the same barrier sequence (T, Y per iteration; iterations 0 .. my_tiles) - one per role, so that the
    // register allocator sees two disjoint live sets (128 weight + 64 accumulator registers on one side, 96 of staging /
    // mask data on the other) instead of their union.  In iteration `it` the compute waves work on tile `it`
    // (it < my_tiles) while the store waves write out tile it - 1 (it >= 1).
    if (compute) {
        for (int it = 0; it <= my_tiles; ++it) {
            if (it < my_tiles) {
                if (it + 1 < my_tiles) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            kw_barrier();                // T: tile `it` landed, ring stage (it-1)%3 retired; staging(it-1) is in registers
            if (it < my_tiles) {
                if (dma_on && it + 2 < my_tiles) issue(it + 2, (it + 2) % KW_STAGES);
                const unsigned char* As = smem + (it % KW_STAGES) * KW_STAGE;
                f32x16 acc[2][2];
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const float* bsrc = biasS + wave * 64 + cb * 32 + 4 * hh;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 b4 = *reinterpret_cast<const float4*>(bsrc + 8 * g);
#pragma unroll
                        for (int rb = 0; rb < 2; ++rb) {
                            acc[cb][rb][4 * g + 0] = b4.x; acc[cb][rb][4 * g + 1] = b4.y;
                            acc[cb][rb][4 * g + 2] = b4.z; acc[cb][rb][4 * g + 3] = b4.w;
                        }
                    }
                }
                if (!(ab & 2)) {
#pragma unroll
                    for (int s = 0; s < KW_K / 16; ++s) {
                        bf16x8 a[2];
#pragma unroll
                        for (int rb = 0; rb < 2; ++rb) {
                            const int arow = rb * 32 + l31;
                            const int pos = (2 * s + hh) ^ (arow & 15);
                            a[rb] = *reinterpret_cast<const bf16x8*>(As + arow * 512 + pos * 16);
                        }
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                            for (int rb = 0; rb < 2; ++rb)
                                acc[cb][rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfrag[cb][s], a[rb], acc[cb][rb], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb) {
                        if (p.relu) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[cb][rb][r] = fmaxf(acc[cb][rb][r], 0.f);
                        }
                        uint16_t* st = stg + (rb * 32 + l31) * SLD + wave * 64 + cb * 32 + 4 * hh;
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            *reinterpret_cast<uint2*>(st + 8 * g) = kw_pack4(acc[cb][rb][4 * g], acc[cb][rb][4 * g + 1],
                                                                             acc[cb][rb][4 * g + 2], acc[cb][rb][4 * g + 3]);
                    }
            }
            kw_barrier();                // Y: staging(it) published
        }
        return;
    }
    // ---- store waves: thread -> rows srow + 8 j (j < 8), 8 columns at sc8
    const int st_tid = tid - 256;
    const int srow = st_tid >> 5, sc8 = (st_tid & 31) * 8;
    uint4 outv[8], pre_m[8];
    uint2 pre_b = make_uint2(0u, 0u);
#pragma unroll
    for (int j = 0; j < 8; ++j) { outv[j] = make_uint4(0u, 0u, 0u, 0u); pre_m[j] = make_uint4(0u, 0u, 0u, 0u); }
    auto bits_of = [&](int it) -> uint2 {          // the thread's 8 x 8 bits of tile `it` (clamped: the last request is one tile past the end)
        const long tile = min(group + it * p.ngroups, p.ntiles - 1);
        return *reinterpret_cast<const uint2*>(p.bits_in + ((tile * p.nchunks + chunk) * 256 + st_tid) * 2);
    };
    auto mask_of = [&](int it, int j) {
        const int row = min((group + it * p.ngroups) * KW_BM + srow + 8 * j, p.M - 1), col = min(ncol0 + sc8, p.N - 8);
        return *reinterpret_cast<const uint4*>(p.mask + (long)row * p.ldmask + col);
    };
    for (int it = 0; it <= my_tiles; ++it) {
        kw_barrier();                    // T
        if (it >= 1 && !(ab & 1)) {
            // tile it - 1 (already in outv) leaves while the compute waves multiply tile it
            const int m0 = (group + (it - 1) * p.ngroups) * KW_BM;
            if (MODE == 2) {
                // request the NEXT tile's mask first, then wait for everything older than those 8 loads: this tile's mask
                // (requested an iteration ago) and the stores of the tile before
                uint4 nxt[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) nxt[j] = mask_of(it, j);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                auto sel = [](uint32_t m) {
                    return (((int32_t)(m << 16) > 0) ? 0x0000FFFFu : 0u) | (((int32_t)(m & 0xFFFF0000u) > 0) ? 0xFFFF0000u : 0u);
                };
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    outv[j].x &= sel(pre_m[j].x); outv[j].y &= sel(pre_m[j].y);
                    outv[j].z &= sel(pre_m[j].z); outv[j].w &= sel(pre_m[j].w);
                    pre_m[j] = nxt[j];
                }
            }
            if (MODE == 4) {                 // the same schedule on the bit form of the mask: one 8-byte load per thread and tile
                const uint2 nxt = bits_of(it);
                asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
#pragma unroll
                for (int j = 0; j < 8; ++j) kw_and_bits(outv[j], (j < 4 ? pre_b.x : pre_b.y) >> (8 * (j & 3)));
                pre_b = nxt;
            }
            if (MODE == 1 && p.bits_out != nullptr) {
                uint2 ob = make_uint2(0u, 0u);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t b = kw_pos_bits(outv[j]) << (8 * (j & 3));
                    if (j < 4) ob.x |= b; else ob.y |= b;
                }
                *reinterpret_cast<uint2*>(p.bits_out + (((long)(m0 / KW_BM) * p.nchunks + chunk) * 256 + st_tid) * 2) = ob;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int row = srow + 8 * j;
                if (m0 + row < p.M && ncol0 + sc8 < p.N)
                    *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(p.C) + (long)(m0 + row) * p.ldc + ncol0 + sc8) = outv[j];
            }
        } else if (MODE == 2 && it == 0 && my_tiles > 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) pre_m[j] = mask_of(0, j);          // the first tile's mask
        } else if (MODE == 4 && it == 0 && my_tiles > 0) {
            pre_b = bits_of(0);
        }
        kw_barrier();                    // Y: staging(it) published
        if (it < my_tiles) {
#pragma unroll
            for (int j = 0; j < 8; ++j) outv[j] = *reinterpret_cast<const uint4*>(stg + (srow + 8 * j) * SLD + sc8);
        }
    }
}

extern "C" {

// 1 if ttsmi_hgemm_tn would route this launch to the weight-stationary kernel (exposed for tests / benchmarks)
int ttsmi_hgemm_k256_eligible(int M, int N, int K) {
    TTSMI_KNOB(mode, "TTSMI_HGEMM_K256", 2);      // TTSMI_HGEMM_K256: 0 = never, 1 = every eligible launch, default = by size
    if (mode == 0 || K != KW_K || N % 8 != 0) return 0;
    if (mode == 1) return 1;
    return M >= 4096 && N >= 256;
}

}  // extern "C"

static void kw_plan(K256P& p) {
    p.nchunks = ttsmi_cdiv(p.N, KW_BN);
    p.ntiles = ttsmi_cdiv(p.M, KW_BM);
    // one workgroup per CU: groups = a multiple of 8 with chunks x groups <= 256 (at least 8)
    int groups = (256 / p.nchunks) / 8 * 8;
    if (groups < 8) groups = 8;
    const int need = (p.ntiles + 7) / 8 * 8;
    if (groups > need) groups = need;
    p.ngroups = groups;
}

// mask != nullptr: bf16 output (out_bf16 must be set) kept where the bf16 mask[M, N] is > 0
static int kw_launch(const uint16_t* a, long lda, const uint16_t* bt, long ldb, const float* bias, void* c, long ldc,
                     int M, int N, int relu, int out_bf16, const uint16_t* mask, long ldmask, uint32_t* bits_out,
                     const uint32_t* bits_in, hipStream_t st);
int ttsmi_hgemm_k256_launch(const uint16_t* a, long lda, const uint16_t* bt, long ldb, const float* bias, void* c, long ldc,
                            int M, int N, int relu, int out_bf16, const uint16_t* mask, long ldmask, hipStream_t st) {
    return kw_launch(a, lda, bt, ldb, bias, c, ldc, M, N, relu, out_bf16, mask, ldmask, nullptr, nullptr, st);
}
static int kw_launch(const uint16_t* a, long lda, const uint16_t* bt, long ldb, const float* bias, void* c, long ldc,
                     int M, int N, int relu, int out_bf16, const uint16_t* mask, long ldmask, uint32_t* bits_out,
                     const uint32_t* bits_in, hipStream_t st) {
    K256P p;
    memset(&p, 0, sizeof(p));
    p.A = a; p.lda = lda; p.Bt = bt; p.ldb = ldb; p.C = c; p.ldc = ldc; p.bias = bias; p.relu = relu; p.M = M; p.N = N;
    p.mask = mask; p.ldmask = ldmask; p.bits_out = bits_out; p.bits_in = bits_in;
    // bf16 output, whole 256-column chunks, enough row tiles per workgroup to amortise its 128 KB of weights: wide variant
    TTSMI_KNOB(wide, "TTSMI_HGEMM_K256_WIDE", 1);               // TTSMI_HGEMM_K256_WIDE=0: always the 128-column kernel (A/B knob)
    if (wide && out_bf16 && N % KWW_BN == 0 && (M >= 16384 || wide > 1)) {
        p.nchunks = N / KWW_BN;
        p.ntiles = ttsmi_cdiv(M, KW_BM);
        int groups = (256 / p.nchunks) / 8 * 8;
        if (groups < 8) groups = 8;
        const int need = (p.ntiles + 7) / 8 * 8;
        if (groups > need) groups = need;
        p.ngroups = groups;
        TTSMI_ABLATE_KNOB(ablate, "TTSMI_K256_ABLATE");
        p.ablate = ablate;
        dim3 gridw(p.nchunks * p.ngroups);
        ttsmi_note_kernel(bits_in ? "gemm_k256_wide_kernel<4>" : mask ? "gemm_k256_wide_kernel<2>" : "gemm_k256_wide_kernel<1>");
        if (bits_in) TTSMI_LAUNCH_EV((gemm_k256_wide_kernel<4>), gridw, dim3(512), 0, st, p);
        else if (mask) TTSMI_LAUNCH_EV((gemm_k256_wide_kernel<2>), gridw, dim3(512), 0, st, p);
        else TTSMI_LAUNCH_EV((gemm_k256_wide_kernel<1>), gridw, dim3(512), 0, st, p);
        return 0;
    }
    kw_plan(p);
    dim3 grid(p.nchunks * p.ngroups);
    ttsmi_note_kernel(bits_in ? "gemm_k256_kernel<4>" : mask ? "gemm_k256_kernel<2>" : out_bf16 ? "gemm_k256_kernel<1>" : "gemm_k256_kernel<0>");
    if (bits_in) TTSMI_LAUNCH_EV((gemm_k256_kernel<4>), grid, dim3(512), 0, st, p);
    else if (mask) TTSMI_LAUNCH_EV((gemm_k256_kernel<2>), grid, dim3(512), 0, st, p);
    else if (out_bf16) TTSMI_LAUNCH_EV((gemm_k256_kernel<1>), grid, dim3(512), 0, st, p);
    else TTSMI_LAUNCH_EV((gemm_k256_kernel<0>), grid, dim3(512), 0, st, p);
    return 0;
}

static int kw_check(const void* a, int64_t lda, const void* bt, int64_t ldb, const void* c, int64_t ldc, const void* bits, int M, int N,
                    const char* who) {
    TTSMI_CHECK_ARG(a && bt && c && bits && (((uintptr_t)bits) & 7) == 0, "%s: null pointer / bit matrix not 8-byte aligned", who);
    TTSMI_CHECK_ARG(M > 0 && N > 0 && N % 8 == 0, "%s: bad shape M=%d N=%d", who, M, N);
    TTSMI_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 && ((((uintptr_t)a) | ((uintptr_t)bt) | ((uintptr_t)c)) & 15) == 0,
                    "%s: operands must be 16-byte aligned with 16-byte row pitches", who);
    if (!ttsmi_hgemm_k256_eligible(M, N, KW_K)) {
        ttsmi_set_error("%s: M=%d N=%d is not a launch of the K = 256 weight-stationary kernel (ttsmi_hgemm_k256_eligible)", who, M, N);
        return TTSMI_ERR_UNSUPPORTED;
    }
    return TTSMI_OK;
}
// one 1 KB block per (64-row tile, 128-column chunk) - the wide kernel's 2 KB blocks per 256 columns are the same total
extern "C" size_t ttsmi_relu_bits_bytes(int M, int N) {
    return M > 0 && N > 0 ? (size_t)ttsmi_cdiv(M, KW_BM) * (size_t)ttsmi_cdiv(N, KW_BN) * 1024 : 0;
}
extern "C" int ttsmi_hgemm_k256_relu_bits(const uint16_t* a, int64_t lda, const uint16_t* bt, int64_t ldb, const float* bias,
                                          uint16_t* c, int64_t ldc, uint8_t* bits, int M, int N, ttsmi_stream_t stream) {
    const int rc = kw_check(a, lda, bt, ldb, c, ldc, bits, M, N, "hgemm_k256_relu_bits");
    if (rc) return rc;
    kw_launch(a, (long)lda, bt, (long)ldb, bias, c, (long)ldc, M, N, 1, 1, nullptr, 0, (uint32_t*)bits, nullptr, (hipStream_t)stream);
    TTSMI_CHECK_LAUNCH("hgemm_k256_relu_bits");
    return TTSMI_OK;
}
extern "C" int ttsmi_hgemm_k256_masked_bits(const uint16_t* a, int64_t lda, const uint16_t* bt, int64_t ldb, const uint8_t* bits,
                                            uint16_t* c, int64_t ldc, int M, int N, ttsmi_stream_t stream) {
    const int rc = kw_check(a, lda, bt, ldb, c, ldc, bits, M, N, "hgemm_k256_masked_bits");
    if (rc) return rc;
    kw_launch(a, (long)lda, bt, (long)ldb, nullptr, c, (long)ldc, M, N, 0, 1, nullptr, 0, nullptr, (const uint32_t*)bits, (hipStream_t)stream);
    TTSMI_CHECK_LAUNCH("hgemm_k256_masked_bits");
    return TTSMI_OK;
}

extern "C" int ttsmi_hgemm_k256_split(const void* a, int64_t lda, const uint16_t* bt, int64_t ldb, float* c_acc, int64_t ldc_acc,
                                      int n_acc, void* c_bf16, int64_t ldc_bf16, int M, int N, ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(a && bt && c_acc && c_bf16, "hgemm_k256_split: null pointer");
    TTSMI_CHECK_ARG(M > 0 && N > n_acc && n_acc > 0 && n_acc % KW_BN == 0 && N % 8 == 0, "hgemm_k256_split: bad shape M=%d N=%d n_acc=%d", M, N, n_acc);
    TTSMI_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0 && ldc_acc % 4 == 0 && ldc_bf16 % 8 == 0 &&
                    ((((uintptr_t)a) | ((uintptr_t)bt) | ((uintptr_t)c_acc) | ((uintptr_t)c_bf16)) & 15) == 0,
                    "hgemm_k256_split: operands must be 16-byte aligned with 16-byte row pitches");
    K256P p;
    memset(&p, 0, sizeof(p));
    p.A = (const uint16_t*)a; p.lda = lda; p.Bt = bt; p.ldb = ldb; p.C = c_acc; p.ldc = ldc_acc; p.M = M; p.N = N;
    p.C2 = c_bf16; p.ldc2 = ldc_bf16; p.n_acc = n_acc;
    kw_plan(p);
    ttsmi_note_kernel("gemm_k256_kernel<3>");
    TTSMI_LAUNCH_EV((gemm_k256_kernel<3>), dim3(p.nchunks * p.ngroups), dim3(512), 0, (hipStream_t)stream, p);
    TTSMI_CHECK_LAUNCH("hgemm_k256_split");
    return TTSMI_OK;
}
