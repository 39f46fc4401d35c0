// TTSMI_BF16 GEMM family: bf16 operands (rounded to nearest even), fp32 accumulate on
// v_mfma_f32_32x32x16_bf16 (2.5 PFLOP/s dense chip peak), fp32 storage everywhere in HBM.
//
// One "TN" kernel:  C[M,N] = epilogue( sum_k A[m,k] * B[n,k] )  with BOTH operands K-contiguous,
// which is the only shape an MFMA bf16 fragment (8 consecutive k per lane) can be fetched in with
// 16-byte LDS reads.  The callers arrange that:
//   forward  : A = activations fp32 [M,K] (converted to bf16 while staging), B = W^T bf16 [N,K]
//   dgrad    : A = dy fp32 [M,N_out],  B = W bf16 [K_in, N_out] as stored (rows = k_in)
//   wgrad    : A = x^T bf16 [K_in, M], B = dy^T bf16 [N, M] (ttsmi_cast_transpose_bf16), split over M
//   Conv1D   : forward/dgrad read the contiguous k*C window of frame (b,t) (zero outside the sequence),
//              weights come pre-laid-out ([Cout][k*Cin] / [Cin][k'*Cout] with flipped taps); wgrad
//              uses the transposed im2col the cast kernel writes.
//
// Tile 128x128x64, 256 threads = 4 waves (2x2), 64x64 per wave = 2x2 MFMA 32x32 tiles (64 fp32
// accumulators), LDS images [128][64+8] bf16 (144-byte rows: the 16 lanes of a ds_read_b128 group
// hit 16 different 16-byte slots -> conflict free), register prefetch of the next k-tile.
// With fp32 activations in HBM these GEMMs are L2/HBM-bound (44 FLOP/B at this tile), not MFMA-bound.
#include <stdlib.h>

#include <type_traits>
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define HBM_ 128
#define HBN_ 128
#define HBK_ 64
#define HLD_ (HBK_ + 8)

struct HGemmP {
    const void* A; long lda;            // fp32 or bf16 rows
    const void* A2; long lda2; int K1;  // second K segment (fp32 A only)
    const uint16_t* B; long ldb;        // bf16 [N][K]
    float* C; long ldc;
    const float* bias;
    const float* relu_src; long ld_relu;
    int M, N, K;
    int relu, accumulate;               // accumulate: C += result (fused gradient accumulation)
    int c_bf16, mask_bf16;              // C stored as bf16 / relu_src is bf16
    int a_taps, T, Cw, pad;             // conv windowing on a fp32 A (a_taps == 1: none)
    int k_per_split;
    float* ws; float* colsum; float* colsum_ws;
    int tiles_m, tiles_n;
    int dma_burst;                      // gemm_bf16_dma256_kernel: a k-tile's DMA pieces all at once behind the barrier (A/B knob)
};

__device__ __forceinline__ uint2 pack4(float4 v) {
    bf16x4 h;
    h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
    return *reinterpret_cast<uint2*>(&h);
}

// ---- A tile fetch: fp32 source (BM/16 float4 per thread) -----------------------------------------
#define HC4_ (HBK_ / 4)      // float4 chunks per tile row
#define HC8_ (HBK_ / 8)      // 8 x bf16 chunks per tile row
#define HNA_(BM) ((BM) * HBK_ / 1024)   // float4 per thread of a BM-row fp32 tile
#define HNH_(R) ((R) * HBK_ / 2048)     // uint4 per thread of an R-row bf16 tile
template <int BM>
__device__ __forceinline__ void fetch_a_f32(const HGemmP& p, int m0, int k0, int kend, int tid,
                                            float4 (&r)[HNA_(BM)]) {
    const float* A = (const float*)p.A;
    const float* A2 = (const float*)p.A2;
#pragma unroll
    for (int i = 0; i < HNA_(BM); ++i) {
        int id = tid + 256 * i;
        int row = id / HC4_, c4 = id % HC4_;
        int m = m0 + row, kk = k0 + c4 * 4;
        bool ok = (m < p.M) && (kk < kend);
        const float* ptr;
        if (p.a_taps > 1) {
            int t = m % p.T, tap = kk / p.Cw, tt = t + tap - p.pad;
            ok = ok && (tt >= 0) && (tt < p.T);
            ptr = A + ((long)m - p.pad) * p.Cw + kk;
        } else if (A2 != nullptr && kk >= p.K1) {
            ptr = A2 + (long)m * p.lda2 + (kk - p.K1);
        } else {
            ptr = A + (long)m * p.lda + kk;
        }
        r[i] = ok ? *reinterpret_cast<const float4*>(ptr) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
template <int BM>
__device__ __forceinline__ void stash_a_f32(uint16_t (*S)[HLD_], int tid, const float4 (&r)[HNA_(BM)]) {
#pragma unroll
    for (int i = 0; i < HNA_(BM); ++i) {
        int id = tid + 256 * i;
        int row = id / HC4_, c4 = id % HC4_;
        *reinterpret_cast<uint2*>(&S[row][c4 * 4]) = pack4(r[i]);
    }
}
// ---- bf16 source tile of ROWS rows (ROWS/32 uint4 per thread) -----------------------------------
// A32: the chunk address is the (wave-uniform) matrix base plus a 32-bit BYTE offset, so the compiler keeps the
// base in SGPRs and one VGPR per chunk instead of a 64-bit pointer pair per chunk - 9 pointer pairs are what
// the 128-register build of the kernel spills.  Valid while the operand is smaller than 4 GB (checked at launch).
template <int ROWS, bool A32 = false>
__device__ __forceinline__ void fetch_h(const uint16_t* base, long ld, int rows, int r0, int k0,
                                        int kend, int tid, uint4 (&r)[HNH_(ROWS)]) {
#pragma unroll
    for (int i = 0; i < HNH_(ROWS); ++i) {
        int id = tid + 256 * i;
        int row = id / HC8_, c8 = id % HC8_;
        int m = r0 + row, kk = k0 + c8 * 8;
        bool ok = (m < rows) && (kk < kend);
        if constexpr (A32) {
            const uint32_t boff = ((uint32_t)m * (uint32_t)ld + (uint32_t)kk) * 2u;
            r[i] = ok ? *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(base) + boff) : make_uint4(0, 0, 0, 0);
        } else {
            r[i] = ok ? *reinterpret_cast<const uint4*>(base + (long)m * ld + kk) : make_uint4(0, 0, 0, 0);
        }
    }
}
template <int ROWS>
__device__ __forceinline__ void stash_h(uint16_t (*S)[HLD_], int tid, const uint4 (&r)[HNH_(ROWS)]) {
#pragma unroll
    for (int i = 0; i < HNH_(ROWS); ++i) {
        int id = tid + 256 * i;
        int row = id / HC8_, c8 = id % HC8_;
        *reinterpret_cast<uint4*>(&S[row][c8 * 8]) = r[i];
    }
}

#define EPLD 68     // fp32 row stride of the per-wave epilogue patch [32][64+4]

// BM = 128: wave tile 64 rows (2 MFMA tiles); BM = 64: 32 rows - half the registers, twice the resident
// workgroups.  BN = 128: wave tile 64 columns (2 MFMA tiles); BN = 256: 128 columns (4 tiles).
// (BN = 256 is a measurement knob, see hlaunch.)
// MINB = 4 (measurement knob TTSMI_HGEMM_OCC4=1, bf16 A only): a 128-register build so that four workgroups fit a
// CU instead of three (LDS allows it: 4 x 36 KB); it uses the 32-bit offset addressing above to get there
// without spills.  Prepared and checked statically in round 1 (register / spill counts), not yet run.
template <bool A_F32, int BM, int BN = HBN_, int MINB = 0>
__global__ __launch_bounds__(256, MINB) void gemm_bf16_kernel(HGemmP p) {
    constexpr bool A32 = MINB >= 4;
    constexpr int MI = BM / 64;
    constexpr int NJ = BN / 64;
    constexpr int TILE_BYTES = (BM + BN) * HLD_ * 2;
    constexpr int PATCH_BYTES = 4 * 32 * EPLD * 4;
    constexpr int SMEM_BYTES = TILE_BYTES > PATCH_BYTES ? TILE_BYTES : PATCH_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];
    uint16_t(*As)[HLD_] = reinterpret_cast<uint16_t(*)[HLD_]>(smem);
    uint16_t(*Bs)[HLD_] = As + BM;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = bid % p.tiles_n, tm = bid / p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = blockIdx.z * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const int l31 = lane & 31, kg = lane >> 5;

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[HNA_(BM)];   // fp32-A prefetch registers (dead in the bf16-A instantiation)
    uint4 rah[HNH_(BM)];   // bf16-A prefetch registers (dead in the fp32-A instantiation)
    uint4 rb[HNH_(BN)];
    // bf16 A with a second K segment (concat([q_in, ctx]) . W): K1 is a multiple of the k-step, so a
    // step lies wholly inside one segment
    auto fetch_a_h = [&](int k0) {
        if (p.A2 != nullptr && k0 >= p.K1)
            fetch_h<BM, A32>((const uint16_t*)p.A2, p.lda2, p.M, m0, k0 - p.K1, kend - p.K1, tid, rah);
        else
            fetch_h<BM, A32>((const uint16_t*)p.A, p.lda, p.M, m0, k0, p.A2 != nullptr ? min(kend, p.K1) : kend, tid, rah);
    };
    if (kbeg < kend) {
        if constexpr (A_F32) fetch_a_f32<BM>(p, m0, kbeg, kend, tid, ra);
        else fetch_a_h(kbeg);
        fetch_h<BN, A32>(p.B, p.ldb, p.N, n0, kbeg, kend, tid, rb);
        if constexpr (A_F32) stash_a_f32<BM>(As, tid, ra); else stash_h<BM>(As, tid, rah);
        stash_h<BN>(Bs, tid, rb);
    }
    __syncthreads();

    const bool do_colsum = (p.colsum != nullptr) && (tm == 0) && (tid < BN);
    float csum = 0.f;
    for (int k0 = kbeg; k0 < kend; k0 += HBK_) {
        const bool more = (k0 + HBK_) < kend;
        if (more) {
            if constexpr (A_F32) fetch_a_f32<BM>(p, m0, k0 + HBK_, kend, tid, ra);
            else fetch_a_h(k0 + HBK_);
            fetch_h<BN, A32>(p.B, p.ldb, p.N, n0, k0 + HBK_, kend, tid, rb);
        }
        if (do_colsum) {                       // bias gradient from the dy^T tile (wgrad)
#pragma unroll
            for (int c8 = 0; c8 < HBK_ / 8; ++c8) {
                bf16x8 v = *reinterpret_cast<const bf16x8*>(&Bs[tid][c8 * 8]);
#pragma unroll
                for (int e = 0; e < 8; ++e) csum += (float)v[e];
            }
        }
#pragma unroll
        for (int ks = 0; ks < HBK_ / 16; ++ks) {
            const int ko = ks * 16 + kg * 8;
            bf16x8 a[MI], b[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i)
                a[i] = *reinterpret_cast<const bf16x8*>(&As[wr * (BM / 2) + i * 32 + l31][ko]);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                b[j] = *reinterpret_cast<const bf16x8*>(&Bs[wc * (BN / 2) + j * 32 + l31][ko]);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (more) {
            if constexpr (A_F32) stash_a_f32<BM>(As, tid, ra); else stash_h<BM>(As, tid, rah);
            stash_h<BN>(Bs, tid, rb);
        }
        __syncthreads();
    }

    const bool split = gridDim.z > 1;
    if (do_colsum && n0 + tid < p.N) {
        if (split) p.colsum_ws[(long)blockIdx.z * p.N + n0 + tid] = csum;
        else p.colsum[n0 + tid] = csum;
    }
    // ---- epilogue: accumulators -> per-wave LDS patch -> full-row 16-byte stores -----------------
    float* Cb = split ? p.ws + (long)blockIdx.z * p.M * p.N : p.C;
    const long ldc = split ? (long)p.N : p.ldc;
    const bool vec = ((ldc & 3) == 0) && ((p.N & 3) == 0) && ((((uintptr_t)Cb) & 15) == 0);
    const bool fuse = !split;
    float* patch = reinterpret_cast<float*>(smem) + wave * 32 * EPLD;
    const int rl = lane >> 4, c4 = lane & 15;
#pragma unroll
    for (int jh = 0; jh < NJ / 2; ++jh) {              // 64-column halves of the wave tile
    const int col = n0 + wc * (BN / 2) + jh * 64 + c4 * 4;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (fuse && p.bias) {
        if (col + 0 < p.N) bias4.x = p.bias[col + 0];
        if (col + 1 < p.N) bias4.y = p.bias[col + 1];
        if (col + 2 < p.N) bias4.z = p.bias[col + 2];
        if (col + 3 < p.N) bias4.w = p.bias[col + 3];
    }
    // epilogue operands that come from global memory - the accumulate source and the bf16 ReLU' mask - are fetched
    // for all 8 row groups of a 32-row slab BEFORE its accumulators are staged through LDS: their latency hides under
    // the patch round trip.  (Loaded inside the store loop, each was a dependent HBM round trip between a patch read and
    // its store: the masked dgrad ran 87 us against 42 us for the same shape without a mask.)
    const bool pre_acc = fuse && vec && p.accumulate && !p.c_bf16;
    const bool pre_msk = fuse && vec && p.relu_src != nullptr && p.mask_bf16;
    const bool colok = col < p.N;
    const int colc = colok ? col : 0;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        float4 pre_o[8];
        uint2 pre_m[8];
        if (pre_acc || pre_msk) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = min(m0 + wr * (BM / 2) + i * 32 + it * 4 + rl, p.M - 1);
                if (pre_acc) pre_o[it] = *reinterpret_cast<const float4*>(Cb + (long)row * ldc + colc);
                if (pre_msk) pre_m[it] = *reinterpret_cast<const uint2*>((const uint16_t*)p.relu_src + (long)row * p.ld_relu + colc);
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                patch[((r & 3) + 8 * (r >> 2) + 4 * kg) * EPLD + j * 32 + l31] = acc[i][jh * 2 + j][r];
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int prow = it * 4 + rl;
            const int row = m0 + wr * (BM / 2) + i * 32 + prow;
            if (row >= p.M || !colok) continue;
            float4 v = *reinterpret_cast<const float4*>(patch + prow * EPLD + c4 * 4);
            if (fuse) {
                v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
                if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            }
            float* dst = Cb + (long)row * ldc + col;
            if (vec) {
                if (fuse && p.relu_src) {
                    if (p.mask_bf16) {      // bf16 activation: > 0  <=>  sign clear and magnitude non-zero
                        const uint2 mb = pre_m[it];
                        auto pos = [](uint32_t h) { return ((h & 0x8000u) == 0u) && ((h & 0x7FFFu) != 0u); };
                        v.x = pos(mb.x & 0xFFFFu) ? v.x : 0.f; v.y = pos(mb.x >> 16) ? v.y : 0.f;
                        v.z = pos(mb.y & 0xFFFFu) ? v.z : 0.f; v.w = pos(mb.y >> 16) ? v.w : 0.f;
                    } else {
                        float4 m = *reinterpret_cast<const float4*>(p.relu_src + (long)row * p.ld_relu + col);
                        v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f;
                        v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
                    }
                }
                if (fuse && p.c_bf16) {
                    *reinterpret_cast<uint2*>((uint16_t*)p.C + (long)row * p.ldc + col) = pack4(v);
                    continue;
                }
                if (fuse && p.accumulate) {
                    const float4 o = pre_o[it];
                    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                }
                *reinterpret_cast<float4*>(dst) = v;
            } else {
                const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (col + e >= p.N) break;
                    float x = vv[e];
                    if (fuse && p.relu_src) x = p.relu_src[(long)row * p.ld_relu + col + e] > 0.f ? x : 0.f;
                    if (fuse && p.accumulate) x += dst[e];
                    dst[e] = x;
                }
            }
        }
        __syncthreads();
    }
    }
}


// One LDS-DMA instruction, emitted as inline asm ON PURPOSE: for the builtin form the compiler's waitcnt pass
// treats every later LDS read as a possible consumer of the DMA and inserts s_waitcnt vmcnt(0) in front of
// it - i.e. the multiply of stage s would wait for the stage that was just put in flight.  The kernels below
// do their own vmcnt accounting (a fixed number of DMA instructions per step and nothing else on that counter
// inside the loops).  lds_off = wave-uniform byte offset of the 1 KB destination (lane i lands at +16 i).
__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_off) : "memory", "m0");
}
// Barrier of the DMA pipelines.  A bare s_barrier (not __syncthreads(), whose fence is vmcnt(0) and would also
// wait for the stage just put in flight) - but WITH lgkmcnt(0): the "memory" clobber keeps the LDS reads of the
// finished step above the barrier, it does not keep the wait for their DATA there.  hipcc sinks the last MFMAs
// of a step (and the s_waitcnt lgkmcnt in front of them) below the barrier, so a wave would arrive with ds_reads
// still queued while a faster wave already overwrites that stage by DMA - rare wrong operands, found as
// run-to-run different checksums (tools/check_determinism.py).  The DMA counter (vmcnt) is waited by the caller.
__device__ __forceinline__ void lds_stage_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ unsigned lds_offset(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}

// =================================================================================================
// LDS-DMA persistent variant of the TN GEMM (bf16 A, K % 64 == 0): c[M,N] (+)= act(a.b^T + bias).mask
//
//  * tiles arrive by `global_load_lds_dwordx4` (gfx950 LDS-DMA): a wave instruction moves 64 x 16 B straight
//    from global memory into 1 KB of LDS - no staging registers, so a whole k-tile (32 KB) is in flight per
//    workgroup while the previous one is multiplied, and the NEXT TILE's first k-tile is in flight under
//    the epilogue (persistent tile loop) without costing a single live register.  That is what the
//    register-staged kernel above cannot do: its stage ablation shows ~1/3 of a short-K launch is the
//    cold-start fetch every workgroup exposes, and prefetching the next tile through registers spilled.
//  * the DMA image is lane-linear (lane i -> base + 16 i; placement pinned by tools/probes/lds_dma_probe.hip),
//    i.e. dense 128-byte rows [row][8 chunks of 8 bf16].  Dense rows would make the ds_read_b128 fragment
//    reads 8-way bank conflicted, so chunk c of row r is stored at chunk position c ^ ((r >> 1) & 7): the
//    swizzle costs nothing on the load side (each lane just fetches a different global chunk) and makes
//    every b128 lane group hit 16 distinct 16-byte slots.
//  * two 32 KB stages (64 KB LDS, 2 workgroups per CU), ONE barrier per k-step (vmcnt(0) + barrier publishes
//    the stage that was in flight and retires the one just read); rows past M / N are clamped to the last
//    valid row on the load side and dropped at the store.
// =================================================================================================
#define DBM 128
#define DBN 128
#define DSTAGE ((DBM + DBN) * HBK_ * 2)           // bytes per stage: A image then B image
#define DPLD 68                                    // fp32 row stride of the per-wave epilogue patch [16][64+4]
// a patch is private to its wave: wave-scope ordering is all its write -> read hand-off needs
#define HWAVE_SYNC()                                             \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
    } while (0)

__global__ __launch_bounds__(256, 2) void gemm_bf16_dma_kernel(HGemmP p) {
    static_assert(HBK_ == 64, "the swizzled image assumes 8 chunks of 8 bf16 per row");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * DSTAGE];
    typedef __attribute__((address_space(1))) const void* gptr;
    typedef __attribute__((address_space(3))) void* lptr;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, kg = lane >> 5;
    const int kend = p.K;
    const int nk = p.K / HBK_;
    // persistent tile walk (see gemm_bf16_kernel's XCD note): XCD x owns tiles [base, base + len)
    const int T = p.tiles_m * p.tiles_n;
    const int nx = 8, xcd = blockIdx.x % nx, slot = blockIdx.x / nx;
    const int per = (gridDim.x + nx - 1 - xcd) / nx;
    const int tq = T / nx, tr = T % nx;
    const int len = tq + (xcd < tr ? 1 : 0);
    const int base = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;

    const uint16_t* A = (const uint16_t*)p.A;
    const uint16_t* A2 = (const uint16_t*)p.A2;
    // this lane's place in a DMA wave instruction: 8 rows x 8 chunk positions
    const int drow = lane >> 3, dpos = lane & 7;
    auto issue = [&](int m0, int n0, int k0, int st) {
        unsigned char* As = smem + st * DSTAGE;
        unsigned char* Bs = As + DBM * HBK_ * 2;
        const uint16_t* Ab = A;
        long lda = p.lda;
        int ka = k0;
        if (A2 != nullptr && k0 >= p.K1) { Ab = A2; lda = p.lda2; ka = k0 - p.K1; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wave * 32 + i * 8 + drow;
            const int c = dpos ^ ((row >> 1) & 7);
            const int gm = min(m0 + row, p.M - 1);
            lds_dma16(Ab + (long)gm * lda + ka + c * 8, lds_offset(As + (wave * 32 + i * 8) * 128));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wave * 32 + i * 8 + drow;
            const int c = dpos ^ ((row >> 1) & 7);
            const int gn = min(n0 + row, p.N - 1);
            lds_dma16(p.B + (long)gn * p.ldb + k0 + c * 8, lds_offset(Bs + (wave * 32 + i * 8) * 128));
        }
    };
    int st = 0;
    if (slot < len) {
        const int t0 = base + slot;
        issue((t0 / p.tiles_n) * DBM, (t0 % p.tiles_n) * DBN, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0xF70);
    lds_stage_barrier();

    for (int ti = slot; ti < len; ti += per) {
        const int tile = base + ti;
        const int tn = tile % p.tiles_n, tm = tile / p.tiles_n;
        const int m0 = tm * DBM, n0 = tn * DBN;
        const bool has_next = (ti + per) < len;
        const int tnext = tile + per;
        const int m0n = (tnext / p.tiles_n) * DBM, n0n = (tnext % p.tiles_n) * DBN;

        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        for (int ks = 0; ks < nk; ++ks) {
            const bool more = ks + 1 < nk;
            if (more) issue(m0, n0, (ks + 1) * HBK_, st ^ 1);
            else if (has_next) issue(m0n, n0n, 0, st ^ 1);          // next tile's first k-tile, under the epilogue
            const unsigned char* As = smem + st * DSTAGE;
            const unsigned char* Bs = As + DBM * HBK_ * 2;
#pragma unroll
            for (int kk = 0; kk < HBK_ / 16; ++kk) {
                const int c = kk * 2 + kg;
                bf16x8 a[2], b[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = wr * 64 + i * 32 + l31;
                    a[i] = *reinterpret_cast<const bf16x8*>(As + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int row = wc * 64 + j * 32 + l31;
                    b[j] = *reinterpret_cast<const bf16x8*>(Bs + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_s_waitcnt(0xF70);    // vmcnt(0): this wave's DMA pieces of the next stage have landed
            lds_stage_barrier();                  // ... everybody's have, and nobody still reads stage `st`
            st ^= 1;
        }
        // ---- epilogue: the stage just consumed (st ^ 1) is free - its first 17 KB hold the four per-wave patches
        float* patch = reinterpret_cast<float*>(smem + (st ^ 1) * DSTAGE) + wave * 16 * DPLD;
        const long ldc = p.ldc;
        const bool vec = ((ldc & 3) == 0) && ((p.N & 3) == 0) && ((((uintptr_t)p.C) & 15) == 0);
        const int rl = lane >> 4, c4 = lane & 15;
        const int col = n0 + wc * 64 + c4 * 4;
        float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias) {
            if (col + 0 < p.N) bias4.x = p.bias[col + 0];
            if (col + 1 < p.N) bias4.y = p.bias[col + 1];
            if (col + 2 < p.N) bias4.z = p.bias[col + 2];
            if (col + 3 < p.N) bias4.w = p.bias[col + 3];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {                 // registers 8h..8h+7 = rows 16h..16h+15 of the 32-row tile
                // global-memory epilogue operands first (see gemm_bf16_kernel): their latency hides under the patch round trip
                float4 pre_o[4];
                uint2 pre_m[4];
                const bool pre_acc = vec && p.accumulate && !p.c_bf16;
                const bool pre_msk = vec && p.relu_src != nullptr && p.mask_bf16;
                const int colc = col < p.N ? col : 0;
                if (pre_acc || pre_msk) {
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int row = min(m0 + wr * 64 + i * 32 + h * 16 + it * 4 + rl, p.M - 1);
                        if (pre_acc) pre_o[it] = *reinterpret_cast<const float4*>(p.C + (long)row * ldc + colc);
                        if (pre_msk) pre_m[it] = *reinterpret_cast<const uint2*>((const uint16_t*)p.relu_src + (long)row * p.ld_relu + colc);
                    }
                }
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 8; ++r)
                        patch[((r & 3) + 8 * (r >> 2) + 4 * kg) * DPLD + j * 32 + l31] = acc[i][j][8 * h + r];
                HWAVE_SYNC();
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int prow = it * 4 + rl;
                    const int row = m0 + wr * 64 + i * 32 + h * 16 + prow;
                    if (row >= p.M || col >= p.N) continue;
                    float4 v = *reinterpret_cast<const float4*>(patch + prow * DPLD + c4 * 4);
                    v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
                    if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    float* dst = p.C + (long)row * ldc + col;
                    if (vec) {
                        if (p.relu_src) {
                            if (p.mask_bf16) {
                                const uint2 mb = pre_m[it];
                                auto pos = [](uint32_t hh) { return ((hh & 0x8000u) == 0u) && ((hh & 0x7FFFu) != 0u); };
                                v.x = pos(mb.x & 0xFFFFu) ? v.x : 0.f; v.y = pos(mb.x >> 16) ? v.y : 0.f;
                                v.z = pos(mb.y & 0xFFFFu) ? v.z : 0.f; v.w = pos(mb.y >> 16) ? v.w : 0.f;
                            } else {
                                float4 m = *reinterpret_cast<const float4*>(p.relu_src + (long)row * p.ld_relu + col);
                                v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f;
                                v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
                            }
                        }
                        if (p.c_bf16) {
                            *reinterpret_cast<uint2*>((uint16_t*)p.C + (long)row * p.ldc + col) = pack4(v);
                            continue;
                        }
                        if (p.accumulate) {
                            const float4 o = pre_o[it];
                            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                        }
                        *reinterpret_cast<float4*>(dst) = v;
                    } else {
                        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (col + e >= p.N) break;
                            float x = vv[e];
                            if (p.relu_src) x = p.relu_src[(long)row * p.ld_relu + col + e] > 0.f ? x : 0.f;
                            if (p.accumulate) x += dst[e];
                            dst[e] = x;
                        }
                    }
                }
                HWAVE_SYNC();
            }
        }
        __syncthreads();          // the next step's DMA overwrites the stage the patches live in
    }
}

// =================================================================================================
// 256 x 256 tiles for the LARGE launches (the reference-default conv stacks: M = 28 864, N = 1 536, K = 1 152 / 4 608 -
// 100 GFLOP each).  What the LDS-DMA fill of a CU sustains is ~64 GB/s (tools/probes/stream_tile_probe.hip: 16 TB/s over
// the chip, whatever the ring depth); a 128 x 128 x 64 step is 2.1 MFLOP per 32 KB filled = 64 FLOP per byte, i.e. at most
// ~1.05 PFLOP/s however the loop is written - the kernel above measures 0.69-0.78.  A 256 x 256 x 64 step is 8.4 MFLOP per
// 64 KB = 131 FLOP per byte.  One 8-wave workgroup per CU (2 x 4 waves, wave tile 128 x 64 = eight accumulators), two
// 64 KB stages; everything else - images, swizzle, persistent tile walk, the next tile's first k-tile under the epilogue,
// the per-wave epilogue patches - is the kernel above's.
// =================================================================================================
#define D2BM 256
#define D2BN 256
#define D2STAGE ((D2BM + D2BN) * HBK_ * 2)         // 65 536 bytes per stage

__global__ __launch_bounds__(512, 1) void gemm_bf16_dma256_kernel(HGemmP p) {
    static_assert(HBK_ == 64, "the swizzled image assumes 8 chunks of 8 bf16 per row");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * D2STAGE];
    typedef __attribute__((address_space(1))) const void* gptr;
    typedef __attribute__((address_space(3))) void* lptr;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;          // 2 x 4 waves, wave tile 128 x 64
    const int l31 = lane & 31, kg = lane >> 5;
    const int kend = p.K;
    const int nk = p.K / HBK_;
    // persistent tile walk (see gemm_bf16_kernel's XCD note): XCD x owns tiles [base, base + len)
    const int T = p.tiles_m * p.tiles_n;
    const int nx = 8, xcd = blockIdx.x % nx, slot = blockIdx.x / nx;
    const int per = (gridDim.x + nx - 1 - xcd) / nx;
    const int tq = T / nx, tr = T % nx;
    const int len = tq + (xcd < tr ? 1 : 0);
    const int base = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;

    const uint16_t* A = (const uint16_t*)p.A;
    const uint16_t* A2 = (const uint16_t*)p.A2;
    // this lane's place in a DMA wave instruction: 8 rows x 8 chunk positions
    const int drow = lane >> 3, dpos = lane & 7;
    // q < 0: the whole k-tile (8 DMA instructions per wave); q = 0..3: piece q of the A image and of the B image.  Inside the
    // k-loop the pieces are spread over the four 16-wide slices, each pair issued while that slice's fragment reads are in
    // flight: a DMA instruction costs its wave 60-180 cycles of issue time, and eight of them right behind the barrier
    // stalled all eight waves at the same moment with the matrix pipe empty (TTSMI_HGEMM_T256_BURST=1 restores that)
    auto issue = [&](int m0, int n0, int k0, int st, int q) {
        unsigned char* As = smem + st * D2STAGE;
        unsigned char* Bs = As + D2BM * HBK_ * 2;
        const uint16_t* Ab = A;
        long lda = p.lda;
        int ka = k0;
        if (A2 != nullptr && k0 >= p.K1) { Ab = A2; lda = p.lda2; ka = k0 - p.K1; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (q >= 0 && i != q) continue;
            const int row = wave * 32 + i * 8 + drow;
            const int c = dpos ^ ((row >> 1) & 7);
            const int gm = min(m0 + row, p.M - 1);
            lds_dma16(Ab + (long)gm * lda + ka + c * 8, lds_offset(As + (wave * 32 + i * 8) * 128));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (q >= 0 && i != q) continue;
            const int row = wave * 32 + i * 8 + drow;
            const int c = dpos ^ ((row >> 1) & 7);
            const int gn = min(n0 + row, p.N - 1);
            lds_dma16(p.B + (long)gn * p.ldb + k0 + c * 8, lds_offset(Bs + (wave * 32 + i * 8) * 128));
        }
    };
    const bool burst = p.dma_burst != 0;
    int st = 0;
    if (slot < len) {
        const int t0 = base + slot;
        issue((t0 / p.tiles_n) * D2BM, (t0 % p.tiles_n) * D2BN, 0, 0, -1);
    }
    __builtin_amdgcn_s_waitcnt(0xF70);
    lds_stage_barrier();

    for (int ti = slot; ti < len; ti += per) {
        const int tile = base + ti;
        const int tn = tile % p.tiles_n, tm = tile / p.tiles_n;
        const int m0 = tm * D2BM, n0 = tn * D2BN;
        const bool has_next = (ti + per) < len;
        const int tnext = tile + per;
        const int m0n = (tnext / p.tiles_n) * D2BM, n0n = (tnext % p.tiles_n) * D2BN;

        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        for (int ks = 0; ks < nk; ++ks) {
            const bool more = ks + 1 < nk;
            const bool fill = more || has_next;                    // (else: the last k-step of the workgroup's last tile)
            const int fm = more ? m0 : m0n, fn = more ? n0 : n0n, fk = more ? (ks + 1) * HBK_ : 0;   // next tile's first k-tile: under the epilogue
            if (fill && burst) issue(fm, fn, fk, st ^ 1, -1);
            const unsigned char* As = smem + st * D2STAGE;
            const unsigned char* Bs = As + D2BM * HBK_ * 2;
#pragma unroll
            for (int kk = 0; kk < HBK_ / 16; ++kk) {
                const int c = kk * 2 + kg;
                bf16x8 a[4], b[2];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = wr * 128 + i * 32 + l31;
                    a[i] = *reinterpret_cast<const bf16x8*>(As + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int row = wc * 64 + j * 32 + l31;
                    b[j] = *reinterpret_cast<const bf16x8*>(Bs + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
                }
                if (fill && !burst) issue(fm, fn, fk, st ^ 1, kk);
                __builtin_amdgcn_sched_barrier(0);        // the six fragment reads (and two DMA pieces) are in flight before the eight multiplies
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_s_waitcnt(0xF70);    // vmcnt(0): this wave's DMA pieces of the next stage have landed
            lds_stage_barrier();                  // ... everybody's have, and nobody still reads stage `st`
            st ^= 1;
        }
        // ---- epilogue: the stage just consumed (st ^ 1) is free - its first 35 KB hold the eight per-wave patches
        float* patch = reinterpret_cast<float*>(smem + (st ^ 1) * D2STAGE) + wave * 16 * DPLD;
        const long ldc = p.ldc;
        const bool vec = ((ldc & 3) == 0) && ((p.N & 3) == 0) && ((((uintptr_t)p.C) & 15) == 0);
        const int rl = lane >> 4, c4 = lane & 15;
        const int col = n0 + wc * 64 + c4 * 4;
        float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias) {
            if (col + 0 < p.N) bias4.x = p.bias[col + 0];
            if (col + 1 < p.N) bias4.y = p.bias[col + 1];
            if (col + 2 < p.N) bias4.z = p.bias[col + 2];
            if (col + 3 < p.N) bias4.w = p.bias[col + 3];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {                 // registers 8h..8h+7 = rows 16h..16h+15 of the 32-row tile
                // global-memory epilogue operands first (see gemm_bf16_kernel): their latency hides under the patch round trip
                float4 pre_o[4];
                uint2 pre_m[4];
                const bool pre_acc = vec && p.accumulate && !p.c_bf16;
                const bool pre_msk = vec && p.relu_src != nullptr && p.mask_bf16;
                const int colc = col < p.N ? col : 0;
                if (pre_acc || pre_msk) {
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int row = min(m0 + wr * 128 + i * 32 + h * 16 + it * 4 + rl, p.M - 1);
                        if (pre_acc) pre_o[it] = *reinterpret_cast<const float4*>(p.C + (long)row * ldc + colc);
                        if (pre_msk) pre_m[it] = *reinterpret_cast<const uint2*>((const uint16_t*)p.relu_src + (long)row * p.ld_relu + colc);
                    }
                }
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 8; ++r)
                        patch[((r & 3) + 8 * (r >> 2) + 4 * kg) * DPLD + j * 32 + l31] = acc[i][j][8 * h + r];
                HWAVE_SYNC();
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int prow = it * 4 + rl;
                    const int row = m0 + wr * 128 + i * 32 + h * 16 + prow;
                    if (row >= p.M || col >= p.N) continue;
                    float4 v = *reinterpret_cast<const float4*>(patch + prow * DPLD + c4 * 4);
                    v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
                    if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    float* dst = p.C + (long)row * ldc + col;
                    if (vec) {
                        if (p.relu_src) {
                            if (p.mask_bf16) {
                                const uint2 mb = pre_m[it];
                                auto pos = [](uint32_t hh) { return ((hh & 0x8000u) == 0u) && ((hh & 0x7FFFu) != 0u); };
                                v.x = pos(mb.x & 0xFFFFu) ? v.x : 0.f; v.y = pos(mb.x >> 16) ? v.y : 0.f;
                                v.z = pos(mb.y & 0xFFFFu) ? v.z : 0.f; v.w = pos(mb.y >> 16) ? v.w : 0.f;
                            } else {
                                float4 m = *reinterpret_cast<const float4*>(p.relu_src + (long)row * p.ld_relu + col);
                                v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f;
                                v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
                            }
                        }
                        if (p.c_bf16) {
                            *reinterpret_cast<uint2*>((uint16_t*)p.C + (long)row * p.ldc + col) = pack4(v);
                            continue;
                        }
                        if (p.accumulate) {
                            const float4 o = pre_o[it];
                            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                        }
                        *reinterpret_cast<float4*>(dst) = v;
                    } else {
                        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (col + e >= p.N) break;
                            float x = vv[e];
                            if (p.relu_src) x = p.relu_src[(long)row * p.ld_relu + col + e] > 0.f ? x : 0.f;
                            if (p.accumulate) x += dst[e];
                            dst[e] = x;
                        }
                    }
                }
                HWAVE_SYNC();
            }
        }
        __syncthreads();          // the next step's DMA overwrites the stage the patches live in
    }
}

// =================================================================================================
// Deep-ring variant for launches that cannot fill the GPU (inference at batch 1: M = 400 / 2304 rows; the encoder side
// of a training step: M = 6400).  There the k-loop is a chain of exposed memory round trips - measured ~1 us per 64-wide
// k-step whatever M is (K = 256 / 512 / 1024 at M = 400: 7.6 / 10.9 / 20 us) - because a workgroup has nothing
// else resident on its CU to hide behind and the register-staged kernel keeps ONE k-tile in flight.  This kernel keeps
// NST - 1 = 3 k-tiles in flight through an LDS-DMA ring (a whole K = 256 problem is requested before the first multiply)
// and cuts the tile to 64 x 64 so that four times as many CUs take part.  Same swizzled 128-byte-row image as the
// persistent DMA kernel above; one barrier per k-step; the accumulators leave straight from registers (lane = column:
// a register row is 32 consecutive floats = 128-byte segments - fine for a kernel that is latency, not store, bound).
// =================================================================================================
#define SBM 64
#define SBN 64
#define SSTAGE ((SBM + SBN) * HBK_ * 2)
template <int NST>
__global__ __launch_bounds__(256, 2) void gemm_bf16_deep_kernel(HGemmP p) {
    static_assert(HBK_ == 64, "the swizzled image assumes 8 chunks of 8 bf16 per row");
    __shared__ __attribute__((aligned(16))) unsigned char smem[NST * SSTAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, kg = lane >> 5;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = bid % p.tiles_n, tm = bid / p.tiles_n;
    const int m0 = tm * SBM, n0 = tn * SBN;
    const int nk = p.K / HBK_;
    const uint16_t* A = (const uint16_t*)p.A;
    const uint16_t* A2 = (const uint16_t*)p.A2;
    const int drow = lane >> 3, dpos = lane & 7;
    auto issue = [&](int ks, int st) {                      // 4 DMA instructions per wave and stage
        unsigned char* As = smem + st * SSTAGE;
        unsigned char* Bs = As + SBM * HBK_ * 2;
        const int k0 = ks * HBK_;
        const uint16_t* Ab = A;
        long lda = p.lda;
        int ka = k0;
        if (A2 != nullptr && k0 >= p.K1) { Ab = A2; lda = p.lda2; ka = k0 - p.K1; }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wave * 16 + i * 8 + drow;
            const int c = dpos ^ ((row >> 1) & 7);
            const int gm = min(m0 + row, p.M - 1);
            lds_dma16(Ab + (long)gm * lda + ka + c * 8, lds_offset(As + (wave * 16 + i * 8) * 128));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wave * 16 + i * 8 + drow;
            const int c = dpos ^ ((row >> 1) & 7);
            const int gn = min(n0 + row, p.N - 1);
            lds_dma16(p.B + (long)gn * p.ldb + k0 + c * 8, lds_offset(Bs + (wave * 16 + i * 8) * 128));
        }
    };
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < nk) issue(s, s);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int ks = 0; ks < nk; ++ks) {
        // k-tiles requested after tile ks and possibly still in flight: min(NST - 2, nk - 1 - ks) of them, 4 DMAs each
        const int newer = min(NST - 2, nk - 1 - ks);
        if (newer >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (newer == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_stage_barrier();          // everybody's pieces of tile ks have landed, nobody still reads tile ks - 1 ...
        if (ks + NST - 1 < nk) issue(ks + NST - 1, (ks + NST - 1) % NST);       // ... whose stage is refilled
        const unsigned char* As = smem + (ks % NST) * SSTAGE;
        const unsigned char* Bs = As + SBM * HBK_ * 2;
#pragma unroll
        for (int kk = 0; kk < HBK_ / 16; ++kk) {
            const int c = kk * 2 + kg;
            const int ra = wr * 32 + l31, rb = wc * 32 + l31;
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(As + ra * 128 + ((c ^ ((ra >> 1) & 7)) << 4));
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(Bs + rb * 128 + ((c ^ ((rb >> 1) & 7)) << 4));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
    }
    const int col = n0 + wc * 32 + l31;
    if (col >= p.N) return;
    const float bias = p.bias ? p.bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
        if (row >= p.M) continue;
        float v = acc[r] + bias;
        if (p.relu) v = fmaxf(v, 0.f);
        if (p.relu_src) {
            bool on;
            if (p.mask_bf16) {
                const uint32_t h = ((const uint16_t*)p.relu_src)[(long)row * p.ld_relu + col];
                on = ((h & 0x8000u) == 0u) && ((h & 0x7FFFu) != 0u);
            } else
                on = p.relu_src[(long)row * p.ld_relu + col] > 0.f;
            v = on ? v : 0.f;
        }
        if (p.c_bf16) {
            ((__bf16*)p.C)[(long)row * p.ldc + col] = (__bf16)v;
        } else {
            float* dst = p.C + (long)row * p.ldc + col;
            *dst = p.accumulate ? *dst + v : v;
        }
    }
}

// =================================================================================================
// wgrad straight from the row-major fp32 activations: dW[K_in, N] = X[M, K_in]^T . dY[M, N].
// Both operands are fetched as [32 rows][128 cols] tiles (coalesced 512-byte fp32 / 256-byte bf16 rows)
// and rounded to bf16 into row-major LDS images.  The MFMA fragments need 8 consecutive REDUCTION
// indices per lane = 8 consecutive rows of one column of the image: gfx950's transposing LDS read
// (ds_read_b64_tr_b16: per 16-lane group a [4 rows][16 cols] block, lane c gets column c's 4 values)
// delivers exactly that from the row-major image, so no re-layout pass and a single barrier per
// step (images are double buffered).  Row stride 160 bf16 = 80 dwords = 16 (mod 64): the 4 rows a
// half-wave reads fall on disjoint bank groups.
// Versus cast_transpose + TN-GEMM this reads each activation once and writes nothing but dW.
// db (the bias gradient) rides along as one more MFMA with an all-ones A fragment.
// Conv1D: the K_in tile [j*Cin + c0, +128) lies inside one tap j (Cin % 128 == 0), so its source is
// the same X tile shifted by (j - pad) frames, rows outside their sequence zeroed.
// =================================================================================================
#define WR_ROWS 32
#define WR_RLD 160             // bf16 row image stride
#define WR_NI (WR_ROWS / 8)    // fetch items per thread per operand
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
// 8 consecutive rows (r0..r0+7) of one column per lane, through two transposing reads
__device__ __forceinline__ bf16x8 wr_tr8(const uint16_t* p) {
    typedef __attribute__((address_space(3))) s16x4* lds_ptr;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)p);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p + 4 * WR_RLD));
    s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
struct WRowsP {
    const float* X; long ldx; const float* DY; long lddy;
    float* dW; long lddw;
    int M, K, N;                       // rows, K_in (= taps*Cin), N
    int taps, T, Cin, pad;
    int k_per_split;
    float* ws; float* colsum; float* colsum_ws;
    int tiles_k, tiles_n;
    // dual X (no conv window): K_in rows [0, K1) of dW come from X, rows [K1, K) from X2 - the two halves of
    // Dense(concat([q_in, ctx])) (model/layers.py:148) as ONE weight gradient; X2 == nullptr: single operand
    const float* X2; long ldx2; int K1;
};

// Per-thread fetch cursor over a row-major operand tile: item i = row (tid >> 5) + 8 i of the step, 4
// columns at (tid & 31) * 4.  All address arithmetic is done once; a step costs one pointer add (the
// first version recomputed 64-bit row products and a software `m % T` per item per step - with one
// wave per SIMD that instruction stream, not HBM, set the step time).
typedef __attribute__((address_space(1))) const char* wr_gptr;     // forces global_load (not flat_load)
struct WrCursor {
    wr_gptr ptr;          // item 0 of the current step
    long step8, stepS;    // bytes between items (8 rows) / between steps (WR_ROWS rows)
    int m;                // source row of item 0
    int tq;               // m % T (conv windows only)
    bool colok;
};
__device__ __forceinline__ WrCursor wr_cursor(const void* base, int ebytes, long ld, int ncols, int row0, int col0,
                                              int shift, int T, int tid) {
    WrCursor c;
    const int row = tid >> 5, col = col0 + (tid & 31) * 4;
    c.m = row0 + row;
    c.colok = col < ncols;
    c.ptr = (wr_gptr)base + (((long)c.m + shift) * ld + col) * ebytes;
    c.step8 = 8 * ld * ebytes;
    c.stepS = (long)WR_ROWS * ld * ebytes;
    c.tq = T > 0 ? c.m % T : 0;
    return c;
}
__device__ __forceinline__ void wr_advance(WrCursor& c, int T) {
    c.ptr += c.stepS;
    c.m += WR_ROWS;
    if (T > 0) {
        c.tq += WR_ROWS;
        while (c.tq >= T) c.tq -= T;
    }
}
template <typename V> struct wr_raw;
template <> struct wr_raw<float4> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct wr_raw<uint2> { typedef unsigned type __attribute__((ext_vector_type(2))); };
template <typename V>
__device__ __forceinline__ void wr_fetch(const WrCursor& c, int rend, int shift, int T, V (&r)[WR_NI]) {
#pragma unroll
    for (int i = 0; i < WR_NI; ++i) {
        bool ok = c.colok && (c.m + 8 * i < rend);
        if (T > 0) {                      // conv tap: the shifted frame must stay inside its sequence
            int t = c.tq + 8 * i;
            while (t >= T) t -= T;
            ok = ok && ((unsigned)(t + shift) < (unsigned)T);
        }
        typedef typename wr_raw<V>::type R;
        R raw = {};
        if (ok) raw = *(__attribute__((address_space(1))) const R*)(c.ptr + i * c.step8);
        r[i] = __builtin_bit_cast(V, raw);
    }
}
__device__ __forceinline__ void wr_stash_h(uint16_t* S, int tid, const uint2 (&r)[WR_NI]) {
#pragma unroll
    for (int i = 0; i < WR_NI; ++i) {
        int id = tid + 256 * i;
        int row = id >> 5, c4 = id & 31;
        *reinterpret_cast<uint2*>(S + row * WR_RLD + c4 * 4) = r[i];
    }
}
__device__ __forceinline__ void wr_stash(uint16_t* S, int tid, const float4 (&r)[WR_NI]) {
#pragma unroll
    for (int i = 0; i < WR_NI; ++i) {
        int id = tid + 256 * i;
        int row = id >> 5, c4 = id & 31;
        *reinterpret_cast<uint2*>(S + row * WR_RLD + c4 * 4) = pack4(r[i]);
    }
}
template <bool XH, bool YH>
__global__ __launch_bounds__(256) void wgrad_rows_kernel(WRowsP p) {
    constexpr int ROWIMG = WR_ROWS * WR_RLD;             // uint16 elements
    __shared__ __attribute__((aligned(16))) uint16_t smem[2][2][ROWIMG];      // [buffer][X | dY]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, kg = lane >> 5;
    // 1-D grid, XCD-aware: every tile of one row split lands on one XCD, so the split's rows of X and dY
    // are fetched from HBM once and shared through that XCD's L2.
    const int ntiles = p.tiles_k * p.tiles_n, nsplit = gridDim.x / ntiles;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int bid = lid % ntiles, zsplit = lid / ntiles;
    const int tn = bid % p.tiles_n, tk = bid / p.tiles_n;
    const int k0 = tk * 128, n0 = tn * 128;
    const int mbeg = zsplit * p.k_per_split;
    const int mend = min(p.M, mbeg + p.k_per_split);
    // conv: this K_in tile belongs to tap j; read X columns [k0 - j*Cin, +128) shifted by j - pad frames
    int tap = 0, xcol0 = k0, shift = 0, Tw = 0, xcols = p.K;
    if (p.taps > 1) {
        tap = k0 / p.Cin; xcol0 = k0 - tap * p.Cin; shift = tap - p.pad; Tw = p.T; xcols = p.Cin;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 rx[WR_NI], ry[WR_NI];       // fp32-source prefetch registers
    uint2 rxh[WR_NI], ryh[WR_NI];      // bf16-source prefetch registers (the unused set is dead code)
    const bool second = p.X2 != nullptr && k0 >= p.K1;                    // (workgroup-uniform)
    if (p.X2 != nullptr) { xcols = second ? p.K - p.K1 : p.K1; xcol0 = second ? k0 - p.K1 : k0; }
    WrCursor cx = wr_cursor(second ? p.X2 : p.X, XH ? 2 : 4, second ? p.ldx2 : p.ldx, xcols, mbeg, xcol0, shift, Tw, tid);
    WrCursor cy = wr_cursor(p.DY, YH ? 2 : 4, p.lddy, p.N, mbeg, n0, 0, 0, tid);
    auto fetch_tiles = [&]() {
        if constexpr (XH) wr_fetch(cx, mend, shift, Tw, rxh); else wr_fetch(cx, mend, shift, Tw, rx);
        if constexpr (YH) wr_fetch(cy, mend, 0, 0, ryh); else wr_fetch(cy, mend, 0, 0, ry);
        wr_advance(cx, Tw);
        wr_advance(cy, 0);
    };
    if (mbeg < mend) fetch_tiles();
    // bias gradient: wave row 0 of the tk == 0 tiles multiplies an all-ones A fragment with its dY fragments
    const bool do_colsum = (p.colsum != nullptr) && (tk == 0) && (wr == 0);
    f32x16 cs[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) cs[j][r] = 0.f;
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
    // transposing-read lane address: 16-lane group g = lane >> 4 covers rows (g >> 1) * 8 + [0, 4) (second read
    // +4), columns (g & 1) * 16 + [0, 16); lane t of the group supplies row t >> 2, columns 4 * (t & 3)..+3
    const int lane_off = ((lane >> 5) * 8 + ((lane & 15) >> 2)) * WR_RLD + ((lane >> 4) & 1) * 16 + 4 * (lane & 3);
    // two instantiations of the loop: a data-dependent `if (do_colsum)` around MFMAs inside it makes the
    // compiler shuttle every accumulator between AGPRs and VGPRs each step
    auto run = [&](auto colsum_tag) {
        constexpr bool kColsum = decltype(colsum_tag)::value;
        int buf = 0;
        for (int m0 = mbeg; m0 < mend; m0 += WR_ROWS) {
            uint16_t* Xi = smem[buf][0];
            uint16_t* Yi = smem[buf][1];
            if constexpr (XH) wr_stash_h(Xi, tid, rxh); else wr_stash(Xi, tid, rx);
            if constexpr (YH) wr_stash_h(Yi, tid, ryh); else wr_stash(Yi, tid, ry);
            __syncthreads();
            if (m0 + WR_ROWS < mend) fetch_tiles();
            const uint16_t* xa = Xi + lane_off + wr * 64;
            const uint16_t* yb = Yi + lane_off + wc * 64;
    #pragma unroll
            for (int ks = 0; ks < WR_ROWS / 16; ++ks) {
                bf16x8 a0 = wr_tr8(xa + ks * 16 * WR_RLD);
                bf16x8 a1 = wr_tr8(xa + ks * 16 * WR_RLD + 32);
                bf16x8 b0 = wr_tr8(yb + ks * 16 * WR_RLD);
                bf16x8 b1 = wr_tr8(yb + ks * 16 * WR_RLD + 32);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
                if constexpr (kColsum) {
                    cs[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, b0, cs[0], 0, 0, 0);
                    cs[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, b1, cs[1], 0, 0, 0);
                }
            }
            // double-buffered images: the next stash writes the other buffer, whose last readers finished
            // before they arrived at this iteration's barrier
            buf ^= 1;
        }
    };
    if (do_colsum) run(std::true_type{}); else run(std::false_type{});
    const bool split = nsplit > 1;
    if (do_colsum && kg == 0) {            // every row of cs[j] holds the column sums: take row 0
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int col = n0 + wc * 64 + j * 32 + l31;
            if (col < p.N) {
                if (split) p.colsum_ws[(long)zsplit * p.N + col] = cs[j][0];
                else p.colsum[col] = cs[j][0];
            }
        }
    }
    float* Cb = split ? p.ws + (long)zsplit * p.K * p.N : p.dW;
    const long ldc = split ? (long)p.N : p.lddw;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int col = n0 + wc * 64 + j * 32 + l31;
        if (col >= p.N) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = k0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                if (row < p.K) Cb[(long)row * ldc + col] = acc[i][j][r];
            }
        }
    }
}

// =================================================================================================
// LDS-DMA variant of the row-major wgrad (both operands bf16, no conv window, rows % 32 == 0,
// K_in % 128 == 0, N % 128 == 0): the 32-row tiles of X and dY arrive by global_load_lds_dwordx4 into a
// WD_STAGES-stage ring (four: 64 KB), WD_STAGES - 1 steps in flight while one is multiplied.  The register-staged kernel
// above completes one step per memory round trip (its 24 KB of loads are issued after the barrier and
// needed at the next one: ~1.3 us per step measured, 57 steps for the FFN weights); here a step costs its
// MFMAs and transposing reads.  The DMA image is dense (256-byte rows), which would put the 4 rows a
// ds_read_b64_tr_b16 group touches on the same banks; the 16-byte chunk c of row r is therefore stored at
// chunk position c ^ ((r & 3) << 2) (load side, free), and the reads XOR their 8-byte unit index with
// (r & 3) << 3: the 32 lanes of a read phase then cover 32 distinct units = all 64 banks.
// vmcnt bookkeeping: a wave issues exactly 4 DMA instructions per step and nothing else on the vector
// memory counter inside the loop, so "vmcnt(4 j)" = everything but the newest j steps has landed.
// =================================================================================================
// (round 6: FOUR stages, 64 KB - three steps in flight: 66 -> 59 us alone at 28 800 x 1024 x 256, 4.52 -> 4.49 ms per step; a
// fifth stage leaves one workgroup per CU and gives it all back: profiles/r06_wgrad_ring_ab.txt.  -DWD_STAGES=3/5: variants)
#ifndef WD_STAGES
#define WD_STAGES 4
#endif
#define WD_STAGE_BYTES (2 * WR_ROWS * 256)        // X image then dY image, [32 rows][128 bf16]

__device__ __forceinline__ bf16x8 wd_tr8(const unsigned char* img, int row, int unit) {
    typedef __attribute__((address_space(3))) s16x4* lds_ptr;
    // rows row..row+3 / row+4..row+7 share (row & 3) with `row` only if row % 4 == 0 for the +4 read: they do
    // not - the swizzle term is per supplied row, and this lane always supplies rows = row (mod 4)
    const int swz = (row & 3) << 3;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(img + row * 256 + ((unit ^ swz) << 3)));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(img + (row + 4) * 256 + ((unit ^ swz) << 3)));
    s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(256, 2) void wgrad_dma_kernel(WRowsP p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[WD_STAGES * WD_STAGE_BYTES];
    typedef __attribute__((address_space(1))) const void* gptr;
    typedef __attribute__((address_space(3))) void* lptr;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, kg = lane >> 5;
    const int ntiles = p.tiles_k * p.tiles_n, nsplit = gridDim.x / ntiles;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int bid = lid % ntiles, zsplit = lid / ntiles;
    const int tn = bid % p.tiles_n, tk = bid / p.tiles_n;
    const int k0 = tk * 128, n0 = tn * 128;
    const int mbeg = zsplit * p.k_per_split;
    const int mend = min(p.M, mbeg + p.k_per_split);
    // Round 6: the row count need not be a multiple of 32 any more (the reference's bucketed batches: B x T is anything) -
    // the LAST step of the last split loads its missing rows from the last valid row (clamped addresses: finite data) and
    // zeroes their X fragments (and the ones of the bias gradient's all-ones operand), so they add exactly nothing.  Such
    // row counts used to fall back to the register-staged kernel: 37 us per launch at ~12 k rows against ~15.
    const int nsteps = (mend - mbeg + WR_ROWS - 1) / WR_ROWS;
    const int tail = (mend - mbeg) - (nsteps - 1) * WR_ROWS;         // live rows of the last step: 1 .. 32

    f32x16 acc[2][2], cs[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) cs[j][r] = 0.f;
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
    const bool do_colsum = (p.colsum != nullptr) && (tk == 0) && (wr == 0);

    const bool second = p.X2 != nullptr && k0 >= p.K1;                    // (workgroup-uniform) which X this K tile reads
    const uint16_t* X = (const uint16_t*)(second ? p.X2 : p.X);
    const long ldx = second ? p.ldx2 : p.ldx;
    int xk0 = second ? k0 - p.K1 : k0;
    if (p.taps > 1) {
        // SHIFTED-ROWS taps (conv_T == 0: the zero-margin layout of a 'same' Conv1D, ops.ConvStackFn): K_in rows
        // [j Cin, (j + 1) Cin) of dW come from X shifted down by j rows - every tap of the conv in ONE launch (it was a
        // launch and a slab reduction per tap).  Cin % 128 == 0, so a K tile never straddles two taps.
        const int tap = k0 / p.Cin;
        X += (long)tap * ldx;
        xk0 = k0 - tap * p.Cin;
    }
    const uint16_t* DY = (const uint16_t*)p.DY;
    // DMA: a wave instruction = 4 rows x 16 chunks of 16 bytes; wave w owns rows [8w, 8w + 8) of a step
    const int drow = lane >> 4, dpos = lane & 15;
    auto issue = [&](int step, int stage) {
        unsigned char* Xi = smem + stage * WD_STAGE_BYTES;
        unsigned char* Yi = Xi + WR_ROWS * 256;
        const long m0 = mbeg + (long)step * WR_ROWS;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wave * 8 + i * 4 + drow;
            const int c = dpos ^ ((row & 3) << 2);
            lds_dma16(X + min(m0 + row, (long)mend - 1) * ldx + xk0 + c * 8, lds_offset(Xi + (wave * 8 + i * 4) * 256));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wave * 8 + i * 4 + drow;
            const int c = dpos ^ ((row & 3) << 2);
            lds_dma16(DY + min(m0 + row, (long)mend - 1) * p.lddy + n0 + c * 8, lds_offset(Yi + (wave * 8 + i * 4) * 256));
        }
    };
    // transposing-read lane geometry (see wgrad_rows_kernel): rows (lane>>5)*8 + ((lane&15)>>2) (+4),
    // 8-byte unit = column / 4
    const int trow = (lane >> 5) * 8 + ((lane & 15) >> 2);
    const int tunit = ((lane >> 4) & 1) * 4 + (lane & 3);
#pragma unroll
    for (int s_ = 0; s_ < WD_STAGES - 1; ++s_)
        if (s_ < nsteps) issue(s_, s_);
    auto run = [&](auto colsum_tag) {
        constexpr bool kColsum = decltype(colsum_tag)::value;
        for (int s_ = 0; s_ < nsteps; ++s_) {
            // step s_ has landed once at most the newest step's 4 DMA instructions are still outstanding
            // (s_waitcnt imm: vmcnt = bits 3:0 (+ 15:14), expcnt / lgkmcnt untouched)
            if (WD_STAGES > 4 && s_ + 3 < nsteps) __builtin_amdgcn_s_waitcnt(0xF7C);   // vmcnt(12): three newer steps may be in flight
            else if (WD_STAGES > 3 && s_ + 2 < nsteps) __builtin_amdgcn_s_waitcnt(0xF78);   // vmcnt(8): two
            else if (s_ + 1 < nsteps) __builtin_amdgcn_s_waitcnt(0xF74);      // vmcnt(4)
            else __builtin_amdgcn_s_waitcnt(0xF70);                      // vmcnt(0)
            // not __syncthreads(): its fence is vmcnt(0), which would also wait for the step that was just put in
            // flight.  Everybody's pieces of this step have landed; stage (s_-1)%3 is retired (its reads returned)
            lds_stage_barrier();
            if (s_ + WD_STAGES - 1 < nsteps) issue(s_ + WD_STAGES - 1, (s_ + WD_STAGES - 1) % WD_STAGES);
            const unsigned char* Xi = smem + (s_ % WD_STAGES) * WD_STAGE_BYTES;
            const unsigned char* Yi = Xi + WR_ROWS * 256;
#pragma unroll
            for (int ks = 0; ks < WR_ROWS / 16; ++ks) {
                const int row = ks * 16 + trow;
                bf16x8 a0 = wd_tr8(Xi, row, wr * 16 + tunit);
                bf16x8 a1 = wd_tr8(Xi, row, wr * 16 + 8 + tunit);
                bf16x8 b0 = wd_tr8(Yi, row, wc * 16 + tunit);
                bf16x8 b1 = wd_tr8(Yi, row, wc * 16 + 8 + tunit);
                bf16x8 one8 = ones;
                if (tail < WR_ROWS && s_ + 1 == nsteps) {          // (workgroup-uniform, the last step of a ragged split only)
                    // element e of a fragment is row 16 ks + 8 kg + e of the step (the transposing read hands a lane the
                    // 4 + 4 rows of its column): rows past the tail multiply as zeros
                    const int live = tail - (ks * 16 + kg * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (e >= live) { a0[e] = (__bf16)0.f; a1[e] = (__bf16)0.f; one8[e] = (__bf16)0.f; }
                }
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
                if constexpr (kColsum) {
                    cs[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(one8, b0, cs[0], 0, 0, 0);
                    cs[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(one8, b1, cs[1], 0, 0, 0);
                }
            }
        }
    };
    if (do_colsum) run(std::true_type{}); else run(std::false_type{});
    const bool split = nsplit > 1;
    if (do_colsum && kg == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int col = n0 + wc * 64 + j * 32 + l31;
            if (split) p.colsum_ws[(long)zsplit * p.N + col] = cs[j][0];
            else p.colsum[col] = cs[j][0];
        }
    }
    float* Cb = split ? p.ws + (long)zsplit * p.K * p.N : p.dW;
    const long ldc = split ? (long)p.N : p.lddw;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int col = n0 + wc * 64 + j * 32 + l31;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = k0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                Cb[(long)row * ldc + col] = acc[i][j][r];
            }
        }
    }
}

__global__ void hsplit_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, long ldo,
                                     int M, int N, int splits, const float* __restrict__ cs_ws,
                                     float* __restrict__ cs_out) {
    long n = (long)M * N;
    long total = n + (cs_out ? N : 0);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        if (i < n) {
            for (int z = 0; z < splits; ++z) s += ws[(long)z * n + i];
            int r = (int)(i / N), c = (int)(i - (long)r * N);
            out[(long)r * ldo + c] = s;
        } else {
            long c = i - n;
            for (int z = 0; z < splits; ++z) s += cs_ws[(long)z * N + c];
            cs_out[c] = s;
        }
    }
}

// Same reduction, 4 elements per lane and the splits spread over the block's 4 waves: wave w sums the
// contiguous split range [S*w/4, S*(w+1)/4) in order with 8 loads in flight, wave 0 then adds the four
// partial sums in wave order - a fixed association, so the result is run-to-run deterministic.  The
// one-element-per-thread loop above is a serial chain of S dependent-latency loads on a handful of
// waves (15-25 us per weight); this one keeps ~S/4 x 16 B per lane in flight on 4x the waves.
__device__ __forceinline__ void hsplit_reduce4_body(const float* __restrict__ ws, float* __restrict__ out,
                                                    long ldo, int M, int N, int splits,
                                                    const float* __restrict__ cs_ws,
                                                    float* __restrict__ cs_out, float4 (*part)[64]) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const long n = (long)M * N, nq = n >> 2, ncs = cs_out ? (N >> 2) : 0;
    if (blockIdx.x * 64L >= nq + ncs) return;               // (block-uniform: the batched launch sizes its grid for the largest job)
    const long q = blockIdx.x * 64L + l;
    const bool ok = q < nq + ncs;
    const bool main_part = q < nq;
    const float* base = main_part ? ws + q * 4 : cs_ws + (q - nq) * 4;
    const long stride = main_part ? n : (long)N;
    const int z0 = splits * w / 4, z1 = splits * (w + 1) / 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) {
        int z = z0;
        for (; z + 8 <= z1; z += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(base + (long)(z + u) * stride);
#pragma unroll
            for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        for (; z < z1; ++z) {
            float4 v = *reinterpret_cast<const float4*>(base + (long)z * stride);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    if (w > 0) part[w - 1][l] = s;
    __syncthreads();
    if (w == 0 && ok) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            float4 v = part[u][l];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        if (main_part) {
            long i = q * 4;
            int r = (int)(i / N), c = (int)(i - (long)r * N);
            *reinterpret_cast<float4*>(out + (long)r * ldo + c) = s;
        } else {
            *reinterpret_cast<float4*>(cs_out + (q - nq) * 4) = s;
        }
    }
}
__global__ __launch_bounds__(256) void hsplit_reduce4_kernel(const float* __restrict__ ws, float* __restrict__ out,
                                                             long ldo, int M, int N, int splits,
                                                             const float* __restrict__ cs_ws,
                                                             float* __restrict__ cs_out) {
    __shared__ float4 part[3][64];
    hsplit_reduce4_body(ws, out, ldo, M, N, splits, cs_ws, cs_out, part);
}
// the slab reductions of several weight gradients in ONE launch (blockIdx.y = job): a dense block's backward leaves five
// of them on the weight-gradient stream, each a 4-8 us launch that cannot fill the GPU on its own
struct WJobs { ttsmi_wgrad_job j[TTSMI_WGRAD_MAX_JOBS]; };
__global__ __launch_bounds__(256) void hsplit_reduce4_jobs_kernel(WJobs J) {
    __shared__ float4 part[3][64];
    const ttsmi_wgrad_job& b = J.j[blockIdx.y];
    hsplit_reduce4_body(b.ws, b.dw, b.lddw, b.kin, b.n, b.splits, b.cs_ws, b.db, part);
}

static bool hsplit_vec_ok(const float* ws, const float* dw, long lddw, int n, const float* db) {
    return (n % 4 == 0) && (lddw % 4 == 0) && (((uintptr_t)dw & 15) == 0) && (((uintptr_t)ws & 15) == 0) &&
           (!db || ((uintptr_t)db & 15) == 0);
}

static void hsplit_reduce_launch(hipStream_t st, const float* ws, float* dw, long lddw, int kin, int n, int splits,
                                 const float* cs_ws, float* db) {
    const bool vec = (n % 4 == 0) && (lddw % 4 == 0) && (((uintptr_t)dw & 15) == 0) &&
                     (((uintptr_t)ws & 15) == 0) && (!db || ((uintptr_t)db & 15) == 0);
    long tot = (long)kin * n;
    if (vec) {
        long quads = tot / 4 + (db ? n / 4 : 0);
        hipLaunchKernelGGL(hsplit_reduce4_kernel, dim3((unsigned)((quads + 63) / 64)), dim3(256), 0, st, ws, dw, lddw,
                           kin, n, splits, cs_ws, db);
    } else {
        int blocks = (int)((tot + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(hsplit_reduce_kernel, dim3(blocks), dim3(256), 0, st, ws, dw, lddw, kin, n, splits, cs_ws,
                           db);
    }
}

// ---- fp32 [R, C] -> bf16 transposed [taps*C, Rp] (Rp = ldd >= R, tail zero-filled) -------------
// dst[j*C + c][r] = src[r + j - pad][c] when frame (r % T) + j - pad stays inside its sequence,
// else 0.  taps == 1: plain cast-transpose (weights W -> W^T, activations x -> x^T).
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ src, long lds_,
                                                             uint16_t* __restrict__ dst, long ldd,
                                                             int R, int C, int taps, int T, int pad) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64, j = blockIdx.z;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int rr = ty; rr < 64; rr += 4) {
        int r = r0 + rr, c = c0 + tx;
        float v = 0.f;
        if (r < R && c < C) {
            bool ok = true;
            long sr = r;
            if (taps > 1) {
                int tt = (r % T) + j - pad;
                ok = (tt >= 0) && (tt < T);
                sr = (long)r + j - pad;
            }
            if (ok) v = src[sr * lds_ + c];
        }
        tile[rr][tx] = v;
    }
    __syncthreads();
    for (int cc = ty; cc < 64; cc += 4) {
        int c = c0 + cc, r = r0 + tx;
        if (c < C && r < ldd) {
            __bf16 h = (__bf16)tile[tx][cc];
            dst[((long)j * C + c) * ldd + r] = *reinterpret_cast<uint16_t*>(&h);
        }
    }
}

// Batched form for the per-step weight-shadow refresh: ONE launch transposes every GEMM weight
// (descriptor table resident on the device; block -> descriptor by tile prefix).
__global__ __launch_bounds__(256) void cast_transpose_batched_kernel(const ttsmi_transpose_desc* __restrict__ desc,
                                                                     int n_desc) {
    __shared__ float tile[64][65];
    __shared__ int which;
    if (threadIdx.x == 0) {
        int i = 0;
        while (i + 1 < n_desc && desc[i + 1].tile_start <= (int)blockIdx.x) ++i;
        which = i;
    }
    __syncthreads();
    const ttsmi_transpose_desc d = desc[which];
    const int t = blockIdx.x - d.tile_start;
    const int r0 = (t % d.tiles_r) * 64, c0 = (t / d.tiles_r) * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int rr = ty; rr < 64; rr += 4) {
        int r = r0 + rr, c = c0 + tx;
        tile[rr][tx] = (r < d.R && c < d.C) ? d.src[(long)r * d.ld_src + c] : 0.f;
    }
    __syncthreads();
    for (int cc = ty; cc < 64; cc += 4) {
        int c = c0 + cc, r = r0 + tx;
        if (c < d.C && r < d.ld_dst) {
            __bf16 h = (__bf16)tile[tx][cc];
            d.dst[(long)c * d.ld_dst + r] = *reinterpret_cast<uint16_t*>(&h);
        }
    }
}

// conv weight [k, Cin, Cout] fp32 -> dgrad operand bf16 [Cin][k*Cout] with flipped taps:
// dst[ci][j'*Cout + co] = w[k-1-j'][ci][co]
__global__ __launch_bounds__(256) void conv_wdgrad_layout_kernel(const float* __restrict__ w,
                                                                 uint16_t* __restrict__ dst, int k,
                                                                 int Cin, int Cout, int CoutP) {
    long n = (long)k * Cin * CoutP;           // CoutP >= Cout: per-tap column count of dst, zero filled
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        int co = (int)(i % CoutP);
        long t = i / CoutP;
        int jp = (int)(t % k), ci = (int)(t / k);
        __bf16 h = (__bf16)(co < Cout ? w[((long)(k - 1 - jp) * Cin + ci) * Cout + co] : 0.f);
        dst[i] = *reinterpret_cast<uint16_t*>(&h);
    }
}

// ---- host ---------------------------------------------------------------------------------------
static bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

static void hinit(HGemmP& p) {
    memset(&p, 0, sizeof(p));
    p.a_taps = 1;
}

static int hgemm_bm(bool a_f32, int M, int N, int splits) {
    // tuning knob (measurement only): TTSMI_HGEMM_BM=64|128 overrides the default tile height
    TTSMI_KNOB(forced, "TTSMI_HGEMM_BM", 0);
    if (forced == 64 || forced == 128) return forced;
    if (a_f32) return 64;
    // bf16 A: 128-row tiles halve the B re-reads, but a launch that cannot give every CU a workgroup
    // (encoder-side M = 6400) is latency bound per workgroup: take the 64-row tile, twice the workgroups
    const long wgs128 = (long)ttsmi_cdiv(M, 128) * ttsmi_cdiv(N, HBN_) * splits;
    return wgs128 < 256 ? 64 : 128;
}

static int hlaunch(HGemmP& p, bool a_f32, int splits, hipStream_t st, const char* name) {
    // 128x256 tiles (A re-read N/256 instead of N/128 times) exist as a measurement knob only
    // (TTSMI_HGEMM_BN=256): on the decoder shapes they are 0-35 % SLOWER than 128x128 / 64x128
    // (tools/probe_gemm_variants.py) - 252 VGPRs halve the resident workgroups, and these short-K GEMMs
    // live on latency hiding across workgroups, not on L2 re-read volume.
    TTSMI_KNOB(forced_bn, "TTSMI_HGEMM_BN", 0);
    const bool wide = forced_bn == 256 && !a_f32 && splits == 1 && p.N % 256 == 0;
    if (wide) {
        p.tiles_m = ttsmi_cdiv(p.M, 128);
        p.tiles_n = ttsmi_cdiv(p.N, 256);
        dim3 grid(p.tiles_m * p.tiles_n, 1, 1);
        ttsmi_note_kernel("gemm_bf16_kernel<false, 128, 256>");
        hipLaunchKernelGGL((gemm_bf16_kernel<false, 128, 256>), grid, dim3(256), 0, st, p);
        TTSMI_CHECK_LAUNCH(name);
        return TTSMI_OK;
    }
    // LDS-DMA persistent kernel.  TTSMI_HGEMM_DMA: 0 = never, 1 = every eligible decoder-size launch, 2 = every eligible
    // launch, 3 (default) = decoder-size launches with K >= 512.  Round-2 per-shape A/B (tools/kbench.py, M = 28 800):
    // K = 512 / 768 / 1024 with N = 256 run 19.6 / 27.4 / 31.5-38.5 us against 24.2 / 30.9 / 39.8-45.5 us register-staged
    // (a deeper k-loop amortises the persistent pipeline), the K = 256 shapes tie or lose (4 k-steps per tile: the
    // epilogue dominates either way) and the M = 6 400 launches lose (too few tiles per workgroup).
    TTSMI_KNOB(use_dma, "TTSMI_HGEMM_DMA", 3);
    if (use_dma && !a_f32 && splits == 1 && p.colsum == nullptr && p.a_taps == 1 && p.K % HBK_ == 0 && p.lda % 8 == 0 && p.ldb % 8 == 0 &&
        (use_dma != 3 || p.K >= 512) &&
        (p.A2 == nullptr || (p.K1 % HBK_ == 0 && p.lda2 % 8 == 0 && al16(p.A2)))) {
        p.tiles_m = ttsmi_cdiv(p.M, DBM);
        p.tiles_n = ttsmi_cdiv(p.N, DBN);
        int nw = p.tiles_m * p.tiles_n;
        // large launches: 256 x 256 tiles (TTSMI_HGEMM_T256: 0 = never, 1 (default) = N >= 512, K >= 768 and at least two
        // tiles per CU; the benchmark's dense blocks never qualify - their N is 256)
        TTSMI_KNOB(t256, "TTSMI_HGEMM_T256", 1);
        const int nw256 = ttsmi_cdiv(p.M, D2BM) * ttsmi_cdiv(p.N, D2BN);
        if (t256 && p.N >= 512 && p.K >= 768 && nw256 >= 512) {
            p.tiles_m = ttsmi_cdiv(p.M, D2BM);
            p.tiles_n = ttsmi_cdiv(p.N, D2BN);
            ttsmi_note_kernel("gemm_bf16_dma256_kernel");
            TTSMI_KNOB(t256_burst, "TTSMI_HGEMM_T256_BURST", 0);
            p.dma_burst = t256_burst;
            hipLaunchKernelGGL(gemm_bf16_dma256_kernel, dim3(256), dim3(512), 0, st, p);
            TTSMI_CHECK_LAUNCH(name);
            return TTSMI_OK;
        }
        if (nw >= (use_dma == 2 ? 1 : 192)) {         // under-filled launches keep the 64-row register-staged tiles
            if (nw > 512) nw = 512;
            ttsmi_note_kernel("gemm_bf16_dma_kernel");
            hipLaunchKernelGGL(gemm_bf16_dma_kernel, dim3(nw), dim3(256), 0, st, p);
            TTSMI_CHECK_LAUNCH(name);
            return TTSMI_OK;
        }
    }
    // Under-filled launches (fewer 64 x 128 workgroups than 1.5 per CU): the deep-ring 64 x 64 kernel.  TTSMI_HGEMM_DEEP=0
    // keeps the register-staged kernel (A/B knob).
    TTSMI_KNOB(use_deep, "TTSMI_HGEMM_DEEP", 1);
    // Round-2 A/B (tools/kbench.py --only gemm-small, M = 6400 / 2304 / 400): K = 1024 runs 13.2 / 10.3 / 9.9 us against
    // 20.7 / 16.0 / 15.9 us register-staged, K = 768 15.5 / 10.1 / 9.8 against 17.1 / 13.4 / 13.2; the K = 256 shapes (a
    // single round trip either way) tie or lose, so they stay where they were.  Batch-1 predict: 1.14 -> 0.95 ms.
    if (use_deep && !a_f32 && splits == 1 && p.colsum == nullptr && p.a_taps == 1 && p.K % HBK_ == 0 && p.lda % 8 == 0 &&
        (p.K >= 512 || use_deep > 2) &&
        p.ldb % 8 == 0 && al16(p.A) && al16(p.B) && (long)ttsmi_cdiv(p.M, 64) * ttsmi_cdiv(p.N, HBN_) < (use_deep > 1 ? (1L << 40) : 384) &&
        (p.A2 == nullptr || (p.K1 % HBK_ == 0 && p.lda2 % 8 == 0 && al16(p.A2)))) {
        p.tiles_m = ttsmi_cdiv(p.M, SBM);
        p.tiles_n = ttsmi_cdiv(p.N, SBN);
        ttsmi_note_kernel("gemm_bf16_deep_kernel<4>");
        hipLaunchKernelGGL(gemm_bf16_deep_kernel<4>, dim3(p.tiles_m * p.tiles_n), dim3(256), 0, st, p);
        TTSMI_CHECK_LAUNCH(name);
        return TTSMI_OK;
    }
    const int bm = hgemm_bm(a_f32, p.M, p.N, splits);
    p.tiles_m = ttsmi_cdiv(p.M, bm);
    p.tiles_n = ttsmi_cdiv(p.N, HBN_);
    dim3 grid(p.tiles_m * p.tiles_n, 1, splits);
    TTSMI_KNOB(occ4, "TTSMI_HGEMM_OCC4", 0);      // measurement knob, default off: the 4-workgroups-per-CU build (see the kernel's comment)
    const auto fits32 = [](long rows, long ld) { return rows * ld * 2 < (1L << 32); };
    if (occ4 && !a_f32 && bm == 128 && fits32(p.M, p.lda) && fits32(p.N, p.ldb) &&
        (p.A2 == nullptr || fits32(p.M, p.lda2))) {
        ttsmi_note_kernel("gemm_bf16_kernel<false, 128, 128, 4>");
        hipLaunchKernelGGL((gemm_bf16_kernel<false, 128, HBN_, 4>), grid, dim3(256), 0, st, p);
        TTSMI_CHECK_LAUNCH(name);
        return TTSMI_OK;
    }
    ttsmi_note_kernel(a_f32 ? (bm == 64 ? "gemm_bf16_kernel<true, 64>" : "gemm_bf16_kernel<true, 128>")
                            : (bm == 64 ? "gemm_bf16_kernel<false, 64>" : "gemm_bf16_kernel<false, 128>"));
    if (a_f32) {
        if (bm == 64) hipLaunchKernelGGL((gemm_bf16_kernel<true, 64>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemm_bf16_kernel<true, 128>), grid, dim3(256), 0, st, p);
    } else {
        if (bm == 64) hipLaunchKernelGGL((gemm_bf16_kernel<false, 64>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemm_bf16_kernel<false, 128>), grid, dim3(256), 0, st, p);
    }
    TTSMI_CHECK_LAUNCH(name);
    return TTSMI_OK;
}

static int hpick_splits(long rows, int tiles) {
    // ~1 workgroup per CU: wgrad overlaps the main stream, and fewer splits = less slab traffic.
    // Splits come in multiples of 8 so that the XCD-aware 1-D grid gives every XCD whole splits.
    // (big weights - >= 64 output tiles, the conv blocks' [3 x 384, 1536] / [3 x 1536, 384] - are 100 GFLOP launches that
    // make the second stream as long as the main one: there a full-GPU launch pays, 27.3 -> 26.2 ms per ref-default step;
    // the dense blocks' 4-16-tile weights stay at 128: 5.42 ms per configs[1] step against 5.50 at 192 / 5.51 at 256)
    TTSMI_KNOB(target_env, "TTSMI_WGRAD_WGS", 0);
    const int target = target_env >= 8 ? target_env : (tiles >= 64 ? 256 : 128);
    int want = (target + tiles - 1) / tiles;
    if (want > 8) want = (want + 7) / 8 * 8;
    int maxs = (int)((rows + 511) / 512);
    int s = want < maxs ? want : maxs;
    if (s > 128) s = 128;
    if (s < 1) s = 1;
    return s;
}

extern "C" {

int ttsmi_hgemm_tn(const void* a, int a_is_f32, int64_t lda, const void* a2, int64_t lda2, int K1,
                   const uint16_t* b, int64_t ldb, const float* bias, const float* relu_src,
                   int64_t ld_relu, void* c, int64_t ldc, int M, int N, int K, int flags,
                   int conv_taps, int conv_T, int conv_C, int conv_pad, ttsmi_stream_t stream) {
    const int relu = flags & TTSMI_GEMM_RELU, accumulate = flags & TTSMI_GEMM_ACCUMULATE;
    const int c_bf16 = flags & TTSMI_GEMM_OUT_BF16, mask_bf16 = flags & TTSMI_GEMM_MASK_BF16;
    TTSMI_CHECK_ARG(a && b && c, "hgemm_tn: null pointer");
    TTSMI_CHECK_ARG(M >= 0 && N > 0 && K > 0, "hgemm_tn: bad shape M=%d N=%d K=%d", M, N, K);
    if (M == 0) return TTSMI_OK;
    TTSMI_CHECK_ARG(al16(a) && al16(b) && (K % 8 == 0) && (ldb % 8 == 0),
                    "hgemm_tn: operands must be 16-byte aligned with K %% 8 == 0 (K=%d ldb=%ld)", K, (long)ldb);
    if (a_is_f32) TTSMI_CHECK_ARG(lda % 4 == 0, "hgemm_tn: lda %% 4 != 0");
    else TTSMI_CHECK_ARG(lda % 8 == 0 && conv_taps <= 1 && (!a2 || (K1 % HBK_ == 0 && lda2 % 8 == 0)),
                         "hgemm_tn: bf16 A needs lda %% 8 == 0, no conv window, A2 (bf16 too) with K1 %% 64 == 0");
    if (a2) TTSMI_CHECK_ARG(K1 > 0 && K1 < K && K1 % 8 == 0 && al16(a2) && lda2 % 4 == 0, "hgemm_tn: bad A2 segment");
    if (conv_taps > 1) TTSMI_CHECK_ARG(conv_C % 4 == 0 && conv_T > 0 && K == conv_taps * conv_C, "hgemm_tn: bad conv window");
    HGemmP p;
    hinit(p);
    p.A = a; p.lda = lda; p.A2 = a2; p.lda2 = lda2; p.K1 = K1;
    p.B = b; p.ldb = ldb; p.C = (float*)c; p.ldc = ldc; p.bias = bias; p.relu_src = relu_src; p.ld_relu = ld_relu;
    p.c_bf16 = c_bf16 ? 1 : 0; p.mask_bf16 = mask_bf16 ? 1 : 0;
    if (c_bf16)
        TTSMI_CHECK_ARG(!accumulate && N % 4 == 0 && ldc % 4 == 0 && ((((uintptr_t)c) & 7) == 0),
                        "hgemm_tn: bf16 output needs N %% 4 == 0, ldc %% 4 == 0, no accumulate");
    if (mask_bf16) TTSMI_CHECK_ARG(N % 4 == 0 && ld_relu % 4 == 0, "hgemm_tn: bf16 mask needs N %% 4 == 0");
    p.M = M; p.N = N; p.K = K; p.relu = relu ? 1 : 0; p.accumulate = accumulate ? 1 : 0; p.k_per_split = K;
    if (conv_taps > 1) { p.a_taps = conv_taps; p.T = conv_T; p.Cw = conv_C; p.pad = conv_pad; }
    // K = 256 projections without a second segment / accumulation: weight-stationary kernel (gemm_k256.hip); a ReLU' mask
    // is supported in its bf16 -> bf16 form (the masked FFN1 dgrad)
    const bool k256_mask_ok = !relu_src || (mask_bf16 && c_bf16 && !bias && !relu && ld_relu % 8 == 0 && al16(relu_src));
    TTSMI_KNOB(k256_mask, "TTSMI_HGEMM_K256_MASK", 1);          // TTSMI_HGEMM_K256_MASK=0: masked launches stay on the general kernel (A/B knob)
    if (!a_is_f32 && !a2 && k256_mask_ok && (!relu_src || k256_mask) && !accumulate && conv_taps <= 1 && lda % 8 == 0 && ldc % 4 == 0 &&
        (c_bf16 ? ldc % 8 == 0 : true) && al16(c) && ttsmi_hgemm_k256_eligible(M, N, K)) {
        ttsmi_hgemm_k256_launch((const uint16_t*)a, (long)lda, b, (long)ldb, bias, c, (long)ldc, M, N, relu ? 1 : 0,
                                c_bf16 ? 1 : 0, (const uint16_t*)relu_src, (long)ld_relu, (hipStream_t)stream);
        TTSMI_CHECK_LAUNCH("hgemm_tn(k256)");
        return TTSMI_OK;
    }
    return hlaunch(p, a_is_f32 != 0, 1, (hipStream_t)stream, "hgemm_tn");
}

size_t ttsmi_hgemm_wgrad_ws_bytes(int rows, int kin, int n) {
    int tiles = ttsmi_cdiv(kin, HBM_) * ttsmi_cdiv(n, HBN_);
    int splits = hpick_splits(rows, tiles);
    return (size_t)splits * ((size_t)kin * n + n) * sizeof(float) + 256;
}

/* dw[kin, n] = xT[kin, rows] . dyT[n, rows]^T, both bf16 with leading dimension ldt >= rows
 * (multiple of 8, tail zero-filled); db[n] = row sums of dyT (may be NULL). */
int ttsmi_hgemm_wgrad(const uint16_t* xT, const uint16_t* dyT, int64_t ldt, float* dw, int64_t lddw,
                      float* db, int rows, int kin, int n, void* ws, size_t ws_bytes,
                      ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(xT && dyT && dw, "hgemm_wgrad: null pointer");
    TTSMI_CHECK_ARG(rows > 0 && kin > 0 && n > 0 && ldt % 8 == 0 && ldt >= rows, "hgemm_wgrad: bad shape");
    TTSMI_CHECK_ARG(ws && ws_bytes >= ttsmi_hgemm_wgrad_ws_bytes(rows, kin, n), "hgemm_wgrad: workspace too small");
    TTSMI_CHECK_ARG(al16(xT) && al16(dyT), "hgemm_wgrad: operands must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    HGemmP p;
    hinit(p);
    p.A = xT; p.lda = ldt; p.B = dyT; p.ldb = ldt; p.C = dw; p.ldc = lddw;
    p.M = kin; p.N = n; p.K = (int)ldt;            // zero tail contributes nothing
    int tiles = ttsmi_cdiv(kin, HBM_) * ttsmi_cdiv(n, HBN_);
    int splits = hpick_splits(rows, tiles);
    int kps = ttsmi_cdiv(p.K, splits);
    kps = ((kps + HBK_ - 1) / HBK_) * HBK_;
    splits = ttsmi_cdiv(p.K, kps);
    p.k_per_split = kps;
    p.ws = (float*)ws; p.colsum = db; p.colsum_ws = p.ws + (size_t)splits * kin * n;
    int rc = hlaunch(p, false, splits, st, "hgemm_wgrad");
    if (rc) return rc;
    if (splits > 1) {
        hsplit_reduce_launch(st, p.ws, dw, (long)lddw, kin, n, splits, p.colsum_ws, db);
        TTSMI_CHECK_LAUNCH("hgemm_wgrad_reduce");
    }
    return TTSMI_OK;
}

size_t ttsmi_hgemm_wgrad_rows_ws_bytes(int rows, int kin, int n) { return ttsmi_hgemm_wgrad_ws_bytes(rows, kin, n); }

static int wgrad_rows_splits(int rows, int kin, int n, int* kps_out) {
    const int tiles = ttsmi_cdiv(kin, 128) * ttsmi_cdiv(n, 128);
    int splits = hpick_splits(rows, tiles);
    int kps = ttsmi_cdiv(rows, splits);
    kps = ((kps + WR_ROWS - 1) / WR_ROWS) * WR_ROWS;
    if (kps_out) *kps_out = kps;
    return ttsmi_cdiv(rows, kps);
}

// bytes of slab workspace this call really uses (ttsmi_hgemm_wgrad_ws_bytes is the shape-independent upper bound)
size_t ttsmi_hgemm_wgrad_rows_exact_bytes(int rows, int kin, int n, int has_db) {
    const size_t splits = (size_t)wgrad_rows_splits(rows, kin, n, nullptr);
    return ((splits * kin * n + (has_db ? splits * n : 0)) * sizeof(float) + 255) & ~(size_t)255;
}

static int wgrad_rows_impl(const void* x, int x_is_bf16, int64_t ldx, const void* dy, int dy_is_bf16, int64_t lddy,
                           float* dw, int64_t lddw, float* db, int rows, int kin, int n, int conv_taps,
                           int conv_T, int conv_C, int conv_pad, void* ws, size_t ws_bytes,
                           ttsmi_stream_t stream, ttsmi_wgrad_job* job, const void* x2 = nullptr, int64_t ldx2 = 0, int k1 = 0);

int ttsmi_hgemm_wgrad_rows(const void* x, int x_is_bf16, int64_t ldx, const void* dy, int dy_is_bf16, int64_t lddy,
                           float* dw, int64_t lddw, float* db, int rows, int kin, int n, int conv_taps,
                           int conv_T, int conv_C, int conv_pad, void* ws, size_t ws_bytes,
                           ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(ws && ws_bytes >= ttsmi_hgemm_wgrad_ws_bytes(rows, kin, n), "hgemm_wgrad_rows: workspace too small");
    return wgrad_rows_impl(x, x_is_bf16, ldx, dy, dy_is_bf16, lddy, dw, lddw, db, rows, kin, n, conv_taps, conv_T, conv_C,
                           conv_pad, ws, ws_bytes, stream, nullptr);
}

// The same launch with the slab reduction left to the caller: *job describes it (job->splits == 0: nothing is pending -
// a single split wrote dw directly, or the operands do not suit the vector reduce and it ran here).  ws only has to
// hold ttsmi_hgemm_wgrad_rows_exact_bytes, and must stay untouched until ttsmi_hgemm_wgrad_reduce_jobs has run.
int ttsmi_hgemm_wgrad_rows_deferred(const void* x, int x_is_bf16, int64_t ldx, const void* dy, int dy_is_bf16, int64_t lddy,
                                    float* dw, int64_t lddw, float* db, int rows, int kin, int n, void* ws, size_t ws_bytes,
                                    ttsmi_stream_t stream, ttsmi_wgrad_job* job) {
    TTSMI_CHECK_ARG(job, "hgemm_wgrad_rows_deferred: null job");
    TTSMI_CHECK_ARG(ws && ws_bytes >= ttsmi_hgemm_wgrad_rows_exact_bytes(rows, kin, n, db != nullptr),
                    "hgemm_wgrad_rows_deferred: workspace too small");
    return wgrad_rows_impl(x, x_is_bf16, ldx, dy, dy_is_bf16, lddy, dw, lddw, db, rows, kin, n, 1, 0, 0, 0, ws, ws_bytes,
                           stream, job);
}

// dW[0:K1] = x^T dy and dW[K1:kin] = x2^T dy (one dy, one db) as ONE launch: both halves of Wo = Dense(concat([q_in, ctx]))
int ttsmi_hgemm_wgrad_rows_deferred_dual(const void* x, int64_t ldx, const void* x2, int64_t ldx2, int k1, const void* dy,
                                         int64_t lddy, float* dw, int64_t lddw, float* db, int rows, int kin, int n, void* ws,
                                         size_t ws_bytes, ttsmi_stream_t stream, ttsmi_wgrad_job* job) {
    TTSMI_CHECK_ARG(job && x2, "hgemm_wgrad_rows_deferred_dual: null job / x2");
    TTSMI_CHECK_ARG(ws && ws_bytes >= ttsmi_hgemm_wgrad_rows_exact_bytes(rows, kin, n, db != nullptr),
                    "hgemm_wgrad_rows_deferred_dual: workspace too small");
    return wgrad_rows_impl(x, 1, ldx, dy, 1, lddy, dw, lddw, db, rows, kin, n, 1, 0, 0, 0, ws, ws_bytes, stream, job, x2, ldx2, k1);
}

int ttsmi_hgemm_wgrad_reduce_jobs(const ttsmi_wgrad_job* jobs, int njobs, ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(jobs && njobs >= 0 && njobs <= TTSMI_WGRAD_MAX_JOBS, "hgemm_wgrad_reduce_jobs: bad job list");
    if (njobs == 0) return TTSMI_OK;
    WJobs J;
    memset(&J, 0, sizeof(J));
    long blocks = 1;
    for (int i = 0; i < njobs; ++i) {
        J.j[i] = jobs[i];
        const long quads = (long)jobs[i].kin * jobs[i].n / 4 + (jobs[i].db ? jobs[i].n / 4 : 0);
        blocks = blocks > (quads + 63) / 64 ? blocks : (quads + 63) / 64;
    }
    hipLaunchKernelGGL(hsplit_reduce4_jobs_kernel, dim3((unsigned)blocks, njobs), dim3(256), 0, (hipStream_t)stream, J);
    TTSMI_CHECK_LAUNCH("hgemm_wgrad_reduce_jobs");
    return TTSMI_OK;
}

static int wgrad_rows_impl(const void* x, int x_is_bf16, int64_t ldx, const void* dy, int dy_is_bf16, int64_t lddy,
                           float* dw, int64_t lddw, float* db, int rows, int kin, int n, int conv_taps,
                           int conv_T, int conv_C, int conv_pad, void* ws, size_t ws_bytes,
                           ttsmi_stream_t stream, ttsmi_wgrad_job* job, const void* x2, int64_t ldx2, int k1) {
    (void)ws_bytes;
    if (x2) TTSMI_CHECK_ARG(conv_taps <= 1 && k1 > 0 && k1 < kin && k1 % 128 == 0 && al16(x2) && ldx2 % 8 == 0,
                            "hgemm_wgrad_rows: dual X needs no conv window, 0 < K1 < K with K1 %% 128 == 0, an aligned X2");
    if (job) job->splits = 0;
    // conv_taps > 1 with conv_T == 0: shifted-rows taps of the zero-margin layout (bf16 operands, the LDS-DMA kernel only);
    // x must stay readable for conv_taps - 1 rows past `rows`
    const bool shifted = conv_taps > 1 && conv_T == 0;
    TTSMI_CHECK_ARG(!(x_is_bf16 && conv_taps > 1) || shifted, "hgemm_wgrad_rows: conv needs an fp32 x");
    if (shifted)
        TTSMI_CHECK_ARG(!x2 && x_is_bf16 && dy_is_bf16 && conv_pad == 0 && conv_C % 128 == 0 && kin == conv_taps * conv_C && ldx == conv_C &&
                        rows % WR_ROWS == 0 && n % 128 == 0 && ldx % 8 == 0 && lddy % 8 == 0,
                        "hgemm_wgrad_rows: shifted-rows taps need bf16 operands, Cin %% 128 == 0, N %% 128 == 0, rows %% 32 == 0, ldx == Cin");
    TTSMI_CHECK_ARG(x && dy && dw, "hgemm_wgrad_rows: null pointer");
    TTSMI_CHECK_ARG(rows > 0 && kin > 0 && n > 0, "hgemm_wgrad_rows: bad shape");
    TTSMI_CHECK_ARG(al16(x) && al16(dy) && ldx % 4 == 0 && lddy % 4 == 0 && n % 4 == 0,
                    "hgemm_wgrad_rows: operands must be 16-byte aligned with ld %% 4 == 0");
    if (conv_taps > 1 && !shifted)
        TTSMI_CHECK_ARG(conv_C % 128 == 0 && kin == conv_taps * conv_C && conv_T > 0 && ldx == conv_C,
                        "hgemm_wgrad_rows: conv needs Cin %% 128 == 0");
    else
        TTSMI_CHECK_ARG(kin % 4 == 0, "hgemm_wgrad_rows: K %% 4 != 0");
    hipStream_t st = (hipStream_t)stream;
    WRowsP p;
    memset(&p, 0, sizeof(p));
    p.X = (const float*)x; p.ldx = ldx; p.DY = (const float*)dy; p.lddy = lddy; p.dW = dw; p.lddw = lddw;
    p.M = rows; p.K = kin; p.N = n;
    p.taps = conv_taps > 1 ? conv_taps : 1; p.T = conv_T; p.Cin = conv_C; p.pad = conv_pad;
    p.tiles_k = ttsmi_cdiv(kin, 128); p.tiles_n = ttsmi_cdiv(n, 128);
    p.X2 = (const float*)x2; p.ldx2 = ldx2; p.K1 = k1;
    int tiles = p.tiles_k * p.tiles_n;
    int kps = 0;
    const int splits = wgrad_rows_splits(rows, kin, n, &kps);
    p.k_per_split = kps;
    p.ws = (float*)ws; p.colsum = db; p.colsum_ws = p.ws + (size_t)splits * kin * n;
    dim3 wgrid(tiles * splits);
    TTSMI_KNOB(wdma, "TTSMI_WGRAD_DMA", 1);
    const bool dma_ok = (wdma || shifted) && x_is_bf16 && dy_is_bf16 && (conv_taps <= 1 || shifted) && kin % 128 == 0 &&
                        n % 128 == 0 && ldx % 8 == 0 && lddy % 8 == 0 && al16(x) && al16(dy);
    ttsmi_note_kernel(dma_ok ? "wgrad_dma_kernel" : "wgrad_rows_kernel");
    if (dma_ok) hipLaunchKernelGGL(wgrad_dma_kernel, wgrid, dim3(256), 0, st, p);
    else if (x_is_bf16 && dy_is_bf16) hipLaunchKernelGGL((wgrad_rows_kernel<true, true>), wgrid, dim3(256), 0, st, p);
    else if (x_is_bf16) hipLaunchKernelGGL((wgrad_rows_kernel<true, false>), wgrid, dim3(256), 0, st, p);
    else if (dy_is_bf16) hipLaunchKernelGGL((wgrad_rows_kernel<false, true>), wgrid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((wgrad_rows_kernel<false, false>), wgrid, dim3(256), 0, st, p);
    TTSMI_CHECK_LAUNCH("hgemm_wgrad_rows");
    if (splits > 1) {
        if (job && hsplit_vec_ok(p.ws, dw, (long)lddw, n, db)) {          // the caller reduces (batched with its other jobs)
            job->ws = p.ws; job->dw = dw; job->lddw = (long)lddw; job->kin = kin; job->n = n; job->splits = splits;
            job->cs_ws = p.colsum_ws; job->db = db;
            return TTSMI_OK;
        }
        hsplit_reduce_launch(st, p.ws, dw, (long)lddw, kin, n, splits, p.colsum_ws, db);
        TTSMI_CHECK_LAUNCH("hgemm_wgrad_rows_reduce");
    }
    return TTSMI_OK;
}

int ttsmi_cast_transpose_bf16(const float* src, int64_t ld_src, uint16_t* dst, int64_t ld_dst, int R,
                              int C, int taps, int T, int pad, ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(src && dst && R > 0 && C > 0 && taps >= 1 && ld_dst >= R, "cast_transpose_bf16: bad argument");
    if (taps > 1) TTSMI_CHECK_ARG(T > 0, "cast_transpose_bf16: conv needs T");
    dim3 grid(ttsmi_cdiv(ld_dst, 64), ttsmi_cdiv(C, 64), taps);
    hipLaunchKernelGGL(cast_transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, (long)ld_src, dst,
                       (long)ld_dst, R, C, taps, T, pad);
    TTSMI_CHECK_LAUNCH("cast_transpose_bf16");
    return TTSMI_OK;
}

int ttsmi_cast_transpose_bf16_batched(const ttsmi_transpose_desc* desc_dev, int n_desc, int total_tiles,
                                      ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(desc_dev && n_desc > 0 && total_tiles > 0, "cast_transpose_bf16_batched: bad argument");
    hipLaunchKernelGGL(cast_transpose_batched_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream,
                       desc_dev, n_desc);
    TTSMI_CHECK_LAUNCH("cast_transpose_bf16_batched");
    return TTSMI_OK;
}

int ttsmi_conv_wdgrad_layout_bf16(const float* w, uint16_t* dst, int k, int Cin, int Cout, int cout_ld,
                                  ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(w && dst && k > 0 && Cin > 0 && Cout > 0 && cout_ld >= Cout,
                    "conv_wdgrad_layout_bf16: bad argument");
    long n = (long)k * Cin * cout_ld;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(conv_wdgrad_layout_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, dst, k,
                       Cin, Cout, cout_ld);
    TTSMI_CHECK_LAUNCH("conv_wdgrad_layout_bf16");
    return TTSMI_OK;
}

}  // extern "C"
