// GEMM family of the ForwardTransformer hot path on exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
// One templated kernel serves Dense forward / dgrad / wgrad and Conv1D(k,'same') forward / dgrad /
// wgrad (implicit GEMM: with channels-last activations the im2col row of frame (b,t) is just the
// contiguous window x[b, t-pad : t-pad+k, :], so no im2col buffer exists anywhere).
//
//   C[M,N] = epilogue( sum_kk A(m,kk) * B(kk,n) )
//
// A modes:  A_KC  A(m,kk) = a[m*lda + kk]              (activations as rows; K contiguous)
//                 + two K segments (dual-A: concat([q_in, ctx]) . Wo without the concat)
//                 + conv windowing  a[(m-pad)*Cw + kk], zero outside the sequence
//           A_MC  A(m,kk) = a[kk*lda + m]              (x^T for wgrad; M contiguous)
//                 + conv windowing  a[(kk + tap(m) - pad)*Cw + (m % Cw)]
// B modes:  B_NC  B(kk,n) = b[kk*ldb + n]              (Keras [in,out] kernel as is; dy for wgrad)
//           B_KC  B(kk,n) = b[tapoff(kk/Kt) + n*ldb + kk%Kt]   (w^T for dgrad, flipped taps for conv)
//
// Tile: 128x128x16 per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 2x2 MFMA 32x32
// tiles, 64 accumulator VGPRs).  LDS holds A as [k][m] and B as [k][n] so that each MFMA operand
// is ONE conflict-free ds_read_b32 per lane (lane l reads element [k + (l>>5)][x0 + (l&31)]).
// fp32 MFMA runs at 64 FLOP/clk/SIMD (157 TF chip peak), i.e. one 32x32x2 MFMA per 64 cycles per
// SIMD, so the kernel is MFMA-issue bound by a wide margin: 4 ds_read_b32 feed 4 MFMAs (256
// cycles), and one 16 KB k-tile of global loads feeds 2048 cycles of MFMA per wave; a single LDS
// stage with register prefetch of the next k-tile plus >= 3 workgroups per CU covers the latency.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GBM 128
#define GBN 128
#define GBK 16

enum { A_KC = 0, A_MC = 1 };
enum { B_NC = 0, B_KC = 1 };

struct GemmP {
    const float* A; long lda;
    const float* A2; long lda2; int K1;
    const float* B; long ldb;
    float* C; long ldc;
    const float* bias;
    const float* relu_src; long ld_relu;
    int M, N, K;
    int relu, accumulate;              // accumulate: C += result
    int a_taps, T, Cw, pad;            // conv windowing on A (a_taps == 1: none)
    int b_taps, b_kt; long b_tap_stride; int b_flip;   // B_KC tap addressing
    int k_per_split;                   // reduction range per blockIdx.z
    float* ws;                         // split slabs [gridDim.z][M][N] when gridDim.z > 1
    float* colsum;                     // optional: column sums of the B operand (bias gradient)
    float* colsum_ws;                  //           their split slabs [gridDim.z][N]
    int tiles_m, tiles_n;
    int x3;                            // TTSMI_BF16X3: gemm_x3_kernel (host side only)
};

// ---- element address + validity ---------------------------------------------------------------
template <int AM>
__device__ __forceinline__ const float* a_ptr(const GemmP& p, int m, int kk, bool& ok) {
    if (AM == A_KC) {
        ok = (m < p.M) && (kk < p.K);
        if (p.a_taps > 1) {
            int t = m % p.T, tap = kk / p.Cw, tt = t + tap - p.pad;
            ok = ok && (tt >= 0) && (tt < p.T);
            return p.A + ((long)m - p.pad) * p.Cw + kk;
        }
        if (p.A2 != nullptr && kk >= p.K1) return p.A2 + (long)m * p.lda2 + (kk - p.K1);
        return p.A + (long)m * p.lda + kk;
    } else {
        ok = (m < p.M) && (kk < p.K);
        if (p.a_taps > 1) {
            int t = kk % p.T, tap = m / p.Cw, tt = t + tap - p.pad;
            ok = ok && (tt >= 0) && (tt < p.T);
            return p.A + ((long)kk + tap - p.pad) * p.Cw + (m - tap * p.Cw);
        }
        return p.A + (long)kk * p.lda + m;
    }
}

template <int BMODE>
__device__ __forceinline__ const float* b_ptr(const GemmP& p, int kk, int n, bool& ok) {
    ok = (kk < p.K) && (n < p.N);
    if (BMODE == B_NC) return p.B + (long)kk * p.ldb + n;
    if (p.b_taps > 1) {
        int tap = kk / p.b_kt, r = kk - tap * p.b_kt;
        int tp = p.b_flip ? (p.b_taps - 1 - tap) : tap;
        return p.B + (long)tp * p.b_tap_stride + (long)n * p.ldb + r;
    }
    return p.B + (long)n * p.ldb + kk;
}

template <bool VEC>
__device__ __forceinline__ float4 ld4(const float* ptr, bool ok0, const bool (&oks)[4],
                                      const float* const (&ptrs)[4]) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (VEC) {
        if (ok0) v = *reinterpret_cast<const float4*>(ptr);
    } else {
        if (oks[0]) v.x = *ptrs[0];
        if (oks[1]) v.y = *ptrs[1];
        if (oks[2]) v.z = *ptrs[2];
        if (oks[3]) v.w = *ptrs[3];
    }
    return v;
}

// Fetch this thread's two float4 of the A tile (rows m0.., reduction k0..) into registers.
template <int AM, bool VEC>
__device__ __forceinline__ void fetch_a(const GemmP& p, int m0, int k0, int kend, int tid,
                                        float4 (&r)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int id = tid + 256 * i;
        bool oks[4];
        const float* ptrs[4];
        if (AM == A_KC) {
            int row = id & 127, kq = id >> 7;          // lanes of a wave: consecutive rows, same kq
            int m = m0 + row, kk = k0 + kq * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ptrs[e] = a_ptr<AM>(p, m, kk + e, oks[e]);
                oks[e] = oks[e] && (kk + e < kend);
            }
        } else {
            int kl = id >> 5, mq = id & 31;            // lanes: 32 float4 along m = 512 B rows
            int m = m0 + mq * 4, kk = k0 + kl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ptrs[e] = a_ptr<AM>(p, m + e, kk, oks[e]);
                oks[e] = oks[e] && (kk < kend);
            }
        }
        r[i] = ld4<VEC>(ptrs[0], oks[0], oks, ptrs);
    }
}

template <int BMODE, bool VEC>
__device__ __forceinline__ void fetch_b(const GemmP& p, int n0, int k0, int kend, int tid,
                                        float4 (&r)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int id = tid + 256 * i;
        bool oks[4];
        const float* ptrs[4];
        if (BMODE == B_NC) {
            int kl = id >> 5, nq = id & 31;
            int n = n0 + nq * 4, kk = k0 + kl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ptrs[e] = b_ptr<BMODE>(p, kk, n + e, oks[e]);
                oks[e] = oks[e] && (kk < kend);
            }
        } else {
            int col = id & 127, kq = id >> 7;
            int n = n0 + col, kk = k0 + kq * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ptrs[e] = b_ptr<BMODE>(p, kk + e, n, oks[e]);
                oks[e] = oks[e] && (kk + e < kend);
            }
        }
        r[i] = ld4<VEC>(ptrs[0], oks[0], oks, ptrs);
    }
}

// Store the fetched registers into the [k][x] LDS image.
template <bool K_CONTIG>
__device__ __forceinline__ void stash(float (*S)[GBM], int tid, const float4 (&r)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int id = tid + 256 * i;
        if (K_CONTIG) {                       // transposing scatter: 4 x ds_write_b32, conflict free
            int x = id & 127, kq = id >> 7;
            S[kq * 4 + 0][x] = r[i].x;
            S[kq * 4 + 1][x] = r[i].y;
            S[kq * 4 + 2][x] = r[i].z;
            S[kq * 4 + 3][x] = r[i].w;
        } else {                              // straight ds_write_b128
            int kl = id >> 5, xq = id & 31;
            *reinterpret_cast<float4*>(&S[kl][xq * 4]) = r[i];
        }
    }
}

template <int AM, int BMODE, bool VEC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmP p) {
    __shared__ __attribute__((aligned(16))) float smem[2][GBK][GBM];
    float(*As)[GBM] = smem[0];
    float(*Bs)[GBM] = smem[1];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = bid % p.tiles_n, tm = bid / p.tiles_n;
    const int m0 = tm * GBM, n0 = tn * GBN;
    const int kbeg = blockIdx.z * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[2], rb[2];
    if (kbeg < kend) {
        fetch_a<AM, VEC>(p, m0, kbeg, kend, tid, ra);
        fetch_b<BMODE, VEC>(p, n0, kbeg, kend, tid, rb);
        stash<AM == A_KC>(As, tid, ra);
        stash<BMODE == B_KC>(Bs, tid, rb);
    }
    __syncthreads();

    const int l31 = lane & 31, kh = lane >> 5;
    // bias gradient fused into wgrad: the B tile IS dy[rows, n-tile]; the m-tile-0 workgroups add
    // up its columns while it sits in LDS (one conflict-free ds_read per k per thread)
    const bool do_colsum = (p.colsum != nullptr) && (tm == 0) && (tid < GBN);
    float csum = 0.f;
    for (int k0 = kbeg; k0 < kend; k0 += GBK) {
        const bool more = (k0 + GBK) < kend;
        if (more) {
            fetch_a<AM, VEC>(p, m0, k0 + GBK, kend, tid, ra);
            fetch_b<BMODE, VEC>(p, n0, k0 + GBK, kend, tid, rb);
        }
        if (do_colsum) {
#pragma unroll
            for (int kk = 0; kk < GBK; ++kk) csum += Bs[kk][tid];
        }
#pragma unroll
        for (int kk = 0; kk < GBK; kk += 2) {
            float a0 = As[kk + kh][wr * 64 + l31];
            float a1 = As[kk + kh][wr * 64 + 32 + l31];
            float b0 = Bs[kk + kh][wc * 64 + l31];
            float b1 = Bs[kk + kh][wc * 64 + 32 + l31];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
        if (more) {
            stash<AM == A_KC>(As, tid, ra);
            stash<BMODE == B_KC>(Bs, tid, rb);
        }
        __syncthreads();
    }

    // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const bool split = gridDim.z > 1;
    if (do_colsum && n0 + tid < p.N) {
        if (split) p.colsum_ws[(long)blockIdx.z * p.N + n0 + tid] = csum;
        else p.colsum[n0 + tid] = csum;
    }
    float* Cb = split ? p.ws + (long)blockIdx.z * p.M * p.N : p.C;
    const long ldc = split ? (long)p.N : p.ldc;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int col = n0 + wc * 64 + j * 32 + l31;
        if (col >= p.N) continue;
        float bv = (!split && p.bias) ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (row >= p.M) continue;
                float v = acc[i][j][r] + bv;
                if (!split) {
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (p.relu_src) v = p.relu_src[(long)row * p.ld_relu + col] > 0.f ? v : 0.f;
                    if (p.accumulate) v += Cb[(long)row * ldc + col];
                }
                Cb[(long)row * ldc + col] = v;
            }
        }
    }
}

// =================================================================================================
// The same GEMM family on bf16 MFMA with THREE products per pair (round 6, dtype TTSMI_BF16X3): every fp32 operand element
// is split into hi = bf16(v) and lo = bf16(v - hi) when its tile goes to LDS, and a . b ~ a_hi b_hi + a_hi b_lo + a_lo b_hi
// (fp32 accumulate; the dropped lo . lo term and the rounding of lo are ~2^-17 of the product).  Interface, tiling, split-K,
// epilogue and the fused bias gradient are gemm_f32_kernel's; what changes is the LDS image - [x][k] bf16 with k contiguous,
// 80-byte rows (16-byte fragment reads of 16 consecutive rows then cover all 64 banks), hi and lo images side by side - and
// the rate: 12 v_mfma_f32_32x32x16_bf16 (384 cycles) per 32-deep k-tile instead of 64 v_mfma_f32_32x32x2_f32 (4 096).
// Every thread fetches (row x, 4 consecutive k): one float4 where k is the contiguous dimension, four row-coalesced dwords
// where x is.
// =================================================================================================
typedef __bf16 x3_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 x3_bf16x4 __attribute__((ext_vector_type(4)));
#define X3_BK 32
#define X3_LD 40                      // bf16 elements per LDS row: 32 + 8 of padding
#ifndef X3_WGS_PER_CU
#define X3_WGS_PER_CU 3
#endif

// Fetch geometry (both operands, a 128 x 32 tile = 16 elements per thread as four float4):
//   k contiguous in memory (A_KC, B_KC): four passes of (row x = id >> 3, k-quad kq = id & 7) - eight lanes read one row's 128
//     bytes, r[i] = the four k of row x_i;
//   x contiguous in memory (A_MC, B_NC): ONE 4 x 4 block per thread (x-quad xq = tid & 31, k-quad kq = tid >> 5) - 32 lanes read
//     512 contiguous bytes of one k row, r[e] = the four x at k = 4 kq + e; the block is transposed on its way into LDS.
template <bool KC>
__device__ __forceinline__ void x3_coords(int tid, int i, int& x, int& kq) {
    if (KC) { const int id = tid + 256 * i; x = id >> 3; kq = id & 7; }
    else { x = (tid & 31) * 4; kq = tid >> 5; }
}
template <int AM, bool VEC>
__device__ __forceinline__ void x3_fetch_a(const GemmP& p, int m0, int k0, int kend, int tid, float4 (&r)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int x, kq;
        x3_coords<AM == A_KC>(tid, i, x, kq);
        bool oks[4];
        const float* ptrs[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (AM == A_KC) {
                ptrs[e] = a_ptr<AM>(p, m0 + x, k0 + kq * 4 + e, oks[e]);
                oks[e] = oks[e] && (k0 + kq * 4 + e < kend);
            } else {
                ptrs[e] = a_ptr<AM>(p, m0 + x + e, k0 + kq * 4 + i, oks[e]);
                oks[e] = oks[e] && (k0 + kq * 4 + i < kend);
            }
        }
        r[i] = ld4<VEC>(ptrs[0], oks[0], oks, ptrs);
    }
}
template <int BMODE, bool VEC>
__device__ __forceinline__ void x3_fetch_b(const GemmP& p, int n0, int k0, int kend, int tid, float4 (&r)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int x, kq;
        x3_coords<BMODE == B_KC>(tid, i, x, kq);
        bool oks[4];
        const float* ptrs[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (BMODE == B_KC) {
                ptrs[e] = b_ptr<BMODE>(p, k0 + kq * 4 + e, n0 + x, oks[e]);
                oks[e] = oks[e] && (k0 + kq * 4 + e < kend);
            } else {
                ptrs[e] = b_ptr<BMODE>(p, k0 + kq * 4 + i, n0 + x + e, oks[e]);
                oks[e] = oks[e] && (k0 + kq * 4 + i < kend);
            }
        }
        r[i] = ld4<VEC>(ptrs[0], oks[0], oks, ptrs);
    }
}
// FAST path of the fetch (workgroup-uniform: vectorisable operands, no conv window, no second K segment, the tile wholly inside
// M x N): one pointer per thread and operand, set up once - the generic path above spends ~1 000 scalar / vector instructions per
// k-tile on addresses and validity, against the tile's 24 MFMAs (ISA reading).  base = the operand's tile origin plus the thread's
// own offset (KC: row tid >> 3, k-quad tid & 7; x contiguous: x-quad tid & 31 - its k-quad tid >> 5 enters with k).
template <bool KC>
__device__ __forceinline__ void x3_fetch_fast(const float* base, long ld, int k0, int kend, int tid, float4 (&r)[4]) {
    // (UNCONDITIONAL loads - the fast path also requires a reduction range of whole k-tiles: a `valid ? load : 0` consumes the
    // loaded value at once, i.e. puts the wait for the prefetch in front of the multiplies it should hide under)
    if (KC) {
        const float* q = base + k0;
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = *reinterpret_cast<const float4*>(q + (long)(32 * i) * ld);
    } else {
        const int kk = k0 + (tid >> 5) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = *reinterpret_cast<const float4*>(base + (long)(kk + e) * ld);
    }
}
template <bool KC>
__device__ __forceinline__ void x3_stash(uint16_t (*Hi)[X3_LD], uint16_t (*Lo)[X3_LD], int tid, const float4 (&r)[4]) {
    const float v[4][4] = {{r[0].x, r[0].y, r[0].z, r[0].w}, {r[1].x, r[1].y, r[1].z, r[1].w}, {r[2].x, r[2].y, r[2].z, r[2].w},
                           {r[3].x, r[3].y, r[3].z, r[3].w}};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int x, kq;
        x3_coords<KC>(tid, i, x, kq);
        x3_bf16x4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float f = KC ? v[i][e] : v[e][i];            // (x contiguous: row x + i of the 4 x 4 block, its four k)
            hi[e] = (__bf16)f;
            lo[e] = (__bf16)(f - (float)hi[e]);
        }
        const int row = KC ? x : x + i;
        *reinterpret_cast<x3_bf16x4*>(&Hi[row][kq * 4]) = hi;
        *reinterpret_cast<x3_bf16x4*>(&Lo[row][kq * 4]) = lo;
    }
}

template <int AM, int BMODE, bool VEC>
__global__ __launch_bounds__(256, X3_WGS_PER_CU) void gemm_x3_kernel(GemmP p) {
    __shared__ __attribute__((aligned(16))) uint16_t smem[4][GBM][X3_LD];          // A hi, A lo, B hi, B lo: 40 KB
    uint16_t(*Ah)[X3_LD] = smem[0];
    uint16_t(*Al)[X3_LD] = smem[1];
    uint16_t(*Bh)[X3_LD] = smem[2];
    uint16_t(*Bl)[X3_LD] = smem[3];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = bid % p.tiles_n, tm = bid / p.tiles_n;
    const int m0 = tm * GBM, n0 = tn * GBN;
    const int kbeg = blockIdx.z * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // bias gradient fused into wgrad: the B tile IS dy[rows, n-tile]; the m-tile-0 workgroups add up its columns from the
    // fetched registers - four columns per thread, eight threads (the k-quads) per column, reduced through LDS at the end
    const bool do_colsum = (p.colsum != nullptr) && (tm == 0);
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    float4 ra[4], rb[4];
    auto stash_both = [&]() {
        x3_stash<AM == A_KC>(Ah, Al, tid, ra);
        x3_stash<BMODE == B_KC>(Bh, Bl, tid, rb);
        if (do_colsum) {
            if (BMODE == B_KC) {
#pragma unroll
                for (int i = 0; i < 4; ++i) cs[i] += (rb[i].x + rb[i].y) + (rb[i].z + rb[i].w);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) { cs[0] += rb[e].x; cs[1] += rb[e].y; cs[2] += rb[e].z; cs[3] += rb[e].w; }
            }
        }
    };
    const bool fast = VEC && p.a_taps == 1 && p.A2 == nullptr && p.b_taps == 1 && m0 + GBM <= p.M && n0 + GBN <= p.N && (kend - kbeg) % X3_BK == 0;
    const float* fa = AM == A_KC ? p.A + (long)(m0 + (tid >> 3)) * p.lda + (tid & 7) * 4 : p.A + m0 + (tid & 31) * 4;
    const float* fb = BMODE == B_KC ? p.B + (long)(n0 + (tid >> 3)) * p.ldb + (tid & 7) * 4 : p.B + n0 + (tid & 31) * 4;
    auto fetch_both = [&](int k0) {
        if (fast) {
            x3_fetch_fast<AM == A_KC>(fa, p.lda, k0, kend, tid, ra);
            x3_fetch_fast<BMODE == B_KC>(fb, p.ldb, k0, kend, tid, rb);
        } else {
            x3_fetch_a<AM, VEC>(p, m0, k0, kend, tid, ra);
            x3_fetch_b<BMODE, VEC>(p, n0, k0, kend, tid, rb);
        }
    };
    if (kbeg < kend) {
        fetch_both(kbeg);
        stash_both();
    }
    __syncthreads();

    const int l31 = lane & 31, kh = lane >> 5;
    for (int k0 = kbeg; k0 < kend; k0 += X3_BK) {
        const bool more = (k0 + X3_BK) < kend;
        if (more) fetch_both(k0 + X3_BK);
#pragma unroll
        for (int ks = 0; ks < X3_BK / 16; ++ks) {
            const int off = ks * 16 + kh * 8;
            x3_bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                ah[t] = *reinterpret_cast<const x3_bf16x8*>(&Ah[wr * 64 + t * 32 + l31][off]);
                al[t] = *reinterpret_cast<const x3_bf16x8*>(&Al[wr * 64 + t * 32 + l31][off]);
                bh[t] = *reinterpret_cast<const x3_bf16x8*>(&Bh[wc * 64 + t * 32 + l31][off]);
                bl[t] = *reinterpret_cast<const x3_bf16x8*>(&Bl[wc * 64 + t * 32 + l31][off]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
        if (more) stash_both();
        __syncthreads();
    }

    // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const bool split = gridDim.z > 1;
    if (do_colsum) {
        float* red = reinterpret_cast<float*>(&smem[0][0][0]);         // [8 k-quads][128 columns] (the k-loop's last barrier retired the images)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int x, kq;
            x3_coords<BMODE == B_KC>(tid, i, x, kq);
            red[kq * GBN + (BMODE == B_KC ? x : x + i)] = cs[i];
        }
        __syncthreads();
        if (tid < GBN && n0 + tid < p.N) {
            float c2 = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) c2 += red[q * GBN + tid];
            if (split) p.colsum_ws[(long)blockIdx.z * p.N + n0 + tid] = c2;
            else p.colsum[n0 + tid] = c2;
        }
    }
    float* Cb = split ? p.ws + (long)blockIdx.z * p.M * p.N : p.C;
    const long ldc = split ? (long)p.N : p.ldc;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int col = n0 + wc * 64 + j * 32 + l31;
        if (col >= p.N) continue;
        float bv = (!split && p.bias) ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (row >= p.M) continue;
                float v = acc[i][j][r] + bv;
                if (!split) {
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (p.relu_src) v = p.relu_src[(long)row * p.ld_relu + col] > 0.f ? v : 0.f;
                    if (p.accumulate) v += Cb[(long)row * ldc + col];
                }
                Cb[(long)row * ldc + col] = v;
            }
        }
    }
}

// dw[i] = sum_s ws[s][i]
__global__ void split_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out,
                                    long ldo, int M, int N, int splits,
                                    const float* __restrict__ cs_ws, float* __restrict__ cs_out) {
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

// column sums (bias gradient): stage 1 - each block sums a 256-row chunk for 64 columns
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, long ldx,
                                                             float* __restrict__ part, int M,
                                                             int N) {
    __shared__ float red[4][64];
    int c = blockIdx.x * 64 + (threadIdx.x & 63);
    int rl = threadIdx.x >> 6;
    int r0 = blockIdx.y * 256;
    float s = 0.f;
    if (c < N) {
        int rend = min(M, r0 + 256);
        for (int r = r0 + rl; r < rend; r += 4) s += x[(long)r * ldx + c];
    }
    red[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && c < N)
        part[(long)blockIdx.y * N + c] = red[0][threadIdx.x] + red[1][threadIdx.x] +
                                         red[2][threadIdx.x] + red[3][threadIdx.x];
}
__global__ void colsum_final_kernel(const float* __restrict__ part, float* __restrict__ out,
                                    int chunks, int N) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    float s = 0.f;
    for (int k = 0; k < chunks; ++k) s += part[(long)k * N + c];
    out[c] = s;
}

// ---- host side ----------------------------------------------------------------------------------
static bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

template <int AM, int BMODE>
static int launch_gemm(GemmP& p, bool vec, int splits, hipStream_t st, const char* name) {
    p.tiles_m = ttsmi_cdiv(p.M, GBM);
    p.tiles_n = ttsmi_cdiv(p.N, GBN);
    dim3 grid(p.tiles_m * p.tiles_n, 1, splits);
    if (p.x3) {
        ttsmi_note_kernel("gemm_x3_kernel");
        if (vec) hipLaunchKernelGGL((gemm_x3_kernel<AM, BMODE, true>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemm_x3_kernel<AM, BMODE, false>), grid, dim3(256), 0, st, p);
    } else if (vec)
        hipLaunchKernelGGL((gemm_f32_kernel<AM, BMODE, true>), grid, dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL((gemm_f32_kernel<AM, BMODE, false>), grid, dim3(256), 0, st, p);
    TTSMI_CHECK_LAUNCH(name);
    return TTSMI_OK;
}

static void init_p(GemmP& p) {
    memset(&p, 0, sizeof(p));
    p.a_taps = 1;
    p.b_taps = 1;
}

static int pick_splits(int rows, int tiles) {
    int want = (512 + tiles - 1) / tiles;
    int maxs = (rows + 255) / 256;
    int s = want < maxs ? want : maxs;
    if (s > 64) s = 64;
    if (s < 1) s = 1;
    return s;
}

static size_t wgrad_ws_bytes(long rows, long kin, long n) {
    int tiles = ttsmi_cdiv(kin, GBM) * ttsmi_cdiv(n, GBN);
    int splits = pick_splits((int)rows, tiles);
    return (size_t)splits * (kin * n + n) * sizeof(float) + 256;
}

static int colsum(const float* dy, long lddy, float* db, int M, int N, float* ws, hipStream_t st) {
    int chunks = ttsmi_cdiv(M, 256);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(ttsmi_cdiv(N, 64), chunks), dim3(256), 0, st, dy,
                       lddy, ws, M, N);
    TTSMI_CHECK_LAUNCH("colsum_partial");
    hipLaunchKernelGGL(colsum_final_kernel, dim3(ttsmi_cdiv(N, 128)), dim3(128), 0, st, ws, db,
                       chunks, N);
    TTSMI_CHECK_LAUNCH("colsum_final");
    return TTSMI_OK;
}

static void setup_wgrad_split(GemmP& p, int rows, float* ws, float* db, int& splits) {
    int tiles = ttsmi_cdiv(p.M, GBM) * ttsmi_cdiv(p.N, GBN);
    splits = pick_splits(rows, tiles);
    int kps = ttsmi_cdiv(rows, splits);
    kps = ((kps + GBK - 1) / GBK) * GBK;
    splits = ttsmi_cdiv(rows, kps);
    p.k_per_split = kps;
    p.ws = ws;
    p.colsum = db;
    p.colsum_ws = ws + (size_t)splits * p.M * p.N;
}

static int finish_wgrad(GemmP& p, int splits, float* dw, long lddw, hipStream_t st) {
    if (splits > 1) {
        long n = (long)p.M * p.N;
        int blocks = (int)((n + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(split_reduce_kernel, dim3(blocks), dim3(256), 0, st, p.ws, dw, lddw,
                           p.M, p.N, splits, p.colsum_ws, p.colsum);
        TTSMI_CHECK_LAUNCH("split_reduce");
    }
    return TTSMI_OK;
}

extern "C" {

int ttsmi_linear_fwd(const void* x, int64_t ldx, const void* x2, int64_t ldx2, int K1,
                     const void* w, int64_t ldw, const float* bias, void* y, int64_t ldy, int M,
                     int N, int K, int relu, int dtype, ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(x && w && y, "linear_fwd: null pointer");
    TTSMI_CHECK_ARG(M >= 0 && N > 0 && K > 0, "linear_fwd: bad shape M=%d N=%d K=%d", M, N, K);
    TTSMI_CHECK_ARG(dtype == TTSMI_F32 || dtype == TTSMI_BF16X3, "linear_fwd: dtype %d not built", dtype);
    if (x2) TTSMI_CHECK_ARG(K1 > 0 && K1 < K, "linear_fwd: bad K1=%d for K=%d", K1, K);
    if (M == 0) return TTSMI_OK;
    GemmP p;
    init_p(p);
    p.x3 = dtype == TTSMI_BF16X3;
    p.A = (const float*)x; p.lda = ldx;
    p.A2 = (const float*)x2; p.lda2 = ldx2; p.K1 = K1;
    p.B = (const float*)w; p.ldb = ldw;
    p.C = (float*)y; p.ldc = ldy;
    p.bias = bias; p.M = M; p.N = N; p.K = K; p.relu = relu; p.k_per_split = K;
    bool vec = al16(x) && (ldx % 4 == 0) && (K % 4 == 0) && al16(w) && (ldw % 4 == 0) &&
               (N % 4 == 0);
    if (x2) vec = vec && al16(x2) && (ldx2 % 4 == 0) && (K1 % 4 == 0);
    return launch_gemm<A_KC, B_NC>(p, vec, 1, (hipStream_t)stream, "linear_fwd");
}

int ttsmi_linear_dgrad(const void* dy, int64_t lddy, const void* w, int64_t ldw,
                       const void* relu_src, int64_t ld_relu, void* dx, int64_t lddx, int M, int N,
                       int K, int accumulate, int dtype, ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(dy && w && dx, "linear_dgrad: null pointer");
    TTSMI_CHECK_ARG(M >= 0 && N > 0 && K > 0, "linear_dgrad: bad shape");
    TTSMI_CHECK_ARG(dtype == TTSMI_F32 || dtype == TTSMI_BF16X3, "linear_dgrad: dtype %d not built", dtype);
    if (M == 0) return TTSMI_OK;
    GemmP p;
    init_p(p);
    p.x3 = dtype == TTSMI_BF16X3;
    // GEMM view: C[M, K] = dy[M, N] . B(kk = n_out, n = k_in) with B = w[k_in*ldw + n_out]
    p.A = (const float*)dy; p.lda = lddy;
    p.B = (const float*)w; p.ldb = ldw;
    p.C = (float*)dx; p.ldc = lddx;
    p.relu_src = (const float*)relu_src; p.ld_relu = ld_relu;
    p.M = M; p.N = K; p.K = N; p.k_per_split = N; p.accumulate = accumulate;
    p.b_kt = N;
    bool vec = al16(dy) && (lddy % 4 == 0) && (N % 4 == 0) && al16(w) && (ldw % 4 == 0);
    return launch_gemm<A_KC, B_KC>(p, vec, 1, (hipStream_t)stream, "linear_dgrad");
}

size_t ttsmi_linear_wgrad_ws_bytes(int M, int N, int K) { return wgrad_ws_bytes(M, K, N); }

int ttsmi_linear_wgrad(const void* x, int64_t ldx, const void* dy, int64_t lddy, float* dw,
                       int64_t lddw, float* db, int M, int N, int K, void* ws, size_t ws_bytes,
                       int dtype, ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(x && dy && dw, "linear_wgrad: null pointer");
    TTSMI_CHECK_ARG(M > 0 && N > 0 && K > 0, "linear_wgrad: bad shape");
    TTSMI_CHECK_ARG(dtype == TTSMI_F32 || dtype == TTSMI_BF16X3, "linear_wgrad: dtype %d not built", dtype);
    TTSMI_CHECK_ARG(ws_bytes >= wgrad_ws_bytes(M, K, N) && ws, "linear_wgrad: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    GemmP p;
    init_p(p);
    p.x3 = dtype == TTSMI_BF16X3;
    // GEMM view: C[K_in, N] = A(m = k_in, kk = row) . B(kk = row, n); reduction over the M rows
    p.A = (const float*)x; p.lda = ldx;
    p.B = (const float*)dy; p.ldb = lddy;
    p.C = dw; p.ldc = lddw;
    p.M = K; p.N = N; p.K = M;
    int splits;
    setup_wgrad_split(p, M, (float*)ws, db, splits);
    bool vec = al16(x) && (ldx % 4 == 0) && (K % 4 == 0) && al16(dy) && (lddy % 4 == 0) &&
               (N % 4 == 0);
    int rc = launch_gemm<A_MC, B_NC>(p, vec, splits, st, "linear_wgrad");
    if (rc) return rc;
    return finish_wgrad(p, splits, dw, lddw, st);
}

int ttsmi_conv1d_fwd(const void* x, const void* w, const float* bias, void* y, int B, int T,
                     int Cin, int Cout, int k, int relu, int dtype, ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(x && w && y, "conv1d_fwd: null pointer");
    TTSMI_CHECK_ARG(B >= 0 && T > 0 && Cin > 0 && Cout > 0 && k > 0, "conv1d_fwd: bad shape");
    TTSMI_CHECK_ARG(dtype == TTSMI_F32 || dtype == TTSMI_BF16X3, "conv1d_fwd: dtype %d not built", dtype);
    if (B == 0) return TTSMI_OK;
    GemmP p;
    init_p(p);
    p.x3 = dtype == TTSMI_BF16X3;
    p.A = (const float*)x; p.lda = Cin;
    p.a_taps = k; p.T = T; p.Cw = Cin; p.pad = (k - 1) / 2;
    p.B = (const float*)w; p.ldb = Cout;            // [k*Cin, Cout] contiguous
    p.C = (float*)y; p.ldc = Cout;
    p.bias = bias; p.relu = relu;
    p.M = B * T; p.N = Cout; p.K = k * Cin; p.k_per_split = p.K;
    if (k == 1) p.a_taps = 1;
    bool vec = al16(x) && (Cin % 4 == 0) && al16(w) && (Cout % 4 == 0);
    return launch_gemm<A_KC, B_NC>(p, vec, 1, (hipStream_t)stream, "conv1d_fwd");
}

int ttsmi_conv1d_dgrad(const void* dy, const void* w, const void* relu_src, void* dx, int B, int T,
                       int Cin, int Cout, int k, int dtype, ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(dy && w && dx, "conv1d_dgrad: null pointer");
    TTSMI_CHECK_ARG(B >= 0 && T > 0 && Cin > 0 && Cout > 0 && k > 0, "conv1d_dgrad: bad shape");
    TTSMI_CHECK_ARG(dtype == TTSMI_F32 || dtype == TTSMI_BF16X3, "conv1d_dgrad: dtype %d not built", dtype);
    if (B == 0) return TTSMI_OK;
    GemmP p;
    init_p(p);
    p.x3 = dtype == TTSMI_BF16X3;
    // dx[s, ci] = sum_{j'} sum_co dy[s + j' - (k-1-pl), co] * w[k-1-j', ci, co]
    p.A = (const float*)dy; p.lda = Cout;
    p.a_taps = k; p.T = T; p.Cw = Cout; p.pad = k - 1 - (k - 1) / 2;
    p.B = (const float*)w; p.ldb = Cout;
    p.b_taps = k; p.b_kt = Cout; p.b_tap_stride = (long)Cin * Cout; p.b_flip = 1;
    p.C = (float*)dx; p.ldc = Cin;
    p.relu_src = (const float*)relu_src; p.ld_relu = Cin;
    p.M = B * T; p.N = Cin; p.K = k * Cout; p.k_per_split = p.K;
    if (k == 1) { p.a_taps = 1; p.b_taps = 1; p.b_kt = Cout; }
    bool vec = al16(dy) && (Cout % 4 == 0) && al16(w);
    return launch_gemm<A_KC, B_KC>(p, vec, 1, (hipStream_t)stream, "conv1d_dgrad");
}

size_t ttsmi_conv1d_wgrad_ws_bytes(int B, int T, int Cin, int Cout, int k) {
    return wgrad_ws_bytes((long)B * T, (long)k * Cin, Cout);
}

int ttsmi_conv1d_wgrad(const void* x, const void* dy, float* dw, float* db, int B, int T, int Cin,
                       int Cout, int k, void* ws, size_t ws_bytes, int dtype,
                       ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(x && dy && dw, "conv1d_wgrad: null pointer");
    TTSMI_CHECK_ARG(B > 0 && T > 0 && Cin > 0 && Cout > 0 && k > 0, "conv1d_wgrad: bad shape");
    TTSMI_CHECK_ARG(dtype == TTSMI_F32 || dtype == TTSMI_BF16X3, "conv1d_wgrad: dtype %d not built", dtype);
    int rows = B * T, kin = k * Cin;
    TTSMI_CHECK_ARG(ws && ws_bytes >= wgrad_ws_bytes(rows, kin, Cout),
                    "conv1d_wgrad: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    GemmP p;
    init_p(p);
    p.x3 = dtype == TTSMI_BF16X3;
    // dw[(j,ci), co] = sum_{(b,t)} x[b, t + j - pl, ci] * dy[(b,t), co]
    p.A = (const float*)x; p.lda = Cin;
    p.a_taps = k; p.T = T; p.Cw = Cin; p.pad = (k - 1) / 2;
    if (k == 1) p.a_taps = 1;
    p.B = (const float*)dy; p.ldb = Cout;
    p.C = dw; p.ldc = Cout;
    p.M = kin; p.N = Cout; p.K = rows;
    int splits;
    setup_wgrad_split(p, rows, (float*)ws, db, splits);
    bool vec = al16(x) && (Cin % 4 == 0) && al16(dy) && (Cout % 4 == 0);
    int rc = launch_gemm<A_MC, B_NC>(p, vec, splits, st, "conv1d_wgrad");
    if (rc) return rc;
    return finish_wgrad(p, splits, dw, Cout, st);
}

}  // extern "C"
