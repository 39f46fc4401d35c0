// Full-row GEMMs of the dense block with the neighbouring LayerNormalization fused into the epilogue (TTSMI_BF16 path).
//
//   forward :  y = rowmask( LN( keep(a . W^T + bias) + res ) * gamma + beta )        (o-projection + res-norm 1,
//                                                                                     FFN2 + res-norm 2; layers.py:148-150,
//                                                                                     211,229 and :100-102,230)
//   backward:  da = dy_part + a . W^T ;  (d_o, dres) = LN'(da)                        (dgrad of FFN1 + backward of res-norm 1)
//
// Both GEMMs have N = d = 256 output columns, i.e. one workgroup tile (64 rows x 256 columns) holds COMPLETE rows of the
// result, so the row statistics of the LayerNorm are available in the epilogue and the pre-norm tensor (o / f forward,
// da backward) never exists in HBM: per LayerNorm that removes one launch, a 29.5 MB write and a 29.5 MB read at
// M = 28 800 (the standalone kernels moved 103 MB forward / 133 MB backward).
//
// The accumulators are kept TRANSPOSED (C^T = W . A^T: the MFMA A operand is the weight fragment, the B operand the
// activation fragment), so that a lane owns ONE row m and 64 of its 256 columns (4 column tiles x 16 registers; the
// other 64 sit in lane ^ 32, the other 128 in the partner wave): the row reductions are 64 local adds, one shuffle and
// one two-wave LDS exchange instead of a 32-lane shuffle tree per register.  Lane-private rows also mean the residual,
// gamma / beta and x^ are plain float4 / 8-byte accesses with no LDS staging of the output tile.
#include <stdlib.h>

#include "common.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define RG_BM 64
#define RG_N 256
#define RG_BK 64
#define RG_LD (RG_BK + 8)          // 144-byte LDS rows: conflict-free ds_read_b128 (see gemm_bf16.hip)

struct RowGemmP {
    const uint16_t* A; long lda; const uint16_t* A2; long lda2; int K1;     // bf16 rows, optional second K segment
    const uint16_t* Bt; long ldb;                                           // bf16 [256][K] (W^T forward, W rows backward)
    const uint16_t* Bt2; long ldb2;                                         // optional: the second K segment has its own [256][K - K1] matrix
    int M, K;
    const float* bias;                 // [256] or NULL
    const float* res;                  // fwd: residual [M,256];  bwd: the partial gradient the GEMM result is added to
    const float* gamma; const float* beta;
    const uint8_t* row_pad;            // [M] 1 = padded row
    uint32_t thr_in; float inv_in; uint64_t seed; const int64_t* step_dev; uint32_t site_in;
    float eps;
    // forward outputs
    float* y; uint16_t* y_bf; uint16_t* xhat; float* rstd;
    // backward inputs / outputs
    const uint16_t* xhat_in; const float* rstd_in;
    uint16_t* dx_bf; float* dres;
    float* part; int nparts;           // per-workgroup partial sums of dgamma / dbeta: [nparts][256] + [nparts][256] (+ [nparts])
    int ablate;                        // measurement only (TTSMI_ROWGEMM_ABLATE, LDS-DMA kernel): 1 no multiply, 2 no DMA, 4 no epilogue, 8 no forward-epilogue stores
};

__device__ __forceinline__ uint2 rg_pack4(float a, float b, float c, float d) {
    bf16x4 h;
    h[0] = (__bf16)a; h[1] = (__bf16)b; h[2] = (__bf16)c; h[3] = (__bf16)d;
    return *reinterpret_cast<uint2*>(&h);
}
__device__ __forceinline__ void rg_unpack4(uint2 v, float (&o)[4]) {
    o[0] = __builtin_bit_cast(float, v.x << 16); o[1] = __builtin_bit_cast(float, v.x & 0xFFFF0000u);
    o[2] = __builtin_bit_cast(float, v.y << 16); o[3] = __builtin_bit_cast(float, v.y & 0xFFFF0000u);
}

template <int ROWS>
__device__ __forceinline__ void rg_fetch(const uint16_t* base, long ld, int rows, int r0, int k0, int tid,
                                         uint4 (&r)[ROWS * RG_BK / 2048]) {
#pragma unroll
    for (int i = 0; i < ROWS * RG_BK / 2048; ++i) {
        const int id = tid + 256 * i;
        const int row = id >> 3, c8 = id & 7;
        const int m = r0 + row;                                // (a clamped row index instead of the select sends these
        r[i] = m < rows ? *reinterpret_cast<const uint4*>(base + (long)m * ld + k0 + c8 * 8)   // arrays to scratch: hipcc 7.2)
                        : make_uint4(0, 0, 0, 0);
    }
}
template <int ROWS>
__device__ __forceinline__ void rg_stash(uint16_t (*S)[RG_LD], int tid, const uint4 (&r)[ROWS * RG_BK / 2048]) {
#pragma unroll
    for (int i = 0; i < ROWS * RG_BK / 2048; ++i) {
        const int id = tid + 256 * i;
        *reinterpret_cast<uint4*>(&S[id >> 3][(id & 7) * 8]) = r[i];
    }
}

// ---- the LayerNorm epilogues.  The transposed accumulators (lane = row m, 4 consecutive columns per register quad)
// go to an fp32 LDS tile Z[BM][260] with conflict-free 16-byte writes; then every wave takes whole ROWS of the tile
// (lane = 4 consecutive columns, exactly the geometry of the standalone LayerNorm kernels): residual / x^ / outputs move
// as full 1 KB (fp32) or 512 B (bf16) rows - one coalesced wave instruction each - and the row statistics are wave
// shuffles.  (A first version kept the rows lane-private and accessed global memory straight from the accumulator
// layout: 32 rows x 32 B per instruction; its epilogue alone took 47 us of a 55 us launch.)
// EPI: 0 = LayerNorm forward, 1 = LayerNorm backward.  NW = waves of the workgroup, BM = rows of the tile.
#define RG_ZLD 260
// The epilogue's global-memory operands of one wave: its RPW residual rows (fp32, 4 columns per lane) and, for the
// backward form, the x^ rows.  The LDS-DMA kernel requests them BEFORE its k-loop (rg_prefetch) so that these 30-44 MB
// per launch stream in underneath the loop, which is bound by the L2 -> LDS fill and leaves HBM two thirds idle; left
// to the epilogue they queue behind nothing and in front of 59 MB of stores, and every workgroup of the launch reaches
// that phase at the same moment.
// A wave owns the CONTIGUOUS rows [wave * RPW, (wave + 1) * RPW) of the tile, so their per-row scalars (padding flag, and
// rstd for the backward) are one load by lanes 0..RPW-1 each, read back per row with a lane broadcast: no global load is
// left inside the row loop - a load there meant an `s_waitcnt vmcnt(0)` per row, which also waited for the previous
// row's STORES (vmcnt counts both) and serialised the store latency of all 16 rows.
// RT: the types of the residual stream.  bit 0: the residual INPUT (fwd: res, bwd: dy_part) is bf16 instead of fp32;
// bit 1 (bwd): the residual OUTPUT dres is written as bf16.  (fwd: the fp32 y is simply not stored when p.y is NULL.)
template <int EPI, int RPW, int RT>
struct RgPre {
    typename std::conditional<(RT & 1) != 0, uint2, float4>::type rs[RPW];
    uint2 xs[EPI == 1 ? RPW : 1];
    float rstd_l;            // lane i < RPW: rstd of row i (EPI 1)
    int pad_l;               // lane i < RPW: padding flag of row i
};
__device__ __forceinline__ void rg_res4(const float4& v, float (&o)[4]) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
__device__ __forceinline__ void rg_res4(const uint2& v, float (&o)[4]) { rg_unpack4(v, o); }
#define RG_NPRE(EPI, RPW) ((RPW) * ((EPI) == 1 ? 2 : 1) + ((EPI) == 1 ? 2 : 1))      // vector-memory instructions of rg_prefetch
template <int EPI, int NW, int BM, int RT>
__device__ __forceinline__ void rg_prefetch(const RowGemmP& p, int m0, int wave, int lane, RgPre<EPI, BM / NW, RT>& pre) {
    constexpr int RPW = BM / NW;
    const int c4 = lane * 4;
#pragma unroll
    for (int i = 0; i < RPW; ++i) {                    // every row of this wave in flight at once
        const int row = min(m0 + wave * RPW + i, p.M - 1);
        if constexpr ((RT & 1) != 0) pre.rs[i] = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(p.res) + (long)row * RG_N + c4);
        else pre.rs[i] = *reinterpret_cast<const float4*>(p.res + (long)row * RG_N + c4);
        if constexpr (EPI == 1) pre.xs[i] = *reinterpret_cast<const uint2*>(p.xhat_in + (long)row * RG_N + c4);
    }
    // (always the same number of instructions - the LDS-DMA kernel counts them on vmcnt: without a padding mask the
    // load reads a valid dummy and is ignored)
    const int rowl = min(m0 + wave * RPW + (lane & (RPW - 1)), p.M - 1);
    const uint8_t* padp = p.row_pad ? p.row_pad + rowl : reinterpret_cast<const uint8_t*>(p.gamma);
    pre.pad_l = (int)*padp;
    if (p.row_pad == nullptr) pre.pad_l = 0;
    if constexpr (EPI == 1) pre.rstd_l = p.rstd_in[rowl];
    else pre.rstd_l = 0.f;
}

// NJ = 32-column blocks of a wave's accumulator tile (4: waves 2-wide over the 256 columns; 2: 4-wide)
// W64: the wave's tile is 64 rows x 64 columns = acc[mi * 2 + nj] (wm = row half, wc = column quarter) instead of 32 x NJ*32
// What the epilogue needs that does not depend on the tile: the dropout key of the launch (one load of the step counter) and this
// lane's four columns of bias / gamma / beta.  The LDS-DMA kernel requests them in FRONT of its first DMA issue: asked for in
// the epilogue they were two dependent memory round trips (counter -> wait -> vectors -> wait) at the end of every workgroup.
struct RgConst {
    uint64_t key_in;
    float4 bs, gm, bt;
};
template <int EPI>
__device__ __forceinline__ RgConst rg_const(const RowGemmP& p, int lane) {
    RgConst c;
    const int c4 = lane * 4;
    c.key_in = p.thr_in ? ttsmi_drop_key(p.seed, p.step_dev, p.site_in) : 0;
    c.gm = *reinterpret_cast<const float4*>(p.gamma + c4);
    if constexpr (EPI == 0) {
        c.bs = p.bias ? *reinterpret_cast<const float4*>(p.bias + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        c.bt = *reinterpret_cast<const float4*>(p.beta + c4);
    } else {
        c.bs = make_float4(0.f, 0.f, 0.f, 0.f);
        c.bt = c.bs;
    }
    return c;
}

template <int EPI, int NW, int BM, int RT, int NJ = 4, bool W64 = false>
__device__ __forceinline__ void rg_epilogue(const RowGemmP& p, f32x16 (&acc)[NJ], int m0, float* Z, int wave, int wm, int wc,
                                            int lane, const RgPre<EPI, BM / NW, RT>& pre, const RgConst& ec) {
    const int l31 = lane & 31, hh = lane >> 5;
    // ---- accumulators -> Z (all waves finished reading the operand stages: the caller synchronised)
    if constexpr (W64) {
        static_assert(NJ == 4, "64 x 64 wave tiles: four accumulators");
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int nj = 0; nj < 2; ++nj) {
                float* zr = Z + (wm * 64 + mi * 32 + l31) * RG_ZLD + wc * 64 + nj * 32 + 4 * hh;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(zr + 8 * g) = make_float4(acc[mi * 2 + nj][4 * g + 0], acc[mi * 2 + nj][4 * g + 1],
                                                                         acc[mi * 2 + nj][4 * g + 2], acc[mi * 2 + nj][4 * g + 3]);
            }
    } else {
        float* zr = Z + (wm * 32 + l31) * RG_ZLD + wc * (NJ * 32) + 4 * hh;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(zr + j * 32 + 8 * g) =
                    make_float4(acc[j][4 * g + 0], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]);
    }
    __syncthreads();
    constexpr int RPW = BM / NW;                       // rows per wave
    const int c4 = lane * 4;
    const uint64_t key_in = ec.key_in;
    const float invC = 1.0f / (float)RG_N;
    if constexpr (EPI == 0) {
        const float4 bs = ec.bs, gm = ec.gm, bt = ec.bt;
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int rl = wave * RPW + i, row = m0 + rl;
            if (row >= p.M) break;                                     // wave-uniform
            float rs4[4];
            rg_res4(pre.rs[i], rs4);
            const float4 a = *reinterpret_cast<const float4*>(Z + rl * RG_ZLD + c4);
            float v[4] = {a.x + bs.x, a.y + bs.y, a.z + bs.z, a.w + bs.w};
            if (p.thr_in) {
                const uint32_t rb = ttsmi_row_base(key_in, (uint32_t)row);
                const uint32_t h0 = ttsmi_pair_hash(rb, (uint32_t)c4), h1 = ttsmi_pair_hash(rb, (uint32_t)(c4 + 2));
                v[0] *= ((h0 & 0xFFFFu) >= p.thr_in) ? p.inv_in : 0.f;
                v[1] *= ((h0 >> 16) >= p.thr_in) ? p.inv_in : 0.f;
                v[2] *= ((h1 & 0xFFFFu) >= p.thr_in) ? p.inv_in : 0.f;
                v[3] *= ((h1 >> 16) >= p.thr_in) ? p.inv_in : 0.f;
            }
            v[0] += rs4[0]; v[1] += rs4[1]; v[2] += rs4[2]; v[3] += rs4[3];
            const float mean = wave_sum_dpp(v[0] + v[1] + v[2] + v[3]) * invC;
            float q = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] -= mean; q += v[e] * v[e]; }
            // v_rsq_f32 (1 ulp) instead of 1 / sqrtf: the library pair is ~25 dependent instructions in a row loop that is a
            // dependency chain (sum -> statistics -> normalise), 16 rows per wave; this is the bf16 path (bf16 outputs)
            const float rstd = __builtin_amdgcn_rsqf(wave_sum_dpp(q) * invC + p.eps);
            if (lane == 0) p.rstd[row] = rstd;
            const bool padded = __builtin_amdgcn_readlane(pre.pad_l, i) != 0;
            float xh[4], yv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) xh[e] = v[e] * rstd;
            yv[0] = xh[0] * gm.x + bt.x; yv[1] = xh[1] * gm.y + bt.y; yv[2] = xh[2] * gm.z + bt.z; yv[3] = xh[3] * gm.w + bt.w;
            if (padded) { yv[0] = 0.f; yv[1] = 0.f; yv[2] = 0.f; yv[3] = 0.f; }
            const long o = (long)row * RG_N + c4;
            if (TTSMI_ABLATE_BITS(p.ablate) & 8) continue;                     // (measurement: the epilogue without its global stores)
            if (p.y != nullptr) *reinterpret_cast<float4*>(p.y + o) = make_float4(yv[0], yv[1], yv[2], yv[3]);   // (wave-uniform)
            *reinterpret_cast<uint2*>(p.y_bf + o) = rg_pack4(yv[0], yv[1], yv[2], yv[3]);
            *reinterpret_cast<uint2*>(p.xhat + o) = rg_pack4(xh[0], xh[1], xh[2], xh[3]);
        }
    } else {
        // dy = acc + partial;  g = dy * rowmask;  t = g * gamma;  dz = rstd (t - mean(t) - x^ mean(t x^));
        // d_o = keep(dz) (bf16), dres = dz (fp32); dgamma += g x^, dbeta += g summed over the tile's rows
        const float4 gm = ec.gm;
        float ag[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
        const uint2 (&xs)[RPW] = pre.xs;
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int rl = wave * RPW + i, row = m0 + rl;
            if (row >= p.M) break;
            const float4 a = *reinterpret_cast<const float4*>(Z + rl * RG_ZLD + c4);
            const bool padded = __builtin_amdgcn_readlane(pre.pad_l, i) != 0;
            const float rstd = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pre.rstd_l), i));
            float rs4[4];
            rg_res4(pre.rs[i], rs4);
            float gv[4] = {a.x + rs4[0], a.y + rs4[1], a.z + rs4[2], a.w + rs4[3]};
            if (padded) { gv[0] = 0.f; gv[1] = 0.f; gv[2] = 0.f; gv[3] = 0.f; }
            float xh[4];
            rg_unpack4(xs[i], xh);
            const long o = (long)row * RG_N + c4;
#pragma unroll
            for (int e = 0; e < 4; ++e) { ab[e] += gv[e]; ag[e] += gv[e] * xh[e]; }
            const float t[4] = {gv[0] * gm.x, gv[1] * gm.y, gv[2] * gm.z, gv[3] * gm.w};
            const float m1 = wave_sum_dpp(t[0] + t[1] + t[2] + t[3]) * invC;
            const float m2 = wave_sum_dpp(t[0] * xh[0] + t[1] * xh[1] + t[2] * xh[2] + t[3] * xh[3]) * invC;
            float dz[4], dx[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) dz[e] = rstd * (t[e] - m1 - xh[e] * m2);
            if (p.thr_in) {
                const uint32_t rb = ttsmi_row_base(key_in, (uint32_t)row);
                const uint32_t h0 = ttsmi_pair_hash(rb, (uint32_t)c4), h1 = ttsmi_pair_hash(rb, (uint32_t)(c4 + 2));
                dx[0] = dz[0] * (((h0 & 0xFFFFu) >= p.thr_in) ? p.inv_in : 0.f);
                dx[1] = dz[1] * (((h0 >> 16) >= p.thr_in) ? p.inv_in : 0.f);
                dx[2] = dz[2] * (((h1 & 0xFFFFu) >= p.thr_in) ? p.inv_in : 0.f);
                dx[3] = dz[3] * (((h1 >> 16) >= p.thr_in) ? p.inv_in : 0.f);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) dx[e] = dz[e];
            }
            if constexpr ((RT & 2) != 0) *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.dres) + o) = rg_pack4(dz[0], dz[1], dz[2], dz[3]);
            else *reinterpret_cast<float4*>(p.dres + o) = make_float4(dz[0], dz[1], dz[2], dz[3]);
            *reinterpret_cast<uint2*>(p.dx_bf + o) = rg_pack4(dx[0], dx[1], dx[2], dx[3]);
        }
        // parameter-gradient partials of the tile: waves summed through LDS in wave order (deterministic), one partial
        // row per workgroup in the layout ttsmi_layernorm_param_reduce_batched_nw reads
        __syncthreads();                                   // every wave is done with its rows of Z
        float* red = Z;                                    // [NW - 1][2][256]
        if (wave > 0) {
            *reinterpret_cast<float4*>(red + ((wave - 1) * 2 + 0) * RG_N + c4) = make_float4(ag[0], ag[1], ag[2], ag[3]);
            *reinterpret_cast<float4*>(red + ((wave - 1) * 2 + 1) * RG_N + c4) = make_float4(ab[0], ab[1], ab[2], ab[3]);
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int u = 0; u < NW - 1; ++u) {
                const float4 a = *reinterpret_cast<const float4*>(red + (u * 2 + 0) * RG_N + c4);
                const float4 b = *reinterpret_cast<const float4*>(red + (u * 2 + 1) * RG_N + c4);
                ag[0] += a.x; ag[1] += a.y; ag[2] += a.z; ag[3] += a.w;
                ab[0] += b.x; ab[1] += b.y; ab[2] += b.z; ab[3] += b.w;
            }
            const int blk = m0 / BM;
            *reinterpret_cast<float4*>(p.part + (long)blk * RG_N + c4) = make_float4(ag[0], ag[1], ag[2], ag[3]);
            *reinterpret_cast<float4*>(p.part + ((long)p.nparts + blk) * RG_N + c4) = make_float4(ab[0], ab[1], ab[2], ab[3]);
        }
    }
}

// EPI: 0 = LayerNorm forward, 1 = LayerNorm backward
template <int EPI, int RT = 0>
__global__ __launch_bounds__(256, 2) void rowgemm_kernel(RowGemmP p) {
    constexpr int TILE_BYTES = (RG_BM + RG_N) * RG_LD * 2, Z_BYTES = RG_BM * RG_ZLD * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[TILE_BYTES > Z_BYTES ? TILE_BYTES : Z_BYTES];
    uint16_t(*As)[RG_LD] = reinterpret_cast<uint16_t(*)[RG_LD]>(smem);
    uint16_t(*Bs)[RG_LD] = As + RG_BM;

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wc = wave & 1;
    const int m0 = blockIdx.x * RG_BM;

    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    uint4 ra[RG_BM * RG_BK / 2048], rb[RG_N * RG_BK / 2048];
#define RG_FETCH(k0_)                                                                                   \
    do {                                                                                                \
        const int kk_ = (k0_);                                                                          \
        const bool second_ = p.A2 != nullptr && kk_ >= p.K1;                                            \
        rg_fetch<RG_BM>(second_ ? p.A2 : p.A, second_ ? p.lda2 : p.lda, p.M, m0, second_ ? kk_ - p.K1 : kk_, tid, ra); \
        if (second_ && p.Bt2 != nullptr) rg_fetch<RG_N>(p.Bt2, p.ldb2, RG_N, 0, kk_ - p.K1, tid, rb);   \
        else rg_fetch<RG_N>(p.Bt, p.ldb, RG_N, 0, kk_, tid, rb);                                        \
    } while (0)
    RG_FETCH(0);
    rg_stash<RG_BM>(As, tid, ra);
    rg_stash<RG_N>(Bs, tid, rb);
    __syncthreads();
    for (int k0 = 0; k0 < p.K; k0 += RG_BK) {
        const bool more = k0 + RG_BK < p.K;
        if (more) RG_FETCH(k0 + RG_BK);
#pragma unroll
        for (int ks = 0; ks < RG_BK / 16; ++ks) {
            const int ko = ks * 16 + hh * 8;
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(&As[wm * 32 + l31][ko]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(&Bs[wc * 128 + j * 32 + l31][ko]);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc[j], 0, 0, 0);      // C^T: rows = n, cols = m
            }
        }
        __syncthreads();
        if (more) {
            rg_stash<RG_BM>(As, tid, ra);
            rg_stash<RG_N>(Bs, tid, rb);
        }
        __syncthreads();
    }

    RgPre<EPI, RG_BM / 4, RT> pre;
    rg_prefetch<EPI, 4, RG_BM, RT>(p, m0, wave, lane, pre);
    const RgConst ec = rg_const<EPI>(p, lane);
    rg_epilogue<EPI, 4, RG_BM, RT>(p, acc, m0, reinterpret_cast<float*>(smem), wave, wm, wc, lane, pre, ec);
}

// =================================================================================================
// 128-row LDS-DMA variant for the decoder-size launches (M >= 256 * 64): ONE 8-wave workgroup per CU owns a
// 128 x 256 tile, halving the per-tile re-read of W (the 64-row kernel above pulls all of W through L2 for every 64
// rows: 289 MB of L2 -> LDS traffic per K = 1024 launch, at which it ran 64 us against 52 us for the unfused pair).
// With a single workgroup per CU nothing else hides memory latency, so the k-tiles arrive by `global_load_lds_dwordx4`
// into a THREE-stage ring (48 KB per stage: A 128 x 64, W 256 x 64 bf16), two k-steps in flight while one is
// multiplied - the structure of wgrad_dma_kernel (gemm_bf16.hip): counted vmcnt, one `lgkmcnt(0) + s_barrier` per step.
// The DMA image is lane-linear: dense 128-byte rows; chunk c of row r is stored at chunk position c ^ ((r >> 1) & 7)
// (source-side swizzle, free), which makes the ds_read_b128 fragment reads conflict free (see gemm_bf16_dma_kernel).
// =================================================================================================
#define RD_BM 128
#define RD_STAGE ((RD_BM + RG_N) * RG_BK * 2)      // 49 152 bytes
#define RD_STAGES 3

__device__ __forceinline__ void rd_dma16(const void* gsrc, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_off) : "memory", "m0");
}
__device__ __forceinline__ void rd_stage_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ unsigned rd_lds_offset(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}

// BM = 128: waves 4 (rows) x 2 (columns), wave tile 32 x 128.  BM = 64: waves 2 x 4, wave tile 32 x 64 - for launches whose
// 128-row tiles would occupy fewer than half the CUs (the encoder side, M = 6 400: 50 workgroups): twice the workgroups,
// each with a shorter fill per k-step (40 KB) and half the epilogue.
// W64 (BM = 128 only): wave tiles of 64 x 64 (waves 2 x 4) - 8 fragment reads per 8 MFMAs instead of 10 (the k-loop is
// paced by LDS traffic together with the fill: 160 KB of fragment reads + 48 KB of DMA writes per k-step and workgroup)
template <int EPI, int BM, int RT = 0, bool W64 = false>
__global__ __launch_bounds__(512, 1) void rowgemm_dma_kernel(RowGemmP p) {
    static_assert(!W64 || BM == 128, "64 x 64 wave tiles need the 128-row workgroup tile");
    constexpr int NJ = BM == 128 ? 4 : 2;                  // 32-column blocks per wave
    constexpr int STAGE = (BM + RG_N) * RG_BK * 2;
    constexpr int DMA_A = BM / 64;                         // A-image DMA instructions per wave and k-step (+ 4 for W)
    static_assert(BM == 128 || BM == 64, "row tiles of 128 or 64");
    static_assert(RD_STAGES * STAGE >= BM * RG_ZLD * 4, "the epilogue tile lives in the retired stages");
    __shared__ __attribute__((aligned(16))) unsigned char smem[RD_STAGES * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);                      // 0..7
    const int wm = W64 ? wave >> 2 : (BM == 128 ? wave >> 1 : wave >> 2), wc = W64 ? wave & 3 : (BM == 128 ? wave & 1 : wave & 3);
    const int m0 = blockIdx.x * BM;
    const int nk = p.K / RG_BK;
    const RgConst ec = rg_const<EPI>(p, lane);          // (older than every DMA instruction: the counted waits below are unaffected)

    f32x16 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // DMA: one wave instruction = 8 rows x 8 chunks of 16 bytes.  Per k-step wave w moves rows [16 w, 16 w + 16) of the
    // A image (2 instructions) and rows [32 w, 32 w + 32) of the W image (4 instructions): 6 per wave and step.
    const int drow = lane >> 3, dpos = lane & 7;
    auto issue = [&](int ks, int stage) {
        unsigned char* As = smem + stage * STAGE;
        unsigned char* Bs = As + BM * 128;
        int k0 = ks * RG_BK;
        const uint16_t* Ab = p.A;
        long lda = p.lda;
        if (p.A2 != nullptr && k0 >= p.K1) { Ab = p.A2; lda = p.lda2; k0 -= p.K1; }
#pragma unroll
        for (int i = 0; i < DMA_A; ++i) {
            const int row = wave * (8 * DMA_A) + i * 8 + drow;
            const int c = dpos ^ ((row >> 1) & 7);
            const int gm = min(m0 + row, p.M - 1);                   // rows past M: clamped on load, dropped on store
            rd_dma16(Ab + (long)gm * lda + k0 + c * 8, rd_lds_offset(As + (wave * (8 * DMA_A) + i * 8) * 128));
        }
        int kb = ks * RG_BK;
        const uint16_t* Bb = p.Bt;
        long ldb = p.ldb;
        if (p.Bt2 != nullptr && kb >= p.K1) { Bb = p.Bt2; ldb = p.ldb2; kb -= p.K1; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wave * 32 + i * 8 + drow;
            const int c = dpos ^ ((row >> 1) & 7);
            rd_dma16(Bb + (long)row * ldb + kb + c * 8, rd_lds_offset(Bs + (wave * 32 + i * 8) * 128));
        }
    };
    const bool dma_on = !(TTSMI_ABLATE_BITS(p.ablate) & 2);
    if (dma_on) {
        issue(0, 0);
        if (nk > 1) issue(1, 1);
    }
    // the epilogue's residual / x^ rows, requested now: NPRE more instructions on this wave's vmcnt, YOUNGER than k-steps
    // 0 and 1 and older than every later one
    constexpr int NPRE = RG_NPRE(EPI, BM / 8), NDMA = 4 + DMA_A;
    static_assert(NDMA + NPRE <= 63, "vmcnt is a 6-bit counter");
    RgPre<EPI, BM / 8, RT> pre;
    rg_prefetch<EPI, 8, BM, RT>(p, m0, wave, lane, pre);
    for (int ks = 0; ks < nk; ++ks) {
        // step ks has landed once at most the newest step's 6 DMA instructions (and, for the first two steps, the
        // prefetch behind them) are outstanding
        if (ks + 1 >= nk) __builtin_amdgcn_s_waitcnt(0xF70);         // vmcnt(0)
        else if (ks < 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA + NPRE) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        rd_stage_barrier();                                          // everybody's pieces landed; stage (ks-1)%3 is retired
        if (dma_on && ks + 2 < nk) issue(ks + 2, (ks + 2) % RD_STAGES);
        const unsigned char* As = smem + (ks % RD_STAGES) * STAGE;
        const unsigned char* Bs = As + BM * 128;
        if (TTSMI_ABLATE_BITS(p.ablate) & 1) continue;
        // Fragments of TWO 16-wide k-slices (10 x 16 B per lane) are requested before the first of their 8 MFMAs issues:
        // written one fragment at a time, hipcc keeps a single ds_read ahead of each MFMA and the matrix pipe waits out an
        // LDS round trip per multiply (23 % busy in the k-loop: it, not the fill, was what bounded this kernel).
        if constexpr (W64) {
#pragma unroll
            for (int k2 = 0; k2 < RG_BK / 32; ++k2) {
                bf16x8 a[2][2], b[2][2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int c = (k2 * 2 + u) * 2 + hh;
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int arow = wm * 64 + t * 32 + l31, brow = wc * 64 + t * 32 + l31;
                        a[u][t] = *reinterpret_cast<const bf16x8*>(As + arow * 128 + ((c ^ ((arow >> 1) & 7)) << 4));
                        b[u][t] = *reinterpret_cast<const bf16x8*>(Bs + brow * 128 + ((c ^ ((brow >> 1) & 7)) << 4));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);        // all eight reads are in flight before the multiplies start
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int nj = 0; nj < 2; ++nj)
                            acc[mi * 2 + nj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[u][nj], a[u][mi], acc[mi * 2 + nj], 0, 0, 0);
            }
            continue;
        }
#pragma unroll
        for (int k2 = 0; k2 < RG_BK / 32; ++k2) {
            bf16x8 a[2], b[2][NJ];
            const int arow = wm * 32 + l31;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int c = (k2 * 2 + u) * 2 + hh;
                a[u] = *reinterpret_cast<const bf16x8*>(As + arow * 128 + ((c ^ ((arow >> 1) & 7)) << 4));
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int brow = wc * (NJ * 32) + j * 32 + l31;
                    b[u][j] = *reinterpret_cast<const bf16x8*>(Bs + brow * 128 + ((c ^ ((brow >> 1) & 7)) << 4));
                }
            }
            __builtin_amdgcn_sched_barrier(0);            // all ten reads are in flight before the multiplies start
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[u][j], a[u], acc[j], 0, 0, 0);
        }
    }
    __syncthreads();
    if (TTSMI_ABLATE_BITS(p.ablate) & 4) return;
    rg_epilogue<EPI, 8, BM, RT, NJ, W64>(p, acc, m0, reinterpret_cast<float*>(smem), wave, wm, wc, lane, pre, ec);
}

// ---- standalone backward of a LayerNorm whose forward kept x^ (bf16) and rstd: the fused forward's counterpart for the
// LayerNorms whose upstream gradient is not a full-row GEMM result.  A wave walks rows (as ln_bwd_kernel), 256 columns:
// 4 per lane; dgamma / dbeta partial sums stay in registers across its rows and leave as one partial row per workgroup.
#define LNX_MAX_BLOCKS 1024
static int lnx_blocks(int M) { int b = ttsmi_cdiv(M, 4); return b > LNX_MAX_BLOCKS ? LNX_MAX_BLOCKS : (b < 1 ? 1 : b); }

template <bool OUT_H>
__global__ __launch_bounds__(256) void ln_bwd_xhat_kernel(RowGemmP p, const float* __restrict__ dy) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwaves = gridDim.x * 4;
    const int c4 = lane * 4;
    const float4 gm = *reinterpret_cast<const float4*>(p.gamma + c4);
    const uint64_t key_in = p.thr_in ? ttsmi_drop_key(p.seed, p.step_dev, p.site_in) : 0;
    float ag[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
    for (int row = blockIdx.x * 4 + wave; row < p.M; row += nwaves) {
        const long base = (long)row * RG_N + c4;
        const bool padded = p.row_pad != nullptr && p.row_pad[row] != 0;
        const float rstd = p.rstd_in[row];
        const float4 g4 = *reinterpret_cast<const float4*>(dy + base);
        float xh[4];
        rg_unpack4(*reinterpret_cast<const uint2*>(p.xhat_in + base), xh);
        float gv[4] = {g4.x, g4.y, g4.z, g4.w};
        if (padded) { gv[0] = 0.f; gv[1] = 0.f; gv[2] = 0.f; gv[3] = 0.f; }
#pragma unroll
        for (int e = 0; e < 4; ++e) { ab[e] += gv[e]; ag[e] += gv[e] * xh[e]; }
        const float t[4] = {gv[0] * gm.x, gv[1] * gm.y, gv[2] * gm.z, gv[3] * gm.w};
        const float s1 = wave_sum_dpp(t[0] + t[1] + t[2] + t[3]) * (1.0f / RG_N);
        const float s2 = wave_sum_dpp(t[0] * xh[0] + t[1] * xh[1] + t[2] * xh[2] + t[3] * xh[3]) * (1.0f / RG_N);
        float dz[4], dx[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) dz[e] = rstd * (t[e] - s1 - xh[e] * s2);
        if (p.thr_in) {
            const uint32_t rb_in = ttsmi_row_base(key_in, (uint32_t)row);
            const uint32_t h0 = ttsmi_pair_hash(rb_in, (uint32_t)c4), h1 = ttsmi_pair_hash(rb_in, (uint32_t)(c4 + 2));
            dx[0] = dz[0] * (((h0 & 0xFFFFu) >= p.thr_in) ? p.inv_in : 0.f);
            dx[1] = dz[1] * (((h0 >> 16) >= p.thr_in) ? p.inv_in : 0.f);
            dx[2] = dz[2] * (((h1 & 0xFFFFu) >= p.thr_in) ? p.inv_in : 0.f);
            dx[3] = dz[3] * (((h1 >> 16) >= p.thr_in) ? p.inv_in : 0.f);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) dx[e] = dz[e];
        }
        if constexpr (OUT_H) *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.dres) + base) = rg_pack4(dz[0], dz[1], dz[2], dz[3]);
        else *reinterpret_cast<float4*>(p.dres + base) = make_float4(dz[0], dz[1], dz[2], dz[3]);
        *reinterpret_cast<uint2*>(p.dx_bf + base) = rg_pack4(dx[0], dx[1], dx[2], dx[3]);
    }
    __shared__ float red[3][2][RG_N];
    if (wave > 0) {
        *reinterpret_cast<float4*>(&red[wave - 1][0][c4]) = make_float4(ag[0], ag[1], ag[2], ag[3]);
        *reinterpret_cast<float4*>(&red[wave - 1][1][c4]) = make_float4(ab[0], ab[1], ab[2], ab[3]);
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const float4 a = *reinterpret_cast<const float4*>(&red[u][0][c4]);
            const float4 b = *reinterpret_cast<const float4*>(&red[u][1][c4]);
            ag[0] += a.x; ag[1] += a.y; ag[2] += a.z; ag[3] += a.w;
            ab[0] += b.x; ab[1] += b.y; ab[2] += b.z; ab[3] += b.w;
        }
        *reinterpret_cast<float4*>(p.part + (long)blockIdx.x * RG_N + c4) = make_float4(ag[0], ag[1], ag[2], ag[3]);
        *reinterpret_cast<float4*>(p.part + ((long)p.nparts + blockIdx.x) * RG_N + c4) = make_float4(ab[0], ab[1], ab[2], ab[3]);
    }
}

static bool rg_al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

static int rg_common(RowGemmP& p, const uint16_t* a, int64_t lda, const uint16_t* a2, int64_t lda2, int K1, const uint16_t* bt,
                     int64_t ldb, int M, int N, int K, const char* who) {
    TTSMI_CHECK_ARG(a && bt && M > 0, "%s: null pointer / empty", who);
    TTSMI_CHECK_ARG(N == RG_N, "%s: built for N = %d output columns (got %d)", who, RG_N, N);
    TTSMI_CHECK_ARG(K > 0 && K % RG_BK == 0 && lda % 8 == 0 && ldb % 8 == 0 && rg_al16(a) && rg_al16(bt),
                    "%s: K %% 64 == 0, leading dimensions %% 8 == 0 and 16-byte aligned operands required", who);
    if (a2) TTSMI_CHECK_ARG(K1 > 0 && K1 < K && K1 % RG_BK == 0 && lda2 % 8 == 0 && rg_al16(a2), "%s: bad second K segment", who);
    memset(&p, 0, sizeof(p));
    p.A = a; p.lda = lda; p.A2 = a2; p.lda2 = lda2; p.K1 = K1; p.Bt = bt; p.ldb = ldb; p.M = M; p.K = K;
    TTSMI_ABLATE_KNOB(ablate, "TTSMI_ROWGEMM_ABLATE");
    p.ablate = ablate;
    return TTSMI_OK;
}

// TTSMI_ROWGEMM_DMA: 0 = always the 64-row register-staged kernel, 1 = always the 128-row LDS-DMA kernel,
// default = by size.  Measured (tools/kbench.py --only rowgemm): the LDS-DMA kernel wins at M = 28 800 (32.7 / 45.2 /
// 44.3 us for o+LN1, FFN2+LN2, dgrad+LN1' against 34.6 / 56.9 / 54.8) and still at M = 6 400, where it fills only 50 CUs
// (26.5 / 33.4 / 30.7 against 27.9 / 37.2 / 34.1): the 64-row kernel is kept for small batches.
// rows per tile of the LDS-DMA kernel: 64 when 128-row tiles would leave more than half of the 256 CUs without a workgroup
static int rg_dma_bm(int M) {
    TTSMI_KNOB(forced, "TTSMI_ROWGEMM_BM", 0);
    if (forced == 64 || forced == 128) return forced;
    return ttsmi_cdiv(M, RD_BM) < 128 ? 64 : RD_BM;
}
static bool rg_use_dma(int M) {
    TTSMI_KNOB(mode, "TTSMI_ROWGEMM_DMA", 2);
    if (mode == 0) return false;
    if (mode == 1) return true;
    return M >= 16 * RD_BM;
}

static void rg_drop(RowGemmP& p, float p_in, uint32_t site_in, uint64_t seed, const int64_t* step_dev) {
    p.thr_in = p_in > 0.f ? ttsmi_drop_threshold(p_in) : 0;
    p.inv_in = p_in > 0.f ? 1.0f / (1.0f - p_in) : 1.f;
    p.seed = seed; p.step_dev = step_dev; p.site_in = site_in;
}

// launch by (rows, residual types): the kernel names reported through ttsmi_last_kernel keep their round-2 spelling for RT = 0
template <int EPI, int RT>
static void rg_launch(const RowGemmP& p, int M, ttsmi_stream_t stream) {
    static const char* const names[3] = {
        EPI == 0 ? (RT ? "rowgemm_dma_kernel<0, 64, 1>" : "rowgemm_dma_kernel<0, 64>")
                 : (RT == 3 ? "rowgemm_dma_kernel<1, 64, 3>" : RT == 1 ? "rowgemm_dma_kernel<1, 64, 1>" : "rowgemm_dma_kernel<1, 64>"),
        EPI == 0 ? (RT ? "rowgemm_dma_kernel<0, 128, 1>" : "rowgemm_dma_kernel<0, 128>")
                 : (RT == 3 ? "rowgemm_dma_kernel<1, 128, 3>" : RT == 1 ? "rowgemm_dma_kernel<1, 128, 1>" : "rowgemm_dma_kernel<1, 128>"),
        EPI == 0 ? (RT ? "rowgemm_kernel<0, 1>" : "rowgemm_kernel<0>")
                 : (RT == 3 ? "rowgemm_kernel<1, 3>" : RT == 1 ? "rowgemm_kernel<1, 1>" : "rowgemm_kernel<1>")};
    if (rg_use_dma(M)) {
        if (rg_dma_bm(M) == 64) {
            ttsmi_note_kernel(names[0]);
            TTSMI_LAUNCH_EV((rowgemm_dma_kernel<EPI, 64, RT>), dim3(ttsmi_cdiv(M, 64)), dim3(512), 0, (hipStream_t)stream, p);
        } else {
            ttsmi_note_kernel(names[1]);
            TTSMI_KNOB(w64, "TTSMI_ROWGEMM_W64", 1);        // A/B knob: 0 = wave tiles of 32 x 128 (rounds 2-3)
            if (w64) TTSMI_LAUNCH_EV((rowgemm_dma_kernel<EPI, RD_BM, RT, true>), dim3(ttsmi_cdiv(M, RD_BM)), dim3(512), 0, (hipStream_t)stream, p);
            else
            TTSMI_LAUNCH_EV((rowgemm_dma_kernel<EPI, RD_BM, RT>), dim3(ttsmi_cdiv(M, RD_BM)), dim3(512), 0, (hipStream_t)stream, p);
        }
    } else {
        ttsmi_note_kernel(names[2]);
        TTSMI_LAUNCH_EV((rowgemm_kernel<EPI, RT>), dim3(ttsmi_cdiv(M, RG_BM)), dim3(256), 0, (hipStream_t)stream, p);
    }
}

static int rg_fwd(const uint16_t* a, int64_t lda, const uint16_t* a2, int64_t lda2, int K1, const uint16_t* bt, int64_t ldb,
                  const float* bias, const void* res, bool res_h, const float* gamma, const float* beta, const uint8_t* row_pad,
                  float p_in, uint32_t site_in, uint64_t seed, const int64_t* step_dev, float eps, float* y, uint16_t* y_bf16,
                  uint16_t* xhat_bf16, float* rstd, int M, int N, int K, ttsmi_stream_t stream) {
    RowGemmP p;
    int rc = rg_common(p, a, lda, a2, lda2, K1, bt, ldb, M, N, K, "hgemm_ln_fwd");
    if (rc) return rc;
    TTSMI_CHECK_ARG(res && gamma && beta && (y || res_h) && y_bf16 && xhat_bf16 && rstd, "hgemm_ln_fwd: null pointer");
    TTSMI_CHECK_ARG(p_in >= 0.f && p_in < 1.f, "hgemm_ln_fwd: dropout rate out of [0,1)");
    p.bias = bias; p.res = (const float*)res; p.gamma = gamma; p.beta = beta; p.row_pad = row_pad; p.eps = eps;
    p.y = y; p.y_bf = y_bf16; p.xhat = xhat_bf16; p.rstd = rstd;
    rg_drop(p, p_in, site_in, seed, step_dev);
    if (res_h) rg_launch<0, 1>(p, M, stream); else rg_launch<0, 0>(p, M, stream);
    TTSMI_CHECK_LAUNCH("hgemm_ln_fwd");
    return TTSMI_OK;
}

static int rg_bwd(const uint16_t* a, int64_t lda, const uint16_t* a2, int64_t lda2, int K1, const uint16_t* bt, int64_t ldb,
                  const uint16_t* bt2, int64_t ldb2, const void* dy_part, bool part_h, const uint16_t* xhat_bf16, const float* rstd,
                  const float* gamma, const uint8_t* row_pad, float p_in, uint32_t site_in, uint64_t seed, const int64_t* step_dev,
                  uint16_t* dx_bf16, void* dres, bool dres_h, void* part_ws, size_t part_ws_bytes, int M, int N, int K,
                  ttsmi_stream_t stream) {
    RowGemmP p;
    int rc = rg_common(p, a, lda, a2, lda2, K1, bt, ldb, M, N, K, "hgemm_ln_bwd");
    if (rc) return rc;
    TTSMI_CHECK_ARG((a2 != nullptr) == (bt2 != nullptr), "hgemm_ln_bwd: the second K segment needs both of its operands");
    if (bt2) TTSMI_CHECK_ARG(ldb2 % 8 == 0 && rg_al16(bt2), "hgemm_ln_bwd: bad second weight matrix");
    TTSMI_CHECK_ARG(part_h || !dres_h, "hgemm_ln_bwd: a bf16 dres needs a bf16 dy_part (residual types: fp32/fp32, bf16/fp32, bf16/bf16)");
    p.Bt2 = bt2; p.ldb2 = ldb2;
    TTSMI_CHECK_ARG(dy_part && xhat_bf16 && rstd && gamma && dx_bf16 && dres && part_ws, "hgemm_ln_bwd: null pointer");
    p.nparts = ttsmi_hgemm_ln_bwd_nparts(M);
    TTSMI_CHECK_ARG(part_ws_bytes >= ttsmi_layernorm_partials_bytes(p.nparts, N), "hgemm_ln_bwd: partial-sum workspace too small");
    p.res = (const float*)dy_part; p.xhat_in = xhat_bf16; p.rstd_in = rstd; p.gamma = gamma; p.row_pad = row_pad;
    p.dx_bf = dx_bf16; p.dres = (float*)dres; p.part = (float*)part_ws;
    rg_drop(p, p_in, site_in, seed, step_dev);
    if (part_h && dres_h) rg_launch<1, 3>(p, M, stream);
    else if (part_h) rg_launch<1, 1>(p, M, stream);
    else rg_launch<1, 0>(p, M, stream);
    TTSMI_CHECK_LAUNCH("hgemm_ln_bwd");
    return TTSMI_OK;
}

static int rg_lnx(const float* dy, const uint16_t* xhat_bf16, const float* rstd, const float* gamma, const uint8_t* row_pad,
                  float p_in, uint32_t site_in, uint64_t seed, const int64_t* step_dev, uint16_t* dx_bf16, void* dres, bool dres_h,
                  void* part_ws, size_t part_ws_bytes, int M, int C, ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(dy && xhat_bf16 && rstd && gamma && dx_bf16 && dres && part_ws && M > 0, "layernorm_bwd_xhat: null pointer");
    TTSMI_CHECK_ARG(C == RG_N, "layernorm_bwd_xhat: built for C = %d (got %d)", RG_N, C);
    RowGemmP p;
    memset(&p, 0, sizeof(p));
    p.nparts = lnx_blocks(M);
    TTSMI_CHECK_ARG(part_ws_bytes >= ttsmi_layernorm_partials_bytes(p.nparts, C), "layernorm_bwd_xhat: partial-sum workspace too small");
    p.M = M; p.xhat_in = xhat_bf16; p.rstd_in = rstd; p.gamma = gamma; p.row_pad = row_pad;
    p.dx_bf = dx_bf16; p.dres = (float*)dres; p.part = (float*)part_ws;
    rg_drop(p, p_in, site_in, seed, step_dev);
    if (dres_h) TTSMI_LAUNCH_EV(ln_bwd_xhat_kernel<true>, dim3(p.nparts), dim3(256), 0, (hipStream_t)stream, p, dy);
    else TTSMI_LAUNCH_EV(ln_bwd_xhat_kernel<false>, dim3(p.nparts), dim3(256), 0, (hipStream_t)stream, p, dy);
    TTSMI_CHECK_LAUNCH("layernorm_bwd_xhat");
    return TTSMI_OK;
}

extern "C" {

int ttsmi_hgemm_ln_fwd(const uint16_t* a, int64_t lda, const uint16_t* a2, int64_t lda2, int K1, const uint16_t* bt,
                       int64_t ldb, const float* bias, const float* res, const float* gamma, const float* beta,
                       const uint8_t* row_pad, float p_in, uint32_t site_in, uint64_t seed, const int64_t* step_dev,
                       float eps, float* y, uint16_t* y_bf16, uint16_t* xhat_bf16, float* rstd, int M, int N, int K,
                       ttsmi_stream_t stream) {
    return rg_fwd(a, lda, a2, lda2, K1, bt, ldb, bias, res, false, gamma, beta, row_pad, p_in, site_in, seed, step_dev, eps, y,
                  y_bf16, xhat_bf16, rstd, M, N, K, stream);
}
int ttsmi_hgemm_ln_fwd_h(const uint16_t* a, int64_t lda, const uint16_t* a2, int64_t lda2, int K1, const uint16_t* bt,
                         int64_t ldb, const float* bias, const uint16_t* res_bf16, const float* gamma, const float* beta,
                         const uint8_t* row_pad, float p_in, uint32_t site_in, uint64_t seed, const int64_t* step_dev,
                         float eps, float* y, uint16_t* y_bf16, uint16_t* xhat_bf16, float* rstd, int M, int N, int K,
                         ttsmi_stream_t stream) {
    return rg_fwd(a, lda, a2, lda2, K1, bt, ldb, bias, res_bf16, true, gamma, beta, row_pad, p_in, site_in, seed, step_dev, eps, y,
                  y_bf16, xhat_bf16, rstd, M, N, K, stream);
}

size_t ttsmi_layernorm_partials_bytes(int nparts, int C) { return (2 * (size_t)nparts * C + nparts) * sizeof(float) + 256; }

int ttsmi_hgemm_ln_bwd_nparts(int M) { return rg_use_dma(M) ? ttsmi_cdiv(M, rg_dma_bm(M)) : ttsmi_cdiv(M, RG_BM); }

int ttsmi_hgemm_ln_bwd(const uint16_t* a, int64_t lda, const uint16_t* bt, int64_t ldb, const float* dy_part,
                       const uint16_t* xhat_bf16, const float* rstd, const float* gamma, const uint8_t* row_pad, float p_in,
                       uint32_t site_in, uint64_t seed, const int64_t* step_dev, uint16_t* dx_bf16, float* dres,
                       void* part_ws, size_t part_ws_bytes, int M, int N, int K, ttsmi_stream_t stream) {
    return rg_bwd(a, lda, nullptr, 0, 0, bt, ldb, nullptr, 0, dy_part, false, xhat_bf16, rstd, gamma, row_pad, p_in, site_in, seed,
                  step_dev, dx_bf16, dres, false, part_ws, part_ws_bytes, M, N, K, stream);
}

int ttsmi_hgemm_ln_bwd_dual(const uint16_t* a, int64_t lda, const uint16_t* a2, int64_t lda2, int K1, const uint16_t* bt,
                            int64_t ldb, const uint16_t* bt2, int64_t ldb2, const float* dy_part, const uint16_t* xhat_bf16,
                            const float* rstd, const float* gamma, const uint8_t* row_pad, float p_in, uint32_t site_in,
                            uint64_t seed, const int64_t* step_dev, uint16_t* dx_bf16, float* dres, void* part_ws,
                            size_t part_ws_bytes, int M, int N, int K, ttsmi_stream_t stream) {
    return rg_bwd(a, lda, a2, lda2, K1, bt, ldb, bt2, ldb2, dy_part, false, xhat_bf16, rstd, gamma, row_pad, p_in, site_in, seed,
                  step_dev, dx_bf16, dres, false, part_ws, part_ws_bytes, M, N, K, stream);
}

int ttsmi_hgemm_ln_bwd_dual_h(const uint16_t* a, int64_t lda, const uint16_t* a2, int64_t lda2, int K1, const uint16_t* bt,
                              int64_t ldb, const uint16_t* bt2, int64_t ldb2, const uint16_t* dy_part_bf16,
                              const uint16_t* xhat_bf16, const float* rstd, const float* gamma, const uint8_t* row_pad, float p_in,
                              uint32_t site_in, uint64_t seed, const int64_t* step_dev, uint16_t* dx_bf16, void* dres,
                              int dres_is_bf16, void* part_ws, size_t part_ws_bytes, int M, int N, int K, ttsmi_stream_t stream) {
    return rg_bwd(a, lda, a2, lda2, K1, bt, ldb, bt2, ldb2, dy_part_bf16, true, xhat_bf16, rstd, gamma, row_pad, p_in, site_in, seed,
                  step_dev, dx_bf16, dres, dres_is_bf16 != 0, part_ws, part_ws_bytes, M, N, K, stream);
}

int ttsmi_layernorm_bwd_xhat_nparts(int M) { return lnx_blocks(M); }

int ttsmi_layernorm_bwd_xhat(const float* dy, const uint16_t* xhat_bf16, const float* rstd, const float* gamma,
                             const uint8_t* row_pad, float p_in, uint32_t site_in, uint64_t seed, const int64_t* step_dev,
                             uint16_t* dx_bf16, float* dres, void* part_ws, size_t part_ws_bytes, int M, int C,
                             ttsmi_stream_t stream) {
    return rg_lnx(dy, xhat_bf16, rstd, gamma, row_pad, p_in, site_in, seed, step_dev, dx_bf16, dres, false, part_ws, part_ws_bytes,
                  M, C, stream);
}
int ttsmi_layernorm_bwd_xhat_h(const float* dy, const uint16_t* xhat_bf16, const float* rstd, const float* gamma,
                               const uint8_t* row_pad, float p_in, uint32_t site_in, uint64_t seed, const int64_t* step_dev,
                               uint16_t* dx_bf16, uint16_t* dres_bf16, void* part_ws, size_t part_ws_bytes, int M, int C,
                               ttsmi_stream_t stream) {
    return rg_lnx(dy, xhat_bf16, rstd, gamma, row_pad, p_in, site_in, seed, step_dev, dx_bf16, dres_bf16, true, part_ws,
                  part_ws_bytes, M, C, stream);
}

}  // extern "C"
