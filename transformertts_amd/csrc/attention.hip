// Fused scaled-dot-product self-attention on exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).
// The [B,H,T,T] logits / weights never touch HBM (flash style, online softmax); they are only
// written when the caller asks for the reference's returned attention maps.
//
// Layout trick used everywhere below: the 32x32 C/D tile of an MFMA has col = lane&31 and
// row = (r&3) + 8*(r>>2) + 4*(lane>>5) over the 16 accumulator registers r.  Since the reduction
// index of an MFMA may be permuted freely as long as A and B agree, accumulator register r of a
// "P" tile can be fed STRAIGHT back as the B operand of step r of the following MFMA chain
// (its reduction index is then exactly row(r, lane>>5)) - P never leaves registers and never
// needs a transpose through LDS:
//   forward / dQ kernel :  S^T = K.Q^T  (col = query, rows = keys)  ->  O^T += V^T.P^T
//   dK/dV kernel        :  S   = Q.K^T  (col = key,   rows = queries) -> dV^T += dO^T.P, dK^T += Q^T.dS
// The head-dim reduction uses the "split-half" index map c(s, hh) = hh*DH/2 + s so that every
// lane's operand values are one contiguous run (16 B LDS / global accesses).
//
// MFMA-bound by construction: per 32x32 score tile a wave issues DH MFMAs of 64 cycles each, and
// only ~16 exp + a handful of VALU ops; LDS traffic is one ds_read per MFMA or less.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

struct AttnP {
    const float* qkv; long ld;      // [B*T, 3*H*DH]
    const uint8_t* key_pad;         // [B, T]
    const int32_t* klen;            // [B]
    float* ctx;                     // [B*T, H*DH]
    float* lse;                     // [B, H, T]
    const float* dctx;              // backward
    const float* octx;
    float* dqkv;
    float* delta;                   // [B, H, T]
    float* weights;                 // [B, H, T, T]
    int B, H, T;
    float sqrt_dk;
    uint32_t thr; float inv_keep; uint64_t seed; const int64_t* step_dev; uint32_t site;
};

__device__ __forceinline__ int rowmap(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

#define KT 64   // keys (or queries) staged per LDS tile

// Cooperative stage of a [KT][DH] slab of rows (row0.., column offset col0) into registers.
template <int DH>
__device__ __forceinline__ void slab_fetch(const float* base, long ld, int row0, int nrows_valid,
                                           int tid, float4 (&r)[KT * DH / 1024]) {
    constexpr int V4 = DH / 4;                   // float4 per row
#pragma unroll
    for (int i = 0; i < KT * DH / 1024; ++i) {
        int id = tid + 256 * i;
        int row = id / V4, c4 = id - row * V4;
        r[i] = (row < nrows_valid)
                   ? *reinterpret_cast<const float4*>(base + (long)(row0 + row) * ld + c4 * 4)
                   : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
template <int DH, int STRIDE>
__device__ __forceinline__ void slab_stash(float* S, int tid, const float4 (&r)[KT * DH / 1024]) {
    constexpr int V4 = DH / 4;
#pragma unroll
    for (int i = 0; i < KT * DH / 1024; ++i) {
        int id = tid + 256 * i;
        int row = id / V4, c4 = id - row * V4;
        *reinterpret_cast<float4*>(S + row * STRIDE + c4 * 4) = r[i];
    }
}

// One wave's 32 rows (one per lane&31), split-half columns [hh*DH/2, +DH/2) into registers.
template <int DH>
__device__ __forceinline__ void rows_to_regs(const float* base, long ld, int row, bool valid, int hh,
                                             float (&q)[DH / 2]) {
    const float* src = base + (long)row * ld + hh * (DH / 2);
#pragma unroll
    for (int u = 0; u < DH / 8; ++u) {
        float4 v = valid ? *reinterpret_cast<const float4*>(src + 4 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
        q[4 * u + 0] = v.x; q[4 * u + 1] = v.y; q[4 * u + 2] = v.z; q[4 * u + 3] = v.w;
    }
}

// acc[x][y] += sum_c A[x][c] * Breg[y][c]: A rows come from a padded LDS slab (b128 reads), the
// other operand from registers.  a_first selects which one is the MFMA A operand.
template <int DH, int STRIDE, bool LDS_IS_A>
__device__ __forceinline__ f32x16 dot_rows(const float* S, int row, int hh, const float (&q)[DH / 2]) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* src = S + row * STRIDE + hh * (DH / 2);
#pragma unroll
    for (int u = 0; u < DH / 8; ++u) {
        float4 kv = *reinterpret_cast<const float4*>(src + 4 * u);
        if (LDS_IS_A) {
            acc = MFMA32(kv.x, q[4 * u + 0], acc);
            acc = MFMA32(kv.y, q[4 * u + 1], acc);
            acc = MFMA32(kv.z, q[4 * u + 2], acc);
            acc = MFMA32(kv.w, q[4 * u + 3], acc);
        } else {
            acc = MFMA32(q[4 * u + 0], kv.x, acc);
            acc = MFMA32(q[4 * u + 1], kv.y, acc);
            acc = MFMA32(q[4 * u + 2], kv.z, acc);
            acc = MFMA32(q[4 * u + 3], kv.w, acc);
        }
    }
    return acc;
}

// out[cb] (rows = columns c of block cb, cols = lane index) += sum_s Slab[rowmap(s,hh)][cb*32+l31] * p[s]
template <int DH, int STRIDE>
__device__ __forceinline__ void accum_T(const float* S, int row0, int l31, int hh, const f32x16& p,
                                        f32x16 (&out)[DH / 32]) {
#pragma unroll
    for (int cb = 0; cb < DH / 32; ++cb) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            float a = S[(row0 + rowmap(s, hh)) * STRIDE + cb * 32 + l31];
            out[cb] = MFMA32(a, p[s], out[cb]);
        }
    }
}

// Transposed accumulators (rows = c, cols = lane row) -> dst rows through a per-wave LDS patch.
// patch: [32][DH+1] floats.  dst row pointer for local row j: dst + (row0 + j) * ld.
template <int DH>
__device__ __forceinline__ void store_T(float* patch, const f32x16 (&o)[DH / 32], float scale_lane,
                                        float* dst, long ld, int row0, int nvalid, int lane) {
    const int l31 = lane & 31, hh = lane >> 5;
    __syncthreads();                      // (block-uniform call site) patch region is free
#pragma unroll
    for (int cb = 0; cb < DH / 32; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) patch[l31 * (DH + 1) + cb * 32 + rowmap(r, hh)] = o[cb][r] * scale_lane;
    __syncthreads();
    for (int j = 0; j < 32; ++j) {
        if (j >= nvalid) break;
        for (int c = lane; c < DH; c += 64) dst[(long)(row0 + j) * ld + c] = patch[j * (DH + 1) + c];
    }
}

template <int DH>
struct Smem {
    static constexpr int KS = DH + 4;               // padded stride for b128 row reads
    static constexpr int SLAB = KT * KS;            // floats
    static constexpr int PATCH = 4 * 32 * (DH + 1); // per-wave transpose patches
    static constexpr int MAIN = (2 * SLAB > PATCH ? 2 * SLAB : PATCH);
};

// =================================================================================================
// forward
// =================================================================================================
template <int DH>
__global__ __launch_bounds__(256, (DH <= 64 ? 2 : 1)) void attn_fwd_kernel(AttnP p) {
    using SM = Smem<DH>;
    __shared__ __attribute__((aligned(16))) float smem[SM::MAIN + KT];
    float* Ks = smem;
    float* Vs = smem + SM::SLAB;
    float* padS = smem + SM::MAIN;            // KT pad flags as floats (0 / 1)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int ntile = (p.T + 127) >> 7;                 // 1-D grid: all q/key tiles of one (b, h) on one XCD
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int bx = lid % ntile, h = (lid / ntile) % p.H, b = lid / (ntile * p.H);
    const int d = p.H * DH;
    const int q = bx * 128 + wave * 32 + l31;
    const bool qok = q < p.T;
    const uint64_t dkey = ttsmi_drop_key(p.seed, p.step_dev, p.site);
    const float* Qb = p.qkv + (long)b * p.T * p.ld + h * DH;
    const float* Kb = Qb + d;
    const float* Vb = Qb + 2 * d;

    float qreg[DH / 2];
    rows_to_regs<DH>(Qb, p.ld, q, qok, hh, qreg);

    f32x16 o[DH / 32];
#pragma unroll
    for (int cb = 0; cb < DH / 32; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[cb][r] = 0.f;
    float m = -INFINITY, l = 0.f;

    const int klen = p.klen[b];
    const uint32_t drop_rb = ttsmi_row_base(dkey, (uint32_t)(((long)b * p.H + h) * p.T + q));

    float4 rk[KT * DH / 1024], rv[KT * DH / 1024];
    float rpad = 0.f;
    {
        int nv = min(KT, klen);
        slab_fetch<DH>(Kb, p.ld, 0, nv, tid, rk);
        slab_fetch<DH>(Vb, p.ld, 0, nv, tid, rv);
        if (tid < KT) rpad = (tid < nv && p.key_pad[(long)b * p.T + tid]) ? 1.f : 0.f;
    }
    for (int k0 = 0; k0 < klen; k0 += KT) {
        __syncthreads();                      // previous tile fully consumed
        slab_stash<DH, SM::KS>(Ks, tid, rk);
        slab_stash<DH, SM::KS>(Vs, tid, rv);
        if (tid < KT) padS[tid] = rpad;
        __syncthreads();
        if (k0 + KT < klen) {                 // prefetch the next tile under this tile's MFMAs
            int nv = min(KT, klen - (k0 + KT));
            slab_fetch<DH>(Kb, p.ld, k0 + KT, nv, tid, rk);
            slab_fetch<DH>(Vb, p.ld, k0 + KT, nv, tid, rv);
            if (tid < KT) rpad = (tid < nv && p.key_pad[(long)b * p.T + k0 + KT + tid]) ? 1.f : 0.f;
        }
#pragma unroll
        for (int kt = 0; kt < KT / 32; ++kt) {
            if (k0 + kt * 32 >= klen) break;
            f32x16 s = dot_rows<DH, SM::KS, true>(Ks, kt * 32 + l31, hh, qreg);   // S^T[key][q]
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int kl = kt * 32 + rowmap(r, hh);
                float v = s[r] / p.sqrt_dk;
                v += padS[kl] * -1e9f;
                if (k0 + kl >= klen) v = -INFINITY;
                s[r] = v;
                mx = fmaxf(mx, v);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float mn = fmaxf(m, mx);
            float alpha = expf(m - mn);
            float rs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float e = expf(s[r] - mn);
                rs += e;
                if (p.thr) {
                    uint32_t key = k0 + kt * 32 + rowmap(r, hh);
                    e *= ttsmi_keep_of(ttsmi_pair_hash(drop_rb, key), key, p.thr, p.inv_keep);
                }
                s[r] = e;
            }
            rs += __shfl_xor(rs, 32, 64);
            l = l * alpha + rs;
            m = mn;
#pragma unroll
            for (int cb = 0; cb < DH / 32; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[cb][r] *= alpha;
            accum_T<DH, SM::KS>(Vs, kt * 32, l31, hh, s, o);                       // O^T += V^T.P^T
        }
    }
    __syncthreads();
    if (qok && hh == 0) p.lse[((long)b * p.H + h) * p.T + q] = m + logf(l);
    float* patch = smem + wave * 32 * (DH + 1);
    int row0 = bx * 128 + wave * 32;
    int nvalid = min(32, p.T - row0);
    store_T<DH>(patch, o, 1.0f / l, p.ctx + (long)b * p.T * d + h * DH, d, row0, nvalid, lane);
}

// =================================================================================================
// backward, kernel A: dQ (+ delta = rowsum(dO * O)), same orientation as the forward
// =================================================================================================
template <int DH>
__global__ __launch_bounds__(256, (DH <= 64 ? 2 : 1)) void attn_bwd_dq_kernel(AttnP p) {
    using SM = Smem<DH>;
    __shared__ __attribute__((aligned(16))) float smem[SM::MAIN + KT];
    float* Ks = smem;
    float* Vs = smem + SM::SLAB;
    float* padS = smem + SM::MAIN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int ntile = (p.T + 127) >> 7;                 // 1-D grid: all q/key tiles of one (b, h) on one XCD
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int bx = lid % ntile, h = (lid / ntile) % p.H, b = lid / (ntile * p.H);
    const int d = p.H * DH;
    const int q = bx * 128 + wave * 32 + l31;
    const bool qok = q < p.T;
    const uint64_t dkey = ttsmi_drop_key(p.seed, p.step_dev, p.site);
    const float* Qb = p.qkv + (long)b * p.T * p.ld + h * DH;
    const float* Kb = Qb + d;
    const float* Vb = Qb + 2 * d;

    float qreg[DH / 2], doreg[DH / 2];
    rows_to_regs<DH>(Qb, p.ld, q, qok, hh, qreg);
    rows_to_regs<DH>(p.dctx + (long)b * p.T * d + h * DH, d, q, qok, hh, doreg);
    float delta = 0.f;
    {
        float oreg[DH / 2];
        rows_to_regs<DH>(p.octx + (long)b * p.T * d + h * DH, d, q, qok, hh, oreg);
#pragma unroll
        for (int i = 0; i < DH / 2; ++i) delta += oreg[i] * doreg[i];
        delta += __shfl_xor(delta, 32, 64);
    }
    const long sidx = ((long)b * p.H + h) * p.T + q;
    if (qok && hh == 0) p.delta[sidx] = delta;
    const float lse = qok ? p.lse[sidx] : INFINITY;

    f32x16 dq[DH / 32];
#pragma unroll
    for (int cb = 0; cb < DH / 32; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[cb][r] = 0.f;

    const int klen = p.klen[b];
    const uint32_t drop_rb = ttsmi_row_base(dkey, (uint32_t)sidx);
    const float inv_sqrt = 1.0f / p.sqrt_dk;

    float4 rk[KT * DH / 1024], rv[KT * DH / 1024];
    float rpad = 0.f;
    {
        int nv = min(KT, klen);
        slab_fetch<DH>(Kb, p.ld, 0, nv, tid, rk);
        slab_fetch<DH>(Vb, p.ld, 0, nv, tid, rv);
        if (tid < KT) rpad = (tid < nv && p.key_pad[(long)b * p.T + tid]) ? 1.f : 0.f;
    }
    for (int k0 = 0; k0 < klen; k0 += KT) {
        __syncthreads();
        slab_stash<DH, SM::KS>(Ks, tid, rk);
        slab_stash<DH, SM::KS>(Vs, tid, rv);
        if (tid < KT) padS[tid] = rpad;
        __syncthreads();
        if (k0 + KT < klen) {
            int nv = min(KT, klen - (k0 + KT));
            slab_fetch<DH>(Kb, p.ld, k0 + KT, nv, tid, rk);
            slab_fetch<DH>(Vb, p.ld, k0 + KT, nv, tid, rv);
            if (tid < KT) rpad = (tid < nv && p.key_pad[(long)b * p.T + k0 + KT + tid]) ? 1.f : 0.f;
        }
#pragma unroll
        for (int kt = 0; kt < KT / 32; ++kt) {
            if (k0 + kt * 32 >= klen) break;
            f32x16 s = dot_rows<DH, SM::KS, true>(Ks, kt * 32 + l31, hh, qreg);     // S^T
            f32x16 dp = dot_rows<DH, SM::KS, true>(Vs, kt * 32 + l31, hh, doreg);   // dP^T = V.dO^T
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int kl = kt * 32 + rowmap(r, hh);
                float v = s[r] / p.sqrt_dk;
                v += padS[kl] * -1e9f;
                float pr = (k0 + kl >= klen) ? 0.f : expf(v - lse);
                float keep = 1.f;
                if (p.thr) keep = ttsmi_keep_of(ttsmi_pair_hash(drop_rb, (uint32_t)(k0 + kl)), (uint32_t)(k0 + kl), p.thr, p.inv_keep);
                s[r] = pr * (keep * dp[r] - delta) * inv_sqrt;                      // dS^T
            }
            accum_T<DH, SM::KS>(Ks, kt * 32, l31, hh, s, dq);                       // dQ^T += K^T.dS^T
        }
    }
    __syncthreads();
    float* patch = smem + wave * 32 * (DH + 1);
    int row0 = bx * 128 + wave * 32;
    int nvalid = min(32, p.T - row0);
    store_T<DH>(patch, dq, 1.0f, p.dqkv + (long)b * p.T * p.ld + h * DH, p.ld, row0, nvalid, lane);
}

// =================================================================================================
// backward, kernel B: dK, dV.  Workgroup owns 128 keys (32 per wave, lane&31 = key), loops queries.
// =================================================================================================
template <int DH>
__global__ __launch_bounds__(256, (DH <= 64 ? 2 : 1)) void attn_bwd_dkv_kernel(AttnP p) {
    using SM = Smem<DH>;
    __shared__ __attribute__((aligned(16))) float smem[SM::MAIN + 2 * KT];
    float* Qs = smem;
    float* Os = smem + SM::SLAB;              // dO tile
    float* lseS = smem + SM::MAIN;
    float* delS = lseS + KT;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int ntile = (p.T + 127) >> 7;                 // 1-D grid: all q/key tiles of one (b, h) on one XCD
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int bx = lid % ntile, h = (lid / ntile) % p.H, b = lid / (ntile * p.H);
    const int d = p.H * DH;
    const int key = bx * 128 + wave * 32 + l31;
    const int klen = p.klen[b];
    const bool kok = key < p.T;
    const bool kact = key < klen;             // keys >= klen took no part in the forward
    const uint64_t dkey = ttsmi_drop_key(p.seed, p.step_dev, p.site);
    const float* Qb = p.qkv + (long)b * p.T * p.ld + h * DH;
    const float* Kb = Qb + d;
    const float* Vb = Qb + 2 * d;
    const float* dOb = p.dctx + (long)b * p.T * d + h * DH;

    float kreg[DH / 2], vreg[DH / 2];
    rows_to_regs<DH>(Kb, p.ld, key, kok, hh, kreg);
    rows_to_regs<DH>(Vb, p.ld, key, kok, hh, vreg);
    const float padterm = (kok && p.key_pad[(long)b * p.T + key]) ? -1e9f : 0.f;

    f32x16 dk[DH / 32], dv[DH / 32];
#pragma unroll
    for (int cb = 0; cb < DH / 32; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[cb][r] = 0.f; dv[cb][r] = 0.f; }
    const float inv_sqrt = 1.0f / p.sqrt_dk;
    const long stat0 = ((long)b * p.H + h) * p.T;

    float4 rq[KT * DH / 1024], ro[KT * DH / 1024];
    float rl = 0.f, rd = 0.f;
    const bool wg_active = bx * 128 < klen;   // block-uniform
    if (wg_active) {
        int nv = min(KT, p.T);
        slab_fetch<DH>(Qb, p.ld, 0, nv, tid, rq);
        slab_fetch<DH>(dOb, d, 0, nv, tid, ro);
        if (tid < KT) {
            rl = tid < nv ? p.lse[stat0 + tid] : INFINITY;
            rd = tid < nv ? p.delta[stat0 + tid] : 0.f;
        }
        for (int q0 = 0; q0 < p.T; q0 += KT) {
            __syncthreads();
            slab_stash<DH, SM::KS>(Qs, tid, rq);
            slab_stash<DH, SM::KS>(Os, tid, ro);
            if (tid < KT) { lseS[tid] = rl; delS[tid] = rd; }
            __syncthreads();
            if (q0 + KT < p.T) {
                int nv = min(KT, p.T - (q0 + KT));
                slab_fetch<DH>(Qb, p.ld, q0 + KT, nv, tid, rq);
                slab_fetch<DH>(dOb, d, q0 + KT, nv, tid, ro);
                if (tid < KT) {
                    rl = tid < nv ? p.lse[stat0 + q0 + KT + tid] : INFINITY;
                    rd = tid < nv ? p.delta[stat0 + q0 + KT + tid] : 0.f;
                }
            }
#pragma unroll
            for (int qt = 0; qt < KT / 32; ++qt) {
                if (q0 + qt * 32 >= p.T) break;
                // A = LDS rows (lane&31 = query row), B = registers (lane&31 = key) -> col = key
                f32x16 s = dot_rows<DH, SM::KS, true>(Qs, qt * 32 + l31, hh, kreg);     // S[q][key]
                f32x16 dp = dot_rows<DH, SM::KS, true>(Os, qt * 32 + l31, hh, vreg);    // dP = dO.V^T
                f32x16 pt;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int ql = qt * 32 + rowmap(r, hh);
                    float v = s[r] / p.sqrt_dk + padterm;
                    float pr = kact ? expf(v - lseS[ql]) : 0.f;      // lse = +inf for q >= T
                    float keep = 1.f;
                    if (p.thr)
                        keep = ttsmi_keep_scale(dkey, (uint32_t)(stat0 + q0 + ql), (uint32_t)key, p.thr, p.inv_keep);
                    pt[r] = pr * keep;                                               // dropped P
                    s[r] = pr * (keep * dp[r] - delS[ql]) * inv_sqrt;                // dS
                }
                accum_T<DH, SM::KS>(Os, qt * 32, l31, hh, pt, dv);                   // dV^T += dO^T.P
                accum_T<DH, SM::KS>(Qs, qt * 32, l31, hh, s, dk);                    // dK^T += Q^T.dS
            }
        }
    }
    __syncthreads();
    float* patch = smem + wave * 32 * (DH + 1);
    int row0 = bx * 128 + wave * 32;
    int nvalid = min(32, p.T - row0);
    float* dst = p.dqkv + (long)b * p.T * p.ld + h * DH;
    store_T<DH>(patch, dk, 1.0f, dst + d, p.ld, row0, nvalid, lane);
    store_T<DH>(patch, dv, 1.0f, dst + 2 * d, p.ld, row0, nvalid, lane);
}

// =================================================================================================
// attention-weight materialisation (only on request): one wave per 32x32 tile, no LDS
// =================================================================================================
template <int DH>
__global__ __launch_bounds__(256) void attn_weights_kernel(AttnP p) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int bh = blockIdx.z, b = bh / p.H, h = bh - b * p.H;
    const int d = p.H * DH;
    const int key = (blockIdx.x * 4 + wave) * 32 + l31;
    const int q0 = blockIdx.y * 32;
    if ((blockIdx.x * 4 + wave) * 32 >= p.T) return;
    const uint64_t dkey = ttsmi_drop_key(p.seed, p.step_dev, p.site);
    const float* Qb = p.qkv + (long)b * p.T * p.ld + h * DH;
    float qreg[DH / 2], kreg[DH / 2];
    rows_to_regs<DH>(Qb, p.ld, q0 + l31, q0 + l31 < p.T, hh, qreg);
    rows_to_regs<DH>(Qb + d, p.ld, key, key < p.T, hh, kreg);
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int i = 0; i < DH / 2; ++i) s = MFMA32(qreg[i], kreg[i], s);
    const float padterm = (key < p.T && p.key_pad[(long)b * p.T + key]) ? -1e9f : 0.f;
    const long stat0 = (long)bh * p.T;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int q = q0 + rowmap(r, hh);
        if (q >= p.T || key >= p.T) continue;
        float v = s[r] / p.sqrt_dk + padterm;
        float pr = expf(v - p.lse[stat0 + q]);
        if (p.thr) pr *= ttsmi_keep_scale(dkey, (uint32_t)(stat0 + q), (uint32_t)key, p.thr, p.inv_keep);
        p.weights[(stat0 + q) * (long)p.T + key] = pr;
    }
}

// ---- host ---------------------------------------------------------------------------------------
static int fill(AttnP& p, const void* qkv, const uint8_t* key_pad, const int32_t* klen, int B, int H,
                int T, int dh, float p_drop, uint64_t seed, const int64_t* step_dev, uint32_t site,
                int dtype, const char* who) {
    TTSMI_CHECK_ARG(qkv && key_pad, "%s: null pointer", who);
    TTSMI_CHECK_ARG(B > 0 && H > 0 && T > 0, "%s: bad shape B=%d H=%d T=%d", who, B, H, T);
    TTSMI_CHECK_ARG(dtype == TTSMI_F32, "%s: dtype %d not built", who, dtype);
    TTSMI_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "%s: dropout rate out of [0,1)", who);
    TTSMI_CHECK_ARG((((uintptr_t)qkv) & 15) == 0, "%s: qkv must be 16-byte aligned", who);
    memset(&p, 0, sizeof(p));
    p.qkv = (const float*)qkv; p.ld = 3L * H * dh; p.key_pad = key_pad; p.klen = klen;
    p.B = B; p.H = H; p.T = T; p.sqrt_dk = sqrtf((float)dh);
    p.thr = p_drop > 0.f ? ttsmi_drop_threshold(p_drop) : 0;
    p.inv_keep = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.f;
    p.seed = seed; p.step_dev = step_dev; p.site = site;
    return TTSMI_OK;
}

#define DISPATCH_DH(dh, KERNEL, grid, st, p)                                                   \
    switch (dh) {                                                                              \
        case 32: hipLaunchKernelGGL((KERNEL<32>), grid, dim3(256), 0, st, p); break;           \
        case 64: hipLaunchKernelGGL((KERNEL<64>), grid, dim3(256), 0, st, p); break;           \
        case 96: hipLaunchKernelGGL((KERNEL<96>), grid, dim3(256), 0, st, p); break;           \
        case 128: hipLaunchKernelGGL((KERNEL<128>), grid, dim3(256), 0, st, p); break;         \
        case 192: hipLaunchKernelGGL((KERNEL<192>), grid, dim3(256), 0, st, p); break;         \
        default:                                                                               \
            ttsmi_set_error("attention: head dim %d not built (32/64/96/128/192)", dh);        \
            return TTSMI_ERR_UNSUPPORTED;                                                      \
    }

extern "C" {

int ttsmi_attention_fwd(const void* qkv, const uint8_t* key_pad, const int32_t* klen, void* ctx,
                        float* lse, int B, int H, int T, int dh, float p_drop, uint64_t seed,
                        const int64_t* step_dev, uint32_t site, int dtype, ttsmi_stream_t stream) {
    if (dtype == TTSMI_BF16 || dtype == TTSMI_BF16_IO)
        return ttsmi_hattention_fwd(qkv, key_pad, klen, ctx, lse, B, H, T, dh, p_drop, seed, step_dev, site,
                                    dtype == TTSMI_BF16_IO, nullptr, (hipStream_t)stream);
    AttnP p;
    int rc = fill(p, qkv, key_pad, klen, B, H, T, dh, p_drop, seed, step_dev, site, dtype, "attention_fwd");
    if (rc) return rc;
    TTSMI_CHECK_ARG(klen && ctx && lse, "attention_fwd: null pointer");
    p.ctx = (float*)ctx; p.lse = lse;
    dim3 grid(ttsmi_cdiv(T, 128) * H * B);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_DH(dh, attn_fwd_kernel, grid, st, p);
    TTSMI_CHECK_LAUNCH("attention_fwd");
    return TTSMI_OK;
}

size_t ttsmi_attention_bwd_ws_bytes(int B, int H, int T, int dh) {
    (void)dh;
    return (size_t)B * H * T * sizeof(float) + 256;
}

int ttsmi_attention_bwd(const void* qkv, const uint8_t* key_pad, const int32_t* klen,
                        const void* ctx, const void* dctx, const float* lse, void* dqkv, int B,
                        int H, int T, int dh, float p_drop, uint64_t seed, const int64_t* step_dev,
                        uint32_t site, void* ws, size_t ws_bytes, int dtype,
                        ttsmi_stream_t stream) {
    if (dtype == TTSMI_BF16 || dtype == TTSMI_BF16_IO) {
        TTSMI_CHECK_ARG(ws && ws_bytes >= ttsmi_attention_bwd_ws_bytes(B, H, T, dh),
                        "attention_bwd: workspace too small");
        return ttsmi_hattention_bwd(qkv, key_pad, klen, ctx, dctx, lse, dqkv, B, H, T, dh, p_drop, seed,
                                    step_dev, site, ws, dtype == TTSMI_BF16_IO, nullptr, (hipStream_t)stream);
    }
    AttnP p;
    int rc = fill(p, qkv, key_pad, klen, B, H, T, dh, p_drop, seed, step_dev, site, dtype, "attention_bwd");
    if (rc) return rc;
    TTSMI_CHECK_ARG(klen && ctx && dctx && lse && dqkv, "attention_bwd: null pointer");
    TTSMI_CHECK_ARG(ws && ws_bytes >= ttsmi_attention_bwd_ws_bytes(B, H, T, dh),
                    "attention_bwd: workspace too small");
    p.octx = (const float*)ctx; p.dctx = (const float*)dctx; p.lse = (float*)lse;
    p.dqkv = (float*)dqkv; p.delta = (float*)ws;
    dim3 grid(ttsmi_cdiv(T, 128) * H * B);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_DH(dh, attn_bwd_dq_kernel, grid, st, p);
    TTSMI_CHECK_LAUNCH("attention_bwd_dq");
    DISPATCH_DH(dh, attn_bwd_dkv_kernel, grid, st, p);
    TTSMI_CHECK_LAUNCH("attention_bwd_dkv");
    return TTSMI_OK;
}

size_t ttsmi_attention_fwd_splitkeys_ws_bytes(int B, int H, int T, int dh) {
    return ttsmi_hattention_fwd_split_ws_bytes(B, H, T, dh);
}

int ttsmi_attention_fwd_splitkeys(const void* qkv, const uint8_t* key_pad, const int32_t* klen, void* ctx, float* lse,
                                  int B, int H, int T, int dh, void* ws, size_t ws_bytes, ttsmi_stream_t stream) {
    return ttsmi_hattention_fwd_split(qkv, key_pad, klen, ctx, lse, B, H, T, dh, ws, ws_bytes, (hipStream_t)stream);
}

size_t ttsmi_attention_dropmask_bytes(int B, int H, int T) { return ttsmi_hattention_dropmask_bytes(B, H, T); }

int ttsmi_attention_dropmask(void* mask, int B, int H, int T, float p_drop, uint64_t seed,
                             const int64_t* step_dev, uint32_t site, ttsmi_stream_t stream) {
    return ttsmi_hattention_dropmask(mask, B, H, T, p_drop, seed, step_dev, site, (hipStream_t)stream);
}

int ttsmi_attention_dropmask_stack(void* const* masks, const uint32_t* sites, int n, int B, int H, int T, float p_drop,
                                   uint64_t seed, const int64_t* step_dev, ttsmi_stream_t stream) {
    return ttsmi_hattention_dropmask_stack(masks, sites, n, B, H, T, p_drop, seed, step_dev, (hipStream_t)stream);
}

int ttsmi_attention_fwd_masked(const void* qkv, const uint8_t* key_pad, const int32_t* klen, void* ctx,
                               float* lse, int B, int H, int T, int dh, float p_drop, const void* dropmask, int dtype,
                               ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(dropmask && p_drop > 0.f, "attention_fwd_masked: needs a keep-bit mask and p_drop > 0");
    TTSMI_CHECK_ARG(dtype == TTSMI_BF16 || dtype == TTSMI_BF16_IO, "attention_fwd_masked: dtype %d (TTSMI_BF16 / TTSMI_BF16_IO)", dtype);
    return ttsmi_hattention_fwd(qkv, key_pad, klen, ctx, lse, B, H, T, dh, p_drop, 0, nullptr, 0, dtype == TTSMI_BF16_IO,
                                dropmask, (hipStream_t)stream);
}

int ttsmi_attention_bwd_masked(const void* qkv, const uint8_t* key_pad, const int32_t* klen,
                               const void* ctx, const void* dctx, const float* lse, void* dqkv, int B,
                               int H, int T, int dh, float p_drop, const void* dropmask, void* ws, size_t ws_bytes,
                               int dtype, ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(dropmask && p_drop > 0.f, "attention_bwd_masked: needs a keep-bit mask and p_drop > 0");
    TTSMI_CHECK_ARG(dtype == TTSMI_BF16 || dtype == TTSMI_BF16_IO, "attention_bwd_masked: dtype %d (TTSMI_BF16 / TTSMI_BF16_IO)", dtype);
    TTSMI_CHECK_ARG(ws && ws_bytes >= ttsmi_attention_bwd_ws_bytes(B, H, T, dh), "attention_bwd_masked: workspace too small");
    return ttsmi_hattention_bwd(qkv, key_pad, klen, ctx, dctx, lse, dqkv, B, H, T, dh, p_drop, 0, nullptr, 0, ws,
                                dtype == TTSMI_BF16_IO, dropmask, (hipStream_t)stream);
}

int ttsmi_attention_weights_masked(const void* qkv, const uint8_t* key_pad, const float* lse, float* weights, int B, int H,
                                   int T, int dh, float p_drop, const void* dropmask, int dtype, ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(dtype == TTSMI_BF16_IO && dropmask, "attention_weights_masked: bf16 tensors and a keep-bit table required");
    return ttsmi_hattention_weights(qkv, key_pad, lse, weights, B, H, T, dh, p_drop, 0, nullptr, 0, dropmask, (hipStream_t)stream);
}

int ttsmi_attention_weights(const void* qkv, const uint8_t* key_pad, const float* lse,
                            float* weights, int B, int H, int T, int dh, float p_drop,
                            uint64_t seed, const int64_t* step_dev, uint32_t site, int dtype,
                            ttsmi_stream_t stream) {
    // TTSMI_BF16_IO: qkv is the bf16 tensor the forward read - recomputed on the bf16 MFMA (attention_bf16.hip).
    // TTSMI_BF16 (fp32 qkv, bf16 forward): recomputed from the fp32 q/k with the exact-fp32 MFMA (the maps are a logging
    // output; rows sum to 1 up to the bf16 rounding of the forward's log-sum-exp)
    if (dtype == TTSMI_BF16_IO)
        return ttsmi_hattention_weights(qkv, key_pad, lse, weights, B, H, T, dh, p_drop, seed, step_dev, site, nullptr,
                                        (hipStream_t)stream);
    AttnP p;
    int rc = fill(p, qkv, key_pad, nullptr, B, H, T, dh, p_drop, seed, step_dev, site, TTSMI_F32, "attention_weights");
    if (rc) return rc;
    TTSMI_CHECK_ARG(lse && weights, "attention_weights: null pointer");
    p.lse = (float*)lse; p.weights = weights;
    dim3 grid(ttsmi_cdiv(T, 128), ttsmi_cdiv(T, 32), B * H);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_DH(dh, attn_weights_kernel, grid, st, p);
    TTSMI_CHECK_LAUNCH("attention_weights");
    return TTSMI_OK;
}

}  // extern "C"
