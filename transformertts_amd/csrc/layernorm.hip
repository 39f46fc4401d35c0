// Fused [dropout ->] residual add -> LayerNorm [-> + s*PE] [-> dropout] [-> row mask], fwd + bwd.
// HBM-bound: one wave64 owns one row, the row lives in registers between the statistics and the
// normalisation (each input element is read exactly once, each output written once), row
// reductions are wave shuffles, loads/stores are 16 B per lane when C % 4 == 0.
#include <stdlib.h>
#include "common.h"

#define LN_WAVES 4   // rows per 256-thread block

template <int VW> struct Vec;
template <> struct Vec<4> { typedef float4 T; };
template <> struct Vec<1> { typedef float T; };

template <int VW>
__device__ __forceinline__ void ldv(const float* p, float (&v)[VW]) {
    if constexpr (VW == 4) {
        float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        v[0] = *p;
    }
}
template <int VW>
__device__ __forceinline__ void stv(float* p, const float (&v)[VW]) {
    if constexpr (VW == 4) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        *p = v[0];
    }
}

typedef __bf16 ln_bf16x4 __attribute__((ext_vector_type(4)));
template <int VW>
__device__ __forceinline__ void sth(uint16_t* p, const float (&v)[VW]) {
    if constexpr (VW == 4) {
        ln_bf16x4 h;
        h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
        *reinterpret_cast<uint2*>(p) = *reinterpret_cast<uint2*>(&h);
    } else {
        __bf16 h = (__bf16)v[0];
        *p = *reinterpret_cast<uint16_t*>(&h);
    }
}

struct LnP {
    const float* x; const float* res; const float* gamma; const float* beta;
    const float* pe; const float* pe_scale; int T;
    const uint8_t* row_pad;
    uint32_t thr_in, thr_out, site_in, site_out; float inv_in, inv_out; uint64_t seed; const int64_t* step_dev;
    float eps;
    float* y; float* mean; float* rstd;
    uint16_t* y_h;          // optional bf16 copy of y (the GEMM operand of the consumers)
    int M, C;
    // backward
    const float* dy; const float* mean_in; const float* rstd_in;
    int relu_in;
    float* dx; float* dres;
    uint16_t* dx_h;         // optional bf16 dx (dx itself may then be NULL: its consumers are GEMMs)
    float* part_g; float* part_b; float* part_s;   // [nwaves][C], [nwaves][C], [nwaves]
};

// lane owns chunks c = (lane + 64*j) * VW, j < NPL
template <int VW, int NPL>
__global__ __launch_bounds__(256) void ln_fwd_kernel(LnP p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * LN_WAVES + (threadIdx.x >> 6);
    if (row >= p.M) return;
    const uint64_t key_in = ttsmi_drop_key(p.seed, p.step_dev, p.site_in);
    const uint64_t key_out = ttsmi_drop_key(p.seed, p.step_dev, p.site_out);
    const long base = (long)row * p.C;
    float z[NPL][VW];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        int c = (lane + 64 * j) * VW;
        if (c < p.C) {
            ldv<VW>(p.x + base + c, z[j]);
            if (p.thr_in) {
#pragma unroll
                for (int e = 0; e < VW; ++e)
                    z[j][e] *= ttsmi_keep_scale(key_in, (uint32_t)row, (uint32_t)(c + e), p.thr_in, p.inv_in);
            }
            if (p.res) {
                float r[VW];
                ldv<VW>(p.res + base + c, r);
#pragma unroll
                for (int e = 0; e < VW; ++e) z[j][e] += r[e];
            }
#pragma unroll
            for (int e = 0; e < VW; ++e) s += z[j][e];
        } else {
#pragma unroll
            for (int e = 0; e < VW; ++e) z[j][e] = 0.f;
        }
    }
    const float invC = 1.0f / (float)p.C;
    const float mean = wave_sum(s) * invC;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        int c = (lane + 64 * j) * VW;
        if (c < p.C) {
#pragma unroll
            for (int e = 0; e < VW; ++e) { float d = z[j][e] - mean; q += d * d; }
        }
    }
    const float var = wave_sum(q) * invC;
    const float rstd = 1.0f / sqrtf(var + p.eps);
    if (lane == 0) { p.mean[row] = mean; p.rstd[row] = rstd; }
    const bool padded = p.row_pad && p.row_pad[row];
    const float ps = p.pe ? p.pe_scale[0] : 0.f;
    const long pbase = p.pe ? (long)(row % p.T) * p.C : 0;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        int c = (lane + 64 * j) * VW;
        if (c < p.C) {
            float g[VW], b[VW], o[VW];
            ldv<VW>(p.gamma + c, g);
            ldv<VW>(p.beta + c, b);
#pragma unroll
            for (int e = 0; e < VW; ++e) o[e] = (z[j][e] - mean) * rstd * g[e] + b[e];
            if (p.pe) {
                float pe[VW];
                ldv<VW>(p.pe + pbase + c, pe);
#pragma unroll
                for (int e = 0; e < VW; ++e) o[e] += ps * pe[e];
            }
            if (p.thr_out) {
#pragma unroll
                for (int e = 0; e < VW; ++e)
                    o[e] *= ttsmi_keep_scale(key_out, (uint32_t)row, (uint32_t)(c + e), p.thr_out, p.inv_out);
            }
            if (padded) {
#pragma unroll
                for (int e = 0; e < VW; ++e) o[e] = 0.f;
            }
            stv<VW>(p.y + base + c, o);
            if (p.y_h) sth<VW>(p.y_h + base + c, o);
        }
    }
}

template <int VW, int NPL>
__global__ __launch_bounds__(256) void ln_bwd_kernel(LnP p) {
    const int lane = threadIdx.x & 63;
    const int wave_g = blockIdx.x * LN_WAVES + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * LN_WAVES;
    const uint64_t key_in = ttsmi_drop_key(p.seed, p.step_dev, p.site_in);
    const uint64_t key_out = ttsmi_drop_key(p.seed, p.step_dev, p.site_out);
    float ag[NPL][VW], ab[NPL][VW];
    float as = 0.f;
#pragma unroll
    for (int j = 0; j < NPL; ++j)
#pragma unroll
        for (int e = 0; e < VW; ++e) { ag[j][e] = 0.f; ab[j][e] = 0.f; }
    const float invC = 1.0f / (float)p.C;

    for (int row = wave_g; row < p.M; row += nwaves) {
        const long base = (long)row * p.C;
        const bool padded = p.row_pad && p.row_pad[row];
        const float mean = p.mean_in[row], rstd = p.rstd_in[row];
        const long pbase = p.pe ? (long)(row % p.T) * p.C : 0;
        float n[NPL][VW], gn[NPL][VW], keep[NPL][VW];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            int c = (lane + 64 * j) * VW;
            if (c < p.C) {
                float xv[VW], g[VW], gam[VW];
                ldv<VW>(p.x + base + c, xv);
                ldv<VW>(p.dy + base + c, g);
                ldv<VW>(p.gamma + c, gam);
#pragma unroll
                for (int e = 0; e < VW; ++e) {
                    float kin = p.thr_in ? ttsmi_keep_scale(key_in, (uint32_t)row, (uint32_t)(c + e), p.thr_in, p.inv_in) : 1.f;
                    // relu'(x) of the producer folded into the x branch
                    keep[j][e] = (p.relu_in && !(xv[e] > 0.f)) ? 0.f : kin;
                    xv[e] *= kin;
                }
                if (p.res) {
                    float r[VW];
                    ldv<VW>(p.res + base + c, r);
#pragma unroll
                    for (int e = 0; e < VW; ++e) xv[e] += r[e];
                }
                if (padded) {
#pragma unroll
                    for (int e = 0; e < VW; ++e) g[e] = 0.f;
                }
                if (p.thr_out) {
#pragma unroll
                    for (int e = 0; e < VW; ++e)
                        g[e] *= ttsmi_keep_scale(key_out, (uint32_t)row, (uint32_t)(c + e), p.thr_out, p.inv_out);
                }
                if (p.pe) {
                    float pe[VW];
                    ldv<VW>(p.pe + pbase + c, pe);
#pragma unroll
                    for (int e = 0; e < VW; ++e) as += g[e] * pe[e];
                }
#pragma unroll
                for (int e = 0; e < VW; ++e) {
                    float nn = (xv[e] - mean) * rstd;
                    n[j][e] = nn;
                    ab[j][e] += g[e];
                    ag[j][e] += g[e] * nn;
                    float t = g[e] * gam[e];
                    gn[j][e] = t;
                    s1 += t;
                    s2 += t * nn;
                }
            } else {
#pragma unroll
                for (int e = 0; e < VW; ++e) { n[j][e] = 0.f; gn[j][e] = 0.f; keep[j][e] = 0.f; }
            }
        }
        s1 = wave_sum(s1) * invC;
        s2 = wave_sum(s2) * invC;
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            int c = (lane + 64 * j) * VW;
            if (c < p.C) {
                float dz[VW], dxv[VW];
#pragma unroll
                for (int e = 0; e < VW; ++e) {
                    dz[e] = rstd * (gn[j][e] - s1 - n[j][e] * s2);
                    dxv[e] = dz[e] * keep[j][e];
                }
                if (p.dres && p.dres != p.dx) stv<VW>(p.dres + base + c, dz);
                if (p.dx) stv<VW>(p.dx + base + c, dxv);
                if (p.dx_h) sth<VW>(p.dx_h + base + c, dxv);
            }
        }
    }
    // parameter-gradient partials: the block's 4 waves are summed through LDS in wave order, so the
    // reduce kernel reads one partial row per block instead of one per wave
    __shared__ float red[LN_WAVES - 1][2][NPL * 64 * VW];
    const int w = threadIdx.x >> 6;
    if (w > 0) {
#pragma unroll
        for (int j = 0; j < NPL; ++j)
#pragma unroll
            for (int e = 0; e < VW; ++e) {
                red[w - 1][0][(lane + 64 * j) * VW + e] = ag[j][e];
                red[w - 1][1][(lane + 64 * j) * VW + e] = ab[j][e];
            }
    }
    as = wave_sum(as);
    __shared__ float red_s[LN_WAVES];
    if (lane == 0) red_s[w] = as;
    __syncthreads();
    if (w == 0) {
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            int c = (lane + 64 * j) * VW;
#pragma unroll
            for (int u = 0; u < LN_WAVES - 1; ++u)
#pragma unroll
                for (int e = 0; e < VW; ++e) {
                    ag[j][e] += red[u][0][c + e];
                    ab[j][e] += red[u][1][c + e];
                }
            if (c < p.C) {
                stv<VW>(p.part_g + (long)blockIdx.x * p.C + c, ag[j]);
                stv<VW>(p.part_b + (long)blockIdx.x * p.C + c, ab[j]);
            }
        }
        if (lane == 0) {
            float t = red_s[0];
#pragma unroll
            for (int u = 1; u < LN_WAVES; ++u) t += red_s[u];
            p.part_s[blockIdx.x] = t;
        }
    }
}

// out[c] = sum_w part[w][c]: block = 16 columns x 16 row-lanes (64 B segments per row-lane), so a
// C=256 reduction runs on 16 workgroups with 16x more loads in flight than a column-per-thread sweep
__global__ __launch_bounds__(256) void ln_param_reduce_kernel(const float* __restrict__ part_g,
                                                              const float* __restrict__ part_b,
                                                              const float* __restrict__ part_s,
                                                              float* dgamma, float* dbeta,
                                                              float* dscale, int nw, int C) {
    __shared__ float red[2][16][17];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float sg = 0.f, sb = 0.f;
    if (c < C) {
#pragma unroll 8
        for (int w = rl; w < nw; w += 16) {
            sg += part_g[(long)w * C + c];
            sb += part_b[(long)w * C + c];
        }
    }
    red[0][rl][cl] = sg;
    red[1][rl][cl] = sb;
    __syncthreads();
    if (rl == 0 && c < C) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { a += red[0][r][cl]; b += red[1][r][cl]; }
        dgamma[c] = a;
        dbeta[c] = b;
    }
    if (blockIdx.x == 0 && dscale) {
        __shared__ float ws4[4];
        __syncthreads();
        float s = 0.f;
        for (int w = threadIdx.x; w < nw; w += 256) s += part_s[w];
        s = wave_sum(s);
        if ((threadIdx.x & 63) == 0) ws4[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) dscale[0] = ws4[0] + ws4[1] + ws4[2] + ws4[3];
    }
}

// The same reduction for up to LN_BATCH_MAX LayerNorm instances in ONE launch: a training step has ~30
// LayerNorm backward calls whose parameter-gradient reductions are 2 MB reads each - as separate launches they
// cost ~6.7 us apiece on the critical path (0.2 ms of a 7.7 ms step), batched they are one ~20 us kernel.
// The descriptors travel by value in the kernel arguments (2.1 KB), so nothing is staged in device memory.
#define LN_BATCH_MAX 48
struct LnReduceItem {
    const float* part;      // [2*nw*C + nw]: partial dgamma rows, partial dbeta rows, partial dscale
    float* dgamma;
    float* dbeta;
    float* dscale;          // NULL when the instance has no positional-encoding scalar
    int nw, C;
};
struct LnReduceBatch {
    LnReduceItem it[LN_BATCH_MAX];
    int blk_off[LN_BATCH_MAX + 1];   // first block of each item (16 columns per block)
    int n;
};

__global__ __launch_bounds__(256) void ln_param_reduce_batched_kernel(LnReduceBatch b) {
    __shared__ float red[2][16][17];
    int item = 0;
    while (item + 1 < b.n && (int)blockIdx.x >= b.blk_off[item + 1]) ++item;
    const LnReduceItem I = b.it[item];
    const int blk = blockIdx.x - b.blk_off[item];
    const int nw = I.nw, C = I.C;
    const float* __restrict__ part_g = I.part;
    const float* __restrict__ part_b = I.part + (long)nw * C;
    const float* __restrict__ part_s = part_b + (long)nw * C;
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blk * 16 + cl;
    float sg = 0.f, sb = 0.f;
    if (c < C) {
#pragma unroll 8
        for (int w = rl; w < nw; w += 16) {
            sg += part_g[(long)w * C + c];
            sb += part_b[(long)w * C + c];
        }
    }
    red[0][rl][cl] = sg;
    red[1][rl][cl] = sb;
    __syncthreads();
    if (rl == 0 && c < C) {
        float a = 0.f, bsum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { a += red[0][r][cl]; bsum += red[1][r][cl]; }
        I.dgamma[c] = a;
        I.dbeta[c] = bsum;
    }
    if (blk == 0 && I.dscale) {                      // uniform per block: every thread takes the branch
        __shared__ float ws4[4];
        __syncthreads();
        float s = 0.f;
        for (int w = threadIdx.x; w < nw; w += 256) s += part_s[w];
        s = wave_sum(s);
        if ((threadIdx.x & 63) == 0) ws4[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) I.dscale[0] = ws4[0] + ws4[1] + ws4[2] + ws4[3];
    }
}

static int ln_bwd_blocks(int M) {
    // rows are walked by a wave one after the other (each row is a dependent load -> reduce -> store chain),
    // so the grid sets how many chains run in parallel; TTSMI_LN_BWD_BLOCKS overrides the cap (measurement)
    TTSMI_KNOB(cap_env, "TTSMI_LN_BWD_BLOCKS", 1024);
    const int cap = cap_env < 1 ? 1024 : cap_env;
    int b = ttsmi_cdiv(M, LN_WAVES);
    return b > cap ? cap : (b < 1 ? 1 : b);
}

template <int VW, int NPL>
static void launch_fwd(const LnP& p, hipStream_t st) {
    hipLaunchKernelGGL((ln_fwd_kernel<VW, NPL>), dim3(ttsmi_cdiv(p.M, LN_WAVES)), dim3(256), 0, st, p);
}
template <int VW, int NPL>
static void launch_bwd(const LnP& p, hipStream_t st) {
    hipLaunchKernelGGL((ln_bwd_kernel<VW, NPL>), dim3(ln_bwd_blocks(p.M)), dim3(256), 0, st, p);
}

template <bool FWD>
static int dispatch(const LnP& p, hipStream_t st) {
    const int C = p.C;
    bool vec = (C % 4 == 0) && ((((uintptr_t)p.x) & 15) == 0);
    if (vec) {
        int chunks = ttsmi_cdiv(C / 4, 64);
        if (chunks <= 1) { if (FWD) launch_fwd<4, 1>(p, st); else launch_bwd<4, 1>(p, st); }
        else if (chunks <= 2) { if (FWD) launch_fwd<4, 2>(p, st); else launch_bwd<4, 2>(p, st); }
        else if (chunks <= 4) { if (FWD) launch_fwd<4, 4>(p, st); else launch_bwd<4, 4>(p, st); }
        else if (chunks <= 8) { if (FWD) launch_fwd<4, 8>(p, st); else launch_bwd<4, 8>(p, st); }
        else return TTSMI_ERR_UNSUPPORTED;
    } else {
        int chunks = ttsmi_cdiv(C, 64);
        if (chunks <= 4) { if (FWD) launch_fwd<1, 4>(p, st); else launch_bwd<1, 4>(p, st); }
        else if (chunks <= 8) { if (FWD) launch_fwd<1, 8>(p, st); else launch_bwd<1, 8>(p, st); }
        else if (chunks <= 16) { if (FWD) launch_fwd<1, 16>(p, st); else launch_bwd<1, 16>(p, st); }
        else if (chunks <= 32) { if (FWD) launch_fwd<1, 32>(p, st); else launch_bwd<1, 32>(p, st); }
        else return TTSMI_ERR_UNSUPPORTED;
    }
    return TTSMI_OK;
}

static void set_drop(LnP& p, float p_in, uint32_t site_in, float p_out, uint32_t site_out,
                     uint64_t seed, const int64_t* step_dev) {
    p.step_dev = step_dev;
    p.thr_in = p_in > 0.f ? ttsmi_drop_threshold(p_in) : 0;
    p.thr_out = p_out > 0.f ? ttsmi_drop_threshold(p_out) : 0;
    p.inv_in = p_in > 0.f ? 1.0f / (1.0f - p_in) : 1.f;
    p.inv_out = p_out > 0.f ? 1.0f / (1.0f - p_out) : 1.f;
    p.site_in = site_in; p.site_out = site_out; p.seed = seed;
}

extern "C" {

int ttsmi_add_layernorm_fwd(const float* x, const float* res, const float* gamma,
                            const float* beta, const float* pe, const float* pe_scale, int T,
                            const uint8_t* row_pad, float p_in, uint32_t site_in, float p_out,
                            uint32_t site_out, uint64_t seed, const int64_t* step_dev, float eps,
                            float* y, float* mean,
                            float* rstd, int M, int C, uint16_t* y_bf16, ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(x && gamma && beta && y && mean && rstd, "add_layernorm_fwd: null pointer");
    TTSMI_CHECK_ARG(M >= 0 && C > 0, "add_layernorm_fwd: bad shape M=%d C=%d", M, C);
    TTSMI_CHECK_ARG(!pe || (pe_scale && T > 0), "add_layernorm_fwd: pe needs pe_scale and T");
    TTSMI_CHECK_ARG(p_in >= 0.f && p_in < 1.f && p_out >= 0.f && p_out < 1.f,
                    "add_layernorm_fwd: dropout rate out of [0,1)");
    if (M == 0) return TTSMI_OK;
    LnP p;
    memset(&p, 0, sizeof(p));
    p.x = x; p.res = res; p.gamma = gamma; p.beta = beta; p.pe = pe; p.pe_scale = pe_scale; p.T = T;
    p.row_pad = row_pad; p.eps = eps; p.y = y; p.mean = mean; p.rstd = rstd; p.M = M; p.C = C;
    p.y_h = y_bf16;
    set_drop(p, p_in, site_in, p_out, site_out, seed, step_dev);
    int rc = dispatch<true>(p, (hipStream_t)stream);
    if (rc) { ttsmi_set_error("add_layernorm_fwd: C=%d too wide", C); return rc; }
    TTSMI_CHECK_LAUNCH("add_layernorm_fwd");
    return TTSMI_OK;
}

size_t ttsmi_add_layernorm_bwd_ws_bytes(int M, int C) {
    size_t nw = (size_t)ln_bwd_blocks(M);
    return (2 * nw * (size_t)C + nw) * sizeof(float) + 256;
}

int ttsmi_add_layernorm_bwd(const float* dy, const float* x, const float* res, const float* gamma,
                            const float* mean, const float* rstd, const float* pe,
                            const float* pe_scale, int T, const uint8_t* row_pad, float p_in,
                            uint32_t site_in, float p_out, uint32_t site_out, uint64_t seed,
                            const int64_t* step_dev, int relu_in, float* dx, float* dres, float* dgamma, float* dbeta,
                            float* dpe_scale, int M, int C, void* ws, size_t ws_bytes,
                            uint16_t* dx_bf16, ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(dy && x && gamma && mean && rstd && (dx || dx_bf16), "add_layernorm_bwd: null pointer");
    TTSMI_CHECK_ARG((dgamma != nullptr) == (dbeta != nullptr),
                    "add_layernorm_bwd: dgamma and dbeta must both be given or both be NULL (deferred reduction)");
    TTSMI_CHECK_ARG(M > 0 && C > 0, "add_layernorm_bwd: bad shape");
    TTSMI_CHECK_ARG(ws && ws_bytes >= ttsmi_add_layernorm_bwd_ws_bytes(M, C),
                    "add_layernorm_bwd: workspace too small");
    TTSMI_CHECK_ARG(!(res && !dres), "add_layernorm_bwd: res given but dres is NULL");
    TTSMI_CHECK_ARG(!(res && dres == dx && (p_in > 0.f || relu_in)),
                    "add_layernorm_bwd: dres may alias dx only when p_in == 0 and !relu_in");
    LnP p;
    memset(&p, 0, sizeof(p));
    p.x = x; p.res = res; p.gamma = gamma; p.pe = pe; p.pe_scale = pe_scale; p.T = T;
    p.row_pad = row_pad; p.M = M; p.C = C;
    p.dy = dy; p.mean_in = mean; p.rstd_in = rstd; p.relu_in = relu_in;
    p.dx = dx; p.dres = res ? dres : nullptr; p.dx_h = dx_bf16;
    set_drop(p, p_in, site_in, p_out, site_out, seed, step_dev);
    size_t nw = (size_t)ln_bwd_blocks(M);
    p.part_g = (float*)ws;
    p.part_b = p.part_g + nw * C;
    p.part_s = p.part_b + nw * C;
    hipStream_t st = (hipStream_t)stream;
    int rc = dispatch<false>(p, st);
    if (rc) { ttsmi_set_error("add_layernorm_bwd: C=%d too wide", C); return rc; }
    TTSMI_CHECK_LAUNCH("add_layernorm_bwd");
    if (!dgamma) return TTSMI_OK;                    // deferred: the partial sums stay in ws for the batched reduce
    hipLaunchKernelGGL(ln_param_reduce_kernel, dim3(ttsmi_cdiv(C, 16)), dim3(256), 0, st, p.part_g,
                       p.part_b, p.part_s, dgamma, dbeta, pe ? dpe_scale : nullptr, (int)nw, C);
    TTSMI_CHECK_LAUNCH("ln_param_reduce");
    return TTSMI_OK;
}

int ttsmi_add_layernorm_bwd_nparts(int M) { return ln_bwd_blocks(M); }

static int reduce_batched(const void* const* ws, float* const* dgamma, float* const* dbeta, float* const* dpe_scale,
                          const int* M, const int* nparts, const int* C, int n, ttsmi_stream_t stream);

int ttsmi_layernorm_param_reduce_batched(const void* const* ws, float* const* dgamma, float* const* dbeta,
                                         float* const* dpe_scale, const int* M, const int* C, int n,
                                         ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(M, "layernorm_param_reduce_batched: null pointer");
    return reduce_batched(ws, dgamma, dbeta, dpe_scale, M, nullptr, C, n, stream);
}

int ttsmi_layernorm_param_reduce_batched_nw(const void* const* ws, float* const* dgamma, float* const* dbeta,
                                            float* const* dpe_scale, const int* nparts, const int* C, int n,
                                            ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(nparts, "layernorm_param_reduce_batched_nw: null pointer");
    return reduce_batched(ws, dgamma, dbeta, dpe_scale, nullptr, nparts, C, n, stream);
}

}  // extern "C"

static int reduce_batched(const void* const* ws, float* const* dgamma, float* const* dbeta, float* const* dpe_scale,
                          const int* M, const int* nparts, const int* C, int n, ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(ws && dgamma && dbeta && dpe_scale && C, "layernorm_param_reduce_batched: null pointer");
    TTSMI_CHECK_ARG(n >= 0, "layernorm_param_reduce_batched: bad count");
    hipStream_t st = (hipStream_t)stream;
    for (int base = 0; base < n; base += LN_BATCH_MAX) {
        LnReduceBatch b;
        memset(&b, 0, sizeof(b));
        b.n = n - base < LN_BATCH_MAX ? n - base : LN_BATCH_MAX;
        int blocks = 0;
        for (int i = 0; i < b.n; ++i) {
            const int j = base + i;
            TTSMI_CHECK_ARG(ws[j] && dgamma[j] && dbeta[j] && (M ? M[j] : nparts[j]) > 0 && C[j] > 0,
                            "layernorm_param_reduce_batched: bad item");
            b.it[i].part = (const float*)ws[j];
            b.it[i].dgamma = dgamma[j];
            b.it[i].dbeta = dbeta[j];
            b.it[i].dscale = dpe_scale[j];
            b.it[i].nw = M ? ln_bwd_blocks(M[j]) : nparts[j];
            b.it[i].C = C[j];
            b.blk_off[i] = blocks;
            blocks += ttsmi_cdiv(C[j], 16);
        }
        b.blk_off[b.n] = blocks;
        hipLaunchKernelGGL(ln_param_reduce_batched_kernel, dim3(blocks), dim3(256), 0, st, b);
        TTSMI_CHECK_LAUNCH("ln_param_reduce_batched");
    }
    return TTSMI_OK;
}
