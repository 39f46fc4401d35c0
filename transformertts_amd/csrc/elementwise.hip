// Small HBM-bound ops of ForwardTransformer.call: padding masks, embedding gather / scatter-add,
// pitch embedding, Dense(C->1) heads, row masks, L1 loss (fwd+bwd fused), TF-form Adam, bf16 cast.
// All reductions are deterministic (fixed partial order, no float atomics).
#include "common.h"

// ---------------------------------------------------------------------------------------------
// padding masks (model/transformer_utils.py:24-32)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void token_pad_mask_kernel(const int32_t* __restrict__ tok,
                                                             const int32_t* __restrict__ len,
                                                             uint8_t* __restrict__ pad,
                                                             int32_t* __restrict__ klen, int T) {
    // one block per batch row
    __shared__ int last[4];
    const int b = blockIdx.x;
    int mylast = -1;
    for (int t = threadIdx.x; t < T; t += 256) {
        bool is_pad = tok ? (tok[(long)b * T + t] == 0) : (t >= len[b]);
        pad[(long)b * T + t] = is_pad ? 1 : 0;
        if (!is_pad) mylast = t;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mylast = max(mylast, __shfl_xor(mylast, o, 64));
    if ((threadIdx.x & 63) == 0) last[threadIdx.x >> 6] = mylast;
    __syncthreads();
    if (threadIdx.x == 0) {
        int l = max(max(last[0], last[1]), max(last[2], last[3]));
        klen[b] = l < 0 ? T : l + 1;       // every key padded -> all keys take part (all at -1e9)
    }
}

// ---------------------------------------------------------------------------------------------
// embedding
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embedding_fwd_kernel(const int32_t* __restrict__ tok,
                                                            const float* __restrict__ table,
                                                            float* __restrict__ y, int M, int V,
                                                            int C) {
    long n = (long)M * C;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        int m = (int)(i / C), c = (int)(i - (long)m * C);
        int t = tok[m];
        y[i] = (t >= 0 && t < V) ? table[(long)t * C + c] : 0.f;
    }
}
// One block per vocabulary row.  Positions are scanned in chunks of 256 x EMB_PER: a thread tests EMB_PER CONSECUTIVE
// positions, an exclusive scan of the 256 match counts (wave shuffles + four wave totals) gives its slot, and the
// matching positions land in an LDS list in ascending order; then only those rows of dy are summed, in that order ->
// deterministic, ~M/V row reads per block, and three barriers per 4 096 positions (the first version compacted 256
// positions per pass with four barriers each: 100 barriers at M = 6 400, most of its 38 us).
#define EMB_PER 16
#define EMB_CHUNK (256 * EMB_PER)
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const int32_t* __restrict__ tok,
                                                            const float* __restrict__ dy,
                                                            float* __restrict__ dtable, int M,
                                                            int C) {
    __shared__ int list[EMB_CHUNK];
    __shared__ int wtot[4];
    const int v = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c0 = 0; c0 < C; c0 += 256) {
        const int c = c0 + tid;
        float s = 0.f;
        for (int m0 = 0; m0 < M; m0 += EMB_CHUNK) {
            const int mb = m0 + tid * EMB_PER;
            unsigned bits = 0;
#pragma unroll
            for (int i = 0; i < EMB_PER; ++i)
                if (mb + i < M && tok[mb + i] == v) bits |= 1u << i;
            const int cnt = __popc(bits);
            int incl = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int t = __shfl_up(incl, d);
                if (lane >= d) incl += t;
            }
            if (lane == 63) wtot[wave] = incl;
            __syncthreads();
            int slot = incl - cnt;
            for (int w = 0; w < wave; ++w) slot += wtot[w];
            const int n = wtot[0] + wtot[1] + wtot[2] + wtot[3];
#pragma unroll
            for (int i = 0; i < EMB_PER; ++i)
                if (bits & (1u << i)) list[slot++] = mb + i;
            __syncthreads();
            if (c < C) {
#pragma unroll 8
                for (int i = 0; i < n; ++i) s += dy[(long)list[i] * C + c];
            }
            __syncthreads();                                 // the list is rewritten by the next chunk
        }
        if (c < C) dtable[(long)v * C + c] = s;
    }
}

// ---------------------------------------------------------------------------------------------
// pitch embedding: y = x + relu(p*w + b)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pitch_embed_fwd_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ p,
                                                              const float* __restrict__ w,
                                                              const float* __restrict__ b,
                                                              float* __restrict__ y, int M, int C) {
    long n = (long)M * C;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        int m = (int)(i / C), c = (int)(i - (long)m * C);
        y[i] = x[i] + fmaxf(p[m] * w[c] + b[c], 0.f);
    }
}
// stage 1: block (by) handles rows [by*256, +256) for 64 columns: partial dw, db; dp per row
__global__ __launch_bounds__(256) void pitch_embed_bwd_kernel(const float* __restrict__ dy,
                                                              const float* __restrict__ p,
                                                              const float* __restrict__ w,
                                                              const float* __restrict__ b,
                                                              float* __restrict__ part_w,
                                                              float* __restrict__ part_b, int M,
                                                              int C) {
    __shared__ float red[2][4][64];
    int c = blockIdx.x * 64 + (threadIdx.x & 63);
    int rl = threadIdx.x >> 6;
    int r0 = blockIdx.y * 256;
    float sw = 0.f, sb = 0.f;
    if (c < C) {
        float wc = w[c], bc = b[c];
        int rend = min(M, r0 + 256);
#pragma unroll 8
        for (int r = r0 + rl; r < rend; r += 4) {        // unconditional loads: eight rows in flight per thread
            const float pr = p[r];
            const float d = dy[(long)r * C + c];
            const float g = (pr * wc + bc) > 0.f ? d : 0.f;
            sw += g * pr;
            sb += g;
        }
    }
    red[0][rl][threadIdx.x & 63] = sw;
    red[1][rl][threadIdx.x & 63] = sb;
    __syncthreads();
    if (rl == 0 && c < C) {
        int t = threadIdx.x;
        part_w[(long)blockIdx.y * C + c] = red[0][0][t] + red[0][1][t] + red[0][2][t] + red[0][3][t];
        part_b[(long)blockIdx.y * C + c] = red[1][0][t] + red[1][1][t] + red[1][2][t] + red[1][3][t];
    }
}
__global__ void colpart_final2_kernel(const float* __restrict__ pa, const float* __restrict__ pb,
                                      float* oa, float* ob, int chunks, int C) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float sa = 0.f, sb = 0.f;
    for (int k = 0; k < chunks; ++k) { sa += pa[(long)k * C + c]; sb += pb[(long)k * C + c]; }
    if (oa) oa[c] = sa;
    if (ob) ob[c] = sb;
}
// dp[m] = sum_c dy[m,c] * w[c] * (p*w+b > 0): one wave per row
__global__ __launch_bounds__(256) void pitch_embed_dp_kernel(const float* __restrict__ dy,
                                                             const float* __restrict__ p,
                                                             const float* __restrict__ w,
                                                             const float* __restrict__ b,
                                                             float* __restrict__ dp, int M, int C) {
    int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    float pr = p[row], s = 0.f;
    for (int c = lane; c < C; c += 64)
        if (pr * w[c] + b[c] > 0.f) s += dy[(long)row * C + c] * w[c];
    s = wave_sum(s);
    if (lane == 0) dp[row] = s;
}

// ---------------------------------------------------------------------------------------------
// Dense(C -> 1) heads
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rowdot_fwd_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ w,
                                                         const float* __restrict__ b,
                                                         const uint8_t* __restrict__ row_pad,
                                                         float* __restrict__ y, int M, int C,
                                                         int relu) {
    int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += x[(long)row * C + c] * w[c];
    s = wave_sum(s);
    if (lane == 0) {
        float v = s + b[0];
        if (relu) v = fmaxf(v, 0.f);
        if (row_pad && row_pad[row]) v = 0.f;
        y[row] = v;
    }
}
// g[m] = dy[m] * (!pad) * (relu ? y>0 : 1);  dx[m,c] = g[m] w[c];  partial dw[c] = sum_m g[m] x[m,c]
__global__ __launch_bounds__(256) void rowdot_bwd_kernel(const float* __restrict__ dy,
                                                         const float* __restrict__ y,
                                                         const float* __restrict__ x,
                                                         const float* __restrict__ w,
                                                         const uint8_t* __restrict__ row_pad,
                                                         float* __restrict__ dx,
                                                         float* __restrict__ part_w,
                                                         float* __restrict__ part_b, int M, int C,
                                                         int relu) {
    __shared__ float red[2][4][64];
    int c = blockIdx.x * 64 + (threadIdx.x & 63);
    int rl = threadIdx.x >> 6;
    int r0 = blockIdx.y * 256;
    float sw = 0.f, sb = 0.f;
    if (c < C) {
        float wc = w[c];
        int rend = min(M, r0 + 256);
        // every load of a row is unconditional and eight rows are in flight per thread (the conditional loads of the first
        // version serialised 64 dependent round trips per thread: 54 us for 6 MB)
        const bool has_pad = row_pad != nullptr;
#pragma unroll 8
        for (int r = r0 + rl; r < rend; r += 4) {
            const float gy = dy[r], yv = y[r], xv = x[(long)r * C + c];
            const bool padded = has_pad && row_pad[r] != 0;
            const float g = (padded || (relu && !(yv > 0.f))) ? 0.f : gy;
            dx[(long)r * C + c] = g * wc;
            sw += g * xv;
            sb += g;
        }
    }
    red[0][rl][threadIdx.x & 63] = sw;
    red[1][rl][threadIdx.x & 63] = sb;
    __syncthreads();
    if (rl == 0 && c < C) {
        int t = threadIdx.x;
        part_w[(long)blockIdx.y * C + c] = red[0][0][t] + red[0][1][t] + red[0][2][t] + red[0][3][t];
        part_b[(long)blockIdx.y * C + c] = red[1][0][t] + red[1][1][t] + red[1][2][t] + red[1][3][t];
    }
}
__global__ void rowdot_final_kernel(const float* __restrict__ pw, const float* __restrict__ pb,
                                    float* dw, float* db, int chunks, int C) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float sa = 0.f, sb = 0.f;
    for (int k = 0; k < chunks; ++k) { sa += pw[(long)k * C + c]; sb += pb[(long)k * C + c]; }
    dw[c] = sa;
    if (c == 0) db[0] = sb;     // every column carries the same sum_m g[m]
}

__global__ __launch_bounds__(256) void rowmask_mul_kernel(const float* __restrict__ x,
                                                          const uint8_t* __restrict__ row_pad,
                                                          float* __restrict__ y, int M, int C) {
    long n = (long)M * C;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        int m = (int)(i / C);
        y[i] = row_pad[m] ? 0.f : x[i];
    }
}

// ---------------------------------------------------------------------------------------------
// L1 loss: block partial sums of |t - p| + gradient in one pass, then a one-block final reduce
// ---------------------------------------------------------------------------------------------
#define L1_BLOCKS 512
__global__ __launch_bounds__(256) void l1_loss_kernel(const float* __restrict__ pred, long ldp,
                                                      const void* __restrict__ target, int tint,
                                                      long rows, long cols, float gscale,
                                                      float* __restrict__ grad, long ldg,
                                                      float* __restrict__ part) {
    __shared__ float red[4];
    long n = rows * cols;
    float s = 0.f;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        long r = i / cols, c = i - r * cols;
        float t = tint ? (float)((const int32_t*)target)[i] : ((const float*)target)[i];
        float d = pred[r * ldp + c] - t;
        s += fabsf(d);
        if (grad) grad[r * ldg + c] = d > 0.f ? gscale : (d < 0.f ? -gscale : 0.f);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void l1_final_kernel(const float* __restrict__ part, int nb,
                                                       float inv_n, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nb; i += 256) s += part[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (red[0] + red[1] + red[2] + red[3]) * inv_n;
}

// ---------------------------------------------------------------------------------------------
// TF-form Adam on the flat parameter buffer (+ optional bf16 shadow copy)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__global__ __launch_bounds__(256) void adam_tf_kernel(float* __restrict__ p,
                                                      const float* __restrict__ g,
                                                      float* __restrict__ m, float* __restrict__ v,
                                                      long n, const float* __restrict__ lr_dev,
                                                      const int64_t* __restrict__ step_dev,
                                                      float b1, float b2, float eps,
                                                      uint16_t* __restrict__ shadow) {
    // lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t), t = iteration (1-based), in double like a host would
    const double t = (double)step_dev[0];
    const float lr_t = (float)((double)lr_dev[0] * sqrt(1.0 - pow((double)b2, t)) /
                               (1.0 - pow((double)b1, t)));
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float gi = g[i];
        float mi = b1 * m[i] + (1.f - b1) * gi;
        float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        float pi = p[i] - lr_t * mi / (sqrtf(vi) + eps);
        m[i] = mi; v[i] = vi; p[i] = pi;
        if (shadow) shadow[i] = f32_to_bf16_rne(pi);
    }
}
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ s,
                                                        uint16_t* __restrict__ d, long n) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        d[i] = f32_to_bf16_rne(s[i]);
}

static int ew_blocks(long n) {
    long b = (n + 255) / 256;
    return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

extern "C" {

int ttsmi_token_pad_mask(const int32_t* tokens, uint8_t* key_pad, int32_t* klen, int B, int T,
                         ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(tokens && key_pad && klen && B >= 0 && T > 0, "token_pad_mask: bad argument");
    if (B == 0) return TTSMI_OK;
    hipLaunchKernelGGL(token_pad_mask_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, tokens,
                       (const int32_t*)nullptr, key_pad, klen, T);
    TTSMI_CHECK_LAUNCH("token_pad_mask");
    return TTSMI_OK;
}
int ttsmi_length_pad_mask(const int32_t* len, uint8_t* key_pad, int32_t* klen, int B, int T,
                          ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(len && key_pad && klen && B >= 0 && T > 0, "length_pad_mask: bad argument");
    if (B == 0) return TTSMI_OK;
    hipLaunchKernelGGL(token_pad_mask_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream,
                       (const int32_t*)nullptr, len, key_pad, klen, T);
    TTSMI_CHECK_LAUNCH("length_pad_mask");
    return TTSMI_OK;
}

int ttsmi_embedding_fwd(const int32_t* tokens, const float* table, float* y, int M, int V, int C,
                        ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(tokens && table && y && M >= 0 && V > 0 && C > 0, "embedding_fwd: bad argument");
    if (M == 0) return TTSMI_OK;
    hipLaunchKernelGGL(embedding_fwd_kernel, dim3(ew_blocks((long)M * C)), dim3(256), 0,
                       (hipStream_t)stream, tokens, table, y, M, V, C);
    TTSMI_CHECK_LAUNCH("embedding_fwd");
    return TTSMI_OK;
}
int ttsmi_embedding_bwd(const int32_t* tokens, const float* dy, float* dtable, int M, int V, int C,
                        ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(tokens && dy && dtable && M >= 0 && V > 0 && C > 0, "embedding_bwd: bad argument");
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3(V), dim3(256), 0, (hipStream_t)stream, tokens, dy,
                       dtable, M, C);
    TTSMI_CHECK_LAUNCH("embedding_bwd");
    return TTSMI_OK;
}

int ttsmi_pitch_embed_fwd(const float* x, const float* p, const float* w, const float* b, float* y,
                          int M, int C, ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(x && p && w && b && y && M >= 0 && C > 0, "pitch_embed_fwd: bad argument");
    if (M == 0) return TTSMI_OK;
    hipLaunchKernelGGL(pitch_embed_fwd_kernel, dim3(ew_blocks((long)M * C)), dim3(256), 0,
                       (hipStream_t)stream, x, p, w, b, y, M, C);
    TTSMI_CHECK_LAUNCH("pitch_embed_fwd");
    return TTSMI_OK;
}
size_t ttsmi_pitch_embed_bwd_ws_bytes(int M, int C) {
    return (size_t)2 * ttsmi_cdiv(M, 256) * C * sizeof(float) + 256;
}
int ttsmi_pitch_embed_bwd(const float* dy, const float* p, const float* w, const float* b,
                          float* dp, float* dw, float* db, int M, int C, void* ws, size_t ws_bytes,
                          ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(dy && p && w && b && dw && db && M > 0 && C > 0, "pitch_embed_bwd: bad argument");
    TTSMI_CHECK_ARG(ws && ws_bytes >= ttsmi_pitch_embed_bwd_ws_bytes(M, C),
                    "pitch_embed_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    int chunks = ttsmi_cdiv(M, 256);
    float* pw = (float*)ws;
    float* pb = pw + (size_t)chunks * C;
    hipLaunchKernelGGL(pitch_embed_bwd_kernel, dim3(ttsmi_cdiv(C, 64), chunks), dim3(256), 0, st,
                       dy, p, w, b, pw, pb, M, C);
    TTSMI_CHECK_LAUNCH("pitch_embed_bwd");
    hipLaunchKernelGGL(colpart_final2_kernel, dim3(ttsmi_cdiv(C, 128)), dim3(128), 0, st, pw, pb,
                       dw, db, chunks, C);
    TTSMI_CHECK_LAUNCH("pitch_embed_bwd_final");
    if (dp) {
        hipLaunchKernelGGL(pitch_embed_dp_kernel, dim3(ttsmi_cdiv(M, 4)), dim3(256), 0, st, dy, p,
                           w, b, dp, M, C);
        TTSMI_CHECK_LAUNCH("pitch_embed_dp");
    }
    return TTSMI_OK;
}

int ttsmi_rowdot_fwd(const float* x, const float* w, const float* b, const uint8_t* row_pad,
                     float* y, int M, int C, int relu, ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(x && w && b && y && M >= 0 && C > 0, "rowdot_fwd: bad argument");
    if (M == 0) return TTSMI_OK;
    hipLaunchKernelGGL(rowdot_fwd_kernel, dim3(ttsmi_cdiv(M, 4)), dim3(256), 0,
                       (hipStream_t)stream, x, w, b, row_pad, y, M, C, relu);
    TTSMI_CHECK_LAUNCH("rowdot_fwd");
    return TTSMI_OK;
}
size_t ttsmi_rowdot_bwd_ws_bytes(int M, int C) {
    return (size_t)2 * ttsmi_cdiv(M, 256) * C * sizeof(float) + 256;
}
int ttsmi_rowdot_bwd(const float* dy, const float* y, const float* x, const float* w,
                     const uint8_t* row_pad, float* dx, float* dw, float* db, int M, int C,
                     int relu, void* ws, size_t ws_bytes, ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(dy && y && x && w && dx && dw && db && M > 0 && C > 0, "rowdot_bwd: bad argument");
    TTSMI_CHECK_ARG(ws && ws_bytes >= ttsmi_rowdot_bwd_ws_bytes(M, C), "rowdot_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    int chunks = ttsmi_cdiv(M, 256);
    float* pw = (float*)ws;
    float* pb = pw + (size_t)chunks * C;
    hipLaunchKernelGGL(rowdot_bwd_kernel, dim3(ttsmi_cdiv(C, 64), chunks), dim3(256), 0, st, dy, y,
                       x, w, row_pad, dx, pw, pb, M, C, relu);
    TTSMI_CHECK_LAUNCH("rowdot_bwd");
    hipLaunchKernelGGL(rowdot_final_kernel, dim3(ttsmi_cdiv(C, 128)), dim3(128), 0, st, pw, pb, dw,
                       db, chunks, C);
    TTSMI_CHECK_LAUNCH("rowdot_bwd_final");
    return TTSMI_OK;
}

int ttsmi_rowmask_mul(const float* x, const uint8_t* row_pad, float* y, int M, int C,
                      ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(x && row_pad && y && M >= 0 && C > 0, "rowmask_mul: bad argument");
    if (M == 0) return TTSMI_OK;
    hipLaunchKernelGGL(rowmask_mul_kernel, dim3(ew_blocks((long)M * C)), dim3(256), 0,
                       (hipStream_t)stream, x, row_pad, y, M, C);
    TTSMI_CHECK_LAUNCH("rowmask_mul");
    return TTSMI_OK;
}

size_t ttsmi_l1_loss_ws_bytes(int64_t n) { (void)n; return L1_BLOCKS * sizeof(float) + 256; }
int ttsmi_l1_loss(const float* pred, int64_t ld_pred, const void* target, int target_is_int,
                  int64_t rows, int64_t cols, float coeff, float* grad, int64_t ld_grad,
                  float* loss_out, void* ws, size_t ws_bytes, ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(pred && target && loss_out && rows > 0 && cols > 0, "l1_loss: bad argument");
    TTSMI_CHECK_ARG(ws && ws_bytes >= ttsmi_l1_loss_ws_bytes(rows * cols), "l1_loss: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    long n = rows * cols;
    int nb = ew_blocks(n);
    if (nb > L1_BLOCKS) nb = L1_BLOCKS;
    float gscale = coeff / (float)n;
    hipLaunchKernelGGL(l1_loss_kernel, dim3(nb), dim3(256), 0, st, pred, (long)ld_pred, target,
                       target_is_int, (long)rows, (long)cols, gscale, grad, (long)ld_grad,
                       (float*)ws);
    TTSMI_CHECK_LAUNCH("l1_loss");
    hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, nb,
                       1.0f / (float)n, loss_out);
    TTSMI_CHECK_LAUNCH("l1_final");
    return TTSMI_OK;
}

// utils/losses.py:63-70 in one call: every term's partial sums are finished by ONE block, which also forms
// total = ((0 + c0 l0) + c1 l1) + ... in the reference's order (products and sums rounded separately, as eager TF does).
#define L1_MAX_TERMS 8
struct L1FinalP {
    int n;
    int nb[L1_MAX_TERMS];
    float inv_n[L1_MAX_TERMS], coeff[L1_MAX_TERMS];
};
__global__ __launch_bounds__(256) void l1_final_multi_kernel(const float* __restrict__ part, L1FinalP q,
                                                             float* __restrict__ losses, float* __restrict__ total) {
    __shared__ float red[4];
    float tot = 0.f;
    for (int t = 0; t < q.n; ++t) {
        float s = 0.f;
        for (int i = threadIdx.x; i < q.nb[t]; i += 256) s += part[(long)t * L1_BLOCKS + i];
        s = wave_sum(s);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        const float loss = (red[0] + red[1] + red[2] + red[3]) * q.inv_n[t];
        __syncthreads();
        tot = __fadd_rn(tot, __fmul_rn(q.coeff[t], loss));
        if (threadIdx.x == 0) losses[t] = loss;
    }
    if (threadIdx.x == 0) total[0] = tot;
}

size_t ttsmi_l1_losses_weighted_ws_bytes(int n_terms) {
    return (size_t)(n_terms > 0 ? n_terms : 1) * L1_BLOCKS * sizeof(float) + 256;
}
int ttsmi_l1_losses_weighted(int n_terms, const float* const* pred, const int64_t* ld_pred, const void* const* target,
                             const int32_t* target_is_int, const int64_t* rows, const int64_t* cols, const float* coeff,
                             const int64_t* denom, float* const* grad, const int64_t* ld_grad, float* losses_out,
                             float* total_out, void* ws, size_t ws_bytes, ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(n_terms >= 1 && n_terms <= L1_MAX_TERMS, "l1_losses_weighted: %d terms (1..%d)", n_terms, L1_MAX_TERMS);
    TTSMI_CHECK_ARG(pred && ld_pred && target && target_is_int && rows && cols && coeff && grad && ld_grad && losses_out &&
                    total_out, "l1_losses_weighted: null pointer");
    TTSMI_CHECK_ARG(ws && ws_bytes >= ttsmi_l1_losses_weighted_ws_bytes(n_terms), "l1_losses_weighted: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    L1FinalP q;
    q.n = n_terms;
    for (int t = 0; t < n_terms; ++t) {
        TTSMI_CHECK_ARG(pred[t] && target[t] && rows[t] > 0 && cols[t] > 0, "l1_losses_weighted: bad term %d", t);
        const long n = rows[t] * cols[t];
        // the mean's divisor: the term's own element count, or the GLOBAL count of a batch sharded over ranks
        // (batch data parallelism: every rank divides its partial sum by B_global * T_max_global * C)
        TTSMI_CHECK_ARG(!denom || denom[t] >= 0, "l1_losses_weighted: negative divisor for term %d", t);
        const double div = denom && denom[t] > 0 ? (double)denom[t] : (double)n;
        int nb = ew_blocks(n);
        if (nb > L1_BLOCKS) nb = L1_BLOCKS;
        q.nb[t] = nb; q.inv_n[t] = (float)(1.0 / div); q.coeff[t] = coeff[t];
        hipLaunchKernelGGL(l1_loss_kernel, dim3(nb), dim3(256), 0, st, pred[t], (long)ld_pred[t], target[t], (int)target_is_int[t],
                           (long)rows[t], (long)cols[t], (float)((double)coeff[t] / div), grad[t], (long)ld_grad[t],
                           (float*)ws + (size_t)t * L1_BLOCKS);
        TTSMI_CHECK_LAUNCH("l1_losses_weighted(term)");
    }
    hipLaunchKernelGGL(l1_final_multi_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, q, losses_out, total_out);
    TTSMI_CHECK_LAUNCH("l1_losses_weighted(final)");
    return TTSMI_OK;
}

__global__ void step_increment_kernel(int64_t* s) { s[0] += 1; }
int ttsmi_step_increment(int64_t* step_dev, ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(step_dev, "step_increment: null pointer");
    hipLaunchKernelGGL(step_increment_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev);
    TTSMI_CHECK_LAUNCH("step_increment");
    return TTSMI_OK;
}

int ttsmi_adam_tf(float* p, const float* g, float* m, float* v, int64_t n, const float* lr_dev,
                  const int64_t* step_dev, float b1, float b2, float eps, uint16_t* bf16_copy,
                  ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(p && g && m && v && lr_dev && step_dev && n >= 0, "adam_tf: bad argument");
    if (n == 0) return TTSMI_OK;
    hipLaunchKernelGGL(adam_tf_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, p, g,
                       m, v, (long)n, lr_dev, step_dev, b1, b2, eps, bf16_copy);
    TTSMI_CHECK_LAUNCH("adam_tf");
    return TTSMI_OK;
}

int ttsmi_cast_f32_to_bf16(const float* src, uint16_t* dst, int64_t n, ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(src && dst && n >= 0, "cast_f32_to_bf16: bad argument");
    if (n == 0) return TTSMI_OK;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, src,
                       dst, (long)n);
    TTSMI_CHECK_LAUNCH("cast_f32_to_bf16");
    return TTSMI_OK;
}

}  // extern "C"
