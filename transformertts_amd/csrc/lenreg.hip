// Length regulator (Expand, model/layers.py:527-565) as an integer index build + row gather, with a
// contiguous segment-sum backward.  The reference tiles x max_dur times and boolean-masks through
// RaggedTensors (a B*Tp*max_dur*C temporary); semantically the output row j of sample b is x[b, i]
// for the phoneme i whose cumulative-duration interval contains j.  Integer contract is bit-exact:
// dims = int32(round_half_even(dur)), negative -> 0.
#include "common.h"

// One block per sample: block-wide inclusive scan of the rounded durations (Tp <= a few thousand),
// then every thread fills the idx entries of its own phonemes.
__global__ __launch_bounds__(256) void lenreg_index_kernel(const void* __restrict__ dur, int is_int,
                                                           int32_t* __restrict__ idx,
                                                           int32_t* __restrict__ cum,
                                                           int32_t* __restrict__ len, int Tp,
                                                           int cap) {
    __shared__ int wsum[4];
    __shared__ int carry_s;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    int32_t* idx_b = idx + (long)b * cap;
    int32_t* cum_b = cum + (long)b * (Tp + 1);
    for (int i0 = 0; i0 < Tp; i0 += 256) {
        int i = i0 + tid;
        int d = 0;
        if (i < Tp) {
            if (is_int) {
                d = ((const int32_t*)dur)[(long)b * Tp + i];
            } else {
                // rintf = round-half-to-even under the default rounding mode (tf.math.round)
                d = (int)rintf(((const float*)dur)[(long)b * Tp + i]);
            }
            if (d < 0) d = 0;
        }
        // inclusive scan inside the wave
        int s = d;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int t = __shfl_up(s, o, 64);
            if (lane >= o) s += t;
        }
        if (lane == 63) wsum[wave] = s;
        __syncthreads();
        int base = carry_s;
        for (int w = 0; w < wave; ++w) base += wsum[w];
        int incl = base + s, excl = incl - d;
        if (i < Tp) {
            cum_b[i] = excl;
            int e = incl < cap ? incl : cap;
            for (int j = excl; j < e; ++j) idx_b[j] = i;
        }
        __syncthreads();
        if (tid == 255) carry_s = incl;
        __syncthreads();
    }
    const int total = carry_s;
    if (tid == 0) { cum_b[Tp] = total; len[b] = total; }
    for (int j = (total < cap ? total : cap) + tid; j < cap; j += 256) idx_b[j] = -1;
}

// y[b,j,:] = x[b, idx[b,j], :] or 0; one wave per output row, 16 B per lane when C % 4 == 0
__global__ __launch_bounds__(256) void lenreg_fwd_kernel(const float* __restrict__ x,
                                                         const int32_t* __restrict__ idx,
                                                         float* __restrict__ y, long rows, int Tp,
                                                         int cap, int C) {
    const int lane = threadIdx.x & 63;
    long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    int b = (int)(row / cap);
    int i = idx[row];
    float* yo = y + row * C;
    if ((C & 3) == 0) {
        const float4* xi = i >= 0 ? reinterpret_cast<const float4*>(x + ((long)b * Tp + i) * C) : nullptr;
        for (int c = lane; c < C / 4; c += 64)
            reinterpret_cast<float4*>(yo)[c] = xi ? xi[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        const float* xi = i >= 0 ? x + ((long)b * Tp + i) * C : nullptr;
        for (int c = lane; c < C; c += 64) yo[c] = xi ? xi[c] : 0.f;
    }
}

// dx[b,i,:] = sum of the contiguous rows dy[b, cum[i] .. min(cum[i+1], cap)); one wave per phoneme
__global__ __launch_bounds__(256) void lenreg_bwd_kernel(const float* __restrict__ dy,
                                                         const int32_t* __restrict__ cum,
                                                         float* __restrict__ dx, long rows, int Tp,
                                                         int cap, int C) {
    const int lane = threadIdx.x & 63;
    long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    int b = (int)(row / Tp), i = (int)(row - (long)b * Tp);
    const int32_t* cb = cum + (long)b * (Tp + 1);
    int j0 = cb[i], j1 = cb[i + 1];
    if (j1 > cap) j1 = cap;
    const float* src = dy + (long)b * cap * C;
    if ((C & 3) == 0 && ((((uintptr_t)dy) | ((uintptr_t)dx)) & 15) == 0) {
        // 16 bytes per lane: a 256-channel row is ONE wave-wide load, four frames in flight (same ascending-frame sums)
        for (int c = lane * 4; c < C; c += 256) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
            for (int j = j0; j < j1; ++j) {
                const float4 v = *reinterpret_cast<const float4*>(src + (long)j * C + c);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            *reinterpret_cast<float4*>(dx + row * C + c) = s;
        }
        return;
    }
    for (int c = lane; c < C; c += 64) {
        float s = 0.f;
        for (int j = j0; j < j1; ++j) s += src[(long)j * C + c];
        dx[row * C + c] = s;
    }
}

extern "C" {

int ttsmi_lenreg_index(const void* dur, int dur_is_int, int32_t* idx, int32_t* cum, int32_t* len,
                       int B, int Tp, int cap, ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(dur && idx && cum && len, "lenreg_index: null pointer");
    TTSMI_CHECK_ARG(B >= 0 && Tp > 0 && cap > 0, "lenreg_index: bad shape B=%d Tp=%d cap=%d", B, Tp, cap);
    if (B == 0) return TTSMI_OK;
    hipLaunchKernelGGL(lenreg_index_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dur,
                       dur_is_int, idx, cum, len, Tp, cap);
    TTSMI_CHECK_LAUNCH("lenreg_index");
    return TTSMI_OK;
}

int ttsmi_lenreg_fwd(const float* x, const int32_t* idx, float* y, int B, int Tp, int cap, int C,
                     ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(x && idx && y, "lenreg_fwd: null pointer");
    TTSMI_CHECK_ARG(B >= 0 && Tp > 0 && cap > 0 && C > 0, "lenreg_fwd: bad shape");
    if (B == 0) return TTSMI_OK;
    long rows = (long)B * cap;
    hipLaunchKernelGGL(lenreg_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, x, idx, y, rows, Tp, cap, C);
    TTSMI_CHECK_LAUNCH("lenreg_fwd");
    return TTSMI_OK;
}

int ttsmi_lenreg_bwd(const float* dy, const int32_t* cum, float* dx, int B, int Tp, int cap, int C,
                     ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(dy && cum && dx, "lenreg_bwd: null pointer");
    TTSMI_CHECK_ARG(B >= 0 && Tp > 0 && cap > 0 && C > 0, "lenreg_bwd: bad shape");
    if (B == 0) return TTSMI_OK;
    long rows = (long)B * Tp;
    hipLaunchKernelGGL(lenreg_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, dy, cum, dx, rows, Tp, cap, C);
    TTSMI_CHECK_LAUNCH("lenreg_bwd");
    return TTSMI_OK;
}

}  // extern "C"
