// mel -> linear magnitudes: the non-negative least squares problem of librosa.feature.inverse.mel_to_stft (0.7.1 [3P]),
// which data/audio.py:94-110 (reconstruct_waveform) solves on the CPU before Griffin-Lim:
//
//       for every frame t:   minimise 1/2 || B x_t - m_t ||^2   subject to x_t >= 0
//
// B [n_mels, n_bins] is the Slaney filterbank (sparse: triangles), m_t the de-normalised mel frame.  librosa starts from
// the clipped least-squares solution and runs scipy's L-BFGS-B over blocks of 127 frames at once (5 - 30 s for a 300 frame
// mel on the host, tools/debug/nnls_probe.py).  The frames are INDEPENDENT problems with 513 unknowns each, so here one
// wave64 owns one frame and runs an accelerated projected gradient (FISTA with the gradient restart of O'Donoghue &
// Candes) entirely on chip: the iterate lives in registers (lane l owns bins l, l + 64, ...), the extrapolated point and
// the residual in LDS, the filterbank's ~750 non-zeros in LDS.  Nothing is read from or written to HBM inside the loop.
//
//       x_0 = max(0, pinv(B) m)                               librosa's start point
//       r   = B y - m                                         lanes own filter rows (widest rows first)
//       x+  = max(0, y - (1/L) B^T r)                         lanes own bins; <= 2 filters cover a bin
//       restart if <y - x+, x+ - x> > 0 (momentum points uphill): y = x+, t = 1
//       else y = x+ + (t - 1) / t+ (x+ - x),  t+ = (1 + sqrt(1 + 4 t^2)) / 2
//
// The minimiser is not unique in x (80 equations, 513 unknowns; B x is): like L-BFGS-B, this iteration moves from the
// start point along combinations of rows of B and projections, and lands close to scipy's answer (3 - 9 % in norm) at a
// LOWER objective - scipy stops at a projected gradient of 1e-5.  Summation orders are fixed: the result is
// bit-reproducible.
#include "common.h"

struct NnlsP {
    const float* mel;      // [T][nm] amplitudes
    const float* pinvT;    // [nm][nb] pinv(B) transposed (the basis' own layout)
    const int* lo;         // [nm] first bin of filter row j
    const int* cnt;        // [nm] bins of row j
    const int* ptr;        // [nm] offset of row j in w
    const float* w;        // [nw] the rows' non-zero runs
    float* x;              // [T][nb]
    int T, nm, nb, nw, n_iter;
    float step;            // 1 / ||B||_2^2
    float inv_power;
};

__device__ __forceinline__ float nnls_wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);         // symmetric butterflies: every lane gets the same bits
    return v;
}

template <int NBL>
__global__ __launch_bounds__(64) void mel_nnls_kernel(NnlsP p) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x, nm = p.nm, nb = p.nb;
    float* ys = lds;                           // [NBL * 64] extrapolated point
    float* rs = ys + NBL * 64;                 // [nm + 2] residual (two zero pads: a bin's second filter may not exist)
    float* ms = rs + nm + 2;                   // [nm] the frame's mel
    float* wl = ms + nm;                       // [nw]
    int* rlo = (int*)(wl + p.nw);              // [nm] x 3
    int* rcnt = rlo + nm;
    int* rptr = rcnt + nm;
    const long f = blockIdx.x;
    for (int j = lane; j < nm; j += 64) {
        ms[j] = p.mel[f * nm + j];
        rlo[j] = p.lo[j]; rcnt[j] = p.cnt[j]; rptr[j] = p.ptr[j];
    }
    for (int q = lane; q < p.nw; q += 64) wl[q] = p.w[q];
    if (lane < 2) rs[nm + lane] = 0.f;
    __syncthreads();

    // the filters that cover this lane's bins: rows [j0, j0 + c) (consecutive for a filterbank; a row of the range that
    // does not cover the bin gets weight 0), the first two weights in registers
    int j0[NBL], jc[NBL];
    float w0[NBL], w1[NBL];
    bool wide = false;
    auto weight = [&](int j, int k) -> float {
        if (j >= nm) return 0.f;
        const unsigned d = (unsigned)(k - rlo[j]);
        return d < (unsigned)rcnt[j] ? wl[rptr[j] + (int)d] : 0.f;
    };
#pragma unroll
    for (int i = 0; i < NBL; ++i) {
        const int k = lane + 64 * i;
        int a = nm, b = -1;
        if (k < nb)
            for (int j = 0; j < nm; ++j)
                if ((unsigned)(k - rlo[j]) < (unsigned)rcnt[j]) { a = min(a, j); b = max(b, j); }
        j0[i] = b < 0 ? nm : a;                                             // nm: the zero pad
        jc[i] = b < 0 ? 0 : b - a + 1;
        w0[i] = b < 0 ? 0.f : weight(a, k);
        w1[i] = jc[i] > 1 ? weight(a + 1, k) : 0.f;
        wide |= jc[i] > 2;
    }
    const bool any_wide = __ballot(wide) != 0ull;

    // start point
    float x[NBL], y[NBL];
#pragma unroll
    for (int i = 0; i < NBL; ++i) {
        const int k = lane + 64 * i;
        float acc = 0.f;
        if (k < nb)
            for (int j = 0; j < nm; ++j) acc = fmaf(p.pinvT[(long)j * nb + k], ms[j], acc);
        x[i] = y[i] = fmaxf(acc, 0.f);
        ys[k] = y[i];
    }
    __syncthreads();

    float t = 1.f;
    for (int it = 0; it < p.n_iter; ++it) {
        for (int j = nm - 1 - lane; j >= 0; j -= 64) {                      // widest filters in the first pass
            const int a = rlo[j], c = rcnt[j];
            const float* wr = wl + rptr[j];
            float acc = 0.f;
            for (int q = 0; q < c; ++q) acc = fmaf(wr[q], ys[a + q], acc);
            rs[j] = acc - ms[j];
        }
        __syncthreads();
        float s = 0.f, dx[NBL];
#pragma unroll
        for (int i = 0; i < NBL; ++i) {
            float g = fmaf(w1[i], rs[j0[i] + 1], w0[i] * rs[j0[i]]);
            if (any_wide)
                for (int q = 2; q < jc[i]; ++q) g = fmaf(weight(j0[i] + q, lane + 64 * i), rs[j0[i] + q], g);
            const float xn = fmaxf(0.f, fmaf(-p.step, g, y[i]));
            dx[i] = xn - x[i];
            s = fmaf(y[i] - xn, dx[i], s);
            x[i] = xn;
        }
        s = nnls_wave_sum(s);
        const float tn = 0.5f * (1.f + sqrtf(fmaf(4.f * t, t, 1.f)));
        const bool restart = s > 0.f;
        const float beta = restart ? 0.f : (t - 1.f) / tn;
        t = restart ? 1.f : tn;
#pragma unroll
        for (int i = 0; i < NBL; ++i) {
            y[i] = fmaf(beta, dx[i], x[i]);
            ys[lane + 64 * i] = y[i];
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < NBL; ++i) {
        const int k = lane + 64 * i;
        if (k < nb) p.x[f * nb + k] = p.inv_power == 1.f ? x[i] : powf(x[i], p.inv_power);
    }
}

extern "C" {

int ttsmi_mel_nnls(const float* mel, const float* pinv_t, const int* row_lo, const int* row_cnt, const int* row_ptr,
                   const float* w, int n_w, float* x, int T, int n_mels, int n_bins, float inv_lipschitz, int n_iter,
                   float inv_power, ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(T >= 0, "mel_nnls: T = %d", T);
    if (T == 0) return TTSMI_OK;                                            // an empty mel: nothing to solve, no pointers read
    TTSMI_CHECK_ARG(mel && pinv_t && row_lo && row_cnt && row_ptr && w && x, "mel_nnls: null pointer");
    TTSMI_CHECK_ARG(n_mels > 0 && n_mels <= 256 && n_bins > 0 && n_bins <= 17 * 64 && n_w > 0 && n_w <= 8192,
                    "mel_nnls: %d mels / %d bins / %d weights outside what is built (<= 256 / <= 1088 / <= 8192)", n_mels,
                    n_bins, n_w);
    TTSMI_CHECK_ARG(n_iter >= 0 && inv_lipschitz > 0.f && inv_power > 0.f, "mel_nnls: bad n_iter / step / power");
    NnlsP p;
    p.mel = mel; p.pinvT = pinv_t; p.lo = row_lo; p.cnt = row_cnt; p.ptr = row_ptr; p.w = w; p.x = x;
    p.T = T; p.nm = n_mels; p.nb = n_bins; p.nw = n_w; p.n_iter = n_iter; p.step = inv_lipschitz; p.inv_power = inv_power;
    const int nbl = n_bins <= 9 * 64 ? 9 : 17;
    const size_t lds = ((size_t)nbl * 64 + (size_t)n_mels * 2 + 2 + n_w + (size_t)n_mels * 3) * 4;
    hipStream_t st = (hipStream_t)stream;
    if (nbl == 9) hipLaunchKernelGGL(mel_nnls_kernel<9>, dim3(T), dim3(64), lds, st, p);
    else hipLaunchKernelGGL(mel_nnls_kernel<17>, dim3(T), dim3(64), lds, st, p);
    TTSMI_CHECK_LAUNCH("mel_nnls");
    return TTSMI_OK;
}

}  // extern "C"
