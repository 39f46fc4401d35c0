// Griffin-Lim phase reconstruction as a GPU iSTFT / STFT loop (data/audio.py:94-110 -> librosa 0.7.1
// core.griffinlim: "fast" Griffin-Lim with momentum 0.99, n_fft = 1024 [3P]).
//
//   angles <- given (random phases drawn by the caller: the reference's are unseeded)
//   repeat n_iter times:
//       inverse = istft(S * angles)                  K1 frames + K2 overlap-add
//       rebuilt = stft(inverse)                      K3 (center = True: reflect padding of n_fft / 2)
//       angles  = rebuilt - momentum / (1 + momentum) * rebuilt_prev;  angles /= |angles| + 1e-16
//   return istft(S * angles)
//
// One wave64 owns one frame in K1 and K3; the 1024-point real transforms are 512-point complex transforms (fft512.h)
// plus the usual even / odd (un)tangling.  The overlap-add is a GATHER: output sample n sums the (<= n_fft / hop) frames
// that cover it in ascending frame order - the order librosa's accumulation loop uses - and divides by the window
// sum-square envelope, so the result is deterministic and needs no atomics.  Everything stays in HBM between the three
// launches of an iteration (a 10 s clip: 2.3 MB of frames + 0.9 MB of spectra, L2 resident).
#include <stdlib.h>

#include "common.h"
#include "fft512.h"

#define GL_NFFT 1024
#define GL_BINS 513
#define GL_WAVES 4

struct GlP {
    const float* mag;          // [T][513]
    float2* ang;               // [T][513] unit phases (in / out)
    float2* prev;              // [T][513] previous rebuilt spectrum
    const float* window;       // [1024] synthesis = analysis window (periodic Hann centred in n_fft)
    const float* wss;          // [n_fft + hop (T - 1)] window sum-square envelope
    float* frames;             // [T][1024] windowed inverse transforms
    float* y;                  // [hop (T - 1)] trimmed signal
    int T, hop;
    float mom;                 // momentum / (1 + momentum)
    float tiny;                // smallest normal float32: envelope values above it divide (librosa util.tiny)
};

__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }

__device__ __forceinline__ void gl_twiddles(float2* tw, int tid) {
    for (int k = tid; k < GL_NFFT; k += 64 * GL_WAVES) {
        float s, c;
        sincospif(-2.0f * (float)k / (float)GL_NFFT, &s, &c);
        tw[k] = make_float2(c, s);
    }
    __syncthreads();
}

// K1: frames[f][n] = window[n] * irfft(mag[f] * ang[f])[n]
__global__ __launch_bounds__(64 * GL_WAVES) void gl_istft_frames_kernel(GlP p) {
    __shared__ float2 tw[GL_NFFT];
    __shared__ float2 buf[GL_WAVES][ZBUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    gl_twiddles(tw, tid);
    const int f = blockIdx.x * GL_WAVES + wave;
    if (f >= p.T) return;
    const float* S = p.mag + (long)f * GL_BINS;
    const float2* A = p.ang + (long)f * GL_BINS;
    auto X = [&](int k) {                                   // bins 0 and 512 are real for a real signal
        float2 a = A[k];
        const float s = S[k];
        a = make_float2(a.x * s, a.y * s);
        if (k == 0 || k == 512) a.y = 0.f;
        return a;
    };
    // Z[k] = E[k] + i O[k],  E = (X[k] + conj X[512-k]) / 2,  O = (X[k] - conj X[512-k]) / 2 * w^-k;  z = ifft512(Z)
    // holds x[2n] + i x[2n+1].  ifft via the forward transform: ifft(Z) = conj(fft(conj Z)) / 512.
    float2 u[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int k = lane + 64 * r;
        const float2 a = X(k), b = cconj(X(512 - k));
        const float2 E = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y + b.y));
        const float2 D = make_float2(0.5f * (a.x - b.x), 0.5f * (a.y - b.y));
        const float2 O = cmul(D, cconj(tw[k]));            // w^-k
        u[r] = cconj(make_float2(E.x - O.y, E.y + O.x));   // conj(E + i O)
    }
    float2* zb = buf[wave];
    fft512<GL_NFFT>(u, zb, tw, lane, true);
    float* out = p.frames + (long)f * GL_NFFT;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int n = lane + 64 * r;
        const float2 z = zb[ZP(n)];
        const float2 w = *reinterpret_cast<const float2*>(p.window + 2 * n);
        *reinterpret_cast<float2*>(out + 2 * n) = make_float2(w.x * (z.x * (1.0f / 512.0f)), w.y * (-z.y * (1.0f / 512.0f)));
    }
}

// K2: y[j] = sum_i frames[i][j + 512 - i hop] / wss[j + 512]   (frames i with 0 <= j + 512 - i hop < 1024, ascending i)
__global__ __launch_bounds__(256) void gl_overlap_add_kernel(GlP p) {
    const long L = (long)p.hop * (p.T - 1);
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= L) return;
    const long n = j + GL_NFFT / 2;                         // position in the untrimmed signal
    long i0 = (n - (GL_NFFT - 1) + p.hop - 1) / p.hop;      // first frame with n - i hop <= 1023
    if (n < GL_NFFT) i0 = 0;
    long i1 = n / p.hop;
    if (i1 > p.T - 1) i1 = p.T - 1;
    float acc = 0.f;
    for (long i = i0; i <= i1; ++i) acc += p.frames[i * GL_NFFT + (n - i * p.hop)];
    const float e = p.wss[n];
    p.y[j] = e > p.tiny ? acc / e : acc;
}

// K3: rebuilt = rfft(window * reflect_pad(y)[f hop : f hop + 1024]);  angles update
__global__ __launch_bounds__(64 * GL_WAVES) void gl_stft_update_kernel(GlP p) {
    __shared__ float2 tw[GL_NFFT];
    __shared__ float2 buf[GL_WAVES][ZBUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    gl_twiddles(tw, tid);
    const int f = blockIdx.x * GL_WAVES + wave;
    if (f >= p.T) return;
    const long L = (long)p.hop * (p.T - 1);
    auto sample = [&](long q) {                             // q = index into the trimmed signal before padding
        if (q < 0) q = -q;
        if (q >= L) q = 2 * (L - 1) - q;
        return p.y[q];
    };
    float2 u[8];
    const long q0 = (long)f * p.hop - GL_NFFT / 2;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int n = lane + 64 * r;
        const float2 w = *reinterpret_cast<const float2*>(p.window + 2 * n);
        u[r] = make_float2(w.x * sample(q0 + 2 * n), w.y * sample(q0 + 2 * n + 1));
    }
    float2* zb = buf[wave];
    fft512<GL_NFFT>(u, zb, tw, lane, true);
    float2* A = p.ang + (long)f * GL_BINS;
    float2* P = p.prev + (long)f * GL_BINS;
    auto bin = [&](int k) {
        // X[k] = E + w^k O,  E = (Z[k] + conj Z[512-k]) / 2,  O = -i (Z[k] - conj Z[512-k]) / 2
        const float2 a = zb[ZP(k & 511)], b = cconj(zb[ZP((512 - k) & 511)]);
        const float2 E = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y + b.y));
        const float2 D = make_float2(0.5f * (a.x - b.x), 0.5f * (a.y - b.y));
        const float2 O = cmul_mi(D);
        const float2 wk = k < 512 ? tw[k] : make_float2(-1.f, 0.f);
        float2 X = cadd(E, cmul(wk, O));
        if (k == 0 || k == 512) X.y = 0.f;
        const float2 old = P[k];
        P[k] = X;
        float2 a2 = make_float2(X.x - p.mom * old.x, X.y - p.mom * old.y);
        const float inv = 1.0f / (sqrtf(a2.x * a2.x + a2.y * a2.y) + 1e-16f);
        A[k] = make_float2(a2.x * inv, a2.y * inv);
    };
#pragma unroll
    for (int r = 0; r < 8; ++r) bin(lane + 64 * r);
    if (lane == 0) bin(512);
}

extern "C" {

size_t ttsmi_griffinlim_ws_bytes(int T) {
    if (T < 3) return 0;
    return (size_t)T * GL_BINS * sizeof(float2) + (size_t)T * GL_NFFT * sizeof(float) + 256;
}

int ttsmi_griffinlim(const float* mag, float* angles, const float* window, const float* wss, int T, int n_fft, int hop,
                     int n_iter, float momentum, float* wav, void* ws, size_t ws_bytes, ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(mag && angles && window && wss && wav && ws, "griffinlim: null pointer");
    TTSMI_CHECK_ARG(n_fft == GL_NFFT, "griffinlim: n_fft %d not built (1024)", n_fft);
    TTSMI_CHECK_ARG(hop > 0 && hop <= n_fft && n_iter >= 0 && momentum >= 0.f, "griffinlim: bad hop / n_iter / momentum");
    // the reflect padding of the analysis step needs more than n_fft / 2 samples: hop (T - 1) > 512
    TTSMI_CHECK_ARG(T >= 3 && (long)hop * (T - 1) > n_fft / 2, "griffinlim: %d frames are too few for centred frames", T);
    TTSMI_CHECK_ARG(ws_bytes >= ttsmi_griffinlim_ws_bytes(T) && (((uintptr_t)ws) & 15) == 0 && (((uintptr_t)angles) & 7) == 0 &&
                    (((uintptr_t)window) & 7) == 0, "griffinlim: workspace too small / unaligned");
    hipStream_t st = (hipStream_t)stream;
    GlP p;
    p.mag = mag; p.ang = (float2*)angles; p.window = window; p.wss = wss; p.y = wav; p.T = T; p.hop = hop;
    p.prev = (float2*)ws;
    p.frames = (float*)((char*)ws + (((size_t)T * GL_BINS * sizeof(float2) + 255) & ~(size_t)255));
    p.mom = momentum / (1.0f + momentum);
    p.tiny = 1.17549435e-38f;
    if (hipMemsetAsync(p.prev, 0, (size_t)T * GL_BINS * sizeof(float2), st) != hipSuccess) {   // rebuilt = 0 before the first pass
        ttsmi_set_error("griffinlim: hipMemsetAsync failed");
        return TTSMI_ERR_LAUNCH;
    }
    const dim3 gf(ttsmi_cdiv(T, GL_WAVES)), bf(64 * GL_WAVES);
    const long L = (long)hop * (T - 1);
    const dim3 go(ttsmi_cdiv(L, 256));
    for (int it = 0; it <= n_iter; ++it) {
        hipLaunchKernelGGL(gl_istft_frames_kernel, gf, bf, 0, st, p);
        hipLaunchKernelGGL(gl_overlap_add_kernel, go, dim3(256), 0, st, p);
        if (it < n_iter) hipLaunchKernelGGL(gl_stft_update_kernel, gf, bf, 0, st, p);
    }
    TTSMI_CHECK_LAUNCH("griffinlim");
    return TTSMI_OK;
}

}  // extern "C"
