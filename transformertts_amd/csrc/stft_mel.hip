// wav -> reflect-padded frames -> periodic-Hann window -> 1024-pt real FFT -> |.| -> sparse Slaney
// mel filterbank -> clip + log, as ONE batched kernel (data/audio.py:72-92,209-242).
//
// One wave64 owns one frame.  The 1024-pt real FFT is a 512-pt complex FFT of z[n] = x[2n] + i x[2n+1]
// done as three radix-8 Stockham passes (512 = 8^3): each lane keeps 8 complex points in registers
// per pass, the two inter-pass exchanges go through a per-wave 4 KB LDS buffer, twiddles come from
// an LDS table of exp(-2 pi i k / 1024) built once per workgroup with sincospi.  The magnitude
// spectrum (513 bins) stays in LDS; the mel filterbank is applied in its sparse form (each filter
// is one contiguous run of bins, 727 non-zeros for the LJSpeech setting) and the [frames, n_mels]
// output row is written with one coalesced store.  HBM traffic: every sample is fetched ~once
// (the 4x frame overlap is served by L2 because consecutive frames run in the same workgroup),
// 4*n_mels bytes written per frame.
#include "common.h"

#define NFFT 1024
#define NC 512          // complex points
#define FR_PER_WG 4     // waves per workgroup = frames in flight

struct MelP {
    const float* wav; const int64_t* clip_off; const int64_t* frame_off;
    int n_clips; long total_frames; int hop;
    const float* window;
    int n_mels; const int32_t* mel_lo; const int32_t* mel_cnt; const int32_t* mel_ptr;
    const float* mel_w;
    int normalizer; float clip_min;
    float* out;
    int groups_per_wg;
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// multiply by -i
__device__ __forceinline__ float2 cmul_mi(float2 a) { return make_float2(a.y, -a.x); }

// in-place radix-8 DIF butterfly; X[r] ends up in u[rev3(r)]
__device__ __forceinline__ void fft8(float2 (&u)[8]) {
    const float h = 0.70710678118654752440f;
    float2 a[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = cadd(u[i], u[i + 4]);
        a[i + 4] = csub(u[i], u[i + 4]);
    }
    // a[4+i] *= w8^i,  w8 = exp(-i pi/4)
    a[5] = make_float2(h * (a[5].x + a[5].y), h * (a[5].y - a[5].x));
    a[6] = cmul_mi(a[6]);
    a[7] = make_float2(h * (a[7].y - a[7].x), -h * (a[7].x + a[7].y));
    float2 b[8];
#pragma unroll
    for (int base = 0; base < 8; base += 4) {
        b[base + 0] = cadd(a[base + 0], a[base + 2]);
        b[base + 2] = csub(a[base + 0], a[base + 2]);
        b[base + 1] = cadd(a[base + 1], a[base + 3]);
        b[base + 3] = cmul_mi(csub(a[base + 1], a[base + 3]));
    }
#pragma unroll
    for (int base = 0; base < 8; base += 2) {
        u[base] = cadd(b[base], b[base + 1]);
        u[base + 1] = csub(b[base], b[base + 1]);
    }
}

__device__ __forceinline__ int clip_of_frame(const int64_t* frame_off, int n_clips, long f) {
    int lo = 0, hi = n_clips;        // find c with frame_off[c] <= f < frame_off[c+1]
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (frame_off[mid] <= f) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void stft_logmel_kernel(MelP p) {
    __shared__ float2 tw[NFFT];                       // exp(-2 pi i k / 1024)
    __shared__ float2 buf[FR_PER_WG][NC + 8];
    __shared__ float mag[FR_PER_WG][NC + 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rev[8] = {0, 4, 2, 6, 1, 5, 3, 7};

    for (int k = tid; k < NFFT; k += 256) {
        float s, c;
        sincospif(-2.0f * (float)k / (float)NFFT, &s, &c);
        tw[k] = make_float2(c, s);
    }
    __syncthreads();

    float2* zb = buf[wave];
    float* mg = mag[wave];
    for (int g = 0; g < p.groups_per_wg; ++g) {
        const long f = ((long)blockIdx.x * p.groups_per_wg + g) * FR_PER_WG + wave;
        const bool active = f < p.total_frames;
        float2 u[8];
        long base = 0;
        int L = 1, t = 0;
        if (active) {
            int c = clip_of_frame(p.frame_off, p.n_clips, f);
            base = p.clip_off[c];
            L = (int)(p.clip_off[c + 1] - base);
            t = (int)(f - p.frame_off[c]);
        }
        // ---- pass 1 (p = 1): lane i loads z[i + 64 r], r = 0..7, straight from global ----------
        const int start = t * p.hop - NFFT / 2;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            int n = 2 * (lane + 64 * r);
            float v[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                int gi = start + n + e;
                if (gi < 0) gi = -gi;                     // np.pad(mode='reflect')
                if (gi >= L) gi = 2 * (L - 1) - gi;
                gi = gi < 0 ? 0 : gi;
                v[e] = active ? p.wav[base + gi] * p.window[n + e] : 0.f;
            }
            u[r] = make_float2(v[0], v[1]);
        }
        fft8(u);
        {
            int j = lane << 3;                            // k = 0
#pragma unroll
            for (int r = 0; r < 8; ++r) zb[j + r] = u[rev[r]];
        }
        __syncthreads();
        // ---- pass 2 (p = 8) --------------------------------------------------------------------
        {
            int k = lane & 7, j = ((lane - k) << 3) + k;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                float2 x = zb[lane + 64 * r];
                u[r] = r ? cmul(x, tw[(16 * k * r) & (NFFT - 1)]) : x;
            }
            __syncthreads();
            fft8(u);
#pragma unroll
            for (int r = 0; r < 8; ++r) zb[j + r * 8] = u[rev[r]];
        }
        __syncthreads();
        // ---- pass 3 (p = 64) -------------------------------------------------------------------
        {
            int k = lane;                                  // j = k
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                float2 x = zb[lane + 64 * r];
                u[r] = r ? cmul(x, tw[(2 * k * r) & (NFFT - 1)]) : x;
            }
            __syncthreads();
            fft8(u);
#pragma unroll
            for (int r = 0; r < 8; ++r) zb[k + r * 64] = u[rev[r]];
        }
        __syncthreads();
        // ---- real-FFT post-processing: |X[k]|, k = 0..512 -------------------------------------
        for (int k = lane; k <= NC; k += 64) {
            float2 zk = zb[k & (NC - 1)];
            float2 zc = zb[(NC - k) & (NC - 1)];
            zc.y = -zc.y;
            float2 e = cadd(zk, zc), o = csub(zk, zc);
            float2 w = (k < NC) ? tw[k] : make_float2(-1.f, 0.f);
            // X = 0.5*e - 0.5i * w * o
            float2 wo = cmul(w, o);
            float xr = 0.5f * (e.x + wo.y), xi = 0.5f * (e.y - wo.x);
            mg[k] = sqrtf(xr * xr + xi * xi);
        }
        __syncthreads();
        // ---- sparse mel + normalisation ---------------------------------------------------------
        for (int m = lane; m < p.n_mels; m += 64) {
            int lo = p.mel_lo[m], cnt = p.mel_cnt[m];
            const float* w = p.mel_w + p.mel_ptr[m];
            float s = 0.f;
            for (int i = 0; i < cnt; ++i) s += w[i] * mg[lo + i];
            float o;
            if (p.normalizer == 0) {
                o = logf(fmaxf(s, p.clip_min));
            } else {
                float db = 20.f * log10f(fmaxf(1e-5f, s));
                float nz = fminf(fmaxf((db + 100.f) / 100.f, 0.f), 1.f);
                o = nz * 8.f - 4.f;
            }
            if (active) p.out[f * p.n_mels + m] = o;
        }
        __syncthreads();
    }
}

extern "C" {

int ttsmi_stft_logmel(const float* wav, const int64_t* clip_off, const int64_t* frame_off,
                      int n_clips, int64_t total_frames, int n_fft, int hop, const float* window,
                      int n_mels, const int32_t* mel_lo, const int32_t* mel_cnt,
                      const int32_t* mel_ptr, const float* mel_w, int normalizer, float clip_min,
                      float* out, ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(wav && clip_off && frame_off && window && mel_lo && mel_cnt && mel_ptr &&
                        mel_w && out, "stft_logmel: null pointer");
    TTSMI_CHECK_ARG(n_clips > 0 && total_frames > 0 && hop > 0 && n_mels > 0,
                    "stft_logmel: bad shape");
    if (n_fft != NFFT) {
        ttsmi_set_error("stft_logmel: n_fft=%d not built (only %d)", n_fft, NFFT);
        return TTSMI_ERR_UNSUPPORTED;
    }
    TTSMI_CHECK_ARG(normalizer == 0 || normalizer == 1, "stft_logmel: unknown normalizer %d", normalizer);
    MelP p;
    p.wav = wav; p.clip_off = clip_off; p.frame_off = frame_off; p.n_clips = n_clips;
    p.total_frames = total_frames; p.hop = hop; p.window = window; p.n_mels = n_mels;
    p.mel_lo = mel_lo; p.mel_cnt = mel_cnt; p.mel_ptr = mel_ptr; p.mel_w = mel_w;
    p.normalizer = normalizer; p.clip_min = clip_min; p.out = out;
    long groups = (total_frames + FR_PER_WG - 1) / FR_PER_WG;
    int gpw = 1;
    while (gpw < 16 && groups / (gpw * 2) >= 2048) gpw *= 2;   // amortise the twiddle build
    p.groups_per_wg = gpw;
    long blocks = (groups + gpw - 1) / gpw;
    hipLaunchKernelGGL(stft_logmel_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       (hipStream_t)stream, p);
    TTSMI_CHECK_LAUNCH("stft_logmel");
    return TTSMI_OK;
}

}  // extern "C"
