// 512-point complex FFT of one wavefront (three radix-8 Stockham passes; 8 points per lane in registers, two exchanges
// through a wave-private LDS buffer) - shared by the mel extraction (stft_mel.hip) and Griffin-Lim (griffinlim.hip).
#pragma once
#include "common.h"

#define SUBN 512        // complex points of one radix-8^3 sub-transform
// The FFT exchange buffers are PRIVATE to a wave: what the passes need between a wave's LDS writes and
// its own later reads is ordering, not a workgroup barrier (LDS executes one wave's accesses in
// order).  A wavefront-scope fence pins the compiler's ordering and costs no instruction, so the
// four waves of a workgroup drift freely and hide each other's memory latency.
#define WAVE_SYNC()                                              \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
    } while (0)
#define ZP(i) ((i) + ((i) >> 3))   // one pad slot per 8 complex points: the radix-8 scatter of pass 1
                                  // (lane stride 8 points) then lands on 16 distinct banks
#define ZBUF (SUBN + SUBN / 8 + 8) // padded float2 slots of one sub-transform buffer


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

// One 512-point complex FFT of the values u[r] = z[lane + 64 r] (three radix-8 Stockham passes, two
// exchanges through the wave-private buffer zb); the result is left in natural order in zb[ZP(k)].
// tw is the table exp(-2 pi i k / N); TS = N / 512 scales a 512-point twiddle index into it.
template <int N>
__device__ __forceinline__ void fft512(float2 (&u)[8], float2* zb, const float2* tw, int lane, bool full) {
    constexpr int TS = N / SUBN;
    const int rev[8] = {0, 4, 2, 6, 1, 5, 3, 7};
    // ---- pass 1 (p = 1): lane i holds z[i + 64 r], r = 0..7 ---------------------------------
    fft8(u);
    {
        int j = lane << 3;                            // k = 0
#pragma unroll
        for (int r = 0; r < 8; ++r) zb[ZP(j + r)] = u[rev[r]];
    }
    WAVE_SYNC();
    // ---- pass 2 (p = 8) --------------------------------------------------------------------
    if (full) {
        int k = lane & 7, j = ((lane - k) << 3) + k;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float2 x = zb[ZP(lane + 64 * r)];
            u[r] = r ? cmul(x, tw[(TS * 8 * k * r) & (N - 1)]) : x;
        }
        WAVE_SYNC();
        fft8(u);
#pragma unroll
        for (int r = 0; r < 8; ++r) zb[ZP(j + r * 8)] = u[rev[r]];
    }
    WAVE_SYNC();
    // ---- pass 3 (p = 64) -------------------------------------------------------------------
    if (full) {
        int k = lane;                                  // j = k
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float2 x = zb[ZP(lane + 64 * r)];
            u[r] = r ? cmul(x, tw[(TS * k * r) & (N - 1)]) : x;
        }
        WAVE_SYNC();
        fft8(u);
#pragma unroll
        for (int r = 0; r < 8; ++r) zb[ZP(k + r * 64)] = u[rev[r]];
    }
    WAVE_SYNC();
}

