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


// Complex arithmetic on the packed fp32 pipe (a float2 = one 64-bit register pair, v_pk_*_f32).  The operand selects
// (op_sel / op_sel_hi: which half of a source feeds the low / high lane) and the PER-LANE negations (neg_lo / neg_hi) make
// a multiplication by -i, and the cross terms of a complex product, part of the add / fma that consumes them.  hipcc uses
// the selects but folds a negation only when it covers BOTH lanes: written in C++, a complex product came out as three
// packed instructions and a v_mov that merges the two half-right results, a radix-8 butterfly with ~20 v_mov (ISA reading,
// round 4: 72 of a frame's 213 vector instructions in the three passes were moves).
typedef float fft_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ fft_f2 fft_v(float2 a) { fft_f2 v = {a.x, a.y}; return v; }
__device__ __forceinline__ float2 fft_s(fft_f2 v) { return make_float2(v.x, v.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    fft_f2 t, r;                       // t = (a.y b.y, a.y b.x);  r = (a.x b.x - t.x, a.x b.y + t.y)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(fft_v(a)), "v"(fft_v(b)));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(fft_v(a)), "v"(fft_v(b)), "v"(t));
    return fft_s(r);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// multiply by -i
__device__ __forceinline__ float2 cmul_mi(float2 a) { return make_float2(a.y, -a.x); }
// a + (-i) b = (a.x + b.y, a.y - b.x)   and   a - (-i) b = (a.x - b.y, a.y + b.x): one packed add each
__device__ __forceinline__ float2 cadd_mi(float2 a, float2 b) {
    fft_f2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(fft_v(a)), "v"(fft_v(b)));
    return fft_s(r);
}
__device__ __forceinline__ float2 csub_mi(float2 a, float2 b) {
    fft_f2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(fft_v(a)), "v"(fft_v(b)));
    return fft_s(r);
}

// in-place radix-8 DIF butterfly; X[r] ends up in u[rev3(r)]
__device__ __forceinline__ void fft8(float2 (&u)[8]) {
    const float h = 0.70710678118654752440f;
    float2 a[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = cadd(u[i], u[i + 4]);
        a[i + 4] = csub(u[i], u[i + 4]);
    }
    // a[4+i] *= w8^i,  w8 = exp(-i pi/4):  a5 w8 = h (a5 + (-i) a5),  a6 w8^2 = (-i) a6 (folded into its two uses below),
    // a7 w8^3 = -h (a7 - (-i) a7)
    float2 t5 = cadd_mi(a[5], a[5]), t7 = csub_mi(a[7], a[7]);
    a[5] = make_float2(h * t5.x, h * t5.y);
    a[7] = make_float2(-h * t7.x, -h * t7.y);
    float2 b[8];
    b[0] = cadd(a[0], a[2]);
    b[2] = csub(a[0], a[2]);
    b[1] = cadd(a[1], a[3]);
    b[3] = csub(a[1], a[3]);                  // times -i: in its uses
    b[4] = cadd_mi(a[4], a[6]);
    b[6] = csub_mi(a[4], a[6]);
    b[5] = cadd(a[5], a[7]);
    b[7] = csub(a[5], a[7]);                  // times -i: in its uses
    u[0] = cadd(b[0], b[1]);
    u[1] = csub(b[0], b[1]);
    u[2] = cadd_mi(b[2], b[3]);
    u[3] = csub_mi(b[2], b[3]);
    u[4] = cadd(b[4], b[5]);
    u[5] = csub(b[4], b[5]);
    u[6] = cadd_mi(b[6], b[7]);
    u[7] = csub_mi(b[6], b[7]);
}

// One 512-point complex FFT of the values u[r] = z[lane + 64 r] (three radix-8 Stockham passes, two
// exchanges through the wave-private buffer zb); the result is left in natural order in zb[ZP(k)].
// tw is the table exp(-2 pi i k / N); TS = N / 512 scales a 512-point twiddle index into it.
// TO_LDS = false: the last pass's results stay in registers instead - u[rev3(r)] = Z[lane + 64 r] (no third exchange).
template <int N, bool TO_LDS = true>
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
        if constexpr (TO_LDS) {
#pragma unroll
            for (int r = 0; r < 8; ++r) zb[ZP(k + r * 64)] = u[rev[r]];
        }
    }
    WAVE_SYNC();
}

