// Shared device/host helpers for libttsmi (gfx950 / CDNA4 only - no other target is supported).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/ttsmi.h"

#define TTSMI_WAVE 64

// ---- error plumbing (thread-local message, no exceptions across the C ABI) -------------------
void ttsmi_set_error(const char* fmt, ...);

#define TTSMI_CHECK_ARG(cond, ...)                 \
    do {                                           \
        if (!(cond)) {                             \
            ttsmi_set_error(__VA_ARGS__);          \
            return TTSMI_ERR_INVALID_ARG;          \
        }                                          \
    } while (0)

#define TTSMI_CHECK_LAUNCH(name)                                                        \
    do {                                                                                \
        hipError_t e__ = hipGetLastError();                                             \
        if (e__ != hipSuccess) {                                                        \
            ttsmi_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
            return TTSMI_ERR_LAUNCH;                                                    \
        }                                                                               \
    } while (0)

static inline int ttsmi_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// bf16-operand attention kernels (attention_bf16.hip), reached through ttsmi_attention_fwd/bwd with
// dtype == TTSMI_BF16
int ttsmi_hattention_fwd(const void* qkv, const uint8_t* key_pad, const int32_t* klen, void* ctx,
                         float* lse, int B, int H, int T, int dh, float p_drop, uint64_t seed,
                         const int64_t* step_dev, uint32_t site, hipStream_t st);
int ttsmi_hattention_bwd(const void* qkv, const uint8_t* key_pad, const int32_t* klen, const void* ctx,
                         const void* dctx, const float* lse, void* dqkv, int B, int H, int T, int dh,
                         float p_drop, uint64_t seed, const int64_t* step_dev, uint32_t site, void* ws,
                         hipStream_t st);

// ---- wave64 reductions ------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- counter-based dropout RNG ----------------------------------------------------------------
// Stateless: keep(seed, site, idx) so that forward and backward regenerate the same mask without
// storing it.  The (seed, site) pair is mixed once per launch with a 64-bit splitmix finaliser
// (loop invariant - the compiler hoists it); the per-element work is one 32-bit avalanche hash
// (two multiply-xorshift rounds, ~8 VALU ops), cheap enough to sit inside the attention inner
// loop next to the MFMAs.  Statistical quality is checked in tests.
__device__ __forceinline__ uint64_t ttsmi_mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// returns a uniform 32-bit word for element `idx` of dropout site `site` under `seed`
__device__ __forceinline__ uint32_t ttsmi_rand32(uint64_t seed, uint32_t site, uint64_t idx) {
    const uint64_t s = ttsmi_mix64(seed + 0x9E3779B97F4A7C15ull * (uint64_t)(site + 1));
    uint32_t h = (uint32_t)idx ^ (uint32_t)s;
    h += (uint32_t)(idx >> 32) * 0x9E3779B1u + (uint32_t)(s >> 32);
    h ^= h >> 16; h *= 0x7feb352du;
    h ^= h >> 15; h *= 0x846ca68bu;
    h ^= h >> 16;
    return h;
}
// keep-scale: 1/(1-p) if kept else 0.  thr = p * 2^32 (precomputed on the host)
__device__ __forceinline__ float ttsmi_keep_scale(uint64_t seed, uint32_t site, uint64_t idx,
                                                  uint32_t thr, float inv_keep) {
    return ttsmi_rand32(seed, site, idx) >= thr ? inv_keep : 0.0f;
}
// Effective seed of a launch: the host seed advanced by a DEVICE step counter, so that a captured
// hipGraph draws fresh masks on every replay (kernel arguments are frozen at capture).
__device__ __forceinline__ uint64_t ttsmi_step_seed(uint64_t seed, const int64_t* step_dev) {
    return step_dev ? seed + 0xA0761D6478BD642Full * (uint64_t)(*step_dev) : seed;
}
static inline uint32_t ttsmi_drop_threshold(float p) {
    double t = (double)p * 4294967296.0;
    if (t < 0) t = 0;
    if (t > 4294967295.0) t = 4294967295.0;
    return (uint32_t)t;
}

// ---- XCD-aware block id remap (bijective for any grid size) ------------------------------------
// Blocks b, b+8, b+16.. run on the same XCD (observed placement; speed only, never correctness):
// give each XCD a contiguous chunk of the logical tile space so neighbouring tiles share its L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int nx = 8;
    int xcd = bid % nx, slot = bid / nx;
    int q = nwg / nx, r = nwg % nx;
    int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}
