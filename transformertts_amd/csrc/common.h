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

// Routers that choose between kernel variants by launch size note the variant they took (a string literal, thread-local
// like the error message): tests assert through ttsmi_last_kernel() that a parity case really ran the variant the
// benchmark launches.
void ttsmi_note_kernel(const char* name);

// Cross-stream hand-off WITHOUT a marker packet.  hipEventRecord between two kernels of a stream costs the stream ~5 us
// (tools/probes/event_gap_probe.hip: 34.6 us per 29.6 us kernel with record + wait, 30.8 us when the event rides on the
// producing kernel's own completion signal); the dense block's backward hands four tensors per block to the
// weight-gradient stream.  The block launcher ARMS the event (thread-local), the launcher of the producing entry point's
// LAST kernel takes it and launches through hipExtLaunchKernelGGL(..., stopEvent); an event nobody took is recorded the
// ordinary way by the block launcher.
#include <hip/hip_ext.h>
void ttsmi_arm_stop_event(hipEvent_t e);
hipEvent_t ttsmi_take_stop_event();
#define TTSMI_LAUNCH_EV(kernel, grid, block, lds, st, ...)                                                     \
    do {                                                                                                       \
        hipEvent_t ev__ = ttsmi_take_stop_event();                                                             \
        if (ev__) hipExtLaunchKernelGGL(kernel, grid, block, lds, st, nullptr, ev__, 0, __VA_ARGS__);          \
        else hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                                    \
    } while (0)

static inline int ttsmi_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- tuning knobs ------------------------------------------------------------------------------
// Environment variables are read ONCE per process, through a function-local `static const` (C++11: initialised by exactly
// one thread), so concurrent first calls do not race.  TTSMI_KNOB: A/B knobs that choose between equivalent kernels.
// TTSMI_ABLATE_KNOB: stage-ablation / skip knobs whose results are WRONG by construction - they only exist in a library
// built with -DTTSMI_ABLATION_BUILD (TTSMI_EXTRA_HIPCC_FLAGS, build.py); the shipped library ignores the variable.
#include <stdlib.h>
static inline int ttsmi_env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
#define TTSMI_KNOB(var, name, dflt) static const int var = ttsmi_env_int(name, dflt)
// Device side: TTSMI_ABLATE_BITS(p.ablate) is the run-time bit field in a measurement build and the CONSTANT 0 in the
// shipped library - a run-time `if (p.ablate & 8) continue;` inside an unrolled row loop gave the compiler a path
// around the loop's first use of two loaded registers, so its wait-count pass put `s_waitcnt vmcnt(0)` in front of every
// row of the fused GEMM + LayerNorm epilogue: each row then waited for the previous row's stores (ISA reading, round 3).
#ifdef TTSMI_ABLATION_BUILD
#define TTSMI_ABLATE_KNOB(var, name) static const int var = ttsmi_env_int(name, 0)
#define TTSMI_ABLATE_BITS(x) (x)
#else
#define TTSMI_ABLATE_KNOB(var, name) static const int var = 0
#define TTSMI_ABLATE_BITS(x) 0
#endif

// weight gradient with the slab reduction deferred to a batched launch (gemm_bf16.hip; used by dense_block.hip -
// library-internal, not part of include/ttsmi.h)
#define TTSMI_WGRAD_MAX_JOBS 8
struct ttsmi_wgrad_job {
    const float* ws;
    float* dw;
    long lddw;
    int kin, n, splits, pad_;
    const float* cs_ws;
    float* db;
};
extern "C" {
size_t ttsmi_hgemm_wgrad_rows_exact_bytes(int rows, int kin, int n, int has_db);
int ttsmi_hgemm_wgrad_rows_deferred(const void* x, int x_is_bf16, int64_t ldx, const void* dy, int dy_is_bf16, int64_t lddy,
                                    float* dw, int64_t lddw, float* db, int rows, int kin, int n, void* ws, size_t ws_bytes,
                                    ttsmi_stream_t stream, ttsmi_wgrad_job* job);
int ttsmi_hgemm_wgrad_rows_deferred_dual(const void* x, int64_t ldx, const void* x2, int64_t ldx2, int k1, const void* dy,
                                         int64_t lddy, float* dw, int64_t lddw, float* db, int rows, int kin, int n, void* ws,
                                         size_t ws_bytes, ttsmi_stream_t stream, ttsmi_wgrad_job* job);
int ttsmi_hgemm_wgrad_reduce_jobs(const ttsmi_wgrad_job* jobs, int njobs, ttsmi_stream_t stream);
}

// bf16-operand attention kernels (attention_bf16.hip), reached through ttsmi_attention_fwd/bwd with
// dtype == TTSMI_BF16
int ttsmi_hattention_fwd(const void* qkv, const uint8_t* key_pad, const int32_t* klen, void* ctx,
                         float* lse, int B, int H, int T, int dh, float p_drop, uint64_t seed,
                         const int64_t* step_dev, uint32_t site, int qkv_is_bf16, const void* dropmask, hipStream_t st);
int ttsmi_hattention_bwd(const void* qkv, const uint8_t* key_pad, const int32_t* klen, const void* ctx,
                         const void* dctx, const float* lse, void* dqkv, int B, int H, int T, int dh,
                         float p_drop, uint64_t seed, const int64_t* step_dev, uint32_t site, void* ws,
                         int qkv_is_bf16, const void* dropmask, hipStream_t st);
int ttsmi_hattention_weights(const void* qkv, const uint8_t* key_pad, const float* lse, float* weights, int B, int H, int T,
                             int dh, float p_drop, uint64_t seed, const int64_t* step_dev, uint32_t site, const void* dropmask,
                             hipStream_t st);
size_t ttsmi_hattention_fwd_split_ws_bytes(int B, int H, int T, int dh);
int ttsmi_hattention_fwd_split(const void* qkv, const uint8_t* key_pad, const int32_t* klen, void* ctx, float* lse,
                               int B, int H, int T, int dh, void* ws, size_t ws_bytes, hipStream_t st);
size_t ttsmi_hattention_dropmask_bytes(int B, int H, int T);
int ttsmi_hattention_dropmask(void* mask, int B, int H, int T, float p_drop, uint64_t seed, const int64_t* step_dev,
                              uint32_t site, hipStream_t st);
int ttsmi_hattention_dropmask_stack(void* const* masks, const uint32_t* sites, int n, int B, int H, int T, float p_drop,
                                    uint64_t seed, const int64_t* step_dev, hipStream_t st);

// weight-stationary K = 256 GEMM (gemm_k256.hip), reached through ttsmi_hgemm_tn
extern "C" int ttsmi_hgemm_k256_eligible(int M, int N, int K);
int ttsmi_hgemm_k256_launch(const uint16_t* a, long lda, const uint16_t* bt, long ldb, const float* bias, void* c, long ldc,
                            int M, int N, int relu, int out_bf16, const uint16_t* mask, long ldmask, hipStream_t st);

// ---- wave64 reductions ------------------------------------------------------------------------
// wave_sum / wave_max: the xor butterfly.  hipcc lowers each __shfl_xor to ds_bpermute_b32 (an LDS-pipe round trip of
// ~100 cycles; six dependent ones per reduction) - irrelevant in the HBM-bound LayerNorm / loss kernels that use them.
// Every lane ends with its OWN association order of the 64 terms, i.e. the rounding error of the total differs from
// lane to lane.  That matters in one place: the LayerNorm backward forms dz = rstd (t - mean(t) - x^ mean(t x^)), and
// where the two means cancel t almost completely (the pitch predictor's last LayerNorm, whose upstream gradient is a
// rank-one dout * w) dz is the rounding error of the means.  Lane-dependent errors average out in the weight gradient
// that sums dz over rows and columns; a lane-UNIFORM total (wave_sum_dpp below) leaves a systematic per-row error:
// measured on the exact-fp32 path at the benchmark architecture, pitch.conv1.w moved from 1e-4 to 1.5e-3 of the fp64
// oracle and the whole encoder behind it to 1e-3.  So the fp32 kernels keep the butterfly.
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
// wave_sum_dpp: DPP data-parallel-primitive moves - quad / row mirrors sum each row of 16 lanes in four VALU-rate steps,
// two row broadcasts carry the row totals up, lane 63 holds the wave total and is read back as a scalar (every lane gets
// the same bits).  Used by the fused GEMM + LayerNorm epilogues of the bf16 path (rowgemm.hip: two reductions per row, 16
// rows per wave, one after the other = 192 dependent permutes with the butterfly).  Checked against the butterfly by
// tools/probes/dpp_reduce_probe.hip.
#define TTSMI_DPP_QUAD_1032 0xB1
#define TTSMI_DPP_QUAD_2301 0x4E
#define TTSMI_DPP_ROW_HALF_MIRROR 0x141
#define TTSMI_DPP_ROW_MIRROR 0x140
#define TTSMI_DPP_ROW_BCAST15 0x142
#define TTSMI_DPP_ROW_BCAST31 0x143
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float ttsmi_dpp(float v, float fill) {      // lanes outside ROW_MASK receive `fill`
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v),
                                                                 CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += ttsmi_dpp<TTSMI_DPP_QUAD_1032, 0xF>(v, 0.f);
    v += ttsmi_dpp<TTSMI_DPP_QUAD_2301, 0xF>(v, 0.f);
    v += ttsmi_dpp<TTSMI_DPP_ROW_HALF_MIRROR, 0xF>(v, 0.f);
    v += ttsmi_dpp<TTSMI_DPP_ROW_MIRROR, 0xF>(v, 0.f);                  // every lane: the sum of its row of 16
    v += ttsmi_dpp<TTSMI_DPP_ROW_BCAST15, 0xA>(v, 0.f);                 // rows 1, 3 += rows 0, 2
    v += ttsmi_dpp<TTSMI_DPP_ROW_BCAST31, 0xC>(v, 0.f);                 // rows 2, 3 += rows 0 + 1
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// ---- counter-based dropout RNG ----------------------------------------------------------------
// Stateless: keep(seed, step, site, row, col) so that forward and backward regenerate the same mask
// without storing it, in every kernel and in both precisions.  Cost model (it sits inside the
// attention inner loop next to the MFMAs):
//   per launch : one 64-bit splitmix of (seed + step counter, site)          -> ttsmi_drop_key
//   per row    : one 32-bit avalanche of the row index                        -> ttsmi_row_base
//   per PAIR of adjacent columns (2j, 2j+1): one 32-bit avalanche (~8 VALU)   -> ttsmi_pair_hash
//   per element: a 16-bit half of that hash against thr16 = round(p * 65536)  -> ttsmi_keep_of
// Statistical quality (rate, row/column independence, site independence) is checked in tests.
__device__ __forceinline__ uint64_t ttsmi_mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint32_t ttsmi_mix32(uint32_t h) {
    h ^= h >> 16; h *= 0x7feb352du;
    h ^= h >> 15; h *= 0x846ca68bu;
    h ^= h >> 16;
    return h;
}
// The host seed is advanced by a DEVICE step counter so that a captured hipGraph draws fresh masks
// on every replay (kernel arguments are frozen at capture).
// (two steps for kernels with more than one site: the step counter is read once, early - a load in the middle of a kernel
// that feeds LDS by inline-asm DMA makes the compiler wait for EVERYTHING in flight, it does not count those instructions)
__device__ __forceinline__ uint64_t ttsmi_drop_base(uint64_t seed, const int64_t* step_dev) {
    return step_dev ? seed + 0xA0761D6478BD642Full * (uint64_t)(*step_dev) : seed;
}
__device__ __forceinline__ uint64_t ttsmi_drop_key_of(uint64_t base, uint32_t site) {
    return ttsmi_mix64(base + 0x9E3779B97F4A7C15ull * (uint64_t)(site + 1));
}
__device__ __forceinline__ uint64_t ttsmi_drop_key(uint64_t seed, const int64_t* step_dev, uint32_t site) {
    return ttsmi_drop_key_of(ttsmi_drop_base(seed, step_dev), site);
}
__device__ __forceinline__ uint32_t ttsmi_row_base(uint64_t key, uint32_t row) {
    return ttsmi_mix32(row ^ (uint32_t)key) + (uint32_t)(key >> 32);
}
__device__ __forceinline__ uint32_t ttsmi_pair_hash(uint32_t row_base, uint32_t col) {
    return ttsmi_mix32(row_base + (col >> 1) * 0x85EBCA6Bu);
}
__device__ __forceinline__ float ttsmi_keep_of(uint32_t h, uint32_t col, uint32_t thr16, float inv_keep) {
    uint32_t u = (col & 1u) ? (h >> 16) : (h & 0xFFFFu);
    return u >= thr16 ? inv_keep : 0.0f;
}
// convenience: one element
__device__ __forceinline__ float ttsmi_keep_scale(uint64_t key, uint32_t row, uint32_t col, uint32_t thr16,
                                                  float inv_keep) {
    return ttsmi_keep_of(ttsmi_pair_hash(ttsmi_row_base(key, row), col), col, thr16, inv_keep);
}
static inline uint32_t ttsmi_drop_threshold(float p) {      // thr16
    double t = (double)p * 65536.0 + 0.5;
    if (t < 0) t = 0;
    if (t > 65535.0) t = 65535.0;
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
