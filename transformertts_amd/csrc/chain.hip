// The row-local chain of a dense block as ONE persistent-per-tile kernel (TTSMI_BF16 path, d_model = 256):
//
//   a   = LN1( keep([h | ctx] . Wo + bo) + h ) * rowmask                      (model/layers.py:148-150, 211, 229)
//   h1  = relu(a . W1 + b1)                                                   (model/layers.py:99)
//   out = LN2( keep(h1 . W2 + b2) + a ) * rowmask                             (model/layers.py:100-102, 230)
//   qkv'= out . Wqkv' + bqkv'      (the NEXT block's projection, model/layers.py:116-118 - when there is a next block)
//
// Until round 4 these were four launches (full-row GEMM + LN, K = 256 GEMM, full-row GEMM + LN, K = 256 GEMM), each
// re-reading from HBM what the previous one had just written: a_bf twice, h1 once, out_bf once (118 MB per decoder block
// at M = 28 800) and each paying its own launch ramp, epilogue and tail.  Here a workgroup takes 128 rows through the whole
// chain; nothing but the tensors the BACKWARD needs (a_bf, x^1, rstd1, h1, x^2, rstd2, out_bf, qkv') is written, nothing
// is read back.
//
// Layout: a wave owns 32 rows.  Every product is computed transposed (C^T = W . X^T: the MFMA A operand is a 32 x 16
// weight fragment, the B operand the activation fragment), so a lane holds ONE row (lane & 31) and 16 of every 32 output
// features in its accumulator registers - and an accumulator register block IS the B operand of the next product:
// registers 8p .. 8p + 7 of output tile j, rounded to bf16, are the fragment of k-group 2j + p (the register-feedback
// trick of the attention kernels, attention_bf16.hip:to_frags).  The k order inside such a group is the accumulator's
// (features 16q + {0..3, 8..11} + 4 * (lane >> 5)), so the weights are stored pre-permuted to match: ttsmi_dense_chain_pack
// writes the four matrices of a chain as ONE linear stream of 1 KB MFMA A fragments in exactly the order the kernel
// multiplies them (52 stages of 32 fragments at F = 1 024; 1.66 MB per block and step, repacked after every Adam step).
// A row's LayerNorm statistics are 128 local adds and one exchange with lane ^ 32; residuals never leave registers.
//
// The weight stream arrives by LDS-DMA (global_load_lds_dwordx4, lane-linear = fragment order: no swizzle, conflict-free
// ds_read_b128) into a FOUR-stage ring of 32 KB stages: one barrier per stage, three stages in flight.  Every wave issues
// its quarter of a stage (8 pieces of 1 KB), two pieces behind each group of eight MFMAs; inside a stage every multiply is
// followed by the read of the fragment eight multiplies on.  vmcnt counts loads AND stores on gfx950 and stores are not
// ordered against loads, so the counted wait in front of a stage allows exactly the DMA pieces that are YOUNGER than the
// stage's own (16, 8 or 0): loads complete in order, outstanding stores can only make that wait longer, never shorter.
// Bias / gamma / beta vectors are staged in LDS once per workgroup; accumulators start at their bias.
//
// The forward kernel is chain16.h's dense_chain16_kernel: eight 16-row waves on v_mfma_f32_16x16x32_bf16, <= 256 registers, TWO
// waves per SIMD (85.7 us at 28 800 rows against 104.7 for the four launches).  The first form built in round 5 - four
// 32-row waves on v_mfma_f32_32x32x16_bf16, 450-512 registers, ONE wave per SIMD: its MFMA issue, 12 k vector instructions
// and waits simply added, 98-101 us - was correct and tested but never paid; it was removed in round 6 (docs/history.md and
// profiles/r05_chain_*.txt keep its six builds' measurements).  What remains in this file is what both forms shared: the
// parameter block, the LDS / DMA helpers, the entry points.
// chain16b.h holds the BACKWARD chain on the same layout (ttsmi_dense_chain_bwd).
// The host uses the chains from 16 384 rows on (ops.CHAIN_MIN_ROWS): a workgroup's own latency is ~60 us whatever the row
// count, which the four launches beat at encoder sizes (57 us).
#include <stdlib.h>

#include <map>
#include <mutex>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define CH_D 256
#define CH_FRAG_BYTES 1024
#define CH_STAGE_FRAGS 32
#define CH_STAGE_BYTES (CH_STAGE_FRAGS * CH_FRAG_BYTES)
#define CH_NRING 4
#define CH_NW 4
#define CH_ROWS (CH_NW * 32)
#define CH_SLOT_LD 72                                   // bf16 elements per scratch row: 64 + 8 (144-byte rows)
#define CH_SLOT_BYTES (32 * CH_SLOT_LD * 2)             // 4 608: 32 rows x 64 bf16, or 32 rows x 32 fp32 (36-float rows)
#define CH_SCR_BYTES (CH_NW * CH_SLOT_BYTES)            // a private slot per wave
#define CH_NDMA (CH_STAGE_FRAGS / CH_NW)                // DMA pieces per wave and stage: 8
#define CH_MAXF 1024                                    // the staged parameter block holds b1 up to this many features
#define CH_PAR_FLOATS (6 * CH_D + CH_MAXF + 3 * CH_D)   // bo g1 be1 b2 g2 be2 | b1 | bqkv
#define CH_P_BO 0
#define CH_P_G1 (1 * CH_D)
#define CH_P_BE1 (2 * CH_D)
#define CH_P_B2 (3 * CH_D)
#define CH_P_G2 (4 * CH_D)
#define CH_P_BE2 (5 * CH_D)
#define CH_P_B1 (6 * CH_D)
#define CH_P_BQ (6 * CH_D + CH_MAXF)
#define CH_WO_STAGES 8                                  // K = 512 in steps of 64
#define CH_QKV_STAGES 12                                // 768 output features in chunks of 64

struct ChainP {
    const uint16_t* h_bf;            // [M,256] block input: q_in half of the o-projection AND residual of res-norm 1
    const uint16_t* cx;              // [M,256] attention context
    const unsigned char* wpack;      // ttsmi_dense_chain_pack
    int M, F, nchunk, nstages, nhalf;
    const float *bo, *ln1_g, *ln1_b, *b1, *b2, *ln2_g, *ln2_b, *bqkv;
    const uint8_t* row_pad;
    uint32_t thr; float inv_keep; uint64_t seed; const int64_t* step_dev; uint32_t site_ln1, site_ln2;
    float eps;
    uint16_t *a_bf, *xhat1; float* rstd1;
    uint16_t* h1; uint32_t* relu_bits; int bits_wide;
    uint16_t *out_bf, *xhat2; float* rstd2; float* out32;
    uint16_t* qkv;                   // the next block's [M,768], or NULL
    int ablate;                      // measurement build only (TTSMI_CHAIN_ABLATE): 1 no multiplies, 2 no DMA, 8 no in-loop stores
    unsigned long long* dbg;         // measurement build only: per (workgroup, wave) phase stamps, ttsmi_dense_chain_debug
    float* xbuf;                     // SPLIT form: [tiles][2][waves][16 tiles x 64 lanes x 4] fp32 partial sums of the FFN output
    uint32_t* xflag;                 // SPLIT form: [tiles][2][waves], 0 between launches
};

__device__ __forceinline__ void ch_dma16(const void* gsrc, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_off) : "memory", "m0");
}
__device__ __forceinline__ void ch_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void ch_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned ch_lds_offset(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ uint2 ch_pack4(float a, float b, float c, float d) {
    bf16x4 h;
    h[0] = (__bf16)a; h[1] = (__bf16)b; h[2] = (__bf16)c; h[3] = (__bf16)d;
    return *reinterpret_cast<uint2*>(&h);
}
__device__ __forceinline__ float ch_bf(const bf16x8& v, int e) { return (float)v[e]; }
// relu of two packed bf16 values: a negative bf16 is a negative int16 (v_pk_max_i16 against 0; -0.0 becomes +0.0)
typedef short short2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t ch_relu2(uint32_t w) {
    short2v v = __builtin_bit_cast(short2v, w);
    const short2v z = {0, 0};
    v = __builtin_elementwise_max(v, z);
    return __builtin_bit_cast(uint32_t, v);
}
// bit e = bf16 element e of the 8-element group is > 0 (gemm_k256.hip: kw_pos_bits)
__device__ __forceinline__ uint32_t ch_pos_bits(const uint4& v) {
    auto two = [](uint32_t w) { return (((int32_t)(w << 16) > 0) ? 1u : 0u) | (((int32_t)(w & 0xFFFF0000u) > 0) ? 2u : 0u); };
    return two(v.x) | (two(v.y) << 2) | (two(v.z) << 4) | (two(v.w) << 6);
}

// ---- the weight stream ----------------------------------------------------------------------------------------------
struct ChainPackP {
    const uint16_t *wo_t, *w1_t, *w2_t, *wqkv_t;      // [256][512], [F][256], [256][F], [768][256] (W^T as stored by the shadow set)
    uint16_t* out;
    int F, nchunk, nstages;
    int wo_stages;                                    // 8 (forward stream) or 0 (backward stream: no product in front of the FFN pair)
};
#include "chain16.h"
#include "chain16b.h"

// waves per workgroup of both chain kernels by row count: four 16-row waves (64-row tiles, one wave per SIMD) while 128-row
// tiles would leave half of the 256 CUs without a workgroup, eight (two per SIMD) from there on; TTSMI_DENSE_CHAIN_NW = 4 / 8
// forces one form (A/B knob)
static int chain_nw(int M) {
    TTSMI_KNOB(forced, "TTSMI_DENSE_CHAIN_NW", 0);
    if (forced == 4 || forced == 8) return forced;
    return M <= 16384 ? 4 : 8;
}

// loader waves beside the four compute waves of the 64-row form (chain16.h: c16_loader_loop): FOUR by default (two: 45.0 -> 44.3 us
// alone at 6 400 rows, 4.50 -> 4.48 ms per step, lj-dist 2.91 -> 2.88; a 32-row form of two compute + four loader waves is no
// faster - 43.4 us, a workgroup's lifetime is its 52 stages of DMA - and was not kept: profiles/r06_chain_loaders4_ab.txt);
// TTSMI_DENSE_CHAIN_LOADERS = 0 / 2: none / two (A/B knob)
static int chain_loaders() {
    TTSMI_KNOB(on, "TTSMI_DENSE_CHAIN_LOADERS", 4);
    return on == 2 ? 2 : on != 0 ? 4 : 0;
}

// ---- the SPLIT forms' exchange buffers: one set per stream (launches of one stream are ordered; two streams may run chains at
// the same time), allocated on first use outside a capture, never freed.  Flags are zero between launches (chain16.h).
#define CHAIN_SPLIT_MAX_TILES 128                       // 64-row tiles: 8 192 rows, 256 workgroups
struct ChainSplitBuf { float* x; uint32_t* flag; };
static bool chain_split_buffers(hipStream_t st, ChainSplitBuf* out) {
    static std::mutex mu;
    static std::map<hipStream_t, ChainSplitBuf> bufs;
    std::lock_guard<std::mutex> lock(mu);
    auto it = bufs.find(st);
    if (it != bufs.end()) { *out = it->second; return true; }
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return false;   // no allocation inside a capture
    ChainSplitBuf b = {nullptr, nullptr};
    const size_t xbytes = (size_t)CHAIN_SPLIT_MAX_TILES * 2 * 4 * 16 * 256 * sizeof(float), fbytes = (size_t)CHAIN_SPLIT_MAX_TILES * 2 * 4 * sizeof(uint32_t);
    if (hipMalloc((void**)&b.x, xbytes) != hipSuccess) return false;
    if (hipMalloc((void**)&b.flag, fbytes) != hipSuccess || hipMemset(b.flag, 0, fbytes) != hipSuccess) {
        (void)hipFree(b.x);
        return false;
    }
    bufs[st] = b;
    *out = b;
    return true;
}
// two workgroups per 64-row tile (chain16.h: SPLIT) while twice the workgroups still fit the CUs; TTSMI_DENSE_CHAIN_SPLIT=0: never
static bool chain_split(int M, int F) {
    TTSMI_KNOB(on, "TTSMI_DENSE_CHAIN_SPLIT", 1);
    static const int cus = [] { int n = 0, dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n; }();
    const int tiles = ttsmi_cdiv(M, 64);
    return on && chain_nw(M) == 4 && chain_loaders() == 4 && tiles <= CHAIN_SPLIT_MAX_TILES && ttsmi_cdiv(tiles, 8) * 16 <= cus && (F / 64) % 2 == 0;
}

static int chain_bwd_stages(int F) { return 2 * (F / 64) + C16B_CTX_STAGES; }
static int chain_stages(int F, int with_qkv) { return CH_WO_STAGES + 2 * (F / 64) + (with_qkv ? CH_QKV_STAGES : 0); }

#ifdef TTSMI_ABLATION_BUILD
static unsigned long long* g_chain_dbg = nullptr;
extern "C" int ttsmi_dense_chain_debug(void* buf) { g_chain_dbg = (unsigned long long*)buf; return 0; }     // [workgroups][4][8] uint64
#endif

extern "C" {

size_t ttsmi_dense_chain_pack_bytes(int F, int with_qkv) {
    return F > 0 && F % 64 == 0 ? (size_t)chain_stages(F, with_qkv) * CH_STAGE_BYTES : 0;
}

int ttsmi_dense_chain_pack(const uint16_t* wo_t, const uint16_t* w1_t, const uint16_t* w2_t, const uint16_t* wqkv_next_t, int F,
                           void* out, size_t out_bytes, ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(wo_t && w1_t && w2_t && out, "dense_chain_pack: null pointer");
    TTSMI_CHECK_ARG(F > 0 && F % 64 == 0, "dense_chain_pack: F must be a multiple of 64 (got %d)", F);
    TTSMI_CHECK_ARG(out_bytes >= ttsmi_dense_chain_pack_bytes(F, wqkv_next_t != nullptr), "dense_chain_pack: output buffer too small");
    TTSMI_CHECK_ARG(((((uintptr_t)wo_t) | ((uintptr_t)w1_t) | ((uintptr_t)w2_t) | ((uintptr_t)wqkv_next_t)) & 7) == 0 && (((uintptr_t)out) & 15) == 0,
                    "dense_chain_pack: operands must be 8-byte (output: 16-byte) aligned");
    ChainPackP p;
    p.wo_t = wo_t; p.w1_t = w1_t; p.w2_t = w2_t; p.wqkv_t = wqkv_next_t; p.out = (uint16_t*)out;
    p.F = F; p.nchunk = F / 64; p.nstages = chain_stages(F, wqkv_next_t != nullptr);
    p.wo_stages = CH_WO_STAGES;
    const long total = (long)p.nstages * CH_STAGE_FRAGS * 64;
    hipLaunchKernelGGL(dense_chain16_pack_kernel, dim3(ttsmi_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, p);
    TTSMI_CHECK_LAUNCH("dense_chain_pack");
    return TTSMI_OK;
}

/* n weight streams in one launch (ttsmi_chain_pack_job: a forward stream = ttsmi_dense_chain_pack's operands, a backward
 * stream = ttsmi_dense_chain_bwd_pack's); same bytes as the single calls */
int ttsmi_dense_chain_pack_batched(const ttsmi_chain_pack_job* jobs, int n, ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(jobs && n > 0, "dense_chain_pack_batched: bad job list");
    for (int i0 = 0; i0 < n; i0 += C16_PACK_MAX_JOBS) {
        const int m = n - i0 < C16_PACK_MAX_JOBS ? n - i0 : C16_PACK_MAX_JOBS;
        ChainPackJobs J;
        memset(&J, 0, sizeof(J));
        long most = 0;
        for (int i = 0; i < m; ++i) {
            const ttsmi_chain_pack_job* q = &jobs[i0 + i];
            ChainPackP& p = J.job[i];
            TTSMI_CHECK_ARG(q->w1 && q->w2 && q->wo && q->out, "dense_chain_pack_batched: job %d: null pointer", i0 + i);
            TTSMI_CHECK_ARG(q->F > 0 && q->F % 64 == 0, "dense_chain_pack_batched: job %d: F must be a multiple of 64 (got %d)", i0 + i, q->F);
            TTSMI_CHECK_ARG(((((uintptr_t)q->wo) | ((uintptr_t)q->w1) | ((uintptr_t)q->w2) | ((uintptr_t)q->wqkv_next)) & 7) == 0 &&
                                (((uintptr_t)q->out) & 15) == 0,
                            "dense_chain_pack_batched: job %d: operands must be 8-byte (output: 16-byte) aligned", i0 + i);
            if (q->backward) {           // (ttsmi_dense_chain_bwd_pack: the forward packer's roles with the matrices as stored)
                TTSMI_CHECK_ARG(q->out_bytes >= ttsmi_dense_chain_bwd_pack_bytes(q->F), "dense_chain_pack_batched: job %d: output buffer too small", i0 + i);
                p.wo_t = nullptr; p.w1_t = q->w2; p.w2_t = q->w1; p.wqkv_t = q->wo + (long)CH_D * CH_D;
                p.nstages = chain_bwd_stages(q->F); p.wo_stages = 0;
            } else {
                TTSMI_CHECK_ARG(q->out_bytes >= ttsmi_dense_chain_pack_bytes(q->F, q->wqkv_next != nullptr),
                                "dense_chain_pack_batched: job %d: output buffer too small", i0 + i);
                p.wo_t = q->wo; p.w1_t = q->w1; p.w2_t = q->w2; p.wqkv_t = q->wqkv_next;
                p.nstages = chain_stages(q->F, q->wqkv_next != nullptr); p.wo_stages = CH_WO_STAGES;
            }
            p.out = (uint16_t*)q->out; p.F = q->F; p.nchunk = q->F / 64;
            const long total = (long)p.nstages * CH_STAGE_FRAGS * 64;
            most = total > most ? total : most;
        }
        hipLaunchKernelGGL(dense_chain16_pack_jobs_kernel, dim3(ttsmi_cdiv(most, 256), m), dim3(256), 0, (hipStream_t)stream, J);
        TTSMI_CHECK_LAUNCH("dense_chain_pack_batched");
    }
    return TTSMI_OK;
}

int ttsmi_dense_chain_supported(int M, int d, int F) { return M > 0 && d == CH_D && F >= 64 && F % 64 == 0 && F <= CH_MAXF; }

int ttsmi_dense_chain_fwd(const uint16_t* h_bf, const uint16_t* ctx, const void* wpack, size_t wpack_bytes, int M, int F,
                          const float* bo, const float* ln1_g, const float* ln1_b, const float* b1, const float* b2,
                          const float* ln2_g, const float* ln2_b, const float* bqkv_next, const uint8_t* row_pad, float p_drop,
                          uint64_t seed, const int64_t* step_dev, uint32_t site_ln1, uint32_t site_ln2, float eps, uint16_t* a_bf,
                          uint16_t* xhat1, float* rstd1, uint16_t* h1, void* relu_bits, int relu_bits_layout, uint16_t* out_bf,
                          uint16_t* xhat2, float* rstd2, float* out32, uint16_t* qkv_next, ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(h_bf && ctx && wpack && bo && ln1_g && ln1_b && b1 && b2 && ln2_g && ln2_b && a_bf && xhat1 && rstd1 && h1 &&
                        out_bf && xhat2 && rstd2,
                    "dense_chain_fwd: null pointer");
    TTSMI_CHECK_ARG(ttsmi_dense_chain_supported(M, CH_D, F), "dense_chain_fwd: unsupported shape M=%d F=%d", M, F);
    TTSMI_CHECK_ARG((qkv_next == nullptr) == (bqkv_next == nullptr), "dense_chain_fwd: qkv_next and bqkv_next go together");
    TTSMI_CHECK_ARG(wpack_bytes >= ttsmi_dense_chain_pack_bytes(F, qkv_next != nullptr), "dense_chain_fwd: weight stream too short");
    TTSMI_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "dense_chain_fwd: bad dropout rate");
    TTSMI_CHECK_ARG(((((uintptr_t)h_bf) | ((uintptr_t)ctx) | ((uintptr_t)wpack) | ((uintptr_t)a_bf) | ((uintptr_t)xhat1) | ((uintptr_t)h1) |
                      ((uintptr_t)out_bf) | ((uintptr_t)xhat2) | ((uintptr_t)out32) | ((uintptr_t)qkv_next) | ((uintptr_t)bo) |
                      ((uintptr_t)ln1_g) | ((uintptr_t)ln1_b) | ((uintptr_t)b1) | ((uintptr_t)b2) | ((uintptr_t)ln2_g) | ((uintptr_t)ln2_b) |
                      ((uintptr_t)bqkv_next)) & 15) == 0 && (((uintptr_t)relu_bits) & 7) == 0,
                    "dense_chain_fwd: operands must be 16-byte aligned");
    ChainP p;
    memset(&p, 0, sizeof(p));
    p.h_bf = h_bf; p.cx = ctx; p.wpack = (const unsigned char*)wpack; p.M = M; p.F = F; p.nchunk = F / 64;
    p.nstages = chain_stages(F, qkv_next != nullptr);
    p.nhalf = 2;
    p.bo = bo; p.ln1_g = ln1_g; p.ln1_b = ln1_b; p.b1 = b1; p.b2 = b2; p.ln2_g = ln2_g; p.ln2_b = ln2_b; p.bqkv = bqkv_next;
    p.row_pad = row_pad;
    p.thr = p_drop > 0.f ? ttsmi_drop_threshold(p_drop) : 0u;
    p.inv_keep = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    p.seed = seed; p.step_dev = step_dev; p.site_ln1 = site_ln1; p.site_ln2 = site_ln2; p.eps = eps;
    p.a_bf = a_bf; p.xhat1 = xhat1; p.rstd1 = rstd1; p.h1 = h1; p.relu_bits = (uint32_t*)relu_bits;
    // the bit matrix is read back by ttsmi_hgemm_k256_masked_bits: its 256-column variant from 16 384 rows (gemm_k256.hip: kw_launch)
    TTSMI_KNOB(wide, "TTSMI_HGEMM_K256_WIDE", 1);
    p.bits_wide = (wide && F % 256 == 0 && (M >= 16384 || wide > 1)) ? 1 : 0;
    if (relu_bits != nullptr && relu_bits_layout == 1) p.bits_wide = 2;
    if (relu_bits != nullptr && !p.bits_wide) TTSMI_CHECK_ARG(F % 128 == 0, "dense_chain_fwd: the bit matrix needs F %% 128 == 0");
    p.out_bf = out_bf; p.xhat2 = xhat2; p.rstd2 = rstd2; p.out32 = out32; p.qkv = qkv_next;
#ifdef TTSMI_ABLATION_BUILD
    TTSMI_ABLATE_KNOB(ablate, "TTSMI_CHAIN_ABLATE");
    p.ablate = ablate;
    p.dbg = g_chain_dbg;
#endif
#define CHAIN_FWD_LAUNCH(NW_, NL_)                                                                                              \
    do {                                                                                                                        \
        if (out32 != nullptr)                                                                                                   \
            TTSMI_LAUNCH_EV((dense_chain16_kernel<true, NW_, NL_>), dim3(ttsmi_cdiv(M, 16 * NW_)), dim3(64 * (NW_ + NL_)), 0, (hipStream_t)stream, p); \
        else                                                                                                                    \
            TTSMI_LAUNCH_EV((dense_chain16_kernel<false, NW_, NL_>), dim3(ttsmi_cdiv(M, 16 * NW_)), dim3(64 * (NW_ + NL_)), 0, (hipStream_t)stream, p); \
    } while (0)
    const int nw = chain_nw(M), nl = chain_loaders();
    ChainSplitBuf xb = {nullptr, nullptr};
    if (chain_split(M, F) && chain_split_buffers((hipStream_t)stream, &xb)) {
        ttsmi_note_kernel("dense_chain16_kernel<4 waves, split>");
        p.xbuf = xb.x; p.xflag = xb.flag;
        const dim3 grid(ttsmi_cdiv(ttsmi_cdiv(M, 64), 8) * 16);
        if (out32 != nullptr) TTSMI_LAUNCH_EV((dense_chain16_kernel<true, 4, 4, true>), grid, dim3(512), 0, (hipStream_t)stream, p);
        else TTSMI_LAUNCH_EV((dense_chain16_kernel<false, 4, 4, true>), grid, dim3(512), 0, (hipStream_t)stream, p);
    } else if (nw == 4) {
        ttsmi_note_kernel("dense_chain16_kernel<4 waves>");
        if (nl == 4) CHAIN_FWD_LAUNCH(4, 4);
        else if (nl == 2) CHAIN_FWD_LAUNCH(4, 2);
        else CHAIN_FWD_LAUNCH(4, 0);
    } else {
        ttsmi_note_kernel("dense_chain16_kernel");
        CHAIN_FWD_LAUNCH(8, 0);
    }
#undef CHAIN_FWD_LAUNCH
    TTSMI_CHECK_LAUNCH("dense_chain_fwd");
    return TTSMI_OK;
}

// ---- backward chain (chain16b.h) -----------------------------------------------------------------------------------

size_t ttsmi_dense_chain_bwd_pack_bytes(int F) { return F > 0 && F % 64 == 0 ? (size_t)chain_bwd_stages(F) * CH_STAGE_BYTES : 0; }

int ttsmi_dense_chain_bwd_supported(int M, int d, int F) { return ttsmi_dense_chain_supported(M, d, F) && M >= 1; }

/* rows of dgamma / dbeta partials ttsmi_dense_chain_bwd leaves in part_ws: one per 128-row workgroup */
int ttsmi_dense_chain_bwd_nparts(int M) { return ttsmi_cdiv(M, chain_nw(M) * 16); }

/* w1_b [256][F], w2_b [F][256], wo_b [512][256]: the weights AS STORED (bf16 shadows) */
int ttsmi_dense_chain_bwd_pack(const uint16_t* w1_b, const uint16_t* w2_b, const uint16_t* wo_b, int F, void* out, size_t out_bytes,
                               ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(w1_b && w2_b && wo_b && out, "dense_chain_bwd_pack: null pointer");
    TTSMI_CHECK_ARG(F > 0 && F % 64 == 0, "dense_chain_bwd_pack: F must be a multiple of 64 (got %d)", F);
    TTSMI_CHECK_ARG(out_bytes >= ttsmi_dense_chain_bwd_pack_bytes(F), "dense_chain_bwd_pack: output buffer too small");
    TTSMI_CHECK_ARG(((((uintptr_t)w1_b) | ((uintptr_t)w2_b) | ((uintptr_t)wo_b)) & 7) == 0 && (((uintptr_t)out) & 15) == 0,
                    "dense_chain_bwd_pack: operands must be 8-byte (output: 16-byte) aligned");
    // the forward packer's three roles with the backward's matrices: "W1^T [F][256]" = W2 as stored (rows = hidden feature, k = d),
    // "W2^T [256][F]" = W1 as stored (rows = d, k = hidden feature), the tail = the ctx half of Wo as stored ([256][256]: 4 stages)
    ChainPackP p;
    p.wo_t = nullptr; p.w1_t = w2_b; p.w2_t = w1_b; p.wqkv_t = wo_b + (long)CH_D * CH_D; p.out = (uint16_t*)out;
    p.F = F; p.nchunk = F / 64; p.nstages = chain_bwd_stages(F); p.wo_stages = 0;
    const long total = (long)p.nstages * CH_STAGE_FRAGS * 64;
    hipLaunchKernelGGL(dense_chain16_pack_kernel, dim3(ttsmi_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, p);
    TTSMI_CHECK_LAUNCH("dense_chain_bwd_pack");
    return TTSMI_OK;
}

int ttsmi_dense_chain_bwd(const uint16_t* df, const uint16_t* da, const uint16_t* xhat1, const float* rstd1, const float* ln1_g,
                          const uint8_t* row_pad, const void* relu_bits_lane, const void* wpack, size_t wpack_bytes, int M, int F,
                          float p_drop, uint64_t seed, const int64_t* step_dev, uint32_t site_ln1, uint16_t* dh1, uint16_t* d_o,
                          void* dres, int dres_is_bf16, uint16_t* dctx, void* part_ws, size_t part_ws_bytes, ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(df && da && xhat1 && rstd1 && ln1_g && relu_bits_lane && wpack && dh1 && d_o && dres && dctx && part_ws,
                    "dense_chain_bwd: null pointer");
    TTSMI_CHECK_ARG(ttsmi_dense_chain_bwd_supported(M, CH_D, F), "dense_chain_bwd: unsupported shape M=%d F=%d", M, F);
    TTSMI_CHECK_ARG(wpack_bytes >= ttsmi_dense_chain_bwd_pack_bytes(F), "dense_chain_bwd: weight stream too short");
    TTSMI_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "dense_chain_bwd: bad dropout rate");
    const int nparts = ttsmi_dense_chain_bwd_nparts(M);
    TTSMI_CHECK_ARG(part_ws_bytes >= ttsmi_layernorm_partials_bytes(nparts, CH_D), "dense_chain_bwd: partial-sum workspace too small");
    TTSMI_CHECK_ARG(((((uintptr_t)df) | ((uintptr_t)da) | ((uintptr_t)xhat1) | ((uintptr_t)wpack) | ((uintptr_t)dh1) | ((uintptr_t)d_o) |
                      ((uintptr_t)dres) | ((uintptr_t)dctx) | ((uintptr_t)ln1_g) | ((uintptr_t)part_ws)) & 15) == 0 &&
                        (((uintptr_t)relu_bits_lane) & 1) == 0,
                    "dense_chain_bwd: operands must be 16-byte aligned");
    ChainBP p;
    memset(&p, 0, sizeof(p));
    p.df = df; p.da = da; p.xhat = xhat1; p.rstd = rstd1; p.gamma = ln1_g; p.row_pad = row_pad;
    p.bits16 = (const uint16_t*)relu_bits_lane; p.wpack = (const unsigned char*)wpack;
    p.M = M; p.F = F; p.nchunk = F / 64; p.nstages = chain_bwd_stages(F); p.nparts = nparts;
    p.thr = p_drop > 0.f ? ttsmi_drop_threshold(p_drop) : 0u;
    p.inv_keep = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    p.seed = seed; p.step_dev = step_dev; p.site = site_ln1;
    p.dh1 = dh1; p.d_o = d_o; p.dctx = dctx; p.dres = dres; p.dres_bf16 = dres_is_bf16 ? 1 : 0; p.part = (float*)part_ws;
    ttsmi_note_kernel("dense_chain16_bwd_kernel");
#define CHAIN_BWD_LAUNCH(NW_, NL_) TTSMI_LAUNCH_EV((dense_chain16_bwd_kernel<NW_, NL_>), dim3(nparts), dim3(64 * (NW_ + NL_)), 0, (hipStream_t)stream, p)
    const int nw = chain_nw(M), nl = chain_loaders();
    ChainSplitBuf xb = {nullptr, nullptr};
    if (chain_split(M, F) && chain_split_buffers((hipStream_t)stream, &xb)) {
        ttsmi_note_kernel("dense_chain16_bwd_kernel<split>");
        p.xbuf = xb.x; p.xflag = xb.flag;
        TTSMI_LAUNCH_EV((dense_chain16_bwd_kernel<4, 4, true>), dim3(ttsmi_cdiv(nparts, 8) * 16), dim3(512), 0, (hipStream_t)stream, p);
    } else if (nw == 4 && nl == 4) CHAIN_BWD_LAUNCH(4, 4);
    else if (nw == 4 && nl == 2) CHAIN_BWD_LAUNCH(4, 2);
    else if (nw == 4) CHAIN_BWD_LAUNCH(4, 0);
    else CHAIN_BWD_LAUNCH(8, 0);
#undef CHAIN_BWD_LAUNCH
    TTSMI_CHECK_LAUNCH("dense_chain_bwd");
    return TTSMI_OK;
}

}  // extern "C"
