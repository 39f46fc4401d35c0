// ttsmi_dense_block_fwd / _bwd: the launch sequence of one SelfAttentionDenseBlock (model/layers.py:214-230) issued
// from C++ through the library's own entry points - the same calls, in the same order, that ops.DenseBlockFn makes
// from Python (TTSMI_BF16 path), minus ~14 us of interpreter / ctypes / allocator time per launch.
#include <stdlib.h>

#include "common.h"

#define TRY(call)                 \
    do {                          \
        int rc__ = (call);        \
        if (rc__) return rc__;    \
    } while (0)

static const float kLnEps = 1e-6f;     // LayerNormalization(epsilon=1e-6), model/layers.py:27,96,207

// Measurement hook (bench.py's instrumented step): when set, every launch group issued from here is announced to the
// observer before and after it is enqueued, with its stream and its algorithmic FLOPs / bytes, so that the caller can
// bracket it with HIP events exactly as it brackets the entry points it calls itself.  Process-wide, not thread-safe:
// a measurement facility, unset in normal operation.
static ttsmi_launch_observer g_observer = nullptr;
extern "C" int ttsmi_set_launch_observer(ttsmi_launch_observer cb) { g_observer = cb; return 0; }
struct Obs {
    const char* name; double flops, bytes; ttsmi_stream_t st;
    Obs(const char* n, double f, double b, ttsmi_stream_t s) : name(n), flops(f), bytes(b), st(s) { if (g_observer) g_observer(0, name, flops, bytes, st); }
    ~Obs() { if (g_observer) g_observer(1, name, flops, bytes, st); }
};
// algorithmic bytes: every operand read once, every result written once (bench.py:_bytes)
static double gemm_bytes(double M, double N, double K, int out_bytes, bool acc, double extra = 0) {
    return M * K * 2 + 2 * K * N + M * N * out_bytes * (acc ? 2 : 1) + extra;
}
#define OBS(name, flops, bytes, st) Obs obs__(name, flops, bytes, st)

// the FFN's ReLU travels to the backward as a bit matrix when the caller provides one and both FFN GEMMs of the block are
// launches of the K = 256 weight-stationary kernel (the only kernels that write / read it)
static bool relu_bits(const ttsmi_dense_block* D) {
    return D->relu_bits != nullptr && D->fuse_ln && D->d == 256 && ttsmi_hgemm_k256_eligible(D->B * D->T, D->F, D->d);
}

// TTSMI_WGRAD_EVENTS (A/B knob, read once): 2 / 1 = "lazy" hand-offs to the weight-gradient stream (dense_block_bwd_impl)
static bool lazy_wgrad_events() {
    TTSMI_KNOB(wev, "TTSMI_WGRAD_EVENTS", 4);
    return wev == 2 || wev == 1;
}
// THE predicate of the backward chain (csrc/chain16b.h): the forward asks it to choose the ReLU bit layout it writes, the
// backward to choose its path, the host (ttsmi_dense_block_bwd_chained) to size the LayerNorm-partial reduction - one
// definition, so the three cannot disagree (advisor findings, round 5: the forward used to look at chain_bw alone and
// wrote the lane layout for a backward that then read it as the bit matrix under TTSMI_WGRAD_EVENTS=1/2).
static bool chain_bw_path(const ttsmi_dense_block* D) {
    return D->chain_bw != nullptr && D->chain_w != nullptr && D->fuse_ln && D->res16 != 0 && !lazy_wgrad_events() &&
           D->relu_bits != nullptr && ttsmi_dense_chain_bwd_supported(D->B * D->T, D->d, D->F);
}

static int check_desc(const ttsmi_dense_block* D, const char* who) {
    TTSMI_CHECK_ARG(D, "%s: null descriptor", who);
    TTSMI_CHECK_ARG(D->B > 0 && D->H > 0 && D->T > 0 && D->d > 0 && D->F > 0 && D->d % D->H == 0,
                    "%s: bad shape B=%d H=%d T=%d d=%d F=%d", who, D->B, D->H, D->T, D->d, D->F);
    TTSMI_CHECK_ARG(D->d % 64 == 0 && D->F % 8 == 0, "%s: needs d %% 64 == 0 and F %% 8 == 0", who);
    TTSMI_CHECK_ARG(D->pad && D->klen, "%s: null mask", who);       // (main_stream == 0 is HIP's default stream)
    if (D->fuse_ln)
        TTSMI_CHECK_ARG(D->d == 256 && D->xhat1 && D->xhat2 && D->lnp_ws1 && D->lnp_ws2,
                        "%s: fuse_ln needs d == 256 and the x^ / partial-sum buffers", who);
    return TTSMI_OK;
}

// TTSMI_WGRAD_KERNEL_EVENTS=0 (A/B knob): every hand-off to the weight-gradient stream is a hipEventRecord marker again
static bool kernel_events() {
    TTSMI_KNOB(on, "TTSMI_WGRAD_KERNEL_EVENTS", 1);
    return on != 0;
}
// arm hand-off `ev` of block D before the entry point whose last kernel produces the tensor handed over
static thread_local hipEvent_t t_armed = nullptr, t_prerecorded = nullptr;
// (a stream that is being captured into a hipGraph takes the ordinary event record: the graph's cross-stream edge is
// built from hipEventRecord / hipStreamWaitEvent pairs, a kernel's stop event is not captured as a dependency)
static thread_local bool t_capturing = false;
// the block whose qkv projection the last forward chain of this thread computed (ttsmi_dense_block.above): consumed by that
// block's own forward, which then skips its projection
static thread_local const ttsmi_dense_block* t_qkv_written_for = nullptr;
static void note_capture(const ttsmi_dense_block* D) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    t_capturing = hipStreamIsCapturing((hipStream_t)D->main_stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone;
}
static void arm(const ttsmi_dense_block* D, int ev) {
    if (D->side_stream && kernel_events() && !t_capturing) {
        t_armed = (hipEvent_t)D->ev[ev];
        ttsmi_arm_stop_event(t_armed);
    }
}

extern "C" {

int ttsmi_dense_block_fwd(const ttsmi_dense_block* D, const float* h, const uint16_t* h_bf) {
    TRY(check_desc(D, "dense_block_fwd"));
    TTSMI_CHECK_ARG(h_bf && (h || (D->fuse_ln && D->res16)), "dense_block_fwd: null input");
    const int M = D->B * D->T, d = D->d, F = D->F, dh = d / D->H;
    ttsmi_stream_t st = D->main_stream;
    // qkv = h.Wqkv + b                                                     (layers.py:116-118, fused)
    // (qkv_done: written by the row-local chain of the block below, whose `above` is this descriptor)
    // ... and only then: a descriptor whose qkv_done survived a rebind, or that is run out of order, would attend over stale qkv
    // without any error (advisor finding, round 5) - the chain that writes a block's qkv leaves its descriptor's address here
    if (D->qkv_done) {
        TTSMI_CHECK_ARG(t_qkv_written_for == D, "dense_block_fwd: qkv_done is set but the block below did not just run its chain into this block");
        t_qkv_written_for = nullptr;
    }
    if (!D->qkv_done) {
      OBS("ttsmi_hgemm_tn", 2.0 * M * 3 * d * d, gemm_bytes(M, 3 * d, d, 2, false), st);
      TRY(ttsmi_hgemm_tn(h_bf, 0, d, nullptr, 0, 0, D->wqkv_t, d, D->bqkv, nullptr, 0, D->qkv, 3L * d, M, 3 * d, d,
                         TTSMI_GEMM_OUT_BF16, 1, 0, 0, 0, st)); }
    // ctx = softmax(q k^T / sqrt(dh) + mask) v                             (layers.py:176-195)
    {
        const double T2 = (double)D->T * D->T;
        OBS("ttsmi_attention_fwd", 4.0 * D->B * D->H * T2 * dh, (double)M * 3 * d * 2 + (double)M * d * 2 + 4.0 * D->B * D->H * D->T, st);
        if (D->dropmask && D->rate > 0.f)
            TRY(ttsmi_attention_fwd_masked(D->qkv, D->pad, D->klen, D->cx, D->lse, D->B, D->H, D->T, dh, D->rate, D->dropmask, TTSMI_BF16_IO, st));
        else if (D->attn_split && D->rate == 0.f)
            TRY(ttsmi_attention_fwd_splitkeys(D->qkv, D->pad, D->klen, D->cx, D->lse, D->B, D->H, D->T, dh, D->attn_ws,
                                              D->attn_ws_bytes, st));
        else
            TRY(ttsmi_attention_fwd(D->qkv, D->pad, D->klen, D->cx, D->lse, D->B, D->H, D->T, dh, D->rate, D->seed,
                                    D->step_dev, D->site_attn, TTSMI_BF16_IO, st));
    }
    if (D->chain_w != nullptr) {
        // the row-local chain (csrc/chain.hip): o-projection + res-norm 1 -> FFN1 -> FFN2 + res-norm 2 -> the next block's
        // qkv projection in ONE launch; what the backward keeps is written, nothing is read back
        TTSMI_CHECK_ARG(D->fuse_ln && D->res16 && ttsmi_dense_chain_supported(M, d, F),
                        "dense_block_fwd: chain_w needs fuse_ln, res16, d == 256 and F %% 64 == 0");
        const ttsmi_dense_block* A = D->above;
        if (A != nullptr)
            TTSMI_CHECK_ARG(A->qkv_done && A->B == D->B && A->T == D->T && A->d == d && A->qkv && A->bqkv,
                            "dense_block_fwd: `above` is not the next block of the same shape (or its qkv_done is not set)");
        const double wbytes = (double)ttsmi_dense_chain_pack_bytes(F, A != nullptr);
        // bytes: h, ctx in; a_bf, x^1, h1 (+ bits), x^2, out_bf (+ fp32 out, + qkv') out; the weight stream once
        OBS("ttsmi_dense_chain_fwd", 2.0 * M * d * (2.0 * d + 2.0 * F + (A ? 3.0 * d : 0.0)),
            (double)M * 2 * (2.0 * d + 4.0 * d + F + (A ? 3.0 * d : 0.0)) + ((D->res16 & 2) ? 4.0 * M * d : 0.0) +
                (relu_bits(D) ? (double)M * F / 8 : 0.0) + wbytes + 8.0 * M, st);
        // (chain_bw: the backward runs as a chain too and reads the ReLU pattern in the chain's own lane layout)
        const bool lane_bits = chain_bw_path(D);
        t_qkv_written_for = A;
        return ttsmi_dense_chain_fwd(h_bf, D->cx, D->chain_w, D->chain_w_bytes, M, F, D->bo, D->ln1_g, D->ln1_b, D->b1, D->b2, D->ln2_g,
                                     D->ln2_b, A ? A->bqkv : nullptr, D->pad, D->rate, D->seed, D->step_dev, D->site_ln1, D->site_ln2,
                                     kLnEps, D->a_bf, D->xhat1, D->rstd1, D->h1,
                                     lane_bits ? D->relu_bits : (relu_bits(D) ? D->relu_bits : nullptr), lane_bits ? 1 : 0, D->out_bf,
                                     D->xhat2, D->rstd2, (D->res16 & 2) ? D->out : nullptr, A ? A->qkv : nullptr, st);
    }
    TTSMI_CHECK_ARG(D->above == nullptr, "dense_block_fwd: `above` without chain_w");
    if (D->fuse_ln) {
        // res16: the residual stream between the fused kernels is the bf16 tensor the next GEMM reads anyway (h_bf, a_bf,
        // out_bf) - the fp32 copies are neither read nor, except for a requested block output (bit 1), written
        const bool r16 = D->res16 != 0;
        float* out32 = (!r16 || (D->res16 & 2)) ? D->out : nullptr;
        // a = LN(drop([h | ctx].Wo + b) + h) * mask in ONE launch            (layers.py:148-150,211,229)
        { OBS("ttsmi_hgemm_ln_fwd", 2.0 * M * d * 2 * d, gemm_bytes(M, d, 2 * d, r16 ? 0 : 4, false, (double)M * d * ((r16 ? 2 : 4) + 2 + 2)), st);
          if (r16)
              TRY(ttsmi_hgemm_ln_fwd_h(h_bf, d, D->cx, d, d, D->wo_t, 2L * d, D->bo, h_bf, D->ln1_g, D->ln1_b, D->pad, D->rate,
                                       D->site_ln1, D->seed, D->step_dev, kLnEps, nullptr, D->a_bf, D->xhat1, D->rstd1, M, d, 2 * d, st));
          else
              TRY(ttsmi_hgemm_ln_fwd(h_bf, d, D->cx, d, d, D->wo_t, 2L * d, D->bo, h, D->ln1_g, D->ln1_b, D->pad, D->rate,
                                     D->site_ln1, D->seed, D->step_dev, kLnEps, D->a, D->a_bf, D->xhat1, D->rstd1, M, d, 2 * d, st)); }
        { OBS("ttsmi_hgemm_tn", 2.0 * M * F * d, gemm_bytes(M, F, d, 2, false, relu_bits(D) ? (double)M * F / 8 : 0.0), st);
          if (relu_bits(D))        // h1 and, for the backward, the sign of every element as one bit (layers.py:99 ReLU)
              TRY(ttsmi_hgemm_k256_relu_bits(D->a_bf, d, D->w1_t, d, D->b1, D->h1, F, (uint8_t*)D->relu_bits, M, F, st));
          else
              TRY(ttsmi_hgemm_tn(D->a_bf, 0, d, nullptr, 0, 0, D->w1_t, d, D->b1, nullptr, 0, D->h1, F, M, F, d,
                                 TTSMI_GEMM_RELU | TTSMI_GEMM_OUT_BF16, 1, 0, 0, 0, st)); }
        OBS("ttsmi_hgemm_ln_fwd", 2.0 * M * d * F, gemm_bytes(M, d, F, out32 ? 4 : 0, false, (double)M * d * ((r16 ? 2 : 4) + 2 + 2)), st);
        // out = LN(drop(h1.W2 + b2) + a) * mask in ONE launch                (layers.py:100-102,230)
        if (r16)
            TRY(ttsmi_hgemm_ln_fwd_h(D->h1, F, nullptr, 0, 0, D->w2_t, F, D->b2, D->a_bf, D->ln2_g, D->ln2_b, D->pad, D->rate,
                                     D->site_ln2, D->seed, D->step_dev, kLnEps, out32, D->out_bf, D->xhat2, D->rstd2, M, d, F, st));
        else
            TRY(ttsmi_hgemm_ln_fwd(D->h1, F, nullptr, 0, 0, D->w2_t, F, D->b2, D->a, D->ln2_g, D->ln2_b, D->pad, D->rate,
                                   D->site_ln2, D->seed, D->step_dev, kLnEps, D->out, D->out_bf, D->xhat2, D->rstd2, M, d, F, st));
        return TTSMI_OK;
    }
    // o = [h | ctx].Wo + b                                                 (layers.py:148-149)
    TRY(ttsmi_hgemm_tn(h_bf, 0, d, D->cx, d, d, D->wo_t, 2L * d, D->bo, nullptr, 0, D->o, d, M, d, 2 * d, 0, 1, 0, 0, 0, st));
    // a = LN(drop(o) + h) * mask                                           (layers.py:150,211,229)
    TRY(ttsmi_add_layernorm_fwd(D->o, h, D->ln1_g, D->ln1_b, nullptr, nullptr, 0, D->pad, D->rate, D->site_ln1, 0.f, 0,
                                D->seed, D->step_dev, kLnEps, D->a, D->mean1, D->rstd1, M, d, D->a_bf, st));
    // f = relu(a.W1 + b1).W2 + b2                                          (layers.py:99-100)
    TRY(ttsmi_hgemm_tn(D->a_bf, 0, d, nullptr, 0, 0, D->w1_t, d, D->b1, nullptr, 0, D->h1, F, M, F, d,
                       TTSMI_GEMM_RELU | TTSMI_GEMM_OUT_BF16, 1, 0, 0, 0, st));
    TRY(ttsmi_hgemm_tn(D->h1, 0, F, nullptr, 0, 0, D->w2_t, F, D->b2, nullptr, 0, D->f, d, M, d, F, 0, 1, 0, 0, 0, st));
    // out = LN(drop(f) + a) * mask                                         (layers.py:101-102,230)
    TRY(ttsmi_add_layernorm_fwd(D->f, D->a, D->ln2_g, D->ln2_b, nullptr, nullptr, 0, D->pad, D->rate, D->site_ln2, 0.f, 0,
                                D->seed, D->step_dev, kLnEps, D->out, D->mean2, D->rstd2, M, d, D->out_bf, st));
    return TTSMI_OK;
}

// The last launch of a chained block: dy = dh + dqkv.Wqkv^T (+ d_o.Wo[:d]^T when folded) is the upstream gradient of the
// lower block's res-norm 2, whose backward runs in the epilogue -> L->df (bf16), L->da (fp32, or bf16 with res16)
static int chained_ln2(const ttsmi_dense_block* D, const ttsmi_dense_block* L, bool fold, bool dh16, int M, int d, ttsmi_stream_t st) {
    if (dh16 || L->res16) {
        if (!dh16) {
            ttsmi_set_error("dense_block_bwd: a res16 chain needs the folded output-projection dgrad (TTSMI_DENSE_FOLD_DHTO)");
            return TTSMI_ERR_INVALID_ARG;
        }
        return ttsmi_hgemm_ln_bwd_dual_h(D->dqkv, 3L * d, D->d_o, d, 3 * d, D->wqkv_b, 3L * d, D->wo_b, d, (const uint16_t*)D->dh,
                                         L->xhat2, L->rstd2, L->ln2_g, L->pad, L->rate, L->site_ln2, L->seed, L->step_dev, L->df,
                                         L->da, 1, L->lnp_ws2, L->lnp_ws2_bytes, M, d, 4 * d, st);
    }
    return ttsmi_hgemm_ln_bwd_dual(D->dqkv, 3L * d, fold ? D->d_o : nullptr, d, fold ? 3 * d : 0, D->wqkv_b, 3L * d,
                                   fold ? D->wo_b : nullptr, d, D->dh, L->xhat2, L->rstd2, L->ln2_g, L->pad, L->rate, L->site_ln2,
                                   L->seed, L->step_dev, L->df, L->da, L->lnp_ws2, L->lnp_ws2_bytes, M, d, fold ? 4 * d : 3 * d, st);
}

// The block's weight gradients leave their slab reductions to ONE batched launch at the end of the block's backward
// (five 4-8 us launches otherwise, on a stream whose tail is exposed at the end of the step): every call carves its
// slabs out of the shared workspace, and the batch is flushed early if the workspace or the job table runs out.
struct WgradBatch {
    ttsmi_wgrad_job jobs[TTSMI_WGRAD_MAX_JOBS];
    int n;
    size_t used;
};
static int wgrad_flush(const ttsmi_dense_block* D, WgradBatch* wb) {
    if (wb->n == 0) return TTSMI_OK;
    hipStream_t st = D->side_stream ? (hipStream_t)D->side_stream : (hipStream_t)D->main_stream;
    const int rc = ttsmi_hgemm_wgrad_reduce_jobs(wb->jobs, wb->n, (ttsmi_stream_t)st);
    wb->n = 0;
    wb->used = 0;                      // (stream order: the next slabs are written after this reduction has read these)
    return rc;
}

// weight gradient dW[kin,n] = x^T . dy (+ db) on the side stream, ordered after everything the main stream has
// enqueued so far (its operands) through one event
static int wgrad_side(const ttsmi_dense_block* D, WgradBatch* wb, int ev, bool record, const uint16_t* x, int ldx,
                      const uint16_t* dy, int lddy, float* dw, float* db, int kin, int n, const uint16_t* x2 = nullptr,
                      int ldx2 = 0, int k1 = 0) {
    const int M = D->B * D->T;
    TTSMI_ABLATE_KNOB(skip, "TTSMI_DEBUG_SKIP_WGRAD");      // measurement knob (results are then WRONG: no weight gradients): main-stream-only backward time
    if (skip) return TTSMI_OK;
    hipStream_t main_st = (hipStream_t)D->main_stream;
    hipStream_t st = D->side_stream ? (hipStream_t)D->side_stream : main_st;
    if (D->side_stream && record) {
        hipEvent_t e = (hipEvent_t)D->ev[ev];
        TTSMI_CHECK_ARG(e, "dense_block_bwd: null event");
        // the producing entry point may have taken the armed event onto its last kernel (TTSMI_LAUNCH_EV): then it is
        // already recorded; an event still armed was not taken and is recorded the ordinary way
        bool recorded = false;
        if (t_prerecorded == e) {                // (recorded by the block above, on the kernel that produced df)
            recorded = true;
            t_prerecorded = nullptr;
        } else if (t_armed == e) {
            recorded = ttsmi_take_stop_event() != e;
            t_armed = nullptr;
        }
        if ((!recorded && hipEventRecord(e, main_st) != hipSuccess) || hipStreamWaitEvent(st, e, 0) != hipSuccess) {
            ttsmi_set_error("dense_block_bwd: event hand-off to the weight-gradient stream failed");
            return TTSMI_ERR_LAUNCH;
        }
    }
    OBS("ttsmi_hgemm_wgrad_rows", 2.0 * M * kin * n, (double)M * (kin + n) * 2 + 4.0 * kin * n, (ttsmi_stream_t)st);
    TTSMI_KNOB(batched, "TTSMI_WGRAD_BATCH_REDUCE", 1);     // 0: every weight gradient reduces its own slabs (A/B knob)
    const size_t need = ttsmi_hgemm_wgrad_rows_exact_bytes(M, kin, n, db != nullptr);
    if (!batched || need > D->wgrad_ws_bytes) {
        TRY(wgrad_flush(D, wb));
        if (x2) {           // (callers only pass x2 on the batched path; without it: the two halves one after the other)
            TRY(ttsmi_hgemm_wgrad_rows(x, 1, ldx, dy, 1, lddy, dw, n, db, M, k1, n, 1, 0, 0, 0, D->wgrad_ws, D->wgrad_ws_bytes, st));
            return ttsmi_hgemm_wgrad_rows(x2, 1, ldx2, dy, 1, lddy, dw + (long)k1 * n, n, nullptr, M, kin - k1, n, 1, 0, 0, 0,
                                          D->wgrad_ws, D->wgrad_ws_bytes, st);
        }
        return ttsmi_hgemm_wgrad_rows(x, 1, ldx, dy, 1, lddy, dw, n, db, M, kin, n, 1, 0, 0, 0, D->wgrad_ws, D->wgrad_ws_bytes, st);
    }
    if (wb->n == TTSMI_WGRAD_MAX_JOBS || wb->used + need > D->wgrad_ws_bytes) TRY(wgrad_flush(D, wb));
    ttsmi_wgrad_job* job = &wb->jobs[wb->n];
    if (x2)
        TRY(ttsmi_hgemm_wgrad_rows_deferred_dual(x, ldx, x2, ldx2, k1, dy, lddy, dw, n, db, M, kin, n, (char*)D->wgrad_ws + wb->used,
                                                 D->wgrad_ws_bytes - wb->used, st, job));
    else
    TRY(ttsmi_hgemm_wgrad_rows_deferred(x, 1, ldx, dy, 1, lddy, dw, n, db, M, kin, n, (char*)D->wgrad_ws + wb->used,
                                        D->wgrad_ws_bytes - wb->used, st, job));
    if (job->splits > 0) {
        wb->n += 1;
        wb->used += need;
    }
    return TTSMI_OK;
}

static int dense_block_bwd_impl(const ttsmi_dense_block* D, const float* h, const uint16_t* h_bf, const float* dout);

// The armed / pre-recorded hand-off events are thread-local state that must not outlive the call that set them: an error
// return between arm() and the producing launch would otherwise leave an event armed for the next, unrelated
// TTSMI_LAUNCH_EV launch of this thread (advisor finding, round 3).  Every exit path goes through here: nothing stays
// armed; `t_prerecorded` survives a SUCCESSFUL chained call only (the lower block's call consumes it).
// Wo = Dense(concat([q_in, ctx])) is stored [2d][d]: its two halves' gradients read the same d_o - one dual-X launch
// (TTSMI_WGRAD_WO_DUAL=0: two launches, the round-3 form - same products; the row split, hence the order of the fp32
// partial sums, may differ)
static int wgrad_wo(const ttsmi_dense_block* D, WgradBatch* wb, int ev, bool record, const uint16_t* h_bf) {
    const int d = D->d;
    TTSMI_KNOB(dual, "TTSMI_WGRAD_WO_DUAL", 1);
    if (dual && d % 128 == 0)
        return wgrad_side(D, wb, ev, record, h_bf, d, D->d_o, d, D->g_wo, D->g_bo, 2 * d, d, D->cx, d, d);
    TRY(wgrad_side(D, wb, ev, record, h_bf, d, D->d_o, d, D->g_wo, D->g_bo, d, d));
    return wgrad_side(D, wb, ev, false, D->cx, d, D->d_o, d, D->g_wo + (long)d * d, nullptr, d, d);
}

int ttsmi_dense_block_bwd(const ttsmi_dense_block* D, const float* h, const uint16_t* h_bf, const float* dout) {
    const int rc = dense_block_bwd_impl(D, h, h_bf, dout);
    (void)ttsmi_take_stop_event();
    t_armed = nullptr;
    if (rc != TTSMI_OK) t_prerecorded = nullptr;
    return rc;
}

/* 1 when ttsmi_dense_block_bwd(D) will run the backward chain (its res-norm 1 parameter partials are then
 * ttsmi_dense_chain_bwd_nparts(B * T) rows in lnp_ws1, otherwise ttsmi_hgemm_ln_bwd_nparts(B * T)). */
int ttsmi_dense_block_bwd_chained(const ttsmi_dense_block* D) { return D != nullptr && chain_bw_path(D) ? 1 : 0; }

}  // extern "C"

static int dense_block_bwd_impl(const ttsmi_dense_block* D, const float* h, const uint16_t* h_bf, const float* dout) {
    TRY(check_desc(D, "dense_block_bwd"));
    if (t_prerecorded != nullptr && t_prerecorded != (hipEvent_t)D->ev[0]) t_prerecorded = nullptr;   // left by an aborted chain
    TTSMI_CHECK_ARG(h_bf && (h || D->fuse_ln) && (dout || (D->fuse_ln && D->ln2_done)), "dense_block_bwd: null input");
    // res16: the gradient of the residual stream travels as bf16 between the fused kernels - `da` always, `dh` when this
    // block's last launch is the chained full-row kernel that consumes it (otherwise dh is the fp32 result of the call)
    const bool r16 = D->fuse_ln && D->res16 != 0;
    TTSMI_KNOB(fold_knob, "TTSMI_DENSE_FOLD_DHTO", 1);
    const bool fold = fold_knob && D->fuse_ln && D->below != nullptr;
    const bool dh16 = r16 && fold;
    const int M = D->B * D->T, d = D->d, F = D->F, dh = d / D->H;
    ttsmi_stream_t st = D->main_stream;
    const bool dropout = D->rate > 0.f;
    note_capture(D);
    WgradBatch wb;
    wb.n = 0;
    wb.used = 0;
    // TTSMI_WGRAD_EVENTS=2 (A/B knob): two hand-offs to the weight-gradient stream per block instead of four - the FFN
    // pair after dh1, the attention-side three after dqkv (fewer event packets on both queues, later starts on the side stream)
    // TTSMI_WGRAD_EVENTS=1: ONE hand-off for the FFN pair and the output projection, placed right before the attention
    // backward - the full-row GEMM+LN kernels own a CU's LDS (144 KB), so a weight-gradient workgroup sitting on a CU
    // (48 KB) keeps their workgroups off it, while the attention kernels (19 KB, latency bound) share a CU with it
    // (round 6: ONE hand-off per chained block, on the attention backward's last kernel - three kernel-borne hand-offs cost the
    // main queue ~5 us of idle each in the trace - measured 4.567 against 4.534 ms per step, 2.910 against 2.903 on lj-dist:
    // the later start of the weight gradients costs more than the idle; TWO hand-offs - W2 behind the backward chain - 4.491 against
    // 4.479; profiles/r06_one_handoff_per_block_ab.txt; neither kept)
    TTSMI_KNOB(wev, "TTSMI_WGRAD_EVENTS", 4);
    const bool lazy = lazy_wgrad_events();
    const bool pre_attn = wev == 1;
    // ---- LN2 + FFN: df (bf16) = dLN2/dx, da (fp32) = dLN2/dres
    if (D->fuse_ln) {
        if (!D->ln2_done) {         // (chained: the block above already left df / da / the parameter partials)
            TTSMI_CHECK_ARG(dout, "dense_block_bwd: null dout");
            OBS("ttsmi_layernorm_bwd_xhat", 0.0, (double)M * d * (4 + 2 + 2 + (r16 ? 2 : 4)), st);
            if (!lazy) arm(D, 0);
            if (r16)
                TRY(ttsmi_layernorm_bwd_xhat_h(dout, D->xhat2, D->rstd2, D->ln2_g, D->pad, D->rate, D->site_ln2, D->seed, D->step_dev,
                                               D->df, (uint16_t*)D->da, D->lnp_ws2, D->lnp_ws2_bytes, M, d, st));
            else
                TRY(ttsmi_layernorm_bwd_xhat(dout, D->xhat2, D->rstd2, D->ln2_g, D->pad, D->rate, D->site_ln2, D->seed, D->step_dev,
                                             D->df, D->da, D->lnp_ws2, D->lnp_ws2_bytes, M, d, st));
        }
    } else
        TRY(ttsmi_add_layernorm_bwd(dout, D->f, D->a, D->ln2_g, D->mean2, D->rstd2, nullptr, nullptr, 0, D->pad, D->rate,
                                    D->site_ln2, 0.f, 0, D->seed, D->step_dev, 0, dropout ? nullptr : D->da, D->da, nullptr,
                                    nullptr, nullptr, M, d, D->ln_ws2, D->ln_ws_bytes, D->df, st));
    if (!lazy) TRY(wgrad_side(D, &wb, 0, true, D->h1, F, D->df, d, D->g_w2, D->g_b2, F, d));
    // ---- the backward chain (csrc/chain16b.h): FFN2 dgrad + ReLU mask, FFN1 dgrad + res-norm 1 backward and the dctx product
    // in ONE launch; dh1 and d_o are written for the weight-gradient stream, nothing is read back
    const bool chain_bw = chain_bw_path(D);
    if (chain_bw) {
        { OBS("ttsmi_dense_chain_bwd", 2.0 * M * d * (2.0 * F + d),
              (double)M * 2 * (4.0 * d + F + 2.0 * d) + (double)M * F / 8 + (double)ttsmi_dense_chain_bwd_pack_bytes(F) + 4.0 * M, st);
          arm(D, 1);
          TRY(ttsmi_dense_chain_bwd(D->df, (const uint16_t*)D->da, D->xhat1, D->rstd1, D->ln1_g, D->pad, D->relu_bits, D->chain_bw,
                                    D->chain_bw_bytes, M, F, D->rate, D->seed, D->step_dev, D->site_ln1, D->dh1, D->d_o, D->dh,
                                    dh16 ? 1 : 0, D->dctx, D->lnp_ws1, D->lnp_ws1_bytes, st)); }
        TRY(wgrad_side(D, &wb, 1, true, D->a_bf, d, D->dh1, F, D->g_w1, D->g_b1, d, F));
        TRY(wgrad_wo(D, &wb, 2, false, h_bf));
        if (!fold) {         // the bottom block of a stack: the q_in half's gradient joins the fp32 dh here
            OBS("ttsmi_hgemm_tn", 2.0 * M * d * d, gemm_bytes(M, d, d, 4, true), st);
            TRY(ttsmi_hgemm_tn(D->d_o, 0, d, nullptr, 0, 0, D->wo_b, d, nullptr, nullptr, 0, D->dh, d, M, d, d,
                               TTSMI_GEMM_ACCUMULATE, 1, 0, 0, 0, st));
        }
    } else {
    { OBS("ttsmi_hgemm_tn", 2.0 * M * F * d, gemm_bytes(M, F, d, 2, false, relu_bits(D) ? (double)M * F / 8 : (double)M * F * 2), st);
      if (!pre_attn) arm(D, 1);
      if (relu_bits(D))           // relu' from the bit matrix the forward left: 1 / 16 of the bytes of re-reading h1
          TRY(ttsmi_hgemm_k256_masked_bits(D->df, d, D->w2_b, d, (const uint8_t*)D->relu_bits, D->dh1, F, M, F, st));
      else
          TRY(ttsmi_hgemm_tn(D->df, 0, d, nullptr, 0, 0, D->w2_b, d, nullptr, (const float*)D->h1, F, D->dh1, F, M, F, d,
                             TTSMI_GEMM_OUT_BF16 | TTSMI_GEMM_MASK_BF16, 1, 0, 0, 0, st)); }           // relu' fused
    if (lazy && !pre_attn) TRY(wgrad_side(D, &wb, 1, true, D->h1, F, D->df, d, D->g_w2, D->g_b2, F, d));
    if (!pre_attn) TRY(wgrad_side(D, &wb, 1, !lazy, D->a_bf, d, D->dh1, F, D->g_w1, D->g_b1, d, F));
    if (D->fuse_ln) {
        // (da + dh1.W1^T) never reaches HBM: res-norm 1's backward runs in the dgrad's epilogue -> d_o (bf16), dh (fp32)
        // (bytes: dh1 + W1 + the residual gradient in (da) and out (dh) at their stored widths + x^ + d_o)
        OBS("ttsmi_hgemm_ln_bwd", 2.0 * M * d * F, gemm_bytes(M, d, F, 0, false, (double)M * d * ((r16 ? 2 : 4) + (dh16 ? 2 : 4) + 2 + 2)), st);
        if (!lazy) arm(D, 2);
        if (r16)
            TRY(ttsmi_hgemm_ln_bwd_dual_h(D->dh1, F, nullptr, 0, 0, D->w1_b, F, nullptr, 0, (const uint16_t*)D->da, D->xhat1, D->rstd1,
                                          D->ln1_g, D->pad, D->rate, D->site_ln1, D->seed, D->step_dev, D->d_o, D->dh, dh16 ? 1 : 0,
                                          D->lnp_ws1, D->lnp_ws1_bytes, M, d, F, st));
        else
            TRY(ttsmi_hgemm_ln_bwd(D->dh1, F, D->w1_b, F, D->da, D->xhat1, D->rstd1, D->ln1_g, D->pad, D->rate, D->site_ln1,
                                   D->seed, D->step_dev, D->d_o, D->dh, D->lnp_ws1, D->lnp_ws1_bytes, M, d, F, st));
    } else {
        TRY(ttsmi_hgemm_tn(D->dh1, 0, F, nullptr, 0, 0, D->w1_b, F, nullptr, nullptr, 0, D->da, d, M, d, F,
                           TTSMI_GEMM_ACCUMULATE, 1, 0, 0, 0, st));                                   // da += dh1.W1^T
        // ---- LN1 + output projection: d_o (bf16), dh (fp32)
        TRY(ttsmi_add_layernorm_bwd(D->da, D->o, h, D->ln1_g, D->mean1, D->rstd1, nullptr, nullptr, 0, D->pad, D->rate,
                                    D->site_ln1, 0.f, 0, D->seed, D->step_dev, 0, dropout ? nullptr : D->dh, D->dh, nullptr,
                                    nullptr, nullptr, M, d, D->ln_ws1, D->ln_ws_bytes, D->d_o, st));
    }
    if (!lazy) {
        TRY(wgrad_wo(D, &wb, 2, true, h_bf));
    }
    // dh += do.Wo_top^T (fp32) and dctx = do.Wo_ctx^T (bf16): Wo as stored is [2d][d] = both weight halves back to back,
    // and both products read d_o - one weight-stationary launch when the shape suits it (d = 256, decoder-size M)
    TTSMI_KNOB(split_ok, "TTSMI_DENSE_SPLIT_DGRAD", 1);           // TTSMI_DENSE_SPLIT_DGRAD=0: two launches (A/B knob)
    // Chained blocks: the q_in half (do.Wo_top^T) is a second K segment of the full-row kernel that completes dh below
    // (ttsmi_hgemm_ln_bwd_dual, K = 3d + d) - no read-modify-write of the fp32 dh (59 MB at M = 28 800) and a
    // 256-column product at that kernel's rate instead of a launch of its own; only dctx is computed here.
    if (fold) {
        OBS("ttsmi_hgemm_tn", 2.0 * M * d * d, gemm_bytes(M, d, d, 2, false), st);
        if (pre_attn) arm(D, 2);
        TRY(ttsmi_hgemm_tn(D->d_o, 0, d, nullptr, 0, 0, D->wo_b + (long)d * d, d, nullptr, nullptr, 0, D->dctx, d, M, d, d,
                           TTSMI_GEMM_OUT_BF16, 1, 0, 0, 0, st));
    } else if (split_ok && d == 256 && ttsmi_hgemm_k256_eligible(M, 2 * d, d)) {
        OBS("ttsmi_hgemm_tn", 2.0 * M * d * 2 * d, gemm_bytes(M, d, d, 4, true) + gemm_bytes(M, d, d, 2, false) - (double)M * d * 2, st);
        if (pre_attn) arm(D, 2);
        TRY(ttsmi_hgemm_k256_split(D->d_o, d, D->wo_b, d, D->dh, d, d, D->dctx, d, M, 2 * d, st));
    } else {
        { OBS("ttsmi_hgemm_tn", 2.0 * M * d * d, gemm_bytes(M, d, d, 4, true), st);
          TRY(ttsmi_hgemm_tn(D->d_o, 0, d, nullptr, 0, 0, D->wo_b, d, nullptr, nullptr, 0, D->dh, d, M, d, d,
                             TTSMI_GEMM_ACCUMULATE, 1, 0, 0, 0, st)); }                                 // dh += do.Wo_top^T
        { OBS("ttsmi_hgemm_tn", 2.0 * M * d * d, gemm_bytes(M, d, d, 2, false), st);
          TRY(ttsmi_hgemm_tn(D->d_o, 0, d, nullptr, 0, 0, D->wo_b + (long)d * d, d, nullptr, nullptr, 0, D->dctx, d, M, d, d,
                             TTSMI_GEMM_OUT_BF16, 1, 0, 0, 0, st)); }
    }
    }   // (!chain_bw)
    if (pre_attn) {      // everything the weight-gradient stream can do before dqkv exists, handed over in one go
        TRY(wgrad_side(D, &wb, 2, true, D->h1, F, D->df, d, D->g_w2, D->g_b2, F, d));
        TRY(wgrad_side(D, &wb, 2, false, D->a_bf, d, D->dh1, F, D->g_w1, D->g_b1, d, F));
        TRY(wgrad_wo(D, &wb, 2, false, h_bf));
    }
    // ---- attention + qkv projection
    {
    const double T2 = (double)D->T * D->T;
    OBS("ttsmi_attention_bwd", 8.0 * D->B * D->H * T2 * dh, (double)M * 3 * d * 2 * 3 + (double)M * d * 2 * 3 + 16.0 * D->B * D->H * D->T, st);
    arm(D, 3);
    if (D->dropmask && dropout)
        TRY(ttsmi_attention_bwd_masked(D->qkv, D->pad, D->klen, D->cx, D->dctx, D->lse, D->dqkv, D->B, D->H, D->T, dh,
                                       D->rate, D->dropmask, D->attn_ws, D->attn_ws_bytes, TTSMI_BF16_IO, st));
    else
        TRY(ttsmi_attention_bwd(D->qkv, D->pad, D->klen, D->cx, D->dctx, D->lse, D->dqkv, D->B, D->H, D->T, dh, D->rate,
                                D->seed, D->step_dev, D->site_attn, D->attn_ws, D->attn_ws_bytes, TTSMI_BF16_IO, st));
    }
    if (lazy && !pre_attn) {
        TRY(wgrad_wo(D, &wb, 3, true, h_bf));
    }
    TRY(wgrad_side(D, &wb, 3, !lazy || pre_attn, h_bf, d, D->dqkv, 3 * d, D->g_wqkv, D->g_bqkv, d, 3 * d));
    TRY(wgrad_flush(D, &wb));            // the block's five slab reductions, one launch on the weight-gradient stream
    const ttsmi_dense_block* L = D->below;
    if (D->fuse_ln && L != nullptr) {
        // dh + dqkv.Wqkv^T is the upstream gradient of the lower block's res-norm 2: its backward runs in this epilogue
        TTSMI_CHECK_ARG(L->fuse_ln && L->ln2_done && L->B == D->B && L->T == D->T && L->d == d && (L->res16 != 0) == (D->res16 != 0),
                        "dense_block_bwd: `below` is not a chained block of the same shape and residual type");
        OBS("ttsmi_hgemm_ln_bwd", 2.0 * M * d * (fold ? 4 : 3) * d,
            gemm_bytes(M, d, (fold ? 4 : 3) * d, 0, false, (double)M * d * ((dh16 ? 2 : 4) + (L->res16 ? 2 : 4) + 2 + 2)), st);
        // this launch produces the lower block's df: its hand-off 0 rides on this kernel (recorded here either way, so
        // the lower block only waits)
        if (L->side_stream && kernel_events() && !lazy && !t_capturing) {
            arm(L, 0);
            const int rc_ = chained_ln2(D, L, fold, dh16, M, d, st);
            hipEvent_t left = ttsmi_take_stop_event();
            t_armed = nullptr;
            if (rc_) return rc_;
            if (left && hipEventRecord(left, (hipStream_t)st) != hipSuccess) {
                ttsmi_set_error("dense_block_bwd: event record failed");
                return TTSMI_ERR_LAUNCH;
            }
            t_prerecorded = (hipEvent_t)L->ev[0];
            return TTSMI_OK;
        }
        TRY(chained_ln2(D, L, fold, dh16, M, d, st));
        return TTSMI_OK;
    }
    OBS("ttsmi_hgemm_tn", 2.0 * M * d * 3 * d, gemm_bytes(M, d, 3 * d, 4, true), st);
    TRY(ttsmi_hgemm_tn(D->dqkv, 0, 3L * d, nullptr, 0, 0, D->wqkv_b, 3L * d, nullptr, nullptr, 0, D->dh, d, M, d, 3 * d,
                       TTSMI_GEMM_ACCUMULATE, 1, 0, 0, 0, st));                                       // dh += dqkv.Wqkv^T
    return TTSMI_OK;
}

extern "C" {

static int check_stack(const ttsmi_dense_block* const* blocks, int n, const char* who) {
    TTSMI_CHECK_ARG(blocks && n > 0 && n <= 64, "%s: bad block list", who);
    for (int i = 0; i < n; ++i) {
        TTSMI_CHECK_ARG(blocks[i], "%s: null descriptor %d", who, i);
        TTSMI_CHECK_ARG(blocks[i]->B == blocks[0]->B && blocks[i]->T == blocks[0]->T && blocks[i]->d == blocks[0]->d,
                        "%s: block %d has another shape than block 0", who, i);
    }
    return TTSMI_OK;
}

int ttsmi_dense_stack_fwd(const ttsmi_dense_block* const* blocks, int n, const float* h, const uint16_t* h_bf) {
    TRY(check_stack(blocks, n, "dense_stack_fwd"));
    for (int i = 0; i < n; ++i)        // a block that skips its qkv projection must sit right above the chain that wrote it
        TTSMI_CHECK_ARG(!blocks[i]->qkv_done || (i > 0 && blocks[i - 1]->above == blocks[i] && blocks[i - 1]->chain_w != nullptr),
                        "dense_stack_fwd: block %d has qkv_done set but block %d does not run a chain into it", i, i - 1);
    for (int i = 0; i < n; ++i) {
        TRY(ttsmi_dense_block_fwd(blocks[i], h, h_bf));
        h = blocks[i]->out;                       // (not written inside a res16 stack, and then not read either)
        h_bf = blocks[i]->out_bf;
    }
    return TTSMI_OK;
}

int ttsmi_dense_stack_bwd(const ttsmi_dense_block* const* blocks, int n, const float* h, const uint16_t* h_bf, const float* dout) {
    TRY(check_stack(blocks, n, "dense_stack_bwd"));
    for (int i = n - 1; i >= 0; --i) {
        const float* hi = i ? blocks[i - 1]->out : h;
        const uint16_t* hbi = i ? blocks[i - 1]->out_bf : h_bf;
        TRY(ttsmi_dense_block_bwd(blocks[i], hi, hbi, dout));
        dout = blocks[i]->dh;                     // (bf16 behind a chained block, which ignores its dout argument)
    }
    return TTSMI_OK;
}

}  // extern "C"
