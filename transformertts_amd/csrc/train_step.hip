// ttsmi_ft_train_step: the launch sequence of ForwardTransformer._train_step (reference model/models.py:464-482) issued
// from C++ off ONE descriptor - the same entry points, with the same arguments, on the same streams as the per-layer host
// path of transformertts_amd/ops.py + model/models.py (which stays the reference implementation of this sequence and is
// held bit for bit against it by tests/test_cstep_gpu.py), minus the autograd engine and ~1.4 ms of interpreter time
// between the ~290 launches of a step.  See include/ttsmi.h (ttsmi_ft_step) for the contract.
#include "common.h"

#define TRY(call)                 \
    do {                          \
        int rc__ = (call);        \
        if (rc__) return rc__;    \
    } while (0)

static const float kLnEps = 1e-6f;     // LayerNormalization(epsilon=1e-6), model/layers.py:27,96,207,508

// ---- two small kernels the host path took from the framework (an `add` and a constant pad per step) ------------------
__global__ __launch_bounds__(256) void add2_f32_kernel(const float4* __restrict__ a, const float4* __restrict__ b,
                                                       float4* __restrict__ o, long n4, const float* __restrict__ at,
                                                       const float* __restrict__ bt, float* __restrict__ ot, int tail) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 x = a[i], y = b[i];
        o[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < tail) ot[threadIdx.x] = at[threadIdx.x] + bt[threadIdx.x];
}
__global__ __launch_bounds__(256) void pad_cols_f32_kernel(const float* __restrict__ src, int C, float* __restrict__ dst,
                                                           int Cp, long total) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long m = i / Cp;
        const int c = (int)(i - m * Cp);
        dst[i] = c < C ? src[m * C + c] : 0.f;
    }
}
static int ew_grid(long n) {
    long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

static int ev_record(void* ev, ttsmi_stream_t st, const char* what) {
    if (!ev || hipEventRecord((hipEvent_t)ev, (hipStream_t)st) != hipSuccess) {
        ttsmi_set_error("ft_train_step: event record failed (%s)", what);
        return TTSMI_ERR_LAUNCH;
    }
    return TTSMI_OK;
}
static int ev_wait(void* ev, ttsmi_stream_t st, const char* what) {
    if (!ev || hipStreamWaitEvent((hipStream_t)st, (hipEvent_t)ev, 0) != hipSuccess) {
        ttsmi_set_error("ft_train_step: stream wait failed (%s)", what);
        return TTSMI_ERR_LAUNCH;
    }
    return TTSMI_OK;
}
// stream `to` waits for everything enqueued on `from` so far
static int hand_off(void* ev, ttsmi_stream_t from, ttsmi_stream_t to, const char* what) {
    TRY(ev_record(ev, from, what));
    return ev_wait(ev, to, what);
}

// the events of ttsmi_ft_step.ev, by use
enum {
    EV_STEP_START = 0,   // main -> side: the step counter was advanced, last step's readers are done
    EV_MASK_ENC,         // side -> main: the encoder's keep-bit tables (+ every chain weight stream)
    EV_MASK_DEC,         // side -> main: the decoder's keep-bit tables
    EV_ENC_DONE,         // main -> side: the encoder output is complete
    EV_PRED_FWD,         // side -> main: the predictors' outputs
    EV_LOSS,             // main -> side: the loss gradients
    EV_PRED_BWD,         // side -> main: the predictors' gradient wrt the encoder output
    EV_OUT_WGRAD,        // main -> weight-gradient stream: the mel projection's operands
    EV_WGRAD_JOIN,       // weight-gradient stream -> main
    EV_SIDE_JOIN,        // side -> main at the end of the backward
    EV_PACK_START,       // main -> side: the bf16 shadows of this step's weights are written (phase 2)
    EV_PACK,             // side -> main: the chain kernels' weight streams for the next step
    EV_COUNT
};

struct LnBatch {         // deferred LayerNorm parameter reductions of ONE stream (ttsmi_layernorm_param_reduce_batched_nw)
    const void* ws[4 * TTSMI_FT_MAX_BLOCKS + 8];
    float* dg[4 * TTSMI_FT_MAX_BLOCKS + 8];
    float* db[4 * TTSMI_FT_MAX_BLOCKS + 8];
    float* dps[4 * TTSMI_FT_MAX_BLOCKS + 8];
    int nparts[4 * TTSMI_FT_MAX_BLOCKS + 8];
    int C[4 * TTSMI_FT_MAX_BLOCKS + 8];
    int n;
    void add(const void* w, float* g, float* b, float* ps, int np, int c) {
        ws[n] = w; dg[n] = g; db[n] = b; dps[n] = ps; nparts[n] = np; C[n] = c; ++n;
    }
    int flush(ttsmi_stream_t st) {
        if (n == 0) return TTSMI_OK;
        const int rc = ttsmi_layernorm_param_reduce_batched_nw(ws, dg, db, dps, nparts, C, n, st);
        n = 0;
        return rc;
    }
};
// Main-stream reductions pending between phase 0 and phase 1 (the same thread issues both; a data-parallel host flushes the
// decoder's before its all-reduce by running phase 1's head first - see ttsmi_ft_train_step).
static thread_local LnBatch t_ln_main;
static thread_local LnBatch t_ln_side;

static int check_step(const ttsmi_ft_step* S) {
    TTSMI_CHECK_ARG(S, "ft_train_step: null descriptor");
    TTSMI_CHECK_ARG(S->B > 0 && S->Tp > 0 && S->Tm > 0 && S->d > 0 && S->V > 0 && S->n_mel > 0, "ft_train_step: bad shape");
    TTSMI_CHECK_ARG(S->n_enc > 0 && S->n_enc <= TTSMI_FT_MAX_BLOCKS && S->n_dec > 0 && S->n_dec <= TTSMI_FT_MAX_BLOCKS,
                    "ft_train_step: %d + %d blocks (1..%d each)", S->n_enc, S->n_dec, TTSMI_FT_MAX_BLOCKS);
    TTSMI_CHECK_ARG(S->dur.n_layers > 0 && S->dur.n_layers <= TTSMI_FT_MAX_PRED_LAYERS && S->pit.n_layers > 0 &&
                        S->pit.n_layers <= TTSMI_FT_MAX_PRED_LAYERS, "ft_train_step: predictor depth");
    TTSMI_CHECK_ARG(S->side_stream && S->side_stream != S->main_stream, "ft_train_step: needs a side stream of its own");
    for (int i = 0; i < EV_COUNT; ++i) TTSMI_CHECK_ARG(S->ev[i], "ft_train_step: null event %d", i);
    for (int i = 0; i < S->n_enc; ++i)
        TTSMI_CHECK_ARG(S->enc[i] && S->enc[i]->B == S->B && S->enc[i]->T == S->Tp && S->enc[i]->d == S->d,
                        "ft_train_step: encoder block %d is not bound to (B, Tp, d)", i);
    for (int i = 0; i < S->n_dec; ++i)
        TTSMI_CHECK_ARG(S->dec[i] && S->dec[i]->B == S->B && S->dec[i]->T == S->Tm && S->dec[i]->d == S->d,
                        "ft_train_step: decoder block %d is not bound to (B, Tm, d)", i);
    return TTSMI_OK;
}

// keep-bit tables (ttsmi_attention_dropmask) and chain weight streams of a stack, on the side stream
static int side_prepare_stack(const ttsmi_ft_step* S, const ttsmi_dense_block* const* blk, int n) {
    // one launch per stack when its layers share shape and rate (they do: one stack = one (B, T), one dropout rate)
    void* masks[TTSMI_FT_MAX_BLOCKS];
    uint32_t sites[TTSMI_FT_MAX_BLOCKS];
    int m = 0;
    bool same = n <= TTSMI_FT_MAX_BLOCKS;
    for (int i = 0; i < n && same; ++i) {
        const ttsmi_dense_block* D = blk[i];
        if (!(D->dropmask && D->rate > 0.f)) continue;
        same = D->B == blk[0]->B && D->H == blk[0]->H && D->T == blk[0]->T && D->rate == blk[0]->rate && D->seed == blk[0]->seed &&
               D->step_dev == blk[0]->step_dev && blk[0]->dropmask && blk[0]->rate > 0.f;
        masks[m] = (void*)D->dropmask;
        sites[m++] = D->site_attn;
    }
    if (same) {
        if (m == 0) return TTSMI_OK;
        return ttsmi_attention_dropmask_stack(masks, sites, m, blk[0]->B, blk[0]->H, blk[0]->T, blk[0]->rate, blk[0]->seed,
                                              blk[0]->step_dev, S->side_stream);
    }
    for (int i = 0; i < n; ++i) {
        const ttsmi_dense_block* D = blk[i];
        if (D->dropmask && D->rate > 0.f)
            TRY(ttsmi_attention_dropmask((void*)D->dropmask, D->B, D->H, D->T, D->rate, D->seed, D->step_dev, D->site_attn, S->side_stream));
    }
    return TTSMI_OK;
}
// both stacks' chain weight streams (forward and backward: 24 at the benchmark's 6 + 6 blocks) as one launch
static int side_pack_stacks(const ttsmi_ft_step* S) {
    ttsmi_chain_pack_job jobs[4 * TTSMI_FT_MAX_BLOCKS];
    int m = 0;
    for (int s = 0; s < 2; ++s) {
        const ttsmi_dense_block* const* blk = s ? S->dec : S->enc;
        const int n = s ? S->n_dec : S->n_enc;
        for (int i = 0; i < n; ++i) {
            const ttsmi_dense_block* D = blk[i];
            if (!D->chain_w) continue;
            ttsmi_chain_pack_job* q = &jobs[m++];
            q->wo = D->wo_t; q->w1 = D->w1_t; q->w2 = D->w2_t; q->wqkv_next = D->above ? D->above->wqkv_t : nullptr;
            q->out = (void*)D->chain_w; q->out_bytes = D->chain_w_bytes; q->F = D->F; q->backward = 0;
            if (D->chain_bw) {
                q = &jobs[m++];
                q->wo = D->wo_b; q->w1 = D->w1_b; q->w2 = D->w2_b; q->wqkv_next = nullptr;
                q->out = (void*)D->chain_bw; q->out_bytes = D->chain_bw_bytes; q->F = D->F; q->backward = 1;
            }
        }
    }
    return m ? ttsmi_dense_chain_pack_batched(jobs, m, S->side_stream) : TTSMI_OK;
}

// ---- StatPredictor (model/layers.py:481-485, 510-524): ops.StatPredictorFn's launches ---------------------------------
static int predictor_fwd(const ttsmi_ft_step* S, const ttsmi_ft_predictor* P, const float* enc_out, ttsmi_stream_t st) {
    const int M = S->B * S->Tp;
    TRY(ttsmi_rowmask_mul(enc_out, S->pad_e, P->hm, M, S->d, st));                                   // layers.py:482
    const float* h = P->hm;
    for (int j = 0; j < P->n_layers; ++j) {
        const ttsmi_ft_pred_layer* L = &P->layer[j];
        // relu(conv1d(h)) as an implicit GEMM over the channels-last window                          layers.py:512-513
        TRY(ttsmi_hgemm_tn(h, 1, L->Cin, nullptr, 0, 0, L->w_t, (int64_t)L->k * L->Cin, L->bias, nullptr, 0, L->c, L->Cout, M,
                           L->Cout, L->k * L->Cin, TTSMI_GEMM_RELU, L->k, S->Tp, L->Cin, (L->k - 1) / 2, st));
        // LayerNorm + dropout                                                                        layers.py:514-515
        TRY(ttsmi_add_layernorm_fwd(L->c, nullptr, L->ln_g, L->ln_b, nullptr, nullptr, 0, nullptr, 0.f, 0, S->prate, L->site,
                                    S->seed, S->step_dev, kLnEps, L->n, L->mean, L->rstd, M, L->Cout, nullptr, st));
        h = L->n;
    }
    const ttsmi_ft_pred_layer* last = &P->layer[P->n_layers - 1];
    return ttsmi_rowdot_fwd(h, P->lin_w, P->lin_b, S->pad_e, P->y, M, last->Cout, P->relu_head, st);  // layers.py:479,484-485
}

static int predictor_bwd(const ttsmi_ft_step* S, const ttsmi_ft_predictor* P, const float* dy, ttsmi_stream_t st) {
    const int M = S->B * S->Tp;
    const ttsmi_ft_pred_layer* last = &P->layer[P->n_layers - 1];
    TRY(ttsmi_rowdot_bwd(dy, P->y, last->n, P->lin_w, S->pad_e, P->dn, P->g_lin_w, P->g_lin_b, M, last->Cout, P->relu_head,
                         P->rd_ws, P->rd_ws_bytes, st));
    const float* g = P->dn;
    for (int j = P->n_layers - 1; j >= 0; --j) {
        const ttsmi_ft_pred_layer* L = &P->layer[j];
        const float* x = j ? P->layer[j - 1].n : P->hm;                    // the conv's input
        // LayerNorm backward with the ReLU' of the conv output folded in; parameter partials deferred
        TRY(ttsmi_add_layernorm_bwd(g, L->c, nullptr, L->ln_g, L->mean, L->rstd, nullptr, nullptr, 0, nullptr, 0.f, 0, S->prate,
                                    L->site, S->seed, S->step_dev, 1, L->dc, nullptr, nullptr, nullptr, nullptr, M, L->Cout,
                                    L->ln_ws, L->ln_ws_bytes, nullptr, st));
        t_ln_side.add(L->ln_ws, L->g_ln_g, L->g_ln_b, nullptr, ttsmi_add_layernorm_bwd_nparts(M), L->Cout);
        // weight gradient (ops.ConvReluPreMaskedFn._backward)
        const int pad = (L->k - 1) / 2;
        if (L->Cin % 128 == 0 && L->Cout % 4 == 0) {
            TRY(ttsmi_hgemm_wgrad_rows(x, 0, L->Cin, L->dc, 0, L->Cout, L->g_w, L->Cout, L->g_b, M, L->k * L->Cin, L->Cout, L->k,
                                       S->Tp, L->Cin, pad, L->wg_ws, L->wg_ws_bytes, st));
        } else {
            const int64_t ldt = (M + 7) / 8 * 8;
            TRY(ttsmi_cast_transpose_bf16(x, L->Cin, L->xT, ldt, M, L->Cin, L->k, S->Tp, pad, st));
            TRY(ttsmi_cast_transpose_bf16(L->dc, L->Cout, L->dyT, ldt, M, L->Cout, 1, 0, 0, st));
            TRY(ttsmi_hgemm_wgrad(L->xT, L->dyT, ldt, L->g_w, L->Cout, L->g_b, M, L->k * L->Cin, L->Cout, L->wg_ws, L->wg_ws_bytes, st));
        }
        // input gradient: implicit GEMM over the (zero-padded) dy with the flipped-tap layout
        const float* dyp = L->dc;
        if (L->Cout_pad != L->Cout) {
            TRY(ttsmi_pad_cols_f32(L->dc, L->Cout, L->dc_pad, L->Cout_pad, M, st));
            dyp = L->dc_pad;
        }
        TRY(ttsmi_hgemm_tn(dyp, 1, L->Cout_pad, nullptr, 0, 0, L->w_d, (int64_t)L->k * L->Cout_pad, nullptr, nullptr, 0, L->dx, L->Cin,
                           M, L->Cin, L->k * L->Cout_pad, 0, L->k, S->Tp, L->Cout_pad, L->k - 1 - pad, st));
        g = L->dx;
    }
    return ttsmi_rowmask_mul(g, S->pad_e, P->dbranch, M, S->d, st);                                    // d(layers.py:482)
}

// the deferred LayerNorm parameter partials of a stack's blocks (ops.DenseBlockPlan._defer_ln)
static void defer_stack_ln(const ttsmi_dense_block* const* blk, int n) {
    for (int i = n - 1; i >= 0; --i) {
        const ttsmi_dense_block* D = blk[i];
        const int M = D->B * D->T;
        if (D->fuse_ln) {
            const int nw_row = ttsmi_hgemm_ln_bwd_nparts(M);
            t_ln_main.add(D->lnp_ws2, D->g_ln2_g, D->g_ln2_b, nullptr, D->ln2_done ? nw_row : ttsmi_layernorm_bwd_xhat_nparts(M), D->d);
            t_ln_main.add(D->lnp_ws1, D->g_ln1_g, D->g_ln1_b, nullptr,
                          ttsmi_dense_block_bwd_chained(D) ? ttsmi_dense_chain_bwd_nparts(M) : nw_row, D->d);
        } else {
            const int nw = ttsmi_add_layernorm_bwd_nparts(M);
            t_ln_main.add(D->ln_ws2, D->g_ln2_g, D->g_ln2_b, nullptr, nw, D->d);
            t_ln_main.add(D->ln_ws1, D->g_ln1_g, D->g_ln1_b, nullptr, nw, D->d);
        }
    }
}

static int phase0(const ttsmi_ft_step* S) {
    const int B = S->B, Tp = S->Tp, Tm = S->Tm, d = S->d, Me = B * Tp, Md = B * Tm;
    ttsmi_stream_t mn = S->main_stream, sd = S->side_stream;
    t_ln_main.n = 0;
    t_ln_side.n = 0;
    // ---- side stream: keep-bit tables of the attention dropout and the chain kernels' weight streams, ahead of their use
    // (the main stream's first launches are enqueued in front of the side stream's dozen: when the host, not the GPU, bounds
    // the step - small batches, a traced run - the main queue does not sit empty while the side work is being enqueued)
    TRY(ev_record(S->ev[EV_STEP_START], mn, "step start"));
    // ---- forward                                                                                   models.py:521-543
    TRY(ttsmi_token_pad_mask(S->tokens, S->pad_e, S->klen_e, B, Tp, mn));                             // :521
    TRY(ttsmi_embedding_fwd(S->tokens, S->emb, S->x_emb, Me, S->V, d, mn));                           // :522
    TRY(ttsmi_add_layernorm_fwd(S->x_emb, nullptr, S->enc_ln_g, S->enc_ln_b, S->pe_enc, S->enc_ps, Tp, nullptr, 0.f, 0, S->rate,
                                S->site_enc_ln, S->seed, S->step_dev, kLnEps, S->h0, S->mean0, S->rstd0, Me, d, S->h0_bf, mn));
    TRY(ev_wait(S->ev[EV_STEP_START], sd, "step start"));
    TRY(side_prepare_stack(S, S->enc, S->n_enc));
    if (S->pack_now) {             // the weight streams were not packed ahead by the previous step's phase 2 (first step, new binding)
        TRY(side_pack_stacks(S));
    }
    TRY(ev_record(S->ev[EV_MASK_ENC], sd, "encoder tables"));
    TRY(side_prepare_stack(S, S->dec, S->n_dec));
    TRY(ev_record(S->ev[EV_MASK_DEC], sd, "decoder tables"));
    if (!S->pack_now) TRY(ev_wait(S->ev[EV_PACK], mn, "chain weight streams"));
    TRY(ev_wait(S->ev[EV_MASK_ENC], mn, "encoder tables"));
    TRY(ttsmi_dense_stack_fwd(S->enc, S->n_enc, S->h0, S->h0_bf));                                    // :523
    const float* enc_out = S->enc[S->n_enc - 1]->out;
    TRY(hand_off(S->ev[EV_ENC_DONE], mn, sd, "encoder output"));
    TRY(ttsmi_pitch_embed_fwd(enc_out, S->tgt_pitch, S->pit_w, S->pit_b, S->hp, Me, d, mn));          // :527-531
    TRY(ttsmi_lenreg_index(S->tgt_dur, 1, S->idx, S->cum, S->lens, B, Tp, Tm, mn));                   // :540 Expand
    TRY(ttsmi_lenreg_fwd(S->hp, S->idx, S->x_dec, B, Tp, Tm, d, mn));
    TRY(ttsmi_length_pad_mask(S->lens, S->pad_d, S->klen_d, B, Tm, mn));                              // :541
    TRY(ttsmi_add_layernorm_fwd(S->x_dec, nullptr, S->dec_ln_g, S->dec_ln_b, S->pe_dec, S->dec_ps, Tm, nullptr, 0.f, 0, S->rate,
                                S->site_dec_ln, S->seed, S->step_dev, kLnEps, S->h1, S->mean1, S->rstd1, Md, d, S->h1_bf, mn));
    TRY(ev_wait(S->ev[EV_MASK_DEC], mn, "decoder tables"));
    TRY(ttsmi_dense_stack_fwd(S->dec, S->n_dec, S->h1, S->h1_bf));                                    // :542
    const float* dec_out = S->dec[S->n_dec - 1]->out;
    TRY(ttsmi_hgemm_tn(dec_out, 1, d, nullptr, 0, 0, S->out_wt, d, S->out_b, nullptr, 0, S->mel, S->n_mel, Md, S->n_mel, d, 0,
                       1, 0, 0, 0, mn));                                                              // :543
    // the two StatPredictors: ~50 small launches underneath the decoder (teacher forcing: nothing downstream reads them
    // but the losses), issued after it so that their backward is issued first                        models.py:524-526
    TRY(predictor_fwd(S, &S->dur, enc_out, sd));
    TRY(predictor_fwd(S, &S->pit, enc_out, sd));
    TRY(hand_off(S->ev[EV_PRED_FWD], sd, mn, "predictor outputs"));
    // ---- losses                                                                                    models.py:468-478
    {
        const float* pred[3] = {S->mel, S->dur.y, S->pit.y};
        const int64_t ldp[3] = {S->n_mel, 1, 1};
        const void* tgt[3] = {S->tgt_mel, S->tgt_dur, S->tgt_pitch};
        const int32_t tint[3] = {0, 1, 0};
        const int64_t rows[3] = {Md, Me, Me};
        const int64_t cols[3] = {S->n_mel, 1, 1};
        float* grad[3] = {S->g_mel, S->g_dur, S->g_pit};
        const int64_t ldg[3] = {S->n_mel, 1, 1};
        const bool den = S->loss_denom[0] > 0 || S->loss_denom[1] > 0 || S->loss_denom[2] > 0;
        TRY(ttsmi_l1_losses_weighted(3, pred, ldp, tgt, tint, rows, cols, S->loss_w, den ? S->loss_denom : nullptr, grad, ldg,
                                     S->loss_out, S->loss_out + 3, S->loss_ws, S->loss_ws_bytes, mn));
    }
    // ---- backward (the autograd engine's order: newest node first)                                 models.py:480
    TRY(ev_record(S->ev[EV_LOSS], mn, "loss gradients"));
    // mel projection: input gradient, then its weight gradient on the weight-gradient stream
    TRY(ttsmi_hgemm_tn(S->g_mel, 1, S->n_mel, nullptr, 0, 0, S->out_wb, S->n_mel, nullptr, nullptr, 0, S->d_dec_out, d, Md, d,
                       S->n_mel, 0, 1, 0, 0, 0, mn));
    {
        ttsmi_stream_t wg = S->wgrad_stream ? S->wgrad_stream : mn;
        if (S->wgrad_stream) TRY(hand_off(S->ev[EV_OUT_WGRAD], mn, wg, "mel projection operands"));
        TRY(ttsmi_hgemm_wgrad_rows(dec_out, 0, d, S->g_mel, 0, S->n_mel, S->g_out_w, S->n_mel, S->g_out_b, Md, d, S->n_mel, 1, 0, 0,
                                   0, S->wgrad_ws, S->wgrad_ws_bytes, wg));
    }
    // the predictors' backward (~50 small launches on the side stream) is enqueued now - behind the main stream's first
    // backward launches, in front of the decoder stack's hundred - and runs underneath the decoder's backward
    TRY(ev_wait(S->ev[EV_LOSS], sd, "loss gradients"));
    TRY(predictor_bwd(S, &S->pit, S->g_pit, sd));
    TRY(predictor_bwd(S, &S->dur, S->g_dur, sd));
    TRY(ttsmi_add2_f32(S->pit.dbranch, S->dur.dbranch, S->d_branch, (int64_t)Me * d, sd));
    TRY(ev_record(S->ev[EV_PRED_BWD], sd, "predictor gradients"));
    TRY(ttsmi_dense_stack_bwd(S->dec, S->n_dec, S->h1, S->h1_bf, S->d_dec_out));
    defer_stack_ln(S->dec, S->n_dec);
    TRY(ttsmi_add_layernorm_bwd(S->dec[0]->dh, S->x_dec, nullptr, S->dec_ln_g, S->mean1, S->rstd1, S->pe_dec, S->dec_ps, Tm,
                                nullptr, 0.f, 0, S->rate, S->site_dec_ln, S->seed, S->step_dev, 0, S->d_x_dec, nullptr, nullptr,
                                nullptr, nullptr, Md, d, S->ln_ws1, S->ln_ws1_bytes, nullptr, mn));
    t_ln_main.add(S->ln_ws1, S->g_dec_ln_g, S->g_dec_ln_b, S->g_dec_ps, ttsmi_add_layernorm_bwd_nparts(Md), d);
    return TTSMI_OK;
}

static int phase1(const ttsmi_ft_step* S) {
    const int B = S->B, Tp = S->Tp, Tm = S->Tm, d = S->d, Me = B * Tp;
    ttsmi_stream_t mn = S->main_stream, sd = S->side_stream;
    TRY(ttsmi_lenreg_bwd(S->d_x_dec, S->cum, S->d_hp, B, Tp, Tm, d, mn));                             // d(Expand)
    TRY(ttsmi_pitch_embed_bwd(S->d_hp, S->tgt_pitch, S->pit_w, S->pit_b, nullptr, S->g_pit_w, S->g_pit_b, Me, d, S->pit_ws,
                              S->pit_ws_bytes, mn));
    // the encoder output has three consumers: the pitch embedding's residual path and the two predictors
    TRY(ev_wait(S->ev[EV_PRED_BWD], mn, "predictor gradients"));
    TRY(ttsmi_add2_f32(S->d_hp, S->d_branch, S->d_enc_out, (int64_t)Me * d, mn));
    TRY(ttsmi_dense_stack_bwd(S->enc, S->n_enc, S->h0, S->h0_bf, S->d_enc_out));
    defer_stack_ln(S->enc, S->n_enc);
    TRY(ttsmi_add_layernorm_bwd(S->enc[0]->dh, S->x_emb, nullptr, S->enc_ln_g, S->mean0, S->rstd0, S->pe_enc, S->enc_ps, Tp,
                                nullptr, 0.f, 0, S->rate, S->site_enc_ln, S->seed, S->step_dev, 0, S->d_x_emb, nullptr, nullptr,
                                nullptr, nullptr, Me, d, S->ln_ws0, S->ln_ws0_bytes, nullptr, mn));
    t_ln_main.add(S->ln_ws0, S->g_enc_ln_g, S->g_enc_ln_b, S->g_enc_ps, ttsmi_add_layernorm_bwd_nparts(Me), d);
    TRY(ttsmi_embedding_bwd(S->tokens, S->d_x_emb, S->g_emb, Me, S->V, d, mn));
    // LayerNorm parameter gradients: one reduction per stream that produced partials, then the joins
    TRY(t_ln_main.flush(mn));
    TRY(t_ln_side.flush(sd));
    TRY(hand_off(S->ev[EV_SIDE_JOIN], sd, mn, "side join"));
    if (S->wgrad_stream) TRY(hand_off(S->ev[EV_WGRAD_JOIN], S->wgrad_stream, mn, "weight-gradient join"));
    return TTSMI_OK;
}

static int phase2(const ttsmi_ft_step* S) {
    ttsmi_stream_t mn = S->main_stream;
    TTSMI_CHECK_ARG(S->p_flat && S->g_flat && S->m_flat && S->v_flat && S->lr_dev && S->step_rw, "ft_train_step: optimiser state missing");
    TRY(ttsmi_step_increment(S->step_rw, mn));
    TRY(ttsmi_adam_tf(S->p_flat, S->g_flat, S->m_flat, S->v_flat, S->n_flat, S->lr_dev, S->step_rw, S->beta1, S->beta2, S->eps,
                      S->flat_bf16, mn));                               // utils/training_config_manager.py:102-106
    if (S->tr_desc && S->tr_n > 0)
        TRY(ttsmi_cast_transpose_bf16_batched((const ttsmi_transpose_desc*)S->tr_desc, S->tr_n, S->tr_tiles, mn));
    if (S->pack_ahead) {
        // the chain kernels' weight streams of the NEXT step, packed from the shadows just written, on the side stream (one
        // launch, beside the predictors' layout kernels below), so that the next step's start does not have to enqueue
        // them in front of its first kernel (the same blocks, the same chain links: the host clears pack_now for the
        // next step only when that holds)
        TRY(hand_off(S->ev[EV_PACK_START], mn, S->side_stream, "bf16 shadows"));
        TRY(side_pack_stacks(S));
        TRY(ev_record(S->ev[EV_PACK], S->side_stream, "chain weight streams"));
    }
    for (int i = 0; i < S->n_conv_wd; ++i)
        TRY(ttsmi_conv_wdgrad_layout_bf16(S->conv_w[i], S->conv_wd[i], S->conv_k[i], S->conv_cin[i], S->conv_cout[i],
                                          (S->conv_cout[i] + 7) / 8 * 8, mn));
    return TTSMI_OK;
}

extern "C" {

int ttsmi_add2_f32(const float* a, const float* b, float* out, int64_t n, ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(a && b && out && n >= 0, "add2_f32: bad argument");
    TTSMI_CHECK_ARG(((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)out)) & 15) == 0, "add2_f32: operands must be 16-byte aligned");
    if (n == 0) return TTSMI_OK;
    const long n4 = n / 4;
    const int tail = (int)(n - n4 * 4);
    hipLaunchKernelGGL(add2_f32_kernel, dim3(ew_grid(n4)), dim3(256), 0, (hipStream_t)stream, (const float4*)a, (const float4*)b,
                       (float4*)out, n4, a + n4 * 4, b + n4 * 4, out + n4 * 4, tail);
    TTSMI_CHECK_LAUNCH("add2_f32");
    return TTSMI_OK;
}

int ttsmi_pad_cols_f32(const float* src, int C, float* dst, int Cp, int M, ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(src && dst && C > 0 && Cp >= C && M >= 0, "pad_cols_f32: bad argument");
    if (M == 0) return TTSMI_OK;
    const long total = (long)M * Cp;
    hipLaunchKernelGGL(pad_cols_f32_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, src, C, dst, Cp, total);
    TTSMI_CHECK_LAUNCH("pad_cols_f32");
    return TTSMI_OK;
}

int ttsmi_ft_train_step(const ttsmi_ft_step* S, int phase) {
    TRY(check_step(S));
    switch (phase) {
        case 0: return phase0(S);
        case 1: return phase1(S);
        case 2: return phase2(S);
        // a data-parallel host, between phases 0 and 1: the decoder half's LayerNorm gradients are final before its all-reduce
        case 10: return t_ln_main.flush(S->main_stream);
        default:
            ttsmi_set_error("ft_train_step: phase %d (0, 1, 2; 10 = flush the decoder's LayerNorm reductions)", phase);
            return TTSMI_ERR_INVALID_ARG;
    }
}

}  // extern "C"
