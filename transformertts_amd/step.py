"""ForwardTransformer._train_step as ONE descriptor issued from C++ (include/ttsmi.h: ttsmi_ft_step / ttsmi_ft_train_step).

The reference compiles its step once (model/models.py:442-451: tf.function + input signature) and its loop only feeds batches
(train_tts.py:149-160).  The per-layer host path of model/models.py + ops.py drives ~107 C-ABI calls per step through autograd
nodes - ~1.4 ms of interpreter / autograd-engine time that bounds the step at the reference's bucketed batch sizes (~12 k rows).
`TrainStepPlan` owns every activation / gradient buffer of the step (sized for the largest batch seen, like
ops.DenseBlockPlan), fills `ttsmi_ft_step` once per batch shape and runs a step as three C calls (phases 0 / 1 / 2).  The
launches, their arguments and their streams are those of the per-layer path: results are bit-identical
(tests/test_cstep_gpu.py), which stays the path for everything this one does not cover (exact-fp32, conv blocks, attention
maps returned, hipGraph replay, activation taps).

Outputs (`mel`, `duration`, `pitch`, `expanded_lengths`, the losses) are views of persistent buffers: a ring of
`OUT_RING` sets, so a step's outputs stay valid until OUT_RING - 1 more steps have run (the loss scalars: LOSS_RING steps)."""
from __future__ import annotations

import ctypes
import os

import torch

from . import _lib, ops
from ._lib import check

CSTEP = os.environ.get('TTSMI_CSTEP', '1') != '0'          # 0: the per-layer autograd path everywhere (A/B knob)
OUT_RING = 2
LOSS_RING = 256


def _p(t):
    return None if t is None else t.data_ptr()


class _LazyOut(dict):
    """The train step's output dict; `expanded_mask` (the float [B,1,1,Tm] mask the reference returns, models.py:541) is
    built from the uint8 padding flags only when somebody asks for it - one framework launch less per step."""

    def __init__(self, pad_d, *a, **kw):
        super().__init__(*a, **kw)
        self._pad_d = pad_d

    def __missing__(self, key):
        if key == 'expanded_mask':
            v = self._pad_d.to(torch.float32)[:, None, None, :]
            self[key] = v
            return v
        raise KeyError(key)

    def __contains__(self, key):
        return key == 'expanded_mask' or super().__contains__(key)

    def keys(self):
        self['expanded_mask']
        return super().keys()

    def items(self):
        self['expanded_mask']
        return super().items()


def eligible(model) -> bool:
    """The configurations the C++ step covers: bf16 precision, every block of both stacks a planned dense block with fused
    LayerNorms, teacher-forced predictors on their side stream, bf16 conv operands for the predictors."""
    c = model.config
    if not (CSTEP and model.precision == 'bf16' and model.planned_blocks and model.fused_blocks and model.overlap_predictors
            and model.shadow_set is not None):
        return False
    d = c['encoder_model_dimension']
    for prefix, heads, nd in (('enc', c['encoder_num_heads'], c['encoder_dense_blocks']),
                              ('dec', c['decoder_num_heads'], c['decoder_dense_blocks'])):
        if not heads or nd < len(heads) or len(heads) > _lib.FT_MAX_BLOCKS:
            return False
        if not all(model._plan_ok(f'{prefix}.blk{i}', H, d) for i, H in enumerate(heads)):
            return False
    for prefix, filters in (('dur', c['duration_conv_filters']), ('pitch', c['pitch_conv_filters'])):
        if not filters or len(filters) > _lib.FT_MAX_PRED_LAYERS:
            return False
        cin = d
        for j, f in enumerate(filters):
            if cin % 8 != 0 or model.shadow.get(f'{prefix}.conv{j}.w') is None:
                return False
            cin = f
    return 'out.w' in model.shadow and d % 8 == 0 and model.mel_channels % 8 == 0


class TrainStepPlan:
    def __init__(self, model):
        self.m = model
        self.dev = model.device
        self.S = _lib.FtStep()
        self._sref = ctypes.byref(self.S)
        self.cap = (0, 0, 0)                       # (batch, encoder rows, decoder rows)
        self.shape = None
        self.t = {}
        self.events = [torch.cuda.Event() for _ in range(12)]
        for ev in self.events:
            ev.record()                            # materialises the hipEvent_t
        self.loss_ring = torch.zeros((LOSS_RING, 4), dtype=torch.float32, device=self.dev)
        self.keep = None
        self.plans_e, self.plans_d = [], []
        self._wg_ptr = None
        self._packed_version = None                # the weights version the chain kernels' streams were packed ahead for (phase 2)
        self._static()

    # ------------------------------------------------------------------ descriptor parts that never change
    def _static(self):
        m, S = self.m, self.S
        c, W, G, SH = m.config, m.params.w, m.params.g, m.shadow
        d = c['encoder_model_dimension']
        S.d, S.V, S.n_mel = d, m.vocab_size, m.mel_channels
        S.n_enc, S.n_dec = len(c['encoder_num_heads']), len(c['decoder_num_heads'])
        S.seed, S.step_dev = m.drop.seed, _p(m.step_dev)
        for i, ev in enumerate(self.events):
            S.ev[i] = ev.cuda_event
        n_dur, n_pit = len(c['duration_conv_filters']), len(c['pitch_conv_filters'])
        # dropout sites in call()'s order: entry LayerNorm, three per block, one per predictor layer (models.py: _call_front)
        S.site_enc_ln = 1
        self.site_enc_blk = 2
        site = 1 + 3 * S.n_enc
        self.site_dur, self.site_pit = site + 1, site + 1 + n_dur
        S.site_dec_ln = site + n_dur + n_pit + 1
        self.site_dec_blk = S.site_dec_ln + 1
        for k, name in (('emb', 'embedding'), ('enc_ln_g', 'enc.ln.gamma'), ('enc_ln_b', 'enc.ln.beta'), ('enc_ps', 'enc.pos_scalar'),
                        ('dec_ln_g', 'dec.ln.gamma'), ('dec_ln_b', 'dec.ln.beta'), ('dec_ps', 'dec.pos_scalar'),
                        ('pit_w', 'pitch_embed.w'), ('pit_b', 'pitch_embed.b'), ('out_b', 'out.b')):
            setattr(S, k, _p(W[name]))
            setattr(S, 'g_' + k if k != 'out_b' else 'g_out_b', _p(G[name]))
        S.g_out_w = _p(G['out.w'])
        S.out_wt, S.out_wb = _p(SH['out.w'].wt), _p(SH['out.w'].wb)
        S.pe_enc, S.pe_dec = _p(m.pe_enc), _p(m.pe_dec)
        for P, prefix, filters, ksz, relu, site0 in ((S.dur, 'dur', c['duration_conv_filters'], c['duration_kernel_size'], 1, self.site_dur),
                                                     (S.pit, 'pitch', c['pitch_conv_filters'], c['pitch_kernel_size'], 0, self.site_pit)):
            P.n_layers, P.relu_head = len(filters), relu
            cin = d
            for j, f in enumerate(filters):
                L = P.layer[j]
                L.k, L.Cin, L.Cout, L.Cout_pad, L.site = int(ksz), cin, int(f), (int(f) + 7) // 8 * 8, site0 + j
                sh = SH[f'{prefix}.conv{j}.w']
                L.w_t, L.w_d = _p(sh.wt), _p(sh.wd)
                L.bias, L.g_w, L.g_b = _p(W[f'{prefix}.conv{j}.b']), _p(G[f'{prefix}.conv{j}.w']), _p(G[f'{prefix}.conv{j}.b'])
                L.ln_g, L.ln_b = _p(W[f'{prefix}.ln{j}.gamma']), _p(W[f'{prefix}.ln{j}.beta'])
                L.g_ln_g, L.g_ln_b = _p(G[f'{prefix}.ln{j}.gamma']), _p(G[f'{prefix}.ln{j}.beta'])
                cin = int(f)
            P.lin_w, P.lin_b = _p(W[f'{prefix}.lin.w']), _p(W[f'{prefix}.lin.b'])
            P.g_lin_w, P.g_lin_b = _p(G[f'{prefix}.lin.w']), _p(G[f'{prefix}.lin.b'])
        # phase 2: optimiser + bf16 shadows
        Pm, ss = m.params, m.shadow_set
        S.p_flat, S.g_flat, S.m_flat, S.v_flat, S.n_flat = _p(Pm.data), _p(Pm.grad), _p(Pm.m), _p(Pm.v), Pm.data.numel()
        S.lr_dev, S.step_rw = _p(m.lr_dev), _p(m.step_dev)
        S.flat_bf16 = _p(ss.flat_bf16)
        S.tr_desc, S.tr_n, S.tr_tiles = _p(ss.desc), ss.n_desc, ss.total_tiles
        assert len(ss._conv) <= 2 * _lib.FT_MAX_PRED_LAYERS
        S.n_conv_wd = len(ss._conv)
        for i, (w, wd) in enumerate(ss._conv):
            k, cin, cout = w.shape
            S.conv_w[i], S.conv_wd[i], S.conv_k[i], S.conv_cin[i], S.conv_cout[i] = _p(w), _p(wd), k, cin, cout

    # ------------------------------------------------------------------ buffers, sized for a capacity
    def _alloc(self, B, Me, Md):
        """(Re)allocate every buffer for capacities >= (B, Me, Md); geometric growth, like the block plans."""
        cb, ce, cd = self.cap
        grow = lambda cur, need: cur if need <= cur else (need if cur == 0 else max(need, (cur * 3 + 1) // 2))
        nb, ne, nd = grow(cb, B), grow(ce, Me), grow(cd, Md)
        if (nb, ne, nd) == self.cap:
            return
        if self.cap != (0, 0, 0):
            torch.cuda.synchronize()           # nothing in flight reads the buffers that are dropped now
        self.cap = (nb, ne, nd)
        m, S, dev, l = self.m, self.S, self.dev, _lib.lib()
        d, n_mel = S.d, S.n_mel
        f32, bf, i32, u8 = torch.float32, torch.bfloat16, torch.int32, torch.uint8
        e = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)
        t = self.t = {}
        for name, shape, dt in (('pad_e', (ne,), u8), ('klen_e', (nb,), i32), ('pad_d', (nd,), u8), ('klen_d', (nb,), i32),
                                ('x_emb', (ne, d), f32), ('h0', (ne, d), f32), ('h0_bf', (ne, d), bf), ('mean0', (ne,), f32),
                                ('rstd0', (ne,), f32), ('hp', (ne, d), f32), ('idx', (nd,), i32), ('cum', (ne + nb,), i32),
                                ('x_dec', (nd, d), f32), ('h1', (nd, d), f32), ('h1_bf', (nd, d), bf), ('mean1', (nd,), f32),
                                ('rstd1', (nd,), f32), ('g_mel', (nd, n_mel), f32), ('g_dur', (ne,), f32), ('g_pit', (ne,), f32),
                                ('d_dec_out', (nd, d), f32), ('d_x_dec', (nd, d), f32), ('d_hp', (ne, d), f32),
                                ('d_branch', (ne, d), f32), ('d_enc_out', (ne, d), f32), ('d_x_emb', (ne, d), f32)):
            t[name] = e(shape, dt)
        t['mel'] = [e((nd, n_mel), f32) for _ in range(OUT_RING)]
        t['lens'] = [e((nb,), i32) for _ in range(OUT_RING)]
        t['loss_ws'] = ops._ws(l.ttsmi_l1_losses_weighted_ws_bytes(3), dev)
        t['ln_ws0'] = ops._ws(l.ttsmi_add_layernorm_bwd_ws_bytes(ne, d), dev)
        t['ln_ws1'] = ops._ws(l.ttsmi_add_layernorm_bwd_ws_bytes(nd, d), dev)
        t['pit_ws'] = ops._ws(l.ttsmi_pitch_embed_bwd_ws_bytes(ne, d), dev)
        for k in ('pad_e', 'klen_e', 'pad_d', 'klen_d', 'x_emb', 'h0', 'h0_bf', 'mean0', 'rstd0', 'hp', 'idx', 'cum', 'x_dec', 'h1',
                  'h1_bf', 'mean1', 'rstd1', 'g_mel', 'g_dur', 'g_pit', 'd_dec_out', 'd_x_dec', 'd_hp', 'd_branch', 'd_enc_out',
                  'd_x_emb', 'loss_ws', 'ln_ws0', 'ln_ws1', 'pit_ws'):
            setattr(S, k, _p(t[k]))
        S.loss_ws_bytes, S.ln_ws0_bytes, S.ln_ws1_bytes = t['loss_ws'].numel(), t['ln_ws0'].numel(), t['ln_ws1'].numel()
        S.pit_ws_bytes = t['pit_ws'].numel()
        ldt = (ne + 7) // 8 * 8
        for P, key in ((S.dur, 'dur'), (S.pit, 'pit')):
            pt = t[key] = {}
            pt['hm'], pt['dbranch'] = e((ne, d), f32), e((ne, d), f32)
            pt['y'] = [e((ne,), f32) for _ in range(OUT_RING)]
            last = P.layer[P.n_layers - 1]
            pt['dn'] = e((ne, last.Cout), f32)
            pt['rd_ws'] = ops._ws(l.ttsmi_rowdot_bwd_ws_bytes(ne, last.Cout), dev)
            P.hm, P.dbranch, P.dn, P.rd_ws, P.rd_ws_bytes = _p(pt['hm']), _p(pt['dbranch']), _p(pt['dn']), _p(pt['rd_ws']), pt['rd_ws'].numel()
            for j in range(P.n_layers):
                L = P.layer[j]
                lt = pt[j] = {}
                for name, shape in (('c', (ne, L.Cout)), ('n', (ne, L.Cout)), ('mean', (ne,)), ('rstd', (ne,)), ('dc', (ne, L.Cout)),
                                    ('dx', (ne, L.Cin))):
                    lt[name] = e(shape, f32)
                    setattr(L, name, _p(lt[name]))
                lt['dc_pad'] = e((ne, L.Cout_pad), f32) if L.Cout_pad != L.Cout else None
                L.dc_pad = _p(lt['dc_pad'])
                lt['ln_ws'] = ops._ws(l.ttsmi_add_layernorm_bwd_ws_bytes(ne, L.Cout), dev)
                L.ln_ws, L.ln_ws_bytes = _p(lt['ln_ws']), lt['ln_ws'].numel()
                kin = L.k * L.Cin
                if L.Cin % 128 == 0 and L.Cout % 4 == 0:
                    lt['wg_ws'] = ops._ws(l.ttsmi_hgemm_wgrad_rows_ws_bytes(ne, kin, L.Cout), dev)
                    L.xT = L.dyT = None
                else:
                    lt['xT'], lt['dyT'] = e((kin, ldt), bf), e((L.Cout, ldt), bf)
                    L.xT, L.dyT = _p(lt['xT']), _p(lt['dyT'])
                    lt['wg_ws'] = ops._ws(l.ttsmi_hgemm_wgrad_ws_bytes(ne, kin, L.Cout), dev)
                L.wg_ws, L.wg_ws_bytes = _p(lt['wg_ws']), lt['wg_ws'].numel()
        self.shape = None                          # every pointer moved: re-bind

    # ------------------------------------------------------------------ per batch shape
    def _bind_stack(self, prefix, heads, B, T, pad, klen, rate, site0):
        """model._self_attention_blocks' plan wiring (chain links, per-step descriptor fields) for one stack."""
        m = self.m
        l = _lib.lib()
        plans, below = [], None
        n = len(heads)
        dmask_on = rate > 0 and ops._ATTN_DROPBITS
        for i, H in enumerate(heads):
            p = f'{prefix}.blk{i}'
            plan = m._block_plan(p, prefix, B, H, T)
            if below is not None:
                below.chain_above(plan if m.chain_ln else None)
            below = plan
            sites = (site0 + 3 * i, site0 + 3 * i + 1, site0 + 3 * i + 2)
            dmask = None
            if dmask_on:
                need = int(l.ttsmi_attention_dropmask_bytes(B, H, T))
                buf = m._dropmask_bufs.get(p)
                if buf is None or buf.numel() < need:
                    buf = m._dropmask_bufs[p] = torch.empty(need, dtype=torch.uint8, device=self.dev)
                dmask = buf
            last = i == n - 1
            plan.chain_forward(None if last else m._block_plan(f'{prefix}.blk{i + 1}', prefix, B, heads[i + 1], T))
            plan.bind(pad, klen, rate, m.drop, sites, dmask, res16=m.residual_bf16, out32=last)
            plan.bound_by = self                       # (the per-layer path's bind() resets it: its masks are per-step tensors)
            plan.name = p
            plans.append(plan)
        below.chain_above(None)
        return plans

    def _retarget(self, B, Tp, Tm) -> bool:
        """A new batch shape on plans that were bound before, when nothing but the shape changes: every plan has the capacity,
        and each stack stays on the same side of the chain kernels' row threshold (so the chain links, the packed weight
        streams, the ReLU bit layout and every pointer stay as they are).  With the reference's bucketed batches almost
        every step brings a new (B, Tp, Tm): the full re-bind (12 x rebind / chain_above / chain_forward / bind /
        _prepare_bwd) was ~1 ms of interpreter time per step."""
        S, t = self.S, self.t
        if self.shape is None or self.shape[3] != ops._stream() or not self.plans_e:
            return False
        for plans, T in ((self.plans_e, Tp), (self.plans_d, Tm)):
            M = B * T
            for pl in plans:
                if M > pl.cap or (pl.chain and pl.res16 and (M >= ops.CHAIN_MIN_ROWS) != pl.chain_on):
                    return False
            # the keep-bit tables are sized for the largest shape seen
            need = int(_lib.lib().ttsmi_attention_dropmask_bytes(B, plans[0].H, T)) if plans[0].desc.dropmask else 0
            for pl in plans:
                if pl.desc.dropmask and (pl.H != plans[0].H or self.m._dropmask_bufs[pl.name].numel() < need):
                    return False
        for plans, T in ((self.plans_e, Tp), (self.plans_d, Tm)):
            for pl in plans:
                pl.B, pl.T, pl.M = 0, 0, B * T     # (0, 0: whoever binds the plan next - the per-layer path - re-derives everything)
                pl.desc.B, pl.desc.T = B, T
        S.B, S.Tp, S.Tm = B, Tp, Tm
        self.shape = (B, Tp, Tm, self.shape[3])
        return True

    def _bind(self, B, Tp, Tm):
        m, S, t = self.m, self.S, self.t
        c = m.config
        S.B, S.Tp, S.Tm = B, Tp, Tm
        S.rate, S.prate = float(c['dropout_rate']), float(c['predictors_dropout'])
        if Tp > m.pe_enc.shape[0] or Tm > m.pe_dec.shape[0]:
            raise ValueError(f'sequence lengths ({Tp}, {Tm}) exceed the positional-encoding tables '
                             f'({m.pe_enc.shape[0]}, {m.pe_dec.shape[0]} positions)')
        main = ops.cur_stream()
        if m._pred_stream is None:
            m._pred_stream = torch.cuda.Stream(device=self.dev, priority=int(os.environ.get('TTSMI_PRED_PRIO', '0')))
        S.main_stream, S.side_stream = main.cuda_stream, m._pred_stream.cuda_stream
        pad_e, klen_e = t['pad_e'][:B * Tp].view(B, Tp), t['klen_e'][:B]
        pad_d, klen_d = t['pad_d'][:B * Tm].view(B, Tm), t['klen_d'][:B]
        self.plans_e = self._bind_stack('enc', c['encoder_num_heads'], B, Tp, pad_e, klen_e, S.rate, self.site_enc_blk)
        self.plans_d = self._bind_stack('dec', c['decoder_num_heads'], B, Tm, pad_d, klen_d, S.rate, self.site_dec_blk)
        for i, pl in enumerate(self.plans_e):
            S.enc[i] = ctypes.addressof(pl.desc)
        for i, pl in enumerate(self.plans_d):
            S.dec[i] = ctypes.addressof(pl.desc)
        self._bind_wgrad()
        self._packed_version = None                # new chain links / buffers: phase 0 packs
        self.shape = (B, Tp, Tm, main.cuda_stream)     # (not S.main_stream: a c_void_p field reads the null stream back as None)

    def _bind_wgrad(self):
        """The weight-gradient stream / workspace fields of every block descriptor and of the step (ops.DenseBlockPlan._prepare_bwd)."""
        m, S = self.m, self.S
        plans = self.plans_e + self.plans_d
        need = max(pl.wgrad_need for pl in plans)
        ops.enable_wgrad_stream(m.overlap_wgrad)
        try:
            for pl in plans:
                pl._prepare_bwd(self.dev, need)
        finally:
            ops.enable_wgrad_stream(False)
        if m.overlap_wgrad:
            W = ops._WgradStream.cur()
            S.wgrad_stream, S.wgrad_ws, S.wgrad_ws_bytes = W.handle, W.ws.data_ptr(), W.ws.numel()
            self._wg_ptr = (W.ws.data_ptr(), W.ws.numel())
            W.pending = False                      # the step joins the stream itself (phase 1)
        else:
            l = _lib.lib()
            ws = self.t.get('own_wgrad_ws')
            nb = int(l.ttsmi_hgemm_wgrad_rows_ws_bytes(S.B * S.Tm, S.d, S.n_mel))
            if ws is None or ws.numel() < nb:
                ws = self.t['own_wgrad_ws'] = ops._ws(nb, self.dev)
            S.wgrad_stream, S.wgrad_ws, S.wgrad_ws_bytes = None, ws.data_ptr(), ws.numel()
            self._wg_ptr = None

    # ------------------------------------------------------------------ one step
    def run(self, x, ts, td, tp, phases=(0, 1, 2)):
        """x int32 [B,Tp], ts f32 [B,Tm,n_mel], td int32 [B,Tp,1], tp f32 [B,Tp,1] (ForwardTransformer._prep's outputs).
        phases: (0, 1, 2) = the whole step; a data-parallel caller runs (0,), its hook, (1,), its all-reduce, (2,)."""
        m, S, l = self.m, self.S, _lib.lib()
        if 0 in phases:
            B, Tp = int(x.shape[0]), int(x.shape[1])
            Tm = int(ts.shape[1])
            assert ts.shape[0] == B and ts.shape[2] == S.n_mel and td.numel() == B * Tp and tp.numel() == B * Tp
            self._alloc(B, B * Tp, B * Tm)
            main_h = ops._stream()
            if self.shape is not None and any(pl.bound_by is not self for pl in self.plans_e + self.plans_d):
                self.shape = None                  # the per-layer path ran on these plans in between and re-bound them
            if self.shape != (B, Tp, Tm, main_h):
                if not self._retarget(B, Tp, Tm):
                    self._bind(B, Tp, Tm)
            elif self._wg_ptr is not None:
                W = ops._WgradStream.cur()
                if W.ws is None or (W.ws.data_ptr(), W.ws.numel()) != self._wg_ptr:
                    self._bind_wgrad()             # somebody else grew the shared workspace
            t = self.t
            r = m._host_step % OUT_RING
            S.tokens, S.tgt_mel, S.tgt_dur, S.tgt_pitch = _p(x), _p(ts), _p(td), _p(tp)
            S.mel, S.lens = _p(t['mel'][r]), _p(t['lens'][r])
            S.dur.y, S.pit.y = _p(t['dur']['y'][r]), _p(t['pit']['y'][r])
            lo = self.loss_ring[m._host_step % LOSS_RING]
            S.loss_out = _p(lo)
            S.loss_w[0], S.loss_w[1], S.loss_w[2] = [float(w) for w in m.loss_weights]
            den = m.loss_denominators or (0, 0, 0)
            S.loss_denom[0], S.loss_denom[1], S.loss_denom[2] = [int(n) for n in den]
            S.beta1, S.beta2, S.eps = float(m.beta_1), float(m.beta_2), float(m.epsilon)    # (_compile may change them)
            # the chain kernels' weight streams: packed by the previous step's phase 2 (same binding, same weights), else now
            chained = any(pl.chain_on for pl in self.plans_e + self.plans_d)
            S.pack_now = int(chained and self._packed_version != m._weights_version)
            S.pack_ahead = int(chained)
            self.keep = (x, ts, td, tp)            # alive until the next step's inputs replace them
            check(l.ttsmi_ft_train_step(self._sref, 0), 'ft_train_step(0)')
            for pl in self.plans_e + self.plans_d:
                if pl.chain_on:
                    pl.packed_ver, pl.pack_ev = None, None     # (the per-layer path, should it run next, packs for itself)
            pad_d = t['pad_d'][:B * Tm].view(B, Tm)
            self.out = _LazyOut(pad_d, {
                'mel': t['mel'][r][:B * Tm].view(B, Tm, S.n_mel), 'duration': t['dur']['y'][r][:B * Tp].view(B, Tp, 1),
                'pitch': t['pit']['y'][r][:B * Tp].view(B, Tp, 1), 'encoder_attention': {}, 'decoder_attention': {},
                'expanded_lengths': t['lens'][r][:B], 'loss': lo[3],
                'losses': {'mel': lo[0], 'duration': lo[1], 'pitch': lo[2]}})
        if 1 in phases:
            check(l.ttsmi_ft_train_step(self._sref, 1), 'ft_train_step(1)')
        if 2 in phases:
            check(l.ttsmi_ft_train_step(self._sref, 2), 'ft_train_step(2)')
            m._weights_version += 1                # (ForwardTransformer._refresh_shadows' bookkeeping)
            self._packed_version = m._weights_version if S.pack_ahead else None
        return self.out

    def flush_decoder_ln(self):
        """Data parallel, between phases 0 and 1: the decoder half's LayerNorm parameter gradients are final (LenRegFn.backward's
        ln_flush before the hook)."""
        check(_lib.lib().ttsmi_ft_train_step(self._sref, 10), 'ft_train_step(10)')
