"""Batch data-parallel training over one 8xMI355X node: one process per GPU, ONE collective per
step - an all-reduce of the flat fp32 gradient buffer (RCCL over xGMI; torch.distributed backend
"nccl" is RCCL on ROCm).  The reference has no distributed code at all; this
is the only parallelism the path needs: samples are independent, parameters
and Adam state are replicated, every rank applies the identical fused Adam step.

Loss normalisation (reference utils/losses.py:41-49 is an UNMASKED mean over the
whole padded batch [B, T_max, C]): `DataParallel.train_step` pads every rank's shard to the GLOBAL
(Tp_max, Tm_max) - padded positions carry loss and gradient in the reference - and hands the model the
GLOBAL element counts as the divisors of its three means (`ttsmi_l1_losses_weighted(denom=...)`), so a
rank's loss / gradient is its share of the global-batch mean and the all-reduce is a plain SUM: ragged
shards (different local maxima, different local batch sizes) reproduce the single-device step exactly.
`GradAllReduce` on its own (no global counts known) keeps the older contract: equal-shape shards, SUM
followed by one 1/world scaling pass.

The gradient buffer is a single contiguous tensor, so the collective is one large message: on the
xGMI full mesh RCCL can spread it over all 7 links per GPU (direct reduce-scatter + all-gather)
instead of many per-tensor rings that are bound by one link and by launch latency."""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist


def init_process_group(backend: Optional[str] = None) -> tuple:
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torch.distributed.run).
    Returns (rank, local_rank, world)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend is None:
            # TTSMI_DIST_BACKEND=gloo lets several ranks share one GPU (functional test of the DP path)
            backend = os.environ.get('TTSMI_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


class GradAllReduce:
    """Callable handed to ForwardTransformer.grad_sync: averages the flat gradient buffer in place.

    Two buckets on RCCL: the flat buffer is laid out [embedding | encoder | predictors | pitch_embed |
    decoder | mel-out], and backward produces the decoder half first.  `start_tail(flat_grad, split)` is
    called when backward crosses from the decoder into the encoder (model._lenreg_hook, ops.LenRegFn): it
    launches the all-reduce of `flat_grad[split:]` asynchronously, ordered after everything the main and
    the weight-gradient streams have queued so far, so that it runs over xGMI underneath the encoder's
    backward.  `__call__` then reduces the head `[:split]` and joins the tail.  Every rank issues the two
    collectives in the same order.  TTSMI_DP_OVERLAP=0 falls back to the single all-reduce."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # a one-rank group normally issues no collective; TTSMI_DP_FORCE_COLLECTIVES=1 issues them anyway (sum over one
        # rank = identity), which lets a 1-GPU box exercise the RCCL stream ordering of this file (tools/probe_rccl_world1.py)
        self._forced = dist.is_initialized() and os.environ.get('TTSMI_DP_FORCE_COLLECTIVES') == '1'
        # SUM + one 1/world scaling pass on every backend: ReduceOp.AVG is NCCL-only and not worth a
        # backend-dependent code path (the scaling pass is one 44 MB stream, ~15 us)
        self.use_avg = False
        self.scale = True                   # False: the ranks' gradients are already shares of the global mean (plain SUM)
        self.overlap = dist.is_initialized() and os.environ.get('TTSMI_DP_OVERLAP', '1') != '0'
        self._tail = None                   # (work handle, split) of the in-flight decoder bucket
        self._launch_stream = None

    @property
    def active(self) -> bool:
        """Collectives are issued: more than one rank, or a forced one-rank group."""
        return self.world > 1 or self._forced

    def _reduce(self, t: torch.Tensor, async_op: bool = False):
        """Average over ranks (gloo has no AVG: SUM now, the caller scales once the sum has landed)."""
        op = dist.ReduceOp.AVG if self.use_avg else dist.ReduceOp.SUM
        return dist.all_reduce(t, op=op, group=self.group, async_op=async_op)

    def start_tail(self, flat_grad: torch.Tensor, split: int, side_stream=None) -> None:
        if not self.overlap or not self.active or self._tail is not None or split <= 0 or split >= flat_grad.numel():
            return
        tail = flat_grad[split:]
        if flat_grad.is_cuda:
            main = torch.cuda.current_stream()
            if self._launch_stream is None:
                self._launch_stream = torch.cuda.Stream()
            ls = self._launch_stream
            ls.wait_stream(main)                      # LayerNorm / bias gradients written on the main stream
            if side_stream is not None:
                ls.wait_stream(side_stream)           # weight gradients written on the wgrad stream
            with torch.cuda.stream(ls):               # the collective is ordered after `ls` as of now
                work = self._reduce(tail, async_op=True)
        else:
            work = self._reduce(tail, async_op=True)
        self._tail = (work, split)

    def __call__(self, flat_grad: torch.Tensor) -> None:
        if not self.active:
            return
        if self._tail is not None:
            work, split = self._tail
            self._tail = None
            self._reduce(flat_grad[:split])
            work.wait()                               # (CUDA: the current stream waits for the decoder bucket)
        else:
            self._reduce(flat_grad)
        if not self.use_avg and self.scale:
            flat_grad.mul_(1.0 / self.world)


def broadcast_parameters(flat_params: torch.Tensor, src: int = 0, group=None) -> None:
    """Make every replica start from rank `src`'s weights (one broadcast of the flat buffer)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_params, src=src, group=group)


def shard_batch(arrays, rank: int, world: int):
    """Split a global batch evenly on the batch axis (global batch = world * local batch)."""
    out = []
    for a in arrays:
        n = a.shape[0]
        assert n % world == 0, f'global batch {n} is not divisible by world size {world}'
        per = n // world
        out.append(a[rank * per:(rank + 1) * per])
    return out


class DataParallel:
    """Wraps a ForwardTransformer for batch-DP training.  world_size == 1 is the degenerate case."""

    def __init__(self, model, group=None, broadcast: bool = True):
        self.model = model
        self.sync = GradAllReduce(group)
        model.grad_sync = self.sync if self.sync.active else None
        if broadcast:
            broadcast_parameters(model.params.data, 0, group)
            if getattr(model, 'shadow', None) and hasattr(model, '_refresh_shadows'):
                model._refresh_shadows(False)             # bf16 weight copies follow the broadcast weights
        if self.sync.world > 1 and getattr(model, 'drop', None) is not None:
            # replicas share the weight-init seed but must not share dropout masks: every rank draws its own stream
            rank = dist.get_rank(group)
            model.drop.seed = (model.drop.seed ^ (rank * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF
        if self.sync.active and self.sync.overlap:
            self.install_overlap_hook()

    def install_overlap_hook(self):
        """Start the decoder-half all-reduce when backward crosses into the encoder (see GradAllReduce)."""
        from . import ops
        model = self.model
        dec = [o for n, (o, _) in model.params.offsets.items() if n.startswith('dec.')]
        self.split = min(dec) if dec else 0       # flat layout: [... encoder side ... | dec.* | out.*]

        def _hook():
            if model.grad_sync is not None:
                self.sync.start_tail(model.params.grad, self.split, ops.wgrad_stream())
        model._lenreg_hook = _hook           # on the model, not global: other models in the process are untouched

    def __getattr__(self, name):
        return getattr(self.model, name)

    def global_shape(self, B: int, Tp: int, Tm: int) -> tuple:
        """(sum of the ranks' batch sizes, max phoneme length, max frame count) - one small all-gather.  A trainer that
        forms the batch globally knows these on the host and passes them to train_step instead (no collective, no sync)."""
        if self.sync.world == 1:
            return int(B), int(Tp), int(Tm)
        dev = self.model.device if dist.get_backend(self.sync.group) == 'nccl' else 'cpu'
        mine = torch.tensor([B, Tp, Tm], dtype=torch.int64, device=dev)
        every = [torch.empty_like(mine) for _ in range(self.sync.world)]
        dist.all_gather(every, mine, group=self.sync.group)
        every = torch.stack(every).cpu()
        return int(every[:, 0].sum()), int(every[:, 1].max()), int(every[:, 2].max())

    @staticmethod
    def _pad_to(a, length: int, axis: int = 1):
        n = a.shape[axis]
        if n == length:
            return a
        assert n < length, f'local length {n} exceeds the global maximum {length}'
        if torch.is_tensor(a):
            pad = [0, 0] * (a.dim() - 1 - axis) + [0, length - n]
            return torch.nn.functional.pad(a, pad)
        import numpy as np
        width = [(0, 0)] * a.ndim
        width[axis] = (0, length - n)
        return np.pad(a, width)

    def train_step(self, input_sequence, target_sequence, target_durations, target_pitch, global_shape=None,
                   reduce_losses: bool = True):
        """One data-parallel step on this rank's shard, equal to the single-device step on the concatenated batch.
        global_shape = (B_global, Tp_max_global, Tm_max_global); None -> all-gathered from the ranks' local shapes.
        The returned `loss` / `losses` are the GLOBAL batch values when reduce_losses (one 4-float all-reduce),
        otherwise this rank's share of them."""
        B, Tp = int(input_sequence.shape[0]), int(input_sequence.shape[1])
        Tm = int(target_sequence.shape[1])
        Bg, Tpg, Tmg = global_shape if global_shape is not None else self.global_shape(B, Tp, Tm)
        x = self._pad_to(input_sequence, Tpg)
        ts = self._pad_to(target_sequence, Tmg)
        td = self._pad_to(target_durations, Tpg)
        tp = self._pad_to(target_pitch, Tpg)
        model = self.model
        mel_channels = int(target_sequence.shape[2])
        model.loss_denominators = (Bg * Tmg * mel_channels, Bg * Tpg, Bg * Tpg)
        self.sync.scale = False
        try:
            out = model.train_step(x, ts, td, tp)
        finally:
            model.loss_denominators = None
            self.sync.scale = True
        if reduce_losses and self.sync.active:
            vals = torch.stack([out['loss'].reshape(()), out['losses']['mel'].reshape(()),
                                out['losses']['duration'].reshape(()), out['losses']['pitch'].reshape(())])
            dist.all_reduce(vals, op=dist.ReduceOp.SUM, group=self.sync.group)
            out = dict(out.items(), loss=vals[0], losses={'mel': vals[1], 'duration': vals[2], 'pitch': vals[3]})   # (.items(): a lazily built entry of the C-step's dict is built)
        return out
