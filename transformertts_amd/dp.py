"""Batch data-parallel training over one 8xMI355X node: one process per GPU, ONE collective per
step - an all-reduce of the flat fp32 gradient buffer (RCCL over xGMI; torch.distributed backend
"nccl" is RCCL on ROCm).  The reference has no distributed code at all; this
is the only parallelism the path needs: samples are independent, parameters
and Adam state are replicated, every rank applies the identical fused Adam step.

Loss normalisation: each rank's L1 losses are means over ITS [B_local, T_max, C] block.  With equal
per-rank shapes (how the trainer forms a global batch: one bucket, split evenly) the average of the
per-rank gradients equals the gradient of the global-batch mean, so the all-reduce uses AVG (SUM
followed by 1/world on backends without AVG, e.g. gloo in the CPU tests).

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
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


class GradAllReduce:
    """Callable handed to ForwardTransformer.grad_sync: averages the flat gradient buffer in place."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.use_avg = dist.is_initialized() and dist.get_backend(group) == 'nccl'

    def __call__(self, flat_grad: torch.Tensor) -> None:
        if self.world == 1:
            return
        if self.use_avg:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.AVG, group=self.group)
        else:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)
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
        model.grad_sync = self.sync if self.sync.world > 1 else None
        if broadcast:
            broadcast_parameters(model.params.data, 0, group)

    def __getattr__(self, name):
        return getattr(self.model, name)

    def train_step(self, *args, **kw):
        return self.model.train_step(*args, **kw)
