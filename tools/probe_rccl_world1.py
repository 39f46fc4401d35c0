"""The batch-DP step on REAL RCCL with what a 1-GPU box offers: a one-rank "nccl" group.

`TTSMI_DP_FORCE_COLLECTIVES=1` makes transformertts_amd.dp issue its collectives for a one-rank group too (a sum over one
rank is the identity), so everything around them runs exactly as on 8 GPUs: the decoder-half all-reduce launched from
the backward hook on the launch stream, ordered after the main and the weight-gradient streams; the head all-reduce and
the join on the main stream; the parameter broadcast; Adam on the reduced buffer.  A missing stream dependency shows up
as a difference from the plain model, which must be bit-identical after every step.

    python tools/probe_rccl_world1.py [--steps 8]        (one JSON line; exit code 1 on a mismatch)
"""
import argparse
import json
import os
import socket
import sys
import time

os.environ['TTSMI_DP_FORCE_COLLECTIVES'] = '1'
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))

import torch
import torch.distributed as dist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--workload', default='configs[1]')
    args = ap.parse_args()
    from bench import workload_config
    from transformertts_amd import dp
    from transformertts_amd.model.models import ForwardTransformer
    from transformertts_amd.utils.synthetic import synthetic_batch

    torch.cuda.set_device(0)
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
    sock.close()
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1)

    cfg, shape = workload_config(args.workload)
    cfg = dict(cfg, dropout_rate=0.1, predictors_dropout=0.1, device='cuda:0', seed=0, precision='bf16')
    batch = [torch.from_numpy(a).cuda() for a in synthetic_batch(shape['B'], shape['Tp'], shape['Tm'], seed=1234)]

    def make(wrap):
        m = ForwardTransformer.from_config(cfg)
        m._compile(learning_rate=1e-4)
        if not wrap:
            return m, m
        w = dp.DataParallel(m)
        assert m.grad_sync is not None and w.sync.active and w.sync.overlap and m._lenreg_hook is not None
        dist.broadcast(m.params.data, src=0)                # what DataParallel does on world > 1
        return m, w

    plain, _ = make(False)
    forced, wrapped = make(True)
    worst, same = 0.0, True
    for step in range(args.steps):
        a = plain.train_step(*batch)
        b = wrapped.train_step(*batch)
        same = same and torch.equal(plain.params.data, forced.params.data) and float(a['loss']) == float(b['loss'])
        worst = max(worst, float((plain.params.data - forced.params.data).abs().max()))

    def timed(step_fn, n=20):
        for _ in range(3):
            step_fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step_fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n

    ms_plain = timed(lambda: plain.train_step(*batch))
    ms_forced = timed(lambda: wrapped.train_step(*batch))
    print(json.dumps({'probe': 'rccl_world1', 'backend': dist.get_backend(), 'steps': args.steps, 'bit_identical': same,
                      'max_abs_param_diff': worst, 'ms_per_step_plain': ms_plain,
                      'ms_per_step_with_collectives': ms_forced,
                      'grad_bytes': int(forced.params.grad.numel() * 4)}))
    dist.destroy_process_group()
    return 0 if same else 1


if __name__ == '__main__':
    sys.exit(main())
