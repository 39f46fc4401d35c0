#!/usr/bin/env python
"""Where does the host's issue time of a train step go: inside the C-ABI calls (the HIP runtime: launches, events, stream
waits) or in Python around them?  Every C-ABI call is timed on the host through _lib.set_trace (both threads: the forward
runs on the caller's thread, the backward on the autograd engine's); printed per entry point: calls per step, host
microseconds per call and per step.  The step itself is timed without the hook first."""
import collections
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from transformertts_amd import _lib  # noqa: E402
from transformertts_amd.model.models import ForwardTransformer  # noqa: E402
from transformertts_amd.utils.synthetic import synthetic_batch  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    B = int(sys.argv[2]) if len(sys.argv) > 2 else None
    dev = torch.device('cuda', 0)
    cfg, shape = bench.workload_config('configs[1]')
    if B:
        shape = dict(shape, B=B)
    cfg = dict(cfg, dropout_rate=0.1, predictors_dropout=0.1, device=str(dev), seed=0, precision='bf16', use_graph=False)
    model = ForwardTransformer.from_config(cfg)
    model._compile(learning_rate=1e-4)
    tok, mel, dur, pit = synthetic_batch(shape['B'], shape['Tp'], shape['Tm'], seed=1234)
    batch = [torch.from_numpy(a).to(dev) for a in (tok, mel, dur, pit)]

    def run(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            model.train_step(*batch)
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        return 1e3 * host / n, 1e3 * (time.perf_counter() - t0) / n

    # host time of the phases of a step (wrappers around the model's own methods; nested ones are listed as they are)
    phase = collections.defaultdict(float)

    def timed(obj, name, label=None):
        fn = getattr(obj, name)

        def w(*a, **k):
            t0 = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                phase[label or name] += time.perf_counter() - t0
        setattr(obj, name, w)
    from transformertts_amd import ops
    for nm in ('_prep', '_launch_dropmasks', 'call', '_call_front', '_call_back', '_losses', '_apply_gradients', '_join_predictors'):
        timed(model, nm)
    timed(ops, 'ln_flush')
    timed(ops, 'wgrad_join')
    timed(torch.Tensor, 'backward', 'loss.backward()')
    run(5)
    phase.clear()
    host, wall = run(steps)
    print('host ms per step by phase: ' + ', '.join(f'{k} {1e3 * v / steps:.3f}' for k, v in sorted(phase.items(), key=lambda kv: -kv[1])))
    print(f'batch {shape["B"]}: host issue {host:.3f} ms per step, wall {wall:.3f} ms per step (no hook)')
    acc = collections.defaultdict(lambda: [0, 0.0])
    by_thread = collections.defaultdict(float)
    main_id = threading.get_ident()

    def hook(name, args, fn):
        t0 = time.perf_counter()
        rc = fn(*args)
        dt = time.perf_counter() - t0
        a = acc[name]
        a[0] += 1
        a[1] += dt
        by_thread['forward thread' if threading.get_ident() == main_id else 'autograd thread'] += dt
        return rc
    _lib.set_trace(hook)
    host_h, wall_h = run(steps)
    _lib.set_trace(None)
    tot = sum(v[1] for v in acc.values())
    print(f'with the hook: host issue {host_h:.3f} ms per step; inside C-ABI calls {1e3 * tot / steps:.3f} ms per step '
          f'({sum(v[0] for v in acc.values()) / steps:.0f} calls), ' + ', '.join(f'{k} {1e3 * v / steps:.3f} ms' for k, v in by_thread.items()))
    for name, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:24]:
        print(f'  {name:44s} {n / steps:6.1f} calls/step  {1e6 * t / n:8.1f} us/call  {1e3 * t / steps:7.3f} ms/step')


if __name__ == '__main__':
    main()
