#!/usr/bin/env python
"""Does the step get faster when the MAIN stream outranks the weight-gradient stream?  (The step without weight gradients is
4.64 ms against 5.13 ms with them - tools/sessions/r04_o.sh - and they run on a second stream whose 50 us kernels fill the
chip with resident workgroups.)  Runs the configs[1] train step on a stream of priority MAIN (argv[1], default: torch's
default stream) with TTSMI_WGRAD_PRIO from the environment; prints ms per step."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from transformertts_amd.model.models import ForwardTransformer  # noqa: E402
from transformertts_amd.utils.synthetic import synthetic_batch  # noqa: E402


def main():
    prio = sys.argv[1] if len(sys.argv) > 1 else 'default'
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    dev = torch.device('cuda', 0)
    cfg, shape = bench.workload_config('configs[1]')
    cfg = dict(cfg, dropout_rate=0.1, predictors_dropout=0.1, device=str(dev), seed=0, precision='bf16', use_graph=False)
    model = ForwardTransformer.from_config(cfg)
    model._compile(learning_rate=1e-4)
    tok, mel, dur, pit = synthetic_batch(shape['B'], shape['Tp'], shape['Tm'], seed=1234)
    batch = [torch.from_numpy(a).to(dev) for a in (tok, mel, dur, pit)]
    rng = getattr(torch.cuda.Stream, 'priority_range', lambda: None)()
    stream = torch.cuda.current_stream() if prio == 'default' else torch.cuda.Stream(priority=int(prio))
    with torch.cuda.stream(stream):
        for _ in range(5):
            out = model.train_step(*batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = model.train_step(*batch)
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    print(f'main priority {prio} (stream priority {stream.priority}, range {rng}), TTSMI_WGRAD_PRIO={os.environ.get("TTSMI_WGRAD_PRIO", "0")}: '
          f'{1e3 * el / steps:.3f} ms/step, host issue {1e3 * host / steps:.3f} ms, loss {float(out["loss"]):.5f}')


if __name__ == '__main__':
    main()
