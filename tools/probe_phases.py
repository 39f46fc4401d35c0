#!/usr/bin/env python
"""Where does the step's time go on the GPU timeline?  Timing events on the main stream at the phase boundaries
(encoder fwd / decoder fwd / loss+backward / Adam), once with the host racing the GPU (as bench.py runs) and once
with the whole step enqueued behind a GPU-side sleep (the host is then out of the picture)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    from transformertts_amd.model.models import ForwardTransformer
    from transformertts_amd.utils.synthetic import synthetic_batch
    cfg, shape = bench.workload_config('configs[1]')
    kw = {}
    for a in sys.argv[1:]:
        k, v = a.split('=')
        kw[k] = (v == '1')
    batch = [torch.from_numpy(a).cuda() for a in synthetic_batch(shape['B'], shape['Tp'], shape['Tm'], seed=1234)]
    m = ForwardTransformer.from_config(dict(cfg, dropout_rate=0.1, predictors_dropout=0.1, device='cuda:0', seed=0,
                                            precision='bf16', **kw))
    m._compile(learning_rate=1e-4)
    for _ in range(5):
        m.train_step(*batch)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); torch.cuda._sleep(10_000_000); e1.record(); torch.cuda.synchronize()
    cyc_per_ms = 10_000_000 / e0.elapsed_time(e1)
    for S in (0.0, float(os.environ.get("PROBE_SLEEP_MS", "9.0"))):
        n = 8
        allev = []
        for _ in range(n):
            if S:
                torch.cuda._sleep(int(S * cyc_per_ms))
            m._phase_events = []
            m.train_step(*batch)
            allev.append(m._phase_events)
        m._phase_events = None
        torch.cuda.synchronize()
        names = [nm for nm, _ in allev[0]]
        print(f'--- sleep {S} ms before each step (phase durations on the main stream, ms, mean of {n - 2} steps)')
        tot = 0.0
        for i in range(1, len(names)):
            d = sum(ev[i - 1][1].elapsed_time(ev[i][1]) for ev in allev[2:]) / (n - 2)
            tot += d
            print(f'   {names[i - 1]:8s} -> {names[i]:8s} {d:7.3f}')
        print(f'   start -> adam total {tot:7.3f}')


if __name__ == '__main__':
    main()
