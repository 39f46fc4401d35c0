"""Which framework (aten) operators still launch kernels inside a train step, and from where?

Every kernel of the hot path should be one of libttsmi's; whatever torch launches on its own (fills, scalar glue,
copies, casts) is host-driven serial work between them.  Runs one benchmark-shape step under torch.profiler with Python
stacks and prints every aten operator that owns device time: name, input shapes, device microseconds, innermost
repository frame.

    python tools/probe_glue.py [--workload configs[1]] > gpurun_out/glue.txt
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))

import torch
from torch.profiler import ProfilerActivity, profile


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='configs[1]')
    args = ap.parse_args()
    from bench import workload_config
    from transformertts_amd.model.models import ForwardTransformer
    from transformertts_amd.utils.synthetic import synthetic_batch
    cfg, shape = workload_config(args.workload)
    cfg = dict(cfg, dropout_rate=0.1, predictors_dropout=0.1, device='cuda:0', seed=0, precision='bf16')
    model = ForwardTransformer.from_config(cfg)
    model._compile(learning_rate=1e-4)
    batch = [torch.from_numpy(a).cuda() for a in synthetic_batch(shape['B'], shape['Tp'], shape['Tm'], seed=1234)]
    for _ in range(4):
        model.train_step(*batch)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        model.train_step(*batch)
        torch.cuda.synchronize()
    rows = []
    for e in prof.events():
        if not e.name.startswith('aten::'):
            continue
        dev = sum(k.duration for k in e.kernels) if e.kernels else 0
        if dev <= 0 or any(c.kernels for c in (e.cpu_children or [])):       # leaves only (aten::zeros -> aten::fill_)
            continue
        frame = next((s for s in (e.stack or []) if '/transformertts_amd/' in s or 'bench.py' in s), '?')
        rows.append((e.time_range.start, e.name, str(e.input_shapes)[:60], dev, frame.strip()[-90:]))
    rows.sort()
    print(f'{len(rows)} aten operators with device time in one step, {sum(r[3] for r in rows):.0f} us in total')
    for _, name, shapes, dev, frame in rows:
        print(f'  {dev:7.1f} us  {name:28s} {shapes:60s} {frame}')


if __name__ == '__main__':
    main()
