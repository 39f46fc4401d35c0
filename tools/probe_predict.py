#!/usr/bin/env python
"""B=1 predict (400 phonemes, ~2280 frames) replayed N times - the thing to put under rocprofv3 --kernel-trace to see
which kernels the batch-1 latency consists of.  Usage: python tools/probe_predict.py [N] [--eager] [--no-plans] [--maps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from transformertts_amd.model.models import ForwardTransformer  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 50
    dev = torch.device('cuda', 0)
    cfg, _ = bench.workload_config('configs[1]')
    model = ForwardTransformer.from_config(dict(cfg, device=str(dev), seed=0, precision='bf16',
                                                graph_inference='--eager' not in sys.argv,
                                                planned_blocks='--no-plans' not in sys.argv))
    model.return_attention = '--maps' in sys.argv
    rng = np.random.default_rng(1234)
    Tp = 400
    tok = torch.from_numpy(rng.integers(1, 127, size=(1, Tp)).astype(np.int32)).to(dev)
    dur = torch.from_numpy(rng.multinomial(int(5.7 * Tp), np.ones(Tp) / Tp, size=1).astype(np.int32)).to(dev)
    for _ in range(4):
        model.predict(tok, encode=False, phoneme_durations=dur)
    torch.cuda.synchronize()
    lat = []
    for _ in range(n):
        t0 = time.perf_counter()
        model.predict(tok, encode=False, phoneme_durations=dur)
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t0)
    lat = np.sort(lat)
    print(f'p50 {1e3 * lat[len(lat) // 2]:.3f} ms, p10 {1e3 * lat[len(lat) // 10]:.3f} ms over {n} calls '
          f'(argv {sys.argv[1:]})')


if __name__ == '__main__':
    main()
