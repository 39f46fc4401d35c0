"""Stress of the guard allocator itself (no library kernels): many tensors of random sizes, filled and checked on the
current stream and on a side stream, freed in random order.  Prints mismatches and canary violations - both must be 0;
anything else is a false positive of the allocator (or of the virtual-memory path underneath it), not of a kernel."""
import ctypes
import gc
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'tests', '_guard', 'libttsmi_guard_alloc.so')
torch.cuda.memory.change_current_allocator(torch.cuda.memory.CUDAPluggableAllocator(LIB, 'ttsmi_guard_alloc', 'ttsmi_guard_free'))
lib = ctypes.CDLL(LIB)
lib.ttsmi_guard_violations.restype = ctypes.c_long
rng = random.Random(0)
side = torch.cuda.Stream()
bad = 0
live = []
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 400):
    for _ in range(rng.randint(1, 6)):
        n = rng.choice([1, 3, 6, 24, 31, 64, 100, 1000, 4096, 5000, 100000, 262144, 1000003])
        dt = rng.choice([torch.float32, torch.uint8, torch.int64, torch.bfloat16])
        val = rng.randint(1, 7)
        on_side = rng.random() < 0.3
        if on_side:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                t = torch.full((n,), val, dtype=dt, device='cuda')
            torch.cuda.current_stream().wait_stream(side)
        else:
            t = torch.full((n,), val, dtype=dt, device='cuda')
        live.append((t, n, val))
    rng.shuffle(live)
    while len(live) > 12:
        t, n, val = live.pop()
        got = float(t.double().sum())
        if got != float(n * val):
            bad += 1
            print('MISMATCH', n, t.dtype, val, got)
        del t
gc.collect()
torch.cuda.synchronize()
print('RESULT mismatches', bad, 'violations', lib.ttsmi_guard_violations(), 'allocations', lib.ttsmi_guard_allocations())
