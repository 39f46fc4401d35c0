"""SUPERSEDED - kept only because DESIGN.md section 4 cites it.  This probe queued a GPU-side sleep in front of a step to take
the host out of the picture, but calibrated torch.cuda._sleep at idle clocks: its "host-bound" reading was wrong.
tools/probe_phases.py (timing events on the main stream at the phase boundaries) is the measurement to use."""
import torch, time, sys
sys.path.insert(0, '/root/repo')
import bench
from transformertts_amd.model.models import ForwardTransformer
from transformertts_amd.utils.synthetic import synthetic_batch
cfg, shape = bench.workload_config('configs[1]')
batch = [torch.from_numpy(a).cuda() for a in synthetic_batch(shape['B'], shape['Tp'], shape['Tm'], seed=1234)]
m = ForwardTransformer.from_config(dict(cfg, dropout_rate=0.1, predictors_dropout=0.1, device='cuda:0', seed=0, precision='bf16'))
m._compile(learning_rate=1e-4)
for _ in range(5): m.train_step(*batch)
torch.cuda.synchronize()
# calibrate _sleep
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); torch.cuda._sleep(10_000_000); e1.record(); torch.cuda.synchronize()
cyc_per_ms = 10_000_000 / e0.elapsed_time(e1)
print('sleep cycles per ms', cyc_per_ms)
for S in (0.0, 2.0, 5.0, 8.0):
    n = 15
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        if S: torch.cuda._sleep(int(S * cyc_per_ms))
        m.train_step(*batch)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / n * 1e3
    print(f'sleep {S} ms: wall/step {t:.3f} ms -> step without the sleep {t - S:.3f} ms (host issue {t_host / n * 1e3:.3f} ms/step)')
