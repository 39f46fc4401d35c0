import os, sys, cProfile, pstats, torch
sys.path.insert(0, '/root/repo')
import bench
from transformertts_amd.model.models import ForwardTransformer
from transformertts_amd.utils.synthetic import synthetic_batch
cfg, _ = bench.workload_config('configs[1]')
c = dict(cfg, dropout_rate=0.1, predictors_dropout=0.1, device='cuda:0', seed=0, precision='bf16')
m = ForwardTransformer.from_config(c); m._compile(learning_rate=1e-4)
batch = [torch.from_numpy(a).cuda() for a in synthetic_batch(2, 16, 64, seed=1)]
for _ in range(5): m.train_step(*batch)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): m.train_step(*batch)
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(28)
