#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=$R/gpurun_out
export TMPDIR=/tmp
cat > /tmp/maps_step.py <<'PY'
import sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
from transformertts_amd.model.models import ForwardTransformer
from transformertts_amd.utils.synthetic import synthetic_batch
cfg, shape = bench.workload_config('configs[1]')
batch = [torch.from_numpy(a).cuda() for a in synthetic_batch(shape['B'], shape['Tp'], shape['Tm'], seed=1234)]
m = ForwardTransformer.from_config(dict(cfg, dropout_rate=0.1, predictors_dropout=0.1, device='cuda:0', seed=0, precision='bf16', reference_outputs=True))
m._compile(learning_rate=1e-4)
for _ in range(3): out = m.train_step(*batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): out = m.train_step(*batch)
torch.cuda.synchronize()
print('with maps ms/step', (time.perf_counter() - t0) / 5 * 1e3, len(out['decoder_attention']))
PY
python /tmp/maps_step.py 2>&1 | grep -v amdgpu.ids
cd /tmp
timeout 280 rocprofv3 --kernel-trace -d $O/prof_tmp -o trace -- python /tmp/maps_step.py > /dev/null 2>&1
python $R/tools/rocpd_kernel_stats.py $O/prof_tmp/trace_results.db /tmp/ks.csv; head -8 /tmp/ks.csv
rm -rf $O/prof_tmp
