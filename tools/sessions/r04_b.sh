#!/bin/bash
# round 4, session B: the XCD hand-off probe, the one-pass attention backward (parity tests, kernel timings, step A/B),
# the training-curve test on the learnable batch, and a diagnosis of the lj-dist workload (kernel trace + host profile).
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
echo "== xcd_sem_probe"; timeout 120 tools/probes/xcd_sem_probe 2>&1 | tee $O/r04b_xcd_sem_probe.txt
echo "== fused attention backward tests"
timeout 600 python -m pytest tests/test_bench_shapes_gpu.py -x -q -m gpu -k "one_pass or keep_bit" 2>&1 | tail -15
echo "== kbench attn"
timeout 300 python tools/kbench.py --only attn 2>&1 | grep -v amdgpu.ids | tee $O/r04b_kbench_attn.txt
echo "== step A/B"
run() { env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), d['config']['loss_after'])"; }
for i in 1 2; do run TTSMI_ATTN_FUSED_BWD=0; run TTSMI_ATTN_FUSED_BWD=1; run TTSMI_ATTN_FUSED_SC1=1; done
echo "== model-level tests with the one-pass backward on (default)"
timeout 900 python -m pytest tests/test_config1_parity_gpu.py tests/test_training_curve_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "b32 or curve or bit_reproducible or dropout_matches or variable_batch" 2>&1 | tail -8
cat $O/bf16_vs_f32_curve.json; echo
echo "== lj-dist: kernel trace"
( cd /tmp; timeout 280 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof_lj -o trace -- python $OLDPWD/bench.py --workload lj-dist --steps 30 --warmup 10 --lj-samples 1024 > $OLDPWD/$O/r04b_lj_prof.json 2> /dev/null )
python tools/rocpd_kernel_stats.py $O/prof_lj/trace_results.db $O/r04b_lj_kernel_stats.csv; head -25 $O/r04b_lj_kernel_stats.csv
cut -c1-300 $O/r04b_lj_prof.json
echo "== lj-dist: host profile"
timeout 280 python -m cProfile -o /tmp/lj.prof bench.py --workload lj-dist --steps 30 --warmup 10 --lj-samples 1024 > /dev/null 2>&1
python -c "
import pstats; p=pstats.Stats('/tmp/lj.prof'); p.sort_stats('cumulative').print_stats(45)" 2>&1 | tail -60 | cut -c1-180 | tee $O/r04b_lj_host_profile.txt
rm -rf $O/prof_lj
