#!/bin/bash
# The per-round artefact set for profiles/: bash tools/sessions/profiles.sh r02
cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
echo "== kernel trace + PMC passes"
bash tools/gpu_profile.sh ${TAG}_bf16 --no-attention-maps
echo "== SQ counters"
bash tools/gpu_sq_counters.sh ${TAG}_bf16 --no-attention-maps
python tools/rocpd_sq_summary.py $O/sq_${TAG}_bf16_1/pmc_results.db $O/sq_${TAG}_bf16_2/pmc_results.db > $O/${TAG}_sq_counters_bf16.txt 2>&1
echo "== bench (default command, needs profiles/${TAG}_pmc_hbm_traffic_bf16.json for the traffic field: copied in first)"
cp $O/${TAG}_bf16_pmc_traffic.json $R/profiles/${TAG}_pmc_hbm_traffic_bf16.json
python bench.py > $O/${TAG}_bench_bf16.json 2> $O/${TAG}_bench_bf16.err; echo rc=$?
python bench.py --precision f32 --no-attention-maps > $O/${TAG}_bench_f32.json 2>/dev/null; echo rc=$?
python bench.py --workload predict > $O/${TAG}_predict_latency.json 2>/dev/null; echo rc=$?
python bench.py --workload ref-default --no-cpu-baseline > $O/${TAG}_bench_refdefault.json 2>/dev/null; echo rc=$?
python tools/kbench.py --json $O/${TAG}_kbench.jsonl > $O/${TAG}_kbench.txt 2>&1
rm -rf $O/prof_${TAG}_bf16 $O/pmc_${TAG}_bf16_* $O/sq_${TAG}_bf16_*
ls -la $O | tail -20
