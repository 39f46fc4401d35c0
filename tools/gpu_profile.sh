#!/bin/bash
# Collect the per-round profile set on the GPU box: kernel trace + the two PMC passes (separately, as
# the MI355X guide prescribes).  Usage (inside gpurun): bash tools/gpu_profile.sh <tag> [bench args]
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 2 --warmup 2 --no-cpu-baseline --no-roofline $*"
timeout 280 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o trace -- python $R/bench.py $ARGS > $R/gpurun_out/prof_$TAG.log 2>&1
echo trace rc=$?
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 280 rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmc_${TAG}_$c -o pmc -- python $R/bench.py $ARGS > /dev/null 2>&1
  echo $c rc=$?
done
cd $R
python tools/rocpd_kernel_stats.py gpurun_out/prof_$TAG/trace_results.db gpurun_out/${TAG}_kernel_stats.csv
python tools/rocpd_pmc_traffic.py gpurun_out/pmc_${TAG}_FETCH_SIZE/pmc_results.db gpurun_out/pmc_${TAG}_WRITE_SIZE/pmc_results.db gpurun_out/${TAG}_pmc_traffic.json
