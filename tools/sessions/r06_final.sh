#!/bin/bash
# Round 6 artefact session on the FINAL build: the whole GPU suite twice, the suite under the guard allocator, smoke, the PMC
# passes that stamp the traffic files with this build's digest, kernel trace + timeline, SQ counters, lj-dist / ref-default /
# mel traces, the bench lines (default, --steps 20, mel, predict, lj-dist, per-layer path, no chains), kbench, the curve.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
TAG=r06
for i in 1 2; do
  ( time timeout 1200 python -X faulthandler -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -v "dist-packages\|^Extension modules\|amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8 ) > $O/${TAG}_full_gpu_tests_$i.txt 2>&1
  tail -4 $O/${TAG}_full_gpu_tests_$i.txt
done
export TTSMI_GUARD_LOG=$O/${TAG}_guard_log.txt
: > $TTSMI_GUARD_LOG
( time TTSMI_GUARD_ALLOC=1 timeout 1500 python -X faulthandler -m pytest tests -q -m gpu -p no:cacheprovider -n 1 --max-worker-restart 12 -rs 2>&1 \
    | grep -v "dist-packages\|^Extension modules\|amdgpu.ids" | tail -40 ) > $O/${TAG}_guard_suite.txt 2>&1
tail -6 $O/${TAG}_guard_suite.txt; grep -c CANARY $TTSMI_GUARD_LOG
unset TTSMI_GUARD_LOG
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/${TAG}_smoke.txt
echo "== train step: kernel trace + PMC passes"
bash tools/gpu_profile.sh ${TAG}_bf16 --no-attention-maps
cp $O/${TAG}_bf16_pmc_traffic.json $R/profiles/${TAG}_pmc_hbm_traffic_bf16.json
python tools/rocpd_timeline.py $O/prof_${TAG}_bf16/trace_results.db --steps 1 --top 50 --gaps > $O/${TAG}_timeline_bf16.txt 2>&1
echo "== train step: SQ counters"
bash tools/gpu_sq_counters.sh ${TAG}_bf16 --no-attention-maps
python tools/rocpd_sq_summary.py $O/sq_${TAG}_bf16_1/pmc_results.db $O/sq_${TAG}_bf16_2/pmc_results.db > $O/${TAG}_sq_counters_bf16.txt 2>&1
echo "== mel: kernel trace + PMC passes"
MARGS="--workload mel --steps 2 --warmup 1 --no-cpu-baseline --no-roofline"
( cd /tmp
  timeout 280 rocprofv3 --kernel-trace --stats -d $O/prof_${TAG}_mel -o trace -- python $R/bench.py $MARGS > $O/prof_${TAG}_mel.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 280 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_${TAG}_mel_$c -o pmc -- python $R/bench.py $MARGS > /dev/null 2>&1; echo mel $c rc=$?
  done )
python tools/rocpd_kernel_stats.py $O/prof_${TAG}_mel/trace_results.db $O/${TAG}_mel_kernel_stats.csv
python tools/rocpd_pmc_traffic.py $O/pmc_${TAG}_mel_FETCH_SIZE/pmc_results.db $O/pmc_${TAG}_mel_WRITE_SIZE/pmc_results.db $O/${TAG}_mel_pmc_traffic.json
cp $O/${TAG}_mel_pmc_traffic.json $R/profiles/${TAG}_pmc_hbm_traffic_mel.json
echo "== reference-default architecture and lj-dist: kernel traces"
( cd /tmp; timeout 280 rocprofv3 --kernel-trace --stats -d $O/prof_${TAG}_refdef -o trace -- python $R/bench.py --workload ref-default --steps 3 --warmup 3 --no-cpu-baseline --no-roofline --no-attention-maps > /dev/null 2>&1 )
python tools/rocpd_kernel_stats.py $O/prof_${TAG}_refdef/trace_results.db $O/${TAG}_refdefault_kernel_stats.csv
( cd /tmp; timeout 400 rocprofv3 --kernel-trace -d $O/prof_${TAG}_lj -o trace -- python $R/bench.py --workload lj-dist --steps 40 --warmup 10 --lj-skip-max-shape --no-cpu-baseline --no-roofline --no-attention-maps --no-also > /dev/null 2>&1 )
python tools/rocpd_kernel_stats.py $O/prof_${TAG}_lj/trace_results.db $O/${TAG}_ljdist_kernel_stats.csv
python tools/rocpd_timeline.py $O/prof_${TAG}_lj/trace_results.db --steps 24 --top 60 > $O/${TAG}_timeline_ljdist.txt 2>&1
echo "== bench lines (the PMC files just copied into profiles/ feed the traffic fields)"
python bench.py > $O/${TAG}_bench_bf16.json 2> $O/${TAG}_bench_bf16.err; echo rc=$?
python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_bf16_steps20.json 2>/dev/null; echo rc=$?
python bench.py --workload mel > $O/${TAG}_bench_mel.json 2>/dev/null; echo rc=$?
python bench.py --workload predict > $O/${TAG}_predict_latency.json 2>/dev/null; echo rc=$?
python bench.py --workload lj-dist > $O/${TAG}_bench_ljdist.json 2>/dev/null; echo rc=$?
TTSMI_CSTEP=0 python bench.py --workload lj-dist > $O/${TAG}_bench_ljdist_per_layer_path.json 2>/dev/null; echo rc=$?
TTSMI_CSTEP=0 python bench.py --no-cpu-baseline --no-attention-maps --no-also > $O/${TAG}_bench_bf16_per_layer_path.json 2>/dev/null; echo rc=$?
TTSMI_DENSE_CHAIN=0 python bench.py --no-cpu-baseline --no-attention-maps --no-also > $O/${TAG}_bench_bf16_nochain.json 2>/dev/null; echo rc=$?
python tools/kbench.py --only attn > $O/${TAG}_kbench_attn.txt 2>&1
python tools/debug/cstep_host.py 2>&1 | grep -v amdgpu.ids | head -8 > $O/${TAG}_cstep_host.txt
( timeout 120 python tools/probe_chain_split_two_streams.py 6400 50; echo "rc=$?"; timeout 120 python tools/probe_chain_split_two_streams.py 2500 50; echo "rc=$?" ) 2>&1 | grep -v amdgpu.ids > $O/${TAG}_chain_split_two_streams.txt
echo "== the training curve at the benchmarked batch"
TTSMI_CURVE_BATCH=32 timeout 600 python -m pytest tests/test_training_curve_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -2
cp $O/bf16_vs_f32_curve.json $O/${TAG}_bf16_vs_f32_curve_b32.json 2>/dev/null
rm -rf $O/prof_${TAG}_bf16 $O/pmc_${TAG}_bf16_* $O/sq_${TAG}_bf16_* $O/prof_${TAG}_mel $O/pmc_${TAG}_mel_* $O/prof_${TAG}_refdef $O/prof_${TAG}_lj
python - <<'PY'
import json
for f in ('bf16', 'bf16_steps20', 'mel', 'ljdist', 'ljdist_per_layer_path', 'bf16_per_layer_path', 'bf16_nochain'):
    try:
        d = json.loads(open('gpurun_out/r06_bench_' + f + '.json').read().strip().splitlines()[-1])
        print(f, round(d['value'], 1), d['unit'], 'ms', round(d['ms_per_step'], 3), 'host', d.get('host_issue_ms_per_step'), 'burst', d.get('host_issue_burst_ms_per_step'), 'traffic', (d.get('roofline') or {}).get('traffic'))
        if d.get('summary'): print('   summary', d['summary'])
    except Exception as e:
        print(f, 'ERR', e)
PY
ls $O | grep r06_ | wc -l
