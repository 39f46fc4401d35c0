#!/bin/bash
# Round 5 artefact session on the FINAL build: the whole GPU suite twice, the suite under the guard allocator, smoke, the PMC
# passes that stamp the traffic files with this build's digest, kernel trace + timeline, SQ counters, the bench lines.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
TAG=r05
for i in 1 2; do
  ( time timeout 1200 python -X faulthandler -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -v "dist-packages\|^Extension modules\|amdgpu.ids" | tail -12 ) > $O/${TAG}_full_gpu_tests_$i.txt 2>&1
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
echo "== reference-default architecture: kernel trace"
( cd /tmp; timeout 280 rocprofv3 --kernel-trace --stats -d $O/prof_${TAG}_refdef -o trace -- python $R/bench.py --workload ref-default --steps 3 --warmup 3 --no-cpu-baseline --no-roofline --no-attention-maps > /dev/null 2>&1 )
python tools/rocpd_kernel_stats.py $O/prof_${TAG}_refdef/trace_results.db $O/${TAG}_refdefault_kernel_stats.csv
echo "== bench lines (the PMC files just copied into profiles/ feed the traffic fields)"
python bench.py > $O/${TAG}_bench_bf16.json 2> $O/${TAG}_bench_bf16.err; echo rc=$?
python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_bf16_steps20.json 2>/dev/null; echo rc=$?
python bench.py --workload mel > $O/${TAG}_bench_mel.json 2>/dev/null; echo rc=$?
python bench.py --workload predict > $O/${TAG}_predict_latency.json 2>/dev/null; echo rc=$?
python bench.py --graph --no-cpu-baseline --no-roofline --no-attention-maps --steps 30 --warmup 6 > $O/${TAG}_bench_graph.json 2>/dev/null; echo rc=$?
TTSMI_DENSE_CHAIN=0 python bench.py --no-cpu-baseline --no-attention-maps --no-also > $O/${TAG}_bench_bf16_nochain.json 2>/dev/null; echo rc=$?
echo "== the training curve at the benchmarked batch"
TTSMI_CURVE_BATCH=32 timeout 600 python -m pytest tests/test_training_curve_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -2
rm -rf $O/prof_${TAG}_bf16 $O/pmc_${TAG}_bf16_* $O/sq_${TAG}_bf16_* $O/prof_${TAG}_mel $O/pmc_${TAG}_mel_* $O/prof_${TAG}_refdef
python - <<'PY'
import json
for f in ('bf16', 'bf16_steps20', 'mel', 'bf16_nochain'):
    try:
        d = json.loads(open('gpurun_out/r05_bench_' + f + '.json').read().strip().splitlines()[-1])
        print(f, round(d['value'], 1), d['unit'], 'ms', round(d['ms_per_step'], 3), 'maps', d.get('ms_per_step_with_attention_maps'), 'traffic', (d.get('roofline') or {}).get('traffic'))
        for k, v in (d.get('also') or {}).items():
            print('   also', k, {kk: vv for kk, vv in v.items() if kk in ('ms_per_step', 'value', 'error', 'host_issue_ms_per_step')})
    except Exception as e:
        print(f, 'ERR', e)
PY
ls $O | grep r05_ | head -40
