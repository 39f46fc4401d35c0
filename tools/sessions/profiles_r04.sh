#!/bin/bash
# The round-4 artefact set for profiles/: bash tools/sessions/profiles_r04.sh [tag]   (ONE gpurun call, one box)
# (the probes it runs are in-tree binaries: bash tools/probes/build.sh first; the measurement library: tools/build_variant.sh abl "-DTTSMI_ABLATION_BUILD" dense_block.hip attention_bf16.hip)
cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
echo "== train step: kernel trace + PMC passes"
bash tools/gpu_profile.sh ${TAG}_bf16 --no-attention-maps
cp $O/${TAG}_bf16_pmc_traffic.json $R/profiles/${TAG}_pmc_hbm_traffic_bf16.json
python tools/rocpd_timeline.py $O/prof_${TAG}_bf16/trace_results.db --steps 1 --top 50 > $O/${TAG}_timeline_bf16.txt 2>&1
# the raw kernel-trace database of this session, compressed (round-3 review: summaries alone cannot be re-reduced)
gzip -9 -c $O/prof_${TAG}_bf16/trace_results.db > $O/${TAG}_trace_bf16.db.gz; ls -la $O/${TAG}_trace_bf16.db.gz
echo "== train step: SQ counters"
bash tools/gpu_sq_counters.sh ${TAG}_bf16 --no-attention-maps
python tools/rocpd_sq_summary.py $O/sq_${TAG}_bf16_1/pmc_results.db $O/sq_${TAG}_bf16_2/pmc_results.db > $O/${TAG}_sq_counters_bf16.txt 2>&1
echo "== mel: kernel trace + PMC passes (BASELINE configs[3])"
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
python tools/rocpd_timeline.py $O/prof_${TAG}_refdef/trace_results.db --steps 1 --top 40 > $O/${TAG}_refdefault_timeline.txt 2>&1
echo "== bench lines (the committed PMC files feed the traffic fields)"
python bench.py > $O/${TAG}_bench_bf16.json 2> $O/${TAG}_bench_bf16.err; echo rc=$?
python bench.py --workload mel > $O/${TAG}_bench_mel.json 2>/dev/null; echo rc=$?
python bench.py --precision f32 --no-attention-maps > $O/${TAG}_bench_f32.json 2>/dev/null; echo rc=$?
python bench.py --workload predict > $O/${TAG}_predict_latency.json 2>/dev/null; echo rc=$?
python bench.py --workload ref-default --no-cpu-baseline > $O/${TAG}_bench_refdefault.json 2>/dev/null; echo rc=$?
python bench.py --workload lj-dist > $O/${TAG}_bench_ljdist.json 2>/dev/null; echo rc=$?
python bench.py --workload lj-dist --lj-preload > $O/${TAG}_bench_ljdist_preloaded.json 2>/dev/null; echo rc=$?
python bench.py --graph --no-cpu-baseline --no-roofline --no-attention-maps --steps 30 --warmup 6 > $O/${TAG}_bench_graph.json 2>/dev/null; echo rc=$?
python tools/kbench.py --json $O/${TAG}_kbench.jsonl > $O/${TAG}_kbench.txt 2>&1
echo "== the one-pass attention backward: hand-off probe, timeline + stage ablation (measurement build), step A/B"
tools/probes/xcd_sem_probe > $O/${TAG}_xcd_sem_probe.txt 2>&1
( export TTSMI_ALLOW_LIB_OVERRIDE=1 TTSMI_LIB=$R/transformertts_amd/lib/libttsmi_abl.so
  for a in 0 1 3 5 7; do echo "== TTSMI_ATTN_FUSED_ABLATE=$a"; TTSMI_ATTN_FUSED_ABLATE=$a python tools/debug/fused_bwd_timeline.py 2>&1 | grep -v amdgpu.ids; done ) > $O/${TAG}_fused_bwd_timeline.txt 2>&1
run() { env $1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), round(d['host_issue_ms_per_step'],3))"; }
( for i in 1 2; do run A=1; run TTSMI_ATTN_FUSED_BWD=1; run TTSMI_DENSE_STACK=0; run TTSMI_WGRAD_WO_DUAL=0; done ) > $O/${TAG}_step_ab.txt 2>&1
echo "== probes: what paces the kernels"
tools/probes/stream_tile_probe > $O/${TAG}_stream_tile_probe.txt 2>&1
tools/probes/wgrad_probe > $O/${TAG}_wgrad_probe.txt 2>&1
tools/probes/valu_rate_probe > $O/${TAG}_valu_rates.txt 2>&1
python tools/debug/launch_cost.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_launch_cost.txt
python tools/debug/host_split.py 30 2>&1 | grep -v amdgpu.ids > $O/${TAG}_host_split.txt
python tools/debug/conv_gemm_ab.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_conv_gemm.txt
cp $O/bf16_vs_f32_curve.json $O/${TAG}_bf16_vs_f32_curve.json 2>/dev/null
rm -rf $O/prof_${TAG}_bf16 $O/pmc_${TAG}_bf16_* $O/sq_${TAG}_bf16_* $O/prof_${TAG}_mel $O/pmc_${TAG}_mel_* $O/prof_${TAG}_refdef
ls -la $O | tail -34
