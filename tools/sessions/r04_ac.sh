#!/bin/bash
# round 4, session ac: full-row GEMM + LN, 64-row tile with a four-stage ring - tests, isolated timings, step and lj-dist A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bench_shapes_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "rowgemm or hgemm_ln or full_row or dense_block" 2>&1 | tail -3 > gpurun_out/r04ac_tests.txt
cat gpurun_out/r04ac_tests.txt
timeout 600 python tools/kbench.py --only rowgemm --variants TTSMI_ROWGEMM_RING4=0 TTSMI_ROWGEMM_RING4=1 2>&1 | grep -E "rowg|variant" > gpurun_out/r04ac_kbench.txt
cat gpurun_out/r04ac_kbench.txt
: > gpurun_out/r04ac_ab.txt
for one in 0 1 0 1; do
  TTSMI_ROWGEMM_RING4=$one timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null \
    | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('configs[1] ring4', $one, 'ms_per_step', round(d['ms_per_step'], 3), 'host', round(d.get('host_issue_ms_per_step', 0), 3), 'loss', d['config'].get('loss_after'))" | tee -a gpurun_out/r04ac_ab.txt
done
for one in 0 1; do
  TTSMI_ROWGEMM_RING4=$one timeout 600 python bench.py --workload lj-dist 2>/dev/null \
    | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lj-dist ring4', $one, 'ms_per_step', round(d['ms_per_step'], 3), 'host', round(d.get('host_issue_ms_per_step', 0), 3), 'real frames/s', round(d['value']))" | tee -a gpurun_out/r04ac_ab.txt
done
