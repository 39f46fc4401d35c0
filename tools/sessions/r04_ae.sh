#!/bin/bash
# round 4, session ae: 256 x 256-tile GEMM for the reference-default conv stacks - tests, the GEMMs alone, the workload
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bench_shapes_gpu.py -x -q -m gpu -k "256_tile or persistent_dma" 2>&1 | tail -5 > gpurun_out/r04ae_tests.txt
cat gpurun_out/r04ae_tests.txt
: > gpurun_out/r04ae_ab.txt
for t in 0 1; do
  echo "TTSMI_HGEMM_T256=$t" >> gpurun_out/r04ae_ab.txt
  TTSMI_HGEMM_T256=$t timeout 300 python tools/debug/conv_gemm_ab.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04ae_ab.txt
done
for t in 0 1 0 1; do
  TTSMI_HGEMM_T256=$t timeout 600 python bench.py --workload ref-default --steps 15 --warmup 3 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null \
    | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ref-default t256', $t, 'ms_per_step', round(d['ms_per_step'], 3), 'host', round(d.get('host_issue_ms_per_step', 0), 3), 'loss', d['config'].get('loss_after'))" | tee -a gpurun_out/r04ae_ab.txt
done
cat gpurun_out/r04ae_ab.txt
