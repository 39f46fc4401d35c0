#!/bin/bash
# round 4, session v: the one-pass attention backward with the rotated visiting order (tests, isolated timings, step A/B)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bench_shapes_gpu.py -x -q -m gpu -k "keep_bit_attention or one_pass" 2>&1 | tail -4 > gpurun_out/r04v_tests.txt
cat gpurun_out/r04v_tests.txt
timeout 600 python tools/kbench.py --only attn 2>&1 | grep -E "attn   bwd" | head -12 > gpurun_out/r04v_kbench_attn.txt
cat gpurun_out/r04v_kbench_attn.txt
for f in 0 1 0 1; do
  TTSMI_ATTN_FUSED_BWD=$f timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null \
    | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused_bwd', $f, 'ms_per_step', round(d['ms_per_step'], 3), 'host', round(d.get('host_issue_ms_per_step', 0), 3), 'loss', d['config'].get('loss_after'))" \
    | tee -a gpurun_out/r04v_step_ab.txt
done
