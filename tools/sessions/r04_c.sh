#!/bin/bash
# round 4, session C: the ticket form of the one-pass attention backward (parity tests with full output kept, kernel
# timings, step A/B), the training-curve test at the reference's learning rate.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
echo "== fused attention backward tests"
timeout 600 python -m pytest tests/test_bench_shapes_gpu.py -x -q -m gpu -k "one_pass or keep_bit" > $O/r04c_fused_tests.txt 2>&1; echo rc=$?
head -60 $O/r04c_fused_tests.txt | cut -c1-200; echo ...; tail -12 $O/r04c_fused_tests.txt | cut -c1-300
echo "== kbench attn"
timeout 300 python tools/kbench.py --only attn 2>&1 | grep -v amdgpu.ids | grep "900    64\|200    64\|kernel" | tee $O/r04c_kbench_attn.txt
echo "== step A/B"
run() { env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), d['config']['loss_after'])"; }
for i in 1 2 3; do run TTSMI_ATTN_FUSED_BWD=0; run TTSMI_ATTN_FUSED_BWD=1; done
echo "== model-level tests with the one-pass backward on (default)"
timeout 900 python -m pytest tests/test_config1_parity_gpu.py tests/test_training_curve_gpu.py tests/test_model_gpu.py -q -m gpu -k "b32 or curve or bit_reproducible or dropout_matches or variable_batch" > $O/r04c_model_tests.txt 2>&1; echo rc=$?
tail -15 $O/r04c_model_tests.txt | cut -c1-400
python -c "
import json; d=json.load(open('gpurun_out/bf16_vs_f32_curve.json')); print({k:v for k,v in d.items() if 'curve' not in k})"
