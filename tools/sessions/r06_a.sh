#!/bin/bash
# Round 6, first session: (1) which CUs a stream mask selects, (2) parity of the re-built attention kernels (two tile images,
# log-sum-exp folded into the dK/dV accumulator) and of the backward chain's LayerNorm partial count, (3) the attention
# kernels alone, four builds side by side, (4) the train step with the weight-gradient stream confined to whole XCDs.
# (History: the TTSMI_WGRAD_XCDS knob of step 4 and the -DHATTN_* variant builds existed at commit 9c2cbc7 only; the result is
# profiles/r06_wgrad_cu_mask_ab.txt.)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
L=$PWD/transformertts_amd/lib
timeout 120 python tools/probe_cu_mask.py > $O/r06_cu_mask_probe.txt 2>&1; tail -8 $O/r06_cu_mask_probe.txt
timeout 900 python -m pytest tests/test_bench_shapes_gpu.py tests/test_ops_gpu.py tests/test_chain_gpu.py -q -m gpu -p no:cacheprovider -x -k "attention or attn or layernorm_parameter_gradients" 2>&1 | tail -5 | tee $O/r06_attn_parity.txt
export TTSMI_ALLOW_LIB_OVERRIDE=1
timeout 900 python tools/kbench.py --only attn --variants base TTSMI_LIB=$L/libttsmi_nodb.so TTSMI_LIB=$L/libttsmi_nofold.so TTSMI_LIB=$L/libttsmi_r05attn.so 2>&1 | tee $O/r06_kbench_attn_ab.txt
unset TTSMI_ALLOW_LIB_OVERRIDE
OUT=$O/r06_wgrad_cu_mask_ab.txt
one() {
  env $1 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-attention-maps --no-also 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1 ms_per_step', round(d['ms_per_step'],3), 'host', round(d.get('host_issue_ms_per_step') or 0,3))" | tee -a $OUT
}
one A=1; one TTSMI_WGRAD_XCDS=2; one TTSMI_WGRAD_XCDS=3; one TTSMI_WGRAD_XCDS=4; one A=1; one TTSMI_WGRAD_XCDS=5-7
TTSMI_ALLOW_LIB_OVERRIDE=1 TTSMI_LIB=$L/libttsmi_r05attn.so python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-attention-maps --no-also 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('r05 attention kernels ms_per_step', round(d['ms_per_step'],3))" | tee -a $OUT
