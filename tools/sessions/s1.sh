#!/bin/bash
# GPU session 1 (round 2): baseline on this round's box + the experiments staged in round 1.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out; mkdir -p $O; rm -f $O/config1_parity.jsonl
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest config1 parity"; timeout 900 python -m pytest tests/test_config1_parity_gpu.py -x -q -m gpu -s > $O/s1_parity.log 2>&1; echo rc=$?; tail -3 $O/s1_parity.log
echo "== bench default"; timeout 600 python bench.py > $O/s1_bench.json 2> $O/s1_bench.err; echo rc=$?; cut -c1-400 $O/s1_bench.json
echo "== bench --gpus 2 (self-launch, gloo, ranks share the GPU)"; TTSMI_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-attention-maps > $O/s1_bench_g2.json 2> $O/s1_bench_g2.err; echo rc=$?; cut -c1-300 $O/s1_bench_g2.json; tail -2 $O/s1_bench_g2.err
echo "== kbench gemm variants"; timeout 900 python tools/kbench.py --only gemm --variants base TTSMI_HGEMM_OCC4=1 TTSMI_HGEMM_DMA=1 TTSMI_HGEMM_DMA=2 TTSMI_HGEMM_BM=64 --json $O/s1_kbench.jsonl > $O/s1_kbench_gemm.txt 2>&1; echo rc=$?; cat $O/s1_kbench_gemm.txt
echo "== kbench attn/ln/wgrad"; for k in attn ln wgrad; do timeout 600 python tools/kbench.py --only $k --json $O/s1_kbench.jsonl; done > $O/s1_kbench_rest.txt 2>&1; cat $O/s1_kbench_rest.txt
echo "== bench A/B OCC4"; for v in 0 1; do TTSMI_HGEMM_OCC4=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('OCC4=$v', d['ms_per_step'])"; done
echo "== host overhead"; timeout 300 python tools/host_overhead.py 2>&1 | tail -3
echo "== pytest -m gpu (all)"; timeout 1500 python -m pytest tests -x -q -m gpu > $O/s1_pytest.log 2>&1; echo rc=$?; tail -5 $O/s1_pytest.log
