#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
echo "== pytest -m gpu (all)"
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -8
echo "== predict probes"
python tools/probe_predict.py 60 2>/dev/null
python tools/probe_predict.py 60 --no-plans 2>/dev/null
python tools/probe_predict.py 60 --eager 2>/dev/null
TTSMI_HGEMM_BM=64 python tools/probe_predict.py 60 2>/dev/null
echo "== predict kernel trace"
cd /tmp
timeout 280 rocprofv3 --kernel-trace -d $O/prof_tmp -o trace -- python $R/tools/probe_predict.py 100 > $O/prof_tmp.log 2>&1
python $R/tools/rocpd_kernel_stats.py $O/prof_tmp/trace_results.db $O/s12_predict_kernels.csv
head -40 $O/s12_predict_kernels.csv
rm -rf $O/prof_tmp
