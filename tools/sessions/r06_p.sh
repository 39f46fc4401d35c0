#!/bin/bash
# Round 6: the headline step's sequence (every kernel, both queues, idle before) with the loader-wave chain forms; chain backward alone
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( for L in 1 0; do echo "== TTSMI_DENSE_CHAIN_LOADERS=$L"; TTSMI_DENSE_CHAIN_LOADERS=$L timeout 200 python tools/bench_chain_bwd.py 6400; TTSMI_DENSE_CHAIN_LOADERS=$L timeout 200 python tools/bench_chain.py 6400; done ) 2>&1 | grep -v amdgpu.ids | tee $O/r06_loaders_alone.txt
cd /tmp
timeout 280 rocprofv3 --kernel-trace -d $O/prof_hd -o trace -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-roofline --no-attention-maps --no-also > $O/prof_hd.log 2>&1
python $R/tools/rocpd_timeline.py $O/prof_hd/trace_results.db --steps 2 --top 40 --gaps --sequence > $O/r06_timeline_bf16.txt 2>&1
rm -rf $O/prof_hd
head -40 $O/r06_timeline_bf16.txt
