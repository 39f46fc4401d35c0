#!/bin/bash
# Round 6: C-step parity (all cases), kernel table + timeline of the lj-dist workload and of the headline on the C step
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_cstep_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -15 | tee $O/r06_cstep_parity.txt
cd /tmp
timeout 400 rocprofv3 --kernel-trace -d $O/prof_lj -o trace -- python $R/bench.py --workload lj-dist --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --no-attention-maps --no-also > $O/prof_lj.log 2>&1
echo "lj trace rc=$?"; tail -2 $O/prof_lj.log | cut -c1-600
python $R/tools/rocpd_kernel_stats.py $O/prof_lj/trace_results.db $O/r06_ljdist_kernel_stats.csv
head -45 $O/r06_ljdist_kernel_stats.csv
python $R/tools/rocpd_timeline.py $O/prof_lj/trace_results.db --steps 2 --top 12 --gaps > $O/r06_timeline_ljdist.txt 2>&1; head -30 $O/r06_timeline_ljdist.txt
rm -rf $O/prof_lj
timeout 280 rocprofv3 --kernel-trace -d $O/prof_hd -o trace -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-roofline --no-attention-maps --no-also > $O/prof_hd.log 2>&1
echo "headline trace rc=$?"
python $R/tools/rocpd_timeline.py $O/prof_hd/trace_results.db --steps 2 --top 50 --gaps > $O/r06_timeline_bf16_cstep.txt 2>&1; head -24 $O/r06_timeline_bf16_cstep.txt
rm -rf $O/prof_hd
