#!/bin/bash
# Round 6: where the lj-dist step's GPU time goes (ragged steps only)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --kernel-trace -d $O/prof_lj -o trace -- python $R/bench.py --workload lj-dist --steps 40 --warmup 10 --lj-skip-max-shape --no-cpu-baseline --no-roofline --no-attention-maps --no-also > $O/prof_lj.log 2>&1
echo "lj trace rc=$?"; tail -1 $O/prof_lj.log | cut -c1-300
python $R/tools/rocpd_kernel_stats.py $O/prof_lj/trace_results.db $O/r06_ljdist_kernel_stats.csv
python $R/tools/rocpd_timeline.py $O/prof_lj/trace_results.db --steps 24 --top 60 --gaps > $O/r06_timeline_ljdist.txt 2>&1
grep -E "^--- step|queue" $O/r06_timeline_ljdist.txt | head -60
sed -n '/--- per kernel/,$p' $O/r06_timeline_ljdist.txt
rm -rf $O/prof_lj
cd $R; python bench.py --workload lj-dist --no-cpu-baseline --no-roofline --no-attention-maps --no-also 2>/dev/null | tail -1 | cut -c1-1500 | tee $O/r06_bench_ljdist_cstep.json
