#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --kernel-trace -d $O/prof_lj -o trace -- python $R/bench.py --workload lj-dist --steps 40 --warmup 10 --lj-skip-max-shape --no-cpu-baseline --no-roofline --no-attention-maps --no-also > $O/prof_lj.log 2>&1
python $R/tools/rocpd_timeline.py $O/prof_lj/trace_results.db --steps 24 --top 40 > $O/r06_timeline_ljdist.txt 2>&1
sed -n '/--- per kernel/,$p' $O/r06_timeline_ljdist.txt | head -45
rm -rf $O/prof_lj
timeout 280 rocprofv3 --kernel-trace -d $O/prof_hd -o trace -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-roofline --no-attention-maps --no-also > $O/prof_hd.log 2>&1
python $R/tools/rocpd_timeline.py $O/prof_hd/trace_results.db --steps 2 --top 40 > $O/r06_timeline_bf16.txt 2>&1
sed -n '/--- per kernel/,$p' $O/r06_timeline_bf16.txt | grep main | head -30; tail -1 $O/r06_timeline_bf16.txt
rm -rf $O/prof_hd
