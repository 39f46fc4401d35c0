#!/bin/bash
# Round 6: the whole GPU suite after the removals (one-pass attention backward, first chain form, second weight-gradient lane,
# same-thread backward, CU-mask knob), then the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | tail -12 | tee $O/r06_full_gpu_tests_a.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/r06_smoke.txt
python bench.py > $O/r06_bench_default_a.json 2> $O/r06_bench_default_a.err; tail -c 2500 $O/r06_bench_default_a.json
