#!/bin/bash
# round 4: after the out-of-bounds fix of the keep-bit table's row pointer (forward / dQ kernels): the attention tests,
# the PMC passes again (the traffic files carry the library digest) and the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_ops_gpu.py tests/test_bench_shapes_gpu.py -q -m gpu -p no:cacheprovider -x -k "keep_bit or attention" 2>&1 | grep -E "passed|failed|rror" | tail -3 | tee $O/r04_restamp_tests.txt
ARGS="--steps 2 --warmup 2 --no-cpu-baseline --no-roofline --no-attention-maps"
MARGS="--workload mel --steps 2 --warmup 1 --no-cpu-baseline --no-roofline"
( cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_r04_bf16_$c -o pmc -- python $R/bench.py $ARGS > /dev/null 2>&1; echo bf16 $c rc=$?
    timeout 200 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_r04_mel_$c -o pmc -- python $R/bench.py $MARGS > /dev/null 2>&1; echo mel $c rc=$?
  done )
python tools/rocpd_pmc_traffic.py $O/pmc_r04_bf16_FETCH_SIZE/pmc_results.db $O/pmc_r04_bf16_WRITE_SIZE/pmc_results.db $O/r04_bf16_pmc_traffic.json
python tools/rocpd_pmc_traffic.py $O/pmc_r04_mel_FETCH_SIZE/pmc_results.db $O/pmc_r04_mel_WRITE_SIZE/pmc_results.db $O/r04_mel_pmc_traffic.json
cp $O/r04_bf16_pmc_traffic.json $R/profiles/r04_pmc_hbm_traffic_bf16.json
cp $O/r04_mel_pmc_traffic.json $R/profiles/r04_pmc_hbm_traffic_mel.json
python bench.py > $O/r04_bench_bf16.json 2> $O/r04_bench_bf16.err; echo rc=$?
python bench.py --workload mel > $O/r04_bench_mel.json 2>/dev/null; echo rc=$?
rm -rf $O/pmc_r04_bf16_* $O/pmc_r04_mel_*
python -c "
import json
for f in ('bf16', 'mel'):
    d = json.loads(open('$O/r04_bench_' + f + '.json').read().strip().splitlines()[-1]); print(f, round(d['value'], 1), d['unit'], 'ms', round(d['ms_per_step'], 3), 'traffic', d['roofline'].get('traffic'))
"
