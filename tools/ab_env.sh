#!/bin/bash
# A/B of environment knobs on the default bench: bash tools/ab_env.sh "<bench args>" VAR=a VAR=b VAR=a VAR=b ...
# prints one line per run: the setting, ms_per_step, host_issue_ms_per_step, loss_after.  Run alternately (a b a b) -
# consecutive runs on one box differ by a few per cent.
ARGS=$1; shift
for kv in "$@"; do
  env $kv python bench.py --no-cpu-baseline --no-roofline --no-attention-maps $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$kv', round(d['ms_per_step'],3), round(d['host_issue_ms_per_step'],3), d['config'].get('loss_after'))"
done
