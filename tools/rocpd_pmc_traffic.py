#!/usr/bin/env python
"""Summarise two rocprofv3 PMC passes (one `--pmc FETCH_SIZE`, one `--pmc WRITE_SIZE`, each with
`--kernel-trace` only - TCC has 4 counter slots, FETCH_SIZE costs 3 and WRITE_SIZE 2) into HBM bytes
per launch per kernel, corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes:

  * both counters are reported in KiB;
  * on gfx950 FETCH_SIZE reports exactly half the bytes of a wide (16 B/lane) coalesced streaming
    read, so it is doubled.  Calibrated in this very trace on two kernels whose byte counts are
    known exactly: adam_tf_kernel reads 4 fp32 arrays of n_params and writes 3 + one bf16 copy,
    cast_bf16_kernel reads n_params fp32 and writes n_params bf16 - both match 2*FETCH + WRITE to
    < 0.2 % (WRITE_SIZE needs no correction).  Every hot kernel here loads 16 B/lane.

Usage: python tools/rocpd_pmc_traffic.py <fetch.db> <write.db> <out.json>"""
import json
import os
import re
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def short(name: str) -> str:
    name = re.sub(r'\(.*$', '', name)
    return re.sub(r'^void ', '', name).strip()


def load(path):
    db = sqlite3.connect(path)
    agg = {}
    for k, v in db.execute('select kernel_name, value from counters_collection'):
        a = agg.setdefault(short(k), [0, 0.0])
        a[0] += 1
        a[1] += v
    return agg


def main():
    f, w = load(sys.argv[1]), load(sys.argv[2])
    out = {}
    for k, (n, fv) in f.items():
        wn, wv = w.get(k, (n, 0.0))
        out[k] = {'launches': n,
                  'fetch_size_kib_raw_per_launch': fv / n,
                  'write_size_kib_per_launch': wv / max(wn, 1),
                  'hbm_bytes_per_launch': (2.0 * fv / n + wv / max(wn, 1)) * 1024.0}
    out = dict(sorted(out.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['launches']))
    json.dump(with_totals(out), open(sys.argv[3], 'w'), indent=1)


def with_totals(out):
    """Whole-step totals: the trace holds `steps` train steps (one adam_tf_kernel each).  The runtime's own copy
    kernels (__amd_rocclr_*: the parameter uploads of the model construction - a step issues none) are left out of
    both totals and reported separately; the few one-off initialisation launches of libttsmi kernels (weight shadows)
    stay in - a slight over-estimate."""
    steps = max(1, out.get('adam_tf_kernel', {}).get('launches', 1))
    step_k = {k: v for k, v in out.items() if not k.startswith('__amd_rocclr')}
    total = sum(v['hbm_bytes_per_launch'] * v['launches'] for v in step_k.values())
    launches = sum(v['launches'] for v in step_k.values())
    from transformertts_amd.build import library_digest
    return {'lib_digest': library_digest(),       # the build these bytes were measured on (bench.py: null on mismatch)
            'correction': 'bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950, 16 B/lane loads; '
                          'calibrated on adam_tf_kernel and cast_bf16_kernel in this trace)',
            'steps_in_trace': steps, 'hbm_bytes_per_step': total / steps, 'launches_per_step': launches / steps,
            'runtime_copy_launches_in_trace': sum(v['launches'] for k, v in out.items() if k.startswith('__amd_rocclr')),
            'kernels': out}


if __name__ == '__main__':
    if len(sys.argv) == 3 and sys.argv[1] == '--recompute':      # totals of an existing file from its own kernel table
        d = json.load(open(sys.argv[2]))
        json.dump(with_totals(d['kernels']), open(sys.argv[2], 'w'), indent=1)
    else:
        main()
