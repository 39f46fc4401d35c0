#!/usr/bin/env python
"""Per-kernel sums of the SQ counters collected by tools/gpu_sq_counters.sh.
Usage: python tools/rocpd_sq_summary.py <pmc_results.db> [...more dbs] [--filter substr]"""
import re
import sqlite3
import sys


def main():
    dbs = [a for a in sys.argv[1:] if a.endswith('.db')]
    filt = sys.argv[sys.argv.index('--filter') + 1] if '--filter' in sys.argv else ''
    agg = {}
    for path in dbs:
        db = sqlite3.connect(path)
        for k, c, v in db.execute('select kernel_name, counter_name, value from counters_collection'):
            k = re.sub(r'\(.*', '', k).replace('void ', '')[:44]
            if filt and filt not in k:
                continue
            a = agg.setdefault(k, {})
            a[c] = a.get(c, 0.0) + v
            a['_n_' + c] = a.get('_n_' + c, 0) + 1
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0)):
        n = max(v for c, v in a.items() if c.startswith('_n_'))
        print(f'== {k}  (launches {n})')
        wc = a.get('SQ_WAVE_CYCLES', 0) or 1
        for c in sorted(x for x in a if not x.startswith('_n_')):
            print(f'   {c:28s} {a[c] / n:14.0f} /launch   {a[c] / wc:7.3f} of WAVE_CYCLES')


if __name__ == '__main__':
    main()
