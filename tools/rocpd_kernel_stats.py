#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2) `--kernel-trace --stats` rocpd SQLite database into the
per-kernel table rocprofv3 prints as kernel_stats.csv: name, calls, total/avg/min/max ns, %.
Usage: python tools/rocpd_kernel_stats.py <results.db> [out.csv]"""
import csv
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r'\(.*$', '', name)            # drop the argument list of the demangled name
    name = re.sub(r'^void ', '', name)
    return name.strip()


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info('kernels')")]
    namecol = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
    rows = db.execute(f'select {namecol}, start, end from kernels').fetchall()
    agg = {}
    for name, s, e in rows:
        k = short(name)
        a = agg.setdefault(k, [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values()) or 1
    out = sorted(agg.items(), key=lambda kv: -kv[1][1])
    w = csv.writer(open(sys.argv[2], 'w', newline='') if len(sys.argv) > 2 else sys.stdout)
    w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
    for k, (n, t, mn, mx) in out:
        w.writerow([k, n, t, round(t / n, 1), round(100.0 * t / total, 2), mn, mx])


if __name__ == '__main__':
    main()
