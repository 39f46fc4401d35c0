#!/usr/bin/env python
"""Timeline reading of a rocprofv3 --kernel-trace rocpd database of bench.py: where does a train step's WALL time go?

For every complete step (delimited by adam_tf_kernel) and per HIP queue it prints the kernel-busy time (union of the
kernel intervals), the idle time between kernels, the number of launches and the largest gaps - a step whose main
queue shows idle gaps well above the ~1.5 us dependent-launch boundary is waiting for the HOST, not the GPU.  Also
prints the per-kernel totals of the measured steps (main queue) sorted by time.
Usage: python tools/rocpd_timeline.py <trace_results.db> [--steps N] [--top K] [--gaps] [--sequence]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(.*$', '', name)
    return re.sub(r'^void ', '', name).strip()[:60]


def main():
    db = sqlite3.connect(sys.argv[1])
    nsteps = int(sys.argv[sys.argv.index('--steps') + 1]) if '--steps' in sys.argv else 2
    top = int(sys.argv[sys.argv.index('--top') + 1]) if '--top' in sys.argv else 25
    rows = db.execute('select name, start, end, queue_id from kernels order by start').fetchall()
    adam = [i for i, r in enumerate(rows) if 'adam_tf_kernel' in r[0]]
    if len(adam) < nsteps + 1:
        print('not enough complete steps in the trace')
        return
    main_q = rows[adam[0]][3]
    for s in range(nsteps):
        a, b = adam[-(nsteps + 1) + s], adam[-(nsteps + 1) + s + 1]
        t0, t1 = rows[a][2], rows[b][2]                 # end of one Adam -> end of the next
        seg = [r for r in rows if t0 <= r[1] < t1 or t0 < r[2] <= t1]
        print(f'--- step {s}: wall {1e-6 * (t1 - t0):.3f} ms, {len(seg)} kernels')
        for q in sorted({r[3] for r in seg}):
            ks = sorted((max(r[1], t0), min(r[2], t1)) for r in seg if r[3] == q)
            busy, gaps, end = 0, [], ks[0][0]
            gaps.append(ks[0][0] - t0)
            for st, en in ks:
                if st > end:
                    gaps.append(st - end)
                    busy += en - st
                    end = en
                elif en > end:
                    busy += en - end
                    end = en
            big = sorted(gaps, reverse=True)[:5]
            if q == main_q and '--gaps' in sys.argv:
                named = sorted((r for r in seg if r[3] == q), key=lambda r: r[1])
                found = []
                for a_, b_ in zip(named, named[1:]):
                    if b_[1] - a_[2] > 8000:
                        found.append((b_[1] - a_[2], a_[2] - t0, short(a_[0])[:34], short(b_[0])[:34]))
                for g_, at, pa, nb in sorted(found, reverse=True)[:14]:
                    print(f'      gap {g_ / 1e3:7.1f} us at +{at / 1e6:6.3f} ms  after {pa:34s} before {nb}')
            print(f'  queue {q}{" (main)" if q == main_q else ""}: {len(ks)} kernels, busy {1e-6 * busy:.3f} ms, '
                  f'sum of gaps {1e-6 * sum(gaps):.3f} ms, gaps > 5 us: {sum(g > 5000 for g in gaps)} '
                  f'({1e-6 * sum(g for g in gaps if g > 5000):.3f} ms), largest {[round(g / 1e3, 1) for g in big]} us')
    if '--sequence' in sys.argv:                      # every kernel of the last step in start order, both queues
        a, b = adam[-2], adam[-1]
        t0, t1 = rows[a][2], rows[b][2]
        last_end = {}
        print(f'--- sequence of the last step ({1e-6 * (t1 - t0):.3f} ms): start offset us, duration us, idle before (same queue) us')
        for n, st, en, q in rows:
            if t0 <= st < t1:
                gap = st - last_end.get(q, t0)
                print(f'  {"main" if q == main_q else "side"} {1e-3 * (st - t0):9.1f} {1e-3 * (en - st):8.1f} {1e-3 * gap:8.1f}  {short(n)}')
                last_end[q] = max(en, last_end.get(q, t0))
    a, b = adam[-(nsteps + 1)], adam[-1]
    t0, t1 = rows[a][2], rows[b][2]
    agg = {}
    for n, st, en, q in rows:
        if t0 <= st < t1:
            k = (short(n), 'main' if q == main_q else 'side')
            v = agg.setdefault(k, [0, 0])
            v[0] += 1
            v[1] += en - st
    print(f'--- per kernel over {nsteps} steps (per step: launches, total us, average us)')
    for (k, q), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f'  {q:4s} {n / nsteps:7.1f} {t / nsteps / 1e3:9.1f} {t / n / 1e3:8.1f}  {k}')
    tm = sum(t for (k, q), (n, t) in agg.items() if q == 'main') / nsteps / 1e6
    ts = sum(t for (k, q), (n, t) in agg.items() if q == 'side') / nsteps / 1e6
    print(f'  sum of kernel durations per step: main {tm:.3f} ms, side {ts:.3f} ms; launches per step: '
          f'{sum(n for (k, q), (n, t) in agg.items()) / nsteps:.0f}')


if __name__ == '__main__':
    main()
