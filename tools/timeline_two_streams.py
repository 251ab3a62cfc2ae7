#!/usr/bin/env python3
"""What the GPU does while two decodes are in flight, from a rocprofv3 kernel trace of
`bench.py` (the rocpd database `prof_results.db` of `tools/gpu_visit_stats.sh`, two-stream mode).

Takes the steady intervals from one prefix-beam-search start to the next (5.3-6.0 ms apart) and
reports per decode: how long some non-search kernel is running (the encoder chain), how long
none is, the idle gaps by the kernel that ends them, and every kernel's duration when it runs
entirely under a beam search against when none is running.

    python tools/timeline_two_streams.py gpurun_out/<tag>/prof2/prof_results.db
"""
import collections
import sqlite3
import statistics
import sys


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute('select name, queue_id, stream_id, start, end from kernels '
                      'order by start').fetchall()
    beams = [r for r in rows if 'prefix_beam' in r[0]]
    d = [(beams[i + 1][3] - beams[i][3]) / 1e6 for i in range(len(beams) - 1)]
    # steady two-stream intervals: consecutive beam searches `lo`..`hi` ms apart (argv[2:4];
    # a plain decode() leg in the same trace -- bench.py's roofline pass -- is ~0.7 ms longer)
    lo = float(sys.argv[2]) if len(sys.argv) > 2 else 4.7
    hi = float(sys.argv[3]) if len(sys.argv) > 3 else 5.6
    wins = [(beams[i][3], beams[i + 1][3]) for i in range(len(d)) if lo < d[i] < hi]
    if not wins:
        raise SystemExit(f'no steady two-stream intervals (beam searches {lo}-{hi} ms apart)')
    n = len(wins)
    print(f'{n} steady intervals, {statistics.mean((b - a) / 1e6 for a, b in wins):.3f} ms '
          'per decode')
    tot = collections.Counter()
    gaps = collections.defaultdict(lambda: [0, 0.0])
    dur = collections.defaultdict(lambda: [[], []])
    for t0, t1 in wins:
        rs = [r for r in rows if r[4] > t0 and r[3] < t1]
        enc = [(nm, max(s, t0), min(e, t1), s, e) for nm, _, _, s, e in rs
               if 'prefix_beam' not in nm]
        bm = [(s, e) for nm, _, _, s, e in rs if 'prefix_beam' in nm]
        ev = []
        for _, s, e, _, _ in enc:
            ev += [(s, 1), (e, -1)]
        ev.sort()
        busy = conc = depth = 0
        last = t0
        for t, dd in ev:
            if depth > 0:
                busy += t - last
            if depth > 1:
                conc += t - last
            depth += dd
            last = t
        tot['span'] += t1 - t0
        tot['busy'] += busy
        tot['conc'] += conc
        tot['sum'] += sum(e - s for _, s, e, _, _ in enc)
        tot['beam'] += sum(min(e, t1) - max(s, t0) for s, e in bm)
        cur = t0
        for nm, s, e, _, _ in sorted(enc, key=lambda r: r[1]):
            if s > cur:
                gaps[nm[:64]][0] += 1
                gaps[nm[:64]][1] += s - cur
            cur = max(cur, e)
        for nm, _, _, s0, e0 in enc:
            if s0 >= t0 and e0 <= t1:
                if any(bs <= s0 and e0 <= be for bs, be in bm):
                    dur[nm[:64]][0].append((e0 - s0) / 1e3)
                elif not any(bs < e0 and s0 < be for bs, be in bm):
                    dur[nm[:64]][1].append((e0 - s0) / 1e3)
    print(f'per decode: some non-search kernel running {tot["busy"] / n / 1e6:.3f} ms (two at '
          f'once {tot["conc"] / n / 1e6:.3f}), none {(tot["span"] - tot["busy"]) / n / 1e6:.3f} '
          f'ms; sum of their durations {tot["sum"] / n / 1e6:.3f} ms; beam search '
          f'{tot["beam"] / n / 1e6:.3f} ms')
    print('idle gaps of the non-search chain, by the kernel that ends them (per decode):')
    for k, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:6]:
        print(f'  {k:64s} {c / n:5.1f} x, {t / n / 1e3:6.1f} us')
    print('kernel duration (us): entirely under a beam search | with none running')
    for k, (u, a) in sorted(dur.items(), key=lambda kv: -(sum(kv[1][0]) + sum(kv[1][1])))[:16]:
        mu = f'{len(u):4d} x {statistics.mean(u):8.2f}' if u else '   0 x        -'
        ma = f'{len(a):4d} x {statistics.mean(a):8.2f}' if a else '   0 x        -'
        print(f'  {k:64s} {mu} | {ma}')


if __name__ == '__main__':
    main(sys.argv[1])
