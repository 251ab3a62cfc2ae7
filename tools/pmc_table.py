#!/usr/bin/env python3
"""Merge the three PMC passes + the kernel trace of tools/gpu_pmc.sh into one
per-kernel table: average duration, MFMA-busy fraction, HBM read / write bytes
per launch and GB/s.

FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3); on gfx950 FETCH_SIZE counts a
wide coalesced read stream at HALF its bytes (MI355X_MICROARCH.md, section HBM):
the read column is the raw counter x 2 x 1024 and says so."""
import collections
import csv
import glob
import json
import os
import sqlite3
import sys

# the roofline kernel of bench.py: the six-product FFN w_1 GEMM (round 2, gemm_x6.hip; before
# it the fused feed-forward kernel 'ffn_fused_kernel<1, 1, 4>'; round 1: the FFN w_1 GEMM
# 'gemm_f32_kernel<128, 128, 2, 4, 1, false, false, false, 32, 1>')
ROOFLINE_KERNEL = 'ffn_x6f_kernel<1, 3, 16896>'


def load(dirname):
    f = glob.glob(os.path.join(dirname, '**', '*counter_collection.csv'), recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if not f:
        return agg
    for r in csv.DictReader(open(f[0])):
        agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    return agg


def merge(rec_path):
    """python tools/pmc_table.py --merge <dir>/pmc_roofline_kernel.json: file the record under
    its key in profiles/pmc_roofline_kernels.json (what bench.py reports as roofline.traffic)."""
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'profiles',
                        'pmc_roofline_kernels.json')
    rec = json.load(open(rec_path))
    table = json.load(open(root)) if os.path.exists(root) else {}
    table[rec.pop('key')] = rec
    with open(root, 'w') as f:
        json.dump(table, f, indent=1, sort_keys=True)
    print('merged into', os.path.normpath(root), sorted(table))


def main():
    """pmc_table.py DIR [ROOFLINE_REGEX KEY]: KEY = '<workload>:<dtype>' of bench.py; the
    roofline kernel of that workload is the matching kernel with the largest total time."""
    if sys.argv[1] == '--merge':
        return merge(sys.argv[2])
    out = sys.argv[1]
    import re
    roof_re = re.compile(sys.argv[2]) if len(sys.argv) > 2 else re.compile(re.escape(ROOFLINE_KERNEL))
    key = sys.argv[3] if len(sys.argv) > 3 else 'config2:fp32'
    p1, p2, p3 = (load(os.path.join(out, d)) for d in ('p1', 'p2', 'p3'))
    dur = {}
    db = glob.glob(os.path.join(out, 'kt', '**', '*.db'), recursive=True)
    if db:
        cur = sqlite3.connect(db[0]).cursor()
        for name, n, avg, tot in cur.execute(
                'select name, count(*), avg(end-start)/1e3, sum(end-start)/1e3 '
                'from kernels group by name'):
            dur[name] = (n, avg, tot)
    rows = []
    for k, (n, avg, tot) in sorted(dur.items(), key=lambda kv: -kv[1][2]):
        c = p1.get(k, {})
        mean = lambda v: sum(v) / len(v) if v else float('nan')
        gui = mean(c.get('GRBM_GUI_ACTIVE', [])) / 8.0      # summed over 8 XCDs
        mfma = mean(c.get('SQ_VALU_MFMA_BUSY_CYCLES', []))  # summed over SIMDs
        util = mfma / (gui * 1024.0) if gui == gui and gui > 0 else float('nan')
        rd = mean(p2.get(k, {}).get('FETCH_SIZE', [])) * 1024.0 * 2.0
        wr = mean(p3.get(k, {}).get('WRITE_SIZE', [])) * 1024.0
        gbs = (rd + wr) / (avg * 1e-6) / 1e9 if avg > 0 else float('nan')
        rows.append((k, n, avg, tot, util, rd, wr, gbs))
    print('| kernel | launches | avg us | MFMA busy | HBM read MB/launch (FETCH_SIZE x2) | '
          'HBM write MB/launch | HBM GB/s |')
    print('|---|---|---|---|---|---|---|')
    for k, n, avg, tot, util, rd, wr, gbs in rows:
        print(f'| `{k[:110]}` | {n} | {avg:.1f} | {util:.3f} | {rd / 1e6:.1f} | '
              f'{wr / 1e6:.1f} | {gbs:.0f} |')
    # machine-readable record of the roofline kernel: bench.py reports it as
    # roofline.traffic (copy it to profiles/pmc_roofline_kernel.json)
    for k, n, avg, tot, util, rd, wr, gbs in rows:      # rows are sorted by total time
        if roof_re.search(k) and rd == rd and wr == wr:
            rec = dict(key=key, kernel=k[:160], launches=n, avg_us=round(avg, 2),
                       mfma_busy=round(util, 4), hbm_read_bytes_per_launch=int(rd),
                       hbm_write_bytes_per_launch=int(wr),
                       hbm_bytes_per_launch=int(rd) + int(wr),
                       visit=os.path.basename(os.path.dirname(os.path.normpath(out))),
                       method='rocprofv3 --pmc, separate passes for FETCH_SIZE and '
                              'WRITE_SIZE (KiB; FETCH_SIZE x2 on gfx950), --streams 1')
            with open(os.path.join(out, 'pmc_roofline_kernel.json'), 'w') as f:
                json.dump(rec, f, indent=1)
            break


if __name__ == '__main__':
    main()
