#!/usr/bin/env python3
"""Average every PMC counter per kernel from a rocprofv3 counter_collection CSV.
    python tools/pmc_summary.py <csv> [kernel-substring]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
sub = sys.argv[2] if len(sys.argv) > 2 else ''
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if sub in r['Kernel_Name']:
        agg[r['Kernel_Name'][:100]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    print(k)
    for c, vals in sorted(v.items()):
        print(f'    {c:28s} avg {sum(vals) / len(vals):16.1f}  n={len(vals)}')
