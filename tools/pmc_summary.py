#!/usr/bin/env python3
"""tools/pmc_summary.py <counter_collection.csv> [...] -- per-kernel mean of each collected PMC counter, with the
gfx950 HBM correction of MI355X_MICROARCH.md (FETCH_SIZE under-reports wide coalesced reads by 2x; units are KiB)."""
import csv
import sys
from collections import defaultdict

agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
dur = defaultdict(lambda: [0, 0.0])
for path in sys.argv[1:]:
    seen = set()
    for r in csv.DictReader(open(path)):
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:48]
        a = agg[name][r['Counter_Name']]
        a[0] += 1
        a[1] += float(r['Counter_Value'])
        key = (path, r['Dispatch_Id'])
        if key not in seen:
            seen.add(key)
            dur[name][0] += 1
            dur[name][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3
rows = sorted(agg.items(), key=lambda kv: -dur[kv[0]][1])
for name, cs in rows[:14]:
    n, us = dur[name]
    out = '%-48s n=%4d avg %8.1f us |' % (name, n, us / max(n, 1))
    for c, (k, v) in sorted(cs.items()):
        m = v / k
        if c == 'FETCH_SIZE':
            out += ' HBM read %.1f MB (x2 corrected %.1f)' % (m / 1024, 2 * m / 1024)
        elif c == 'WRITE_SIZE':
            out += ' HBM write %.1f MB' % (m / 1024)
        else:
            out += ' %s=%.4g' % (c, m)
    print(out)
