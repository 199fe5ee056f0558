#!/usr/bin/env python3
"""tools/pmc_summary.py <counter_collection.csv> [...] -- per-kernel mean of each collected PMC counter, with the
gfx950 HBM correction of MI355X_MICROARCH.md (FETCH_SIZE under-reports wide coalesced reads by 2x; units are KiB)."""
import csv
import json
import sys
from collections import defaultdict

argv = sys.argv[1:]
json_out = json_tag = json_src = None
if argv and argv[0] == '--json':          # --json <file> <tag> <source text>: merge {tag: {...}} into <file> (bench.py reads it)
    json_out, json_tag, json_src = argv[1:4]
    argv = argv[4:]

agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
dur = defaultdict(lambda: [0, 0.0])
for path in argv:
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
if json_out:
    try:
        doc = json.load(open(json_out))
    except (OSError, ValueError):
        doc = {}
    ks = {}
    for name, cs in rows:
        if 'FETCH_SIZE' in cs and 'WRITE_SIZE' in cs:
            n, us = dur[name]
            ks[name] = {'launches': cs['FETCH_SIZE'][0], 'avg_us': round(us / max(n, 1), 1),
                        'hbm_read_bytes': round(2 * 1024 * cs['FETCH_SIZE'][1] / cs['FETCH_SIZE'][0]),
                        'hbm_write_bytes': round(1024 * cs['WRITE_SIZE'][1] / cs['WRITE_SIZE'][0])}
    doc[json_tag] = {'source': json_src, 'kernels': ks}
    json.dump(doc, open(json_out, 'w'), indent=1, sort_keys=True)
