#!/bin/bash
# kernel trace of the graph-replayed training step: how much of a step's wall time has no kernel running, and the launch-to-launch gaps
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/gaps; mkdir -p $O; cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-stem-ab "$@" > $O/bench.log 2>&1
f=$(find $O/tr -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
# the timed graph replays: take the last 40 % of the trace (the eager second pass comes after the graph pass: use the window with adam_kernel periodicity)
adam = [s for s, e, n in ev if 'adam_kernel' in n]
print('adam launches', len(adam))
# steps between consecutive adam starts; graph-mode steps are the shorter ones
per = [(adam[i + 1] - adam[i]) / 1e6 for i in range(len(adam) - 1)]
print('step periods ms', [round(p, 2) for p in per])
import statistics
best = min(range(len(per)), key=lambda i: per[i])
t0, t1 = adam[best], adam[best + 1]
win = [(s, e, n) for s, e, n in ev if s >= t0 and s < t1]
busy = 0; cur_s, cur_e = None, None
for s, e, n in win:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print('step %.3f ms: %d kernels, union busy %.3f ms, idle %.3f ms, sum of durations %.3f ms' % ((t1 - t0) / 1e6, len(win), busy / 1e6, (t1 - t0 - busy) / 1e6, sum(e - s for s, e, n in win) / 1e6))
gaps = []
prev_e = None
for s, e, n in win:
    if prev_e is not None and s > prev_e: gaps.append((s - prev_e, n))
    prev_e = max(prev_e, e) if prev_e else e
gaps.sort(reverse=True)
print('largest idle gaps (us) before kernel:', [(round(g / 1e3, 1), n[:40]) for g, n in gaps[:12]])
print('gaps > 1us:', sum(1 for g, n in gaps if g > 1000), 'total', round(sum(g for g, n in gaps) / 1e3, 1), 'us')
PY
rm -rf $O/tr
