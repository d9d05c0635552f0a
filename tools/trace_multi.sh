#!/bin/bash
# On the GPU box: kernel trace of the 12-sequence bench -> how much of the wall clock has NO kernel running, and how many run at once
set -e
export TMPDIR=/tmp PYTHONPATH=$PWD
rocprofv3 --kernel-trace -d gpurun_out/tm_prof -o t --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-decode --no-iso --no-cli --no-config4 > gpurun_out/tm_prof.log 2>&1
python3 - $(find gpurun_out/tm_prof -name "*kernel_trace.csv" | head -1) <<'PY'
import csv, sys
ev = []
for r in csv.DictReader(open(sys.argv[1])):
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
t0, t1 = ev[0][0], ev[-1][0]
# last 60 % of the trace = the timed steady state
lo = t0 + (t1 - t0) * 4 // 10
cur = 0; last = None; idle = 0; hist = {}
for t, d in ev:
    if last is not None and t > lo:
        a = max(last, lo)
        if t > a:
            hist[cur] = hist.get(cur, 0) + (t - a)
            if cur == 0: idle += t - a
    cur += d; last = t
tot = t1 - lo
print("window %.1f ms, no kernel running %.2f %%" % (tot / 1e6, 100.0 * idle / tot))
for k in sorted(hist): print("  %2d kernels in flight: %5.1f %%" % (k, 100.0 * hist[k] / tot))
PY
rm -rf gpurun_out/tm_prof
