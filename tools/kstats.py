#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv per 17-picture step: usage kstats.py <csv> [steps=8]
(tools/prof_s1.sh profiles `bench.py --sessions 1 --steps 3 --warmup 1`: 1 + 3 timed-region steps and 1 + 3 of the isolated pass)"""
import csv, re, sys
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r'\(.*', '', r['Name']).replace('void ', '').replace('qsvc::', '')
    t = float(r['TotalDurationNs']) / steps / 1e3
    tot += t
    print(f"{n:42s} calls/step {int(r['Calls'])/steps:5.1f}  us/step {t:9.1f}  avg {float(r['AverageNs'])/1e3:8.1f}  min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
print(f"{'sum':42s} {'':16s} us/step {tot:9.1f}")
