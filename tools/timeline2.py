#!/usr/bin/env python3
"""Timeline of the last complete single-sequence encode of a rocprofv3 --kernel-trace --memory-copy-trace run:
timeline2.py <kernel_trace.csv> <memory_copy_trace.csv> [min_us=20]   (events shorter than min_us are listed only when a gap precedes them)"""
import csv, re, sys
K = list(csv.DictReader(open(sys.argv[1])))
M = list(csv.DictReader(open(sys.argv[2])))
min_ns = int(float(sys.argv[3]) * 1000) if len(sys.argv) > 3 else 20000
ev = []
for r in K:
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("qsvc::", "")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n[:44]))
for r in M:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "").replace("MEMORY_COPY_", "")))
ev.sort()
starts = [i for i, e in enumerate(ev) if e[2].startswith("k_pyr_ll_tile") and (i == 0 or e[0] - max(x[1] for x in ev[max(0, i - 5):i]) > 150000)]
i0, i1 = starts[-2], starts[-1]
t0 = ev[i0][0]
pe = t0
busy = 0
for s, e, n in ev[i0:i1]:
    if (e - s) > min_ns or (s - pe) > 15000:
        print(f"{(s - t0) / 1e6:8.3f} -> {(e - t0) / 1e6:8.3f} ({(e - s) / 1e6:6.3f}) gap {(s - pe) / 1e6:6.3f}  {n}")
    busy += max(0, e - max(s, pe))
    pe = max(pe, e)
print("next encode starts at %.3f ms; device busy %.3f ms" % ((ev[i1][0] - t0) / 1e6, busy / 1e6))
