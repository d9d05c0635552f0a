#!/usr/bin/env python3
"""Kernel timeline of ONE single-sequence encode from a rocprofv3 --kernel-trace csv:  timeline.py <kernel_trace.csv> [step]
(steps are found by the k_pyr_ll_tile launches: the first kernel of each temporal level's motion search, [levels] per encode)"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = [i for i, r in enumerate(rows) if "k_pyr_ll_tile" in r["Kernel_Name"]]
levels = int(sys.argv[3]) if len(sys.argv) > 3 else 3
steps = first[::levels]
st = int(sys.argv[2]) if len(sys.argv) > 2 else len(steps) - 2
i0, i1 = steps[st], steps[st + 1]
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = 0.0
for r in rows[i0:i1]:
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("qsvc::", "")
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    print(f"{s:8.3f} -> {e:8.3f}  ({e - s:6.3f} ms, gap before {s - prev_end:6.3f})  {n[:60]}")
    prev_end = max(prev_end, e)
print("next step starts at %.3f" % ((int(rows[i1]["Start_Timestamp"]) - t0) / 1e6))
