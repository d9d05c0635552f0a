#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs (separate passes, MI355X_MICROARCH.md 'HBM') per kernel.
usage: pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [encodes profiled]
(encodes profiled: how many whole sequence encodes the profiled command ran -- bench.py --sessions 1 --steps 4 --warmup 1
--no-decode --no-iso runs 1 + 4 = 5; gives hbm_bytes_per_step_raw = bytes per encode)
Units: the counters are in KiB.  On gfx950 FETCH_SIZE reads exactly half of a wide coalesced stream's bytes and is
uncalibrated for other access widths, so both the raw sum and the 'fetch doubled' figure are kept."""
import collections, csv, json, sys


def load(path, cname):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == cname:
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("qsvc::", "")
            d[name].append(float(r["Counter_Value"]))
    return d


f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
out = {}
for k in f:
    n = len(f[k])
    fb, wb = sum(f[k]) * 1024 / n, sum(w.get(k, [0])) * 1024 / max(len(w.get(k, [0])), 1)
    out[k] = {"launches_profiled": n, "fetch_bytes_per_launch_raw": int(fb), "write_bytes_per_launch": int(wb),
              "hbm_bytes_per_launch_raw": int(fb + wb), "hbm_bytes_per_launch_fetch_doubled": int(2 * fb + wb),
              "hbm_bytes_per_step_raw": int((sum(f[k]) + sum(w.get(k, [0]))) * 1024 / steps),
              "hbm_bytes_per_step_fetch_doubled": int((2 * sum(f[k]) + sum(w.get(k, [0]))) * 1024 / steps), "encodes_profiled": steps}
json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True)[:1500])
