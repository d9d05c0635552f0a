#!/usr/bin/env python3
"""Per-kernel means of the SQ counters of one or more rocprofv3 --pmc passes.
usage: pmc_sq.py out.json <counter_collection.csv>..."""
import collections, csv, json, sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("qsvc::", "")
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
for k, d in out.items():
    d["launches"] = len(next(iter(acc[k].values())))
    d["encodes_profiled"] = 5   # tools/refresh_profiles.sh / sq_pass.sh: 1 warm-up + 4 timed encodes of one sequence (bench.py --no-iso)
json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
for k in sorted(out):
    print(k, {c: round(v) for c, v in out[k].items()})
