#!/bin/bash
# GPU box: per-launch durations of the kernels whose name contains $1, from a rocprofv3 kernel trace of the program that follows.
# usage: tools/ktrace.sh <name substring> python3 tools/time_mctf.py cfg3 3     (the program directly, no wrapper)
set -e
pat=$1; shift
export TMPDIR=/tmp PYTHONPATH=$PWD
d=gpurun_out/ktrace_$$
rocprofv3 --kernel-trace -d $d -o t --output-format csv -- "$@" > /dev/null 2>&1
python3 - "$pat" $(find $d -name "*kernel_trace.csv") <<'PY'
import csv, sys, collections
pat = sys.argv[1]
acc = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[2])):
    n = r["Kernel_Name"]
    if pat not in n: continue
    key = (n.split("(")[0].replace("void qsvc::", ""), r.get("Grid_Size_X") or r.get("Grid_Size", ""), r.get("LDS_Block_Size", ""), r.get("VGPR_Count", ""))
    acc.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in acc.items():
    v = sorted(v)
    print(f"{k[0]:36s} grid {k[1]:>9s} lds {k[2]:>6s} vgpr {k[3]:>4s}  n {len(v):3d}  median {v[len(v)//2]:8.1f} us  min {v[0]:8.1f}")
PY
rm -rf $d
