#!/bin/bash
# GPU box: per-launch durations (kernel trace) and SQ counters of the DWT kernels of one texture batch.  usage: tools/dwt_prof.sh <tag> [dwt_prof.py args]
set -e
tag=$1; shift
export TMPDIR=/tmp PYTHONPATH=$PWD
mkdir -p gpurun_out
python3 tools/dwt_prof.py "$@" > gpurun_out/${tag}_time.txt 2>&1; cat gpurun_out/${tag}_time.txt
rocprofv3 --kernel-trace -d gpurun_out/${tag}_kt -o t --output-format csv -- python3 tools/dwt_prof.py "$@" > /dev/null 2>&1
python3 - $(find gpurun_out/${tag}_kt -name "*kernel_trace.csv") > gpurun_out/${tag}_levels.txt <<'PY'
import csv, sys, collections
acc = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "dwt" not in n: continue
    key = (n.split("(")[0].replace("void qsvc::", ""), r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""))
    acc.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in acc.items():
    v = sorted(v)
    print(f"{k[0]:36s} grid {k[1]:>9s}  n {len(v):3d}  median {v[len(v)//2]:8.1f} us  min {v[0]:8.1f}")
PY
cat gpurun_out/${tag}_levels.txt
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  d=gpurun_out/${tag}_pm
  rm -rf $d
  rocprofv3 --pmc $grp -d $d -o p --output-format csv -- python3 tools/dwt_prof.py "$@" > /dev/null 2>&1 || { echo "group failed: $grp"; continue; }
  python3 - $(find $d -name "*counter_collection.csv") >> gpurun_out/${tag}_sq.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "dwt" in r["Kernel_Name"]:
        acc[(r["Kernel_Name"].split("(")[0].replace("void qsvc::", ""), r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k[0]:34s} grid {k[1]:>9s} {k[2]:22s} {sum(v)/len(v):16.0f}")
PY
  rm -rf $d
done
rm -rf gpurun_out/${tag}_kt
sort gpurun_out/${tag}_sq.txt
