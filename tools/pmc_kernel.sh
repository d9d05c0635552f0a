#!/bin/bash
# SQ counters of one kernel (name substring $1) over a single-sequence encode:  tools/pmc_kernel.sh k_t1_mq "SQ_INSTS_VALU SQ_INSTS_SALU ..."
# (run on the GPU box; one rocprofv3 pass per group of <= 4 counters; prints per-launch averages)
set -e
export TMPDIR=/tmp
K="$1"; shift
for grp in "$@"; do
  d=gpurun_out/pmck_$$
  rocprofv3 --pmc $grp -d $d -o p --output-format csv -- python3 tools/mq_time.py 2 > /dev/null 2>&1
  python3 - "$K" $(find $d -name "*counter_collection.csv") <<'PY'
import csv, sys, collections
k, f = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if k in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, v in acc.items():
    print(f"{n:28s} per launch {sum(v)/len(v):16.0f}  ({len(v)} launches)")
PY
  rm -rf $d
done
