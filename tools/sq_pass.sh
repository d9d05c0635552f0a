#!/bin/bash
# On the GPU box: only the SQ counter passes of tools/refresh_profiles.sh (instruction mix, cycles, waits per kernel; one sequence in
# flight) -> gpurun_out/<tag>_pmc_sq.json.   usage: tools/sq_pass.sh <tag> [config=3]
set -e
tag=$1; cfg=${2:-3}
export TMPDIR=/tmp PYTHONPATH=$PWD
mkdir -p gpurun_out
ONE="--config $cfg --sessions 1 --steps 4 --warmup 1 --no-cpu-baseline --no-decode --no-iso --no-cli --no-config4"
sq=""; k=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY"; do
  k=$((k+1))
  if rocprofv3 --pmc $grp -d gpurun_out/${tag}_sq$k -o p --output-format csv -- python3 bench.py $ONE > gpurun_out/${tag}_sq$k.log 2>&1; then
    f=$(find gpurun_out/${tag}_sq$k -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && sq="$sq $f"
  fi
done
python3 tools/pmc_sq.py gpurun_out/${tag}_pmc_sq.json $sq > /dev/null
rm -rf gpurun_out/${tag}_sq1 gpurun_out/${tag}_sq2 gpurun_out/${tag}_sq3
python3 - <<P
import json
d = json.load(open("gpurun_out/${tag}_pmc_sq.json"))
for k, v in d.items():
    if isinstance(v, dict) and v.get("SQ_INSTS_VALU", 0) > 1e6:
        print("%-40s launches %3d VALU %7.2fM SALU %6.2fM LDS %6.2fM VMEM %5.2fM busy %6.2fM waves %6d" % (k[:40], v["launches"], v["SQ_INSTS_VALU"]/1e6, v["SQ_INSTS_SALU"]/1e6, v["SQ_INSTS_LDS"]/1e6, (v["SQ_INSTS_VMEM_RD"]+v["SQ_INSTS_VMEM_WR"])/1e6, v["SQ_BUSY_CYCLES"]/1e6, v["SQ_WAVES"]))
P
