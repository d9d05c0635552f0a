#!/bin/bash
# On the GPU box: rocprofv3 kernel stats of tools/time_mctf.py (one sequence, MCTF levels only).  usage: tools/prof_mctf.sh <tag> [cfg3|cfg4]
tag=$1; cfg=${2:-cfg3}
export TMPDIR=/tmp PYTHONPATH=$PWD
rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_prof -o m --output-format csv -- python3 tools/time_mctf.py $cfg 4 > gpurun_out/${tag}_prof.log 2>&1
cp $(find gpurun_out/${tag}_prof -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_${cfg}_kernel_stats.csv
rm -rf gpurun_out/${tag}_prof
python3 tools/kstats.py gpurun_out/${tag}_${cfg}_kernel_stats.csv 5 | head -30
