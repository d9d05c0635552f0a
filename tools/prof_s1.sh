#!/bin/bash
# On the GPU box: ISOLATED (one sequence in flight) bench line + rocprofv3 kernel stats of the same command.
# usage: tools/prof_s1.sh <tag>   -> gpurun_out/<tag>_s1_bench.json, gpurun_out/<tag>_s1_kernel_stats.csv
set -e
tag=$1
export TMPDIR=/tmp PYTHONPATH=$PWD
mkdir -p gpurun_out
python3 bench.py --sessions 1 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_s1_bench.json 2> gpurun_out/${tag}_s1_bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_s1_prof -o b --output-format csv -- python3 bench.py --sessions 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_s1_prof.log 2>&1
cp $(find gpurun_out/${tag}_s1_prof -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_s1_kernel_stats.csv
rm -rf gpurun_out/${tag}_s1_prof
cut -d, -f1-4 gpurun_out/${tag}_s1_kernel_stats.csv | sed 's/(.*)"/"/' | head -24
