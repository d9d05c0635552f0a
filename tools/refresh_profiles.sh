#!/bin/bash
# On the GPU box: bench line, rocprofv3 kernel stats of the same command, and the two PMC passes (FETCH_SIZE / WRITE_SIZE in
# separate runs, MI355X_MICROARCH.md "HBM").  usage: tools/refresh_profiles.sh <tag>   -> gpurun_out/<tag>_*
set -e
tag=$1
R=$PWD
export TMPDIR=/tmp PYTHONPATH=$R
python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -c 600 gpurun_out/${tag}_bench.json; echo
rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_prof -o b --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_prof.log 2>&1
cp $(find gpurun_out/${tag}_prof -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_bench_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/${tag}_pmc_f -o p --output-format csv -- python3 bench.py --sessions 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/${tag}_pmc_w -o p --output-format csv -- python3 bench.py --sessions 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_pmc_w.log 2>&1
python3 tools/pmc_summary.py $(find gpurun_out/${tag}_pmc_f -name "*counter_collection.csv") $(find gpurun_out/${tag}_pmc_w -name "*counter_collection.csv") gpurun_out/${tag}_pmc_hbm.json > /dev/null
rm -rf gpurun_out/${tag}_prof gpurun_out/${tag}_pmc_f gpurun_out/${tag}_pmc_w
head -12 gpurun_out/${tag}_bench_kernel_stats.csv
