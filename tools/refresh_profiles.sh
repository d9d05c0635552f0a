#!/bin/bash
# On the GPU box: the judged bench line, rocprofv3 kernel stats of the same command, ISOLATED kernel stats (one sequence in
# flight) and the two PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs, one sequence in flight: MI355X_MICROARCH.md "HBM").
# The program follows `--` directly (python3 bench.py ...), never a wrapper.
# usage: tools/refresh_profiles.sh <tag>   -> gpurun_out/<tag>_*   (copy what is to be judged into profiles/)
set -e
tag=$1
R=$PWD
export TMPDIR=/tmp PYTHONPATH=$R
mkdir -p gpurun_out
# PMC first: bench.py reads profiles/r02_pmc_hbm.json for roofline.traffic
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/${tag}_pmc_f -o p --output-format csv -- python3 bench.py --sessions 1 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/${tag}_pmc_w -o p --output-format csv -- python3 bench.py --sessions 1 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_pmc_w.log 2>&1
python3 tools/pmc_summary.py $(find gpurun_out/${tag}_pmc_f -name "*counter_collection.csv") $(find gpurun_out/${tag}_pmc_w -name "*counter_collection.csv") gpurun_out/${tag}_pmc_hbm.json > /dev/null
cp gpurun_out/${tag}_pmc_hbm.json profiles/r02_pmc_hbm.json
echo "pmc done"
python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -c 400 gpurun_out/${tag}_bench.json; echo
rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_prof -o b --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_prof.log 2>&1
cp $(find gpurun_out/${tag}_prof -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_bench_kernel_stats.csv
echo "8-session stats done"
python3 bench.py --sessions 1 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_s1_bench.json 2> gpurun_out/${tag}_s1_bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_s1_prof -o b --output-format csv -- python3 bench.py --sessions 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_s1_prof.log 2>&1
cp $(find gpurun_out/${tag}_s1_prof -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_s1_kernel_stats.csv
rm -rf gpurun_out/${tag}_prof gpurun_out/${tag}_s1_prof gpurun_out/${tag}_pmc_f gpurun_out/${tag}_pmc_w
python3 tools/kstats.py gpurun_out/${tag}_s1_kernel_stats.csv 8 | head -24
