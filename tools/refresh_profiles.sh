#!/bin/bash
# On the GPU box: the judged bench line, rocprofv3 kernel stats of the same command, ISOLATED kernel stats (one sequence in
# flight), the two HBM PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs, one sequence in flight: MI355X_MICROARCH.md "HBM")
# and the SQ instruction / cycle counters (three more passes).  The program follows `--` directly (python3 bench.py ...), never a
# wrapper, and the profiled commands run no child process (--no-cli --no-config4).
# usage: tools/refresh_profiles.sh <tag> [config=3]  -> gpurun_out/<tag>_*
# Everything stays under gpurun_out/ (the only directory that travels back from the box); copy what is to be judged into
# profiles/ in the build container afterwards -- in particular <tag>_pmc_hbm.json as profiles/r04_pmc_hbm.json (config 3) or
# profiles/r04_cfg4_pmc_hbm.json (config 4), the files bench.py reads its traffic figures from.
set -e
tag=$1; cfg=${2:-3}
R=$PWD
export TMPDIR=/tmp PYTHONPATH=$R
mkdir -p gpurun_out
ONE="--config $cfg --sessions 1 --steps 4 --warmup 1 --no-cpu-baseline --no-decode --no-iso --no-cli --no-config4"
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/${tag}_pmc_f -o p --output-format csv -- python3 bench.py $ONE > gpurun_out/${tag}_pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/${tag}_pmc_w -o p --output-format csv -- python3 bench.py $ONE > gpurun_out/${tag}_pmc_w.log 2>&1
# 5 = the encodes that command runs: 1 warm-up + 4 timed (no single-sequence pass, no decode extra)
python3 tools/pmc_summary.py $(find gpurun_out/${tag}_pmc_f -name "*counter_collection.csv") $(find gpurun_out/${tag}_pmc_w -name "*counter_collection.csv") gpurun_out/${tag}_pmc_hbm.json 5 > /dev/null
echo "pmc hbm done"
# SQ counters (per-kernel means over the launches of the same command): instruction mix, then cycles, then waits -- a failing group
# (a counter this build of rocprofv3 does not know) is skipped
sq=""
k=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY"; do
  k=$((k+1))
  if rocprofv3 --pmc $grp -d gpurun_out/${tag}_sq$k -o p --output-format csv -- python3 bench.py $ONE > gpurun_out/${tag}_sq$k.log 2>&1; then
    f=$(find gpurun_out/${tag}_sq$k -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && sq="$sq $f"
  else
    echo "SQ group '$grp' failed (see gpurun_out/${tag}_sq$k.log)"
  fi
done
[ -n "$sq" ] && python3 tools/pmc_sq.py gpurun_out/${tag}_pmc_sq.json $sq > /dev/null
echo "pmc sq done"
python3 bench.py --config $cfg > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -c 400 gpurun_out/${tag}_bench.json; echo
rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_prof -o b --output-format csv -- python3 bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-decode --no-iso --no-cli --no-config4 > gpurun_out/${tag}_prof.log 2>&1
cp $(find gpurun_out/${tag}_prof -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_bench_kernel_stats.csv
echo "multi-session stats done"
python3 bench.py --config $cfg --sessions 1 --steps 5 --warmup 2 --no-cpu-baseline --no-cli --no-config4 > gpurun_out/${tag}_s1_bench.json 2> gpurun_out/${tag}_s1_bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_s1_prof -o b --output-format csv -- python3 bench.py --config $cfg --sessions 1 --steps 7 --warmup 1 --no-cpu-baseline --no-decode --no-iso --no-cli --no-config4 > gpurun_out/${tag}_s1_prof.log 2>&1
cp $(find gpurun_out/${tag}_s1_prof -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_s1_kernel_stats.csv
rm -rf gpurun_out/${tag}_prof gpurun_out/${tag}_s1_prof gpurun_out/${tag}_pmc_f gpurun_out/${tag}_pmc_w gpurun_out/${tag}_sq1 gpurun_out/${tag}_sq2 gpurun_out/${tag}_sq3
python3 tools/kstats.py gpurun_out/${tag}_s1_kernel_stats.csv 8 | head -30   # 8 encodes: 1 warm-up + 7 timed
