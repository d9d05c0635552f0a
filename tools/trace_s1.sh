#!/bin/bash
# On the GPU box: kernel + memory-copy trace of single-sequence encodes -> gpurun_out/tl_kernel.csv, gpurun_out/tl_memcpy.csv
# usage: tools/trace_s1.sh [config=3]    (read the timeline with tools/timeline2.py)
set -e
cfg=${1:-3}
export TMPDIR=/tmp PYTHONPATH=$PWD
rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/tl_prof -o t --output-format csv -- python3 bench.py --config $cfg --sessions 1 --steps 3 --warmup 1 --no-cpu-baseline --no-decode --no-iso --no-cli --no-config4 > gpurun_out/tl_prof.log 2>&1
cp $(find gpurun_out/tl_prof -name "*memory_copy_trace.csv" | head -1) gpurun_out/tl_memcpy.csv
cp $(find gpurun_out/tl_prof -name "*kernel_trace.csv" | head -1) gpurun_out/tl_kernel.csv
rm -rf gpurun_out/tl_prof
python3 tools/timeline2.py gpurun_out/tl_kernel.csv gpurun_out/tl_memcpy.csv | tail -70
