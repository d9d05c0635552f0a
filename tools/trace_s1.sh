set -e
export TMPDIR=/tmp PYTHONPATH=$PWD
rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/tl_prof -o t --output-format csv -- python3 bench.py --sessions 1 --steps 3 --warmup 1 --no-cpu-baseline --no-decode > gpurun_out/tl_prof.log 2>&1
ls gpurun_out/tl_prof/*/ | head
python3 tools/timeline.py $(find gpurun_out/tl_prof -name "*kernel_trace.csv" | head -1) > gpurun_out/tl_timeline.txt 2>&1 || true
cp $(find gpurun_out/tl_prof -name "*memory_copy_trace.csv" | head -1) gpurun_out/tl_memcpy.csv || true
cp $(find gpurun_out/tl_prof -name "*kernel_trace.csv" | head -1) gpurun_out/tl_kernel.csv || true
rm -rf gpurun_out/tl_prof
tail -50 gpurun_out/tl_timeline.txt
