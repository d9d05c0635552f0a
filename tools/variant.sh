#!/bin/bash
# A/B of a kernel variant: tools/variant.sh <name> <file.hip> "<-D defs>"  ->  build/libqsvc_<name>.so, the library with that one
# translation unit recompiled with the extra definitions.  Select it at run time with QSVC_HIP_LIB=$PWD/build/libqsvc_<name>.so
# (tools/time_mctf.py, tools/mq_time.py, bench.py).  Experiment switches (#ifndef X / #define X default) stay in the sources only
# when the default is what is committed.
set -e
name=$1; unit=$2; defs=$3
cd "$(dirname "$0")/../qsvc_amd/csrc"
mkdir -p ../../build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-value"
/opt/rocm/bin/hipcc $FLAGS $defs -c $unit -o ../../build/${name}_${unit%.hip}.o
objs=""
for o in *.o; do
  if [ "$o" = "${unit%.hip}.o" ]; then objs="$objs ../../build/${name}_${unit%.hip}.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/libqsvc_$name.so $objs -lpthread
ls -la ../../build/libqsvc_$name.so
