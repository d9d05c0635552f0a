#!/bin/bash
# builds build/libqsvc_mq<N>.so with -DMQ_EXP=<N> for each N given (timing experiments on the MQ coder kernel)
set -e
cd "$(dirname "$0")/../qsvc_amd/csrc"
mkdir -p ../../build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-value"
for n in "$@"; do
  ( /opt/rocm/bin/hipcc $FLAGS -DMQ_RING=2 -DT1_STAGE=116 -DT1_STAGE_FLUSH=48 -DVARIANT=$n -c j2k_kernels.hip -o ../../build/jk_mq$n.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/libqsvc_mq$n.so qsvc_api.o mctf_kernels.o mctf_margin.o ../../build/jk_mq$n.o j2k_dwt.o j2k_api.o j2k_decode.o -lpthread ) &
done
wait
ls -la ../../build/*.so
