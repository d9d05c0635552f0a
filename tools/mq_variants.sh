#!/bin/bash
# A/B of kernel variants: builds build/libqsvc_v<N>.so from j2k_kernels.hip with -DQSVC_VARIANT=<N> (plus EXTRA_DEFS) for each N
# given; a variant is selected at run time by QSVC_HIP_LIB=$PWD/build/libqsvc_v<N>.so (tools/mq_time.py prints kernel times and
# a digest of the streams).  The experiment's #if QSVC_VARIANT == N switches are temporary and do not get committed.
set -e
cd "$(dirname "$0")/../qsvc_amd/csrc"
mkdir -p ../../build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-value"
for n in "$@"; do
  ( /opt/rocm/bin/hipcc $FLAGS -DQSVC_VARIANT=$n $EXTRA_DEFS -c j2k_kernels.hip -o ../../build/jk_v$n.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/libqsvc_v$n.so qsvc_api.o mctf_kernels.o mctf_me.o mctf_pyr.o mctf_pred.o mctf_upd.o mctf_margin.o ../../build/jk_v$n.o j2k_dwt.o j2k_api.o j2k_decode.o -lpthread ) &
done
wait
ls -la ../../build/*.so
