#!/bin/bash
# libdgg with per-wave s_memtime stamps in gemm_x5_kernel (diagnostics only): tools/exp/libdgg_x5stamp.so.
# The stamps are NOT in the shipped source: tools/exp/x5_stamp.patch puts them into a scratch copy of gemm_x5.hip
# (-DDGG_X5_STAMP=1 switches them on).  The other objects are the shipped build's (run csrc/build.sh first).
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
CS=$ROOT/dex_gan_grasp_amd/csrc
T=$(mktemp -d)
cp $CS/*.h $T/ && mkdir -p $T/../../include && cp $ROOT/include/dgg.h $T/../../include/ 2>/dev/null || true
cp $CS/gemm_x5.hip $T/gemm_x5.hip
(cd $T && patch -p3 gemm_x5.hip < $ROOT/tools/exp/x5_stamp.patch)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -fno-slp-vectorize -DDGG_X5_STAMP=1 -I$CS -c $T/gemm_x5.hip -o $T/gemm_x5_stamp.o
OBJS=$(ls $CS/_obj/*.o | grep -v gemm_x5)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/exp/libdgg_x5stamp.so $OBJS $T/gemm_x5_stamp.o
echo built tools/exp/libdgg_x5stamp.so
