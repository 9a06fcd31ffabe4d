#!/bin/bash
# libdgg with per-wave stamps in gemm_x5_kernel (diagnostics): tools/exp/libdgg_x5stamp.so
set -e
cd "$(dirname "$0")/../../dex_gan_grasp_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -fno-slp-vectorize -DDGG_X5_STAMP=1 -c gemm_x5.hip -o _obj/gemm_x5_stamp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/exp/libdgg_x5stamp.so _obj/gemm_f32.o _obj/gemm_x4.o _obj/gemm_x5_stamp.o _obj/elementwise.o _obj/second_order.o _obj/bps_encode.o _obj/dgg_api.o
echo built tools/exp/libdgg_x5stamp.so
