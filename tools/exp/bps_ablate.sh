#!/bin/bash
# Timing builds of bps_mfma_kernel (BPS_ABLATE = 1: no exact stage, 4: matrix instructions only, 7: staging only;
# BPS_NW: waves per workgroup) beside the shipped build.  Run on the GPU box from the repo root:  bash tools/exp/bps_ablate.sh
set -e
cd "$(dirname "$0")/../../dex_gan_grasp_amd/csrc"
cp ../libdgg.so /tmp/libdgg_keep.so
O="_obj/gemm_f32.o _obj/gemm_x4.o _obj/gemm_x5.o _obj/elementwise.o _obj/second_order.o _obj/dgg_api.o"
for a in ${ABL:-"-DBPS_ABLATE=1" "-DBPS_ABLATE=4" "-DBPS_ABLATE=7"}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form $a -c bps_encode.hip -o /tmp/bps_abl.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libdgg.so $O /tmp/bps_abl.o
  echo "BPS_ABLATE $a"; (cd ../.. && python3 tools/exp/bps_ab.py run /tmp/abl.npz)
done
cp /tmp/libdgg_keep.so ../libdgg.so
