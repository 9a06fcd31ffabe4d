#!/bin/bash
# Race hunt: build bps_encode.hip with the given flags, run the A/B comparison R times against the vector-ALU scan.
#   [BN="1024 1024"] bash tools/exp/bps_race.sh "<flags>" [R]      (BN: clouds and points per cloud)
set -e
cd "$(dirname "$0")/../../dex_gan_grasp_amd/csrc"
cp ../libdgg.so /tmp/libdgg_keep.so
O="_obj/gemm_f32.o _obj/gemm_x4.o _obj/gemm_x5.o _obj/elementwise.o _obj/second_order.o _obj/dgg_api.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form $1 -c bps_encode.hip -o /tmp/bps_race.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libdgg.so $O /tmp/bps_race.o
cd ../..
DGG_BPS=valu python3 tools/exp/bps_ab.py run /tmp/bps_valu.npz $BN > /dev/null
echo "flags: $1"
for k in $(seq 1 ${2:-6}); do python3 tools/exp/bps_ab.py run /tmp/bps_mfma.npz $BN | tr '\n' ' '; python3 tools/exp/bps_ab.py cmp /tmp/bps_valu.npz /tmp/bps_mfma.npz x $BN 2>/dev/null | head -1; done
cp /tmp/libdgg_keep.so dex_gan_grasp_amd/libdgg.so
