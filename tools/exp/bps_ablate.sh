#!/bin/bash
# Timing builds of bps_mfma_kernel (BPS_ABLATE = 1: no exact stage, 4: matrix instructions only, 7: staging only;
# BPS_NW: waves per workgroup) beside the shipped build.  The ablation branches are NOT in the shipped source:
# tools/exp/bps_ablate.patch puts them into a scratch copy.  Run on the GPU box from the repo root:  bash tools/exp/bps_ablate.sh
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
CS=$ROOT/dex_gan_grasp_amd/csrc
T=$(mktemp -d)
cp $CS/bps_encode.hip $T/bps_encode.hip
(cd $T && patch -p3 bps_encode.hip < $ROOT/tools/exp/bps_ablate.patch)
cp $CS/../libdgg.so /tmp/libdgg_keep.so
O=$(ls $CS/_obj/*.o | grep -v bps_encode)
for a in ${ABL:-"-DBPS_ABLATE=1" "-DBPS_ABLATE=4" "-DBPS_ABLATE=7"}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form $a -I$CS -c $T/bps_encode.hip -o /tmp/bps_abl.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $CS/../libdgg.so $O /tmp/bps_abl.o
  echo "BPS_ABLATE $a"; (cd $ROOT && python3 tools/exp/bps_ab.py run /tmp/abl.npz)
done
cp /tmp/libdgg_keep.so $CS/../libdgg.so
