#!/bin/bash
# Builds libdgg.so for gfx950 in-tree (the .so travels to the GPU box with the repo snapshot).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
X3FLAGS="-fno-slp-vectorize"  # packed-f32 VALU beside MFMAs costs more than two scalar ops (MI355X_MICROARCH.md)
rm -rf _obj   # no stale objects of deleted kernels (they would ship to the GPU box with the snapshot)
mkdir -p _obj
pids=""
$HIPCC $COMMON -c gemm_f32.hip -o _obj/gemm_f32.o & pids="$pids $!"
$HIPCC $COMMON $X3FLAGS -c gemm_x4.hip -o _obj/gemm_x4.o & pids="$pids $!"
$HIPCC $COMMON $X3FLAGS -c gemm_x5.hip -o _obj/gemm_x5.o & pids="$pids $!"
$HIPCC $COMMON -c elementwise.hip -o _obj/elementwise.o & pids="$pids $!"
$HIPCC $COMMON -c fc2_stage.hip -o _obj/fc2_stage.o & pids="$pids $!"
$HIPCC $COMMON -c second_order.hip -o _obj/second_order.o & pids="$pids $!"
# fixed contraction: only the explicit fmaf() calls of the distance formula fuse (parity contract)
# (-amdgpu-mfma-vgpr-form: the coarse pass reads its accumulators with vector instructions right away — no AGPR copies)
$HIPCC $COMMON -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -c bps_encode.hip -o _obj/bps_encode.o & pids="$pids $!"
$HIPCC $COMMON -c dgg_api.hip -o _obj/dgg_api.o & pids="$pids $!"
for p in $pids; do wait $p || { echo "build.sh: a compile failed" >&2; exit 1; }; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libdgg.so _obj/gemm_f32.o _obj/gemm_x4.o _obj/gemm_x5.o _obj/elementwise.o _obj/fc2_stage.o _obj/second_order.o _obj/bps_encode.o _obj/dgg_api.o
echo "built $(cd .. && pwd)/libdgg.so"
