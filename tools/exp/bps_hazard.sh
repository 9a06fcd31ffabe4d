#!/bin/bash
# Builds tools/exp/libdgg_hazard.so: libdgg.so with bps_encode.hip compiled -DBPS_R3_HAZARD (the round-3 failing form of the
# exact stage: per-lane 12-byte basis loads among 16-byte point loads under the compiler's counted vmcnt, no explicit wait),
# then runs the full-output regression test against it R times:
#     bash tools/exp/bps_hazard.sh build            (here: cross-compiles)
#     bash tools/exp/bps_hazard.sh run [R]          (on the GPU box)
# A run that FAILS shows that the test catches the failure; all runs passing means this build does not reproduce it.
set -e
cd "$(dirname "$0")/../.."
if [ "$1" = "build" ]; then
  cd dex_gan_grasp_amd/csrc
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -DBPS_R3_HAZARD -c bps_encode.hip -o /tmp/bps_hazard.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/exp/libdgg_hazard.so _obj/gemm_f32.o _obj/gemm_x3.o _obj/gemm_p3.o _obj/gemm_p4.o _obj/gemm_x4.o _obj/elementwise.o _obj/second_order.o /tmp/bps_hazard.o _obj/dgg_api.o
  echo built tools/exp/libdgg_hazard.so
else
  for k in $(seq 1 ${2:-4}); do
    DGG_LIB=$PWD/tools/exp/libdgg_hazard.so python -m pytest tests/test_gpu_parity.py -q -s -k "every_output_under_load" 2>&1 | grep "differing\|passed\|failed" | tr '\n' ' '; echo
  done
fi
