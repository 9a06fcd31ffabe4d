#!/bin/bash
# Timing builds of the fused fc2 stage (gemm_f32_kernel<64,64,1,1,BNA>): which part of its 18 us is what.
#   bash tools/exp/fc2_ablate.sh build     (here)      bash tools/exp/fc2_ablate.sh run   (GPU box)
cd "$(dirname "$0")/../.."
VARS="base:-DDGG_ABLATE=0 nostat_pro:-DDGG_FC2_ABLATE=1 nostat_epi:-DDGG_FC2_ABLATE=2 noload:-DDGG_ABLATE=1 nolds:-DDGG_ABLATE=2 nobar:-DDGG_ABLATE=3 nordlds:-DDGG_ABLATE=4"
if [ "$1" = "build" ]; then
  cd dex_gan_grasp_amd/csrc
  for v in $VARS; do n=${v%%:*}; f=${v#*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $f -c gemm_f32.hip -o /tmp/gf_$n.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/exp/libdgg_fc2_$n.so /tmp/gf_$n.o _obj/gemm_x4.o _obj/gemm_x5.o _obj/elementwise.o _obj/second_order.o _obj/bps_encode.o _obj/dgg_api.o
  done; ls tools/exp/libdgg_fc2_* 2>/dev/null; ls ../../tools/exp/libdgg_fc2_*
else
  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
  for v in $VARS; do n=${v%%:*}
    DGG_LIB=$PWD/tools/exp/libdgg_fc2_$n.so rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fc2_$n -o t -- python3 tools/exp/step_prof.py eva 1024 12 > gpurun_out/fc2_$n.log 2>&1
    echo "$n: $(python3 tools/exp/ktrace_groups.py gpurun_out/fc2_$n 'true, true, true' | cut -c60-150)"
  done
fi
