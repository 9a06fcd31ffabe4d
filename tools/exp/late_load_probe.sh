#!/bin/bash
# Builds and runs tools/exp/late_load_probe.hip on the GPU box (both forms of the 12-byte load, two LDS sizes).
set -e
cd "$(dirname "$0")/../.."
for f in "-DX3=1" "-DX3=0" "-DX3=1 -DCLOBBER=1" "-DX3=0 -DCLOBBER=1"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $f -o /tmp/late_load_probe tools/exp/late_load_probe.hip 2>/dev/null
  /tmp/late_load_probe ${ITERS:-2000} 32768 | tail -2
  /tmp/late_load_probe ${ITERS:-2000} 4096 | tail -2
done
