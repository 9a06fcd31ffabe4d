#!/bin/bash
# Builds and runs tools/exp/late_load_probe.hip on the GPU box (both forms of the 12-byte load, two LDS sizes).
set -e
cd "$(dirname "$0")/../.."
for x in 1 0; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -DX3=$x -o /tmp/late_load_probe_$x tools/exp/late_load_probe.hip 2>/dev/null
  /tmp/late_load_probe_$x 2000 32768
  /tmp/late_load_probe_$x 2000 4096
done
