#!/bin/bash
# A / B inside one gpurun call: builds the library of another revision (default HEAD) next to the working tree's, as
# tools/exp/_ab/libdgg_base.so (git-ignored, travels to the GPU box); select it with DGG_LIB=tools/exp/_ab/libdgg_base.so.
set -e
REV=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
TMP=$(mktemp -d /tmp/dgg_base.XXXX)
git -C "$ROOT" archive "$REV" dex_gan_grasp_amd/csrc include | tar -x -C "$TMP"
bash "$TMP/dex_gan_grasp_amd/csrc/build.sh" > "$TMP/build.log" 2>&1 || { tail -20 "$TMP/build.log"; exit 1; }
mkdir -p "$ROOT/tools/exp/_ab"
cp "$TMP/dex_gan_grasp_amd/libdgg.so" "$ROOT/tools/exp/_ab/libdgg_base.so"
rm -rf "$TMP"
echo "built tools/exp/_ab/libdgg_base.so from $REV"
