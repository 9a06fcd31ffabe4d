#!/bin/bash
# build: tools/exp/x4_ablate.sh build   (here, no GPU)  -> scratch/libdgg_x4ab{1..4}.so
# run:   tools/exp/x4_ablate.sh         (GPU box)
cd "$(dirname "$0")/../.."
if [ "$1" = build ]; then
  mkdir -p scratch
  O=dex_gan_grasp_amd/csrc/_obj
  for n in 1 2 3 4; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DDGG_X4_ABLATE=$n -c dex_gan_grasp_amd/csrc/gemm_x4.hip -o scratch/x4ab$n.o &
  done
  wait
  for n in 1 2 3 4; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/libdgg_x4ab$n.so $O/gemm_f32.o $O/gemm_x5.o scratch/x4ab$n.o $O/elementwise.o $O/second_order.o $O/bps_encode.o $O/dgg_api.o
  done
  ls -la scratch/*.so
  exit 0
fi
for n in 0 1 2 3 4; do
  L=dex_gan_grasp_amd/libdgg.so; [ $n != 0 ] && L=scratch/libdgg_x4ab$n.so
  echo "ablate $n (1 no split, 2 no MFMA, 3 no global loads, 4 no LDS stores)"
  DGG_LIB=$PWD/$L python3 tools/microbench.py gemm --k 4120 --splits 4 --iters 50 | tail -1 | cut -c1-120
  DGG_LIB=$PWD/$L python3 tools/microbench.py gemm --mode 2 --m 768 --n 4120 --k 1024 --splits 0 --iters 50 | tail -1 | cut -c1-120
  DGG_LIB=$PWD/$L python3 tools/microbench.py gemm --m 2048 --n 2048 --k 4096 --splits 0 --iters 20 | tail -1 | cut -c1-120
done
