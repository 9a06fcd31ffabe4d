#!/bin/bash
# gemm_x4_kernel against gemm_x3_kernel (DGG_X4=0) on the training shapes: forward NT (split-K 4), dgrad NN, weight gradient TN
for x in 1 0; do
  echo "DGG_X4=$x"
  DGG_X4=$x python3 tools/microbench.py gemm --k 4120 --splits 4 --iters 50 | tail -1
  DGG_X4=$x python3 tools/microbench.py gemm --mode 2 --m 768 --n 4120 --k 1024 --splits 0 --iters 50 | tail -1
  DGG_X4=$x python3 tools/microbench.py gemm --mode 1 --m 1024 --n 512 --k 768 --splits 3 --iters 50 | tail -1
  DGG_X4=$x python3 tools/microbench.py gemm --m 2048 --n 2048 --k 4096 --splits 0 --iters 20 | tail -1
done
