#!/bin/bash
# Round 6: the real pass' small weight-gradient launches (pose columns N = 24, fc2 rider 512 x 256) as single products, alone
cd $GRAFT_REPO_ROOT
for args in "--m 768 --n 24" "--m 768 --n 512" "--m 512 --n 256" "--m 768 --n 4120"; do
  echo "== TN $args K=1024"; python3 tools/microbench.py gemm --mode 2 $args --k 1024 --splits 0 --iters 50 2>/dev/null | tail -1
  echo "== TN $args K=1024 DGG_X5=0"; DGG_X5=0 python3 tools/microbench.py gemm --mode 2 $args --k 1024 --splits 0 --iters 50 2>/dev/null | tail -1
done
