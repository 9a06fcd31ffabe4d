#!/bin/bash
# Round 6 experiment 1: what is the in-kernel operand split of gemm_x5_kernel worth?  libdgg_ns{1,2,3}.so = the shipped kernel with
# the split of B / A / both replaced by a bit-cast (garbage results, timing only).  Upper bound for pre-split planes.
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in base ns1 ns2 ns3; do
  if [ $v = base ]; then unset DGG_LIB; else export DGG_LIB=$GRAFT_REPO_ROOT/tools/exp/libdgg_$v.so; fi
  echo "== $v TN"; python3 tools/microbench.py gemm --mode 2 --m 768 --n 4120 --k 1024 --splits 0 --iters 100 | tail -1
  echo "== $v NT"; python3 tools/microbench.py gemm --mode 0 --m 1024 --n 768 --k 4120 --splits 4 --iters 100 | tail -1
done
done
