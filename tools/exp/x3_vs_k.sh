#!/bin/bash
# gemm_x3_kernel duration against K at a fixed, single-round grid (M x N = 2048 x 2048: 256 tiles, no split): slope =
# steady-state cost per k-tile, intercept = launch + ring fill + epilogue.
for mode in 0 2; do
  for k in 128 256 512 1024 2048 4096; do
    python tools/microbench.py gemm --mode $mode --m 2048 --n 2048 --k $k --splits 0 --iters 50 | grep kernel | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mode', d['mode'], 'K', d['K'], 'us', round(d['ms']*1e3,1))"
  done
done
