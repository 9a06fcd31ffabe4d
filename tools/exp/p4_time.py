"""Planes GEMM timing shapes (run under rocprofv3 --kernel-trace; the round-3 comparison with gemm_p3_kernel is in profiles/r03_experiments.txt):
    rocprofv3 --kernel-trace --stats -d gpurun_out/p4 -o p4 -- python3 tools/exp/p4_time.py
Shapes: the per-ResBlock forward (1024 x 768 x 4120, split-K 4 / 5) and weight gradient (768 x 4120 x 1024), and the
products of all ResBlocks of a network over the shared BPS block in one launch (N = 1536 / 2304 / 3072, K = 4096)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dex_gan_grasp_amd._lib import lib, ptr, check, stream_ptr
dev = "cuda"
g = torch.Generator().manual_seed(0)


def run(M, N, K, akm, bkm, splits, iters=100):
    A = torch.randn((K, M) if akm else (M, K), generator=g).to(dev)
    B = torch.randn((K, N) if bkm else (N, K), generator=g).to(dev)
    nws = int(lib().dgg_gemm_planes_workspace_bytes(M, N, K))
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    slabs = torch.empty((max(splits, 1) * M * N,), device=dev)
    C = torch.empty((M, N), device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(iters + 5):
        if it == 5:
            e0.record()
        check(lib().dgg_gemm_planes(akm, bkm, M, N, K, ptr(A), A.stride(0), ptr(B), B.stride(0), None, N, splits,
                                    ptr(slabs), 0, ptr(ws), nws, stream_ptr()), "planes")
    e1.record()
    torch.cuda.synchronize()
    print("M %d N %d K %d splits %d: %.1f us per call (incl. the two planes launches)" % (M, N, K, splits, e0.elapsed_time(e1) / iters * 1e3), flush=True)


shapes = [(1024, 768, 4120, 0, 0, 4), (1024, 768, 4120, 0, 0, 5), (768, 4120, 1024, 1, 1, 1), (1024, 1536, 4096, 0, 0, 2),
          (1024, 1536, 4096, 0, 0, 5), (1024, 1536, 4096, 0, 0, 8), (1024, 3072, 4096, 0, 0, 1), (1024, 3072, 4096, 0, 0, 4),
          (1024, 2304, 4096, 0, 0, 5), (1024, 2304, 4096, 0, 0, 7)]
for s in shapes:
    run(*s)
