import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dex_gan_grasp_amd import _lib
from dex_gan_grasp_amd._lib import lib, ptr, check, stream_ptr
dev = "cuda"
g = torch.Generator().manual_seed(0)
def run(M, N, K, akm, bkm, splits, iters=200):
    A = torch.randn((K, M) if akm else (M, K), generator=g).to(dev)
    B = torch.randn((K, N) if bkm else (N, K), generator=g).to(dev)
    nws = int(lib().dgg_gemm_planes_workspace_bytes(M, N, K))
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    slabs = torch.empty((max(splits, 1) * M * N,), device=dev)
    C = torch.empty((M, N), device=dev)
    for _ in range(iters):
        check(lib().dgg_gemm_planes(akm, bkm, M, N, K, ptr(A), A.stride(0), ptr(B), B.stride(0), None if splits else ptr(C), N, splits,
                                    ptr(slabs) if splits else None, 0, ptr(ws), nws, stream_ptr()), "p3")
    torch.cuda.synchronize()
run(1024, 768, 4120, 0, 0, 4)
run(768, 4120, 1024, 1, 1, 0)
