"""gemm_p3 (pre-split planes) vs fp64 reference and timing vs the in-register-split kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dex_gan_grasp_amd import _lib
from dex_gan_grasp_amd._lib import lib, ptr, check, stream_ptr
dev = "cuda"
g = torch.Generator().manual_seed(0)
def run(M, N, K, akm, bkm, splits, iters=0):
    A = torch.randn((K, M) if akm else (M, K), generator=g).to(dev)
    B = torch.randn((K, N) if bkm else (N, K), generator=g).to(dev)
    C = torch.empty((M, N), device=dev)
    nws = int(lib().dgg_gemm_planes_workspace_bytes(M, N, K))
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    slabs = torch.empty((max(splits, 1) * M * N,), device=dev)
    call = lambda: check(lib().dgg_gemm_planes(akm, bkm, M, N, K, ptr(A), A.stride(0), ptr(B), B.stride(0), ptr(C), N, splits,
                                               ptr(slabs) if splits else None, 0, ptr(ws), nws, stream_ptr()), "p3")
    call()
    torch.cuda.synchronize()
    ref = ((A.double().t() if akm else A.double()) @ (B.double() if bkm else B.double().t()))
    err = float((C.double() - ref).abs().max() / ref.abs().max())
    print("M %d N %d K %d akm %d bkm %d splits %d: max rel err %.2e" % (M, N, K, akm, bkm, splits, err), flush=True)
    assert err < 2e-6, err
for args in [(128, 128, 32, 0, 0, 0), (128, 128, 64, 0, 0, 0), (128, 128, 96, 0, 0, 0), (128, 128, 128, 0, 0, 0), (256, 384, 4120, 0, 0, 0),
             (1024, 768, 4120, 0, 0, 4), (300, 200, 77, 0, 0, 0), (768, 4120, 1024, 1, 1, 0), (130, 515, 300, 1, 1, 0), (1024, 768, 517, 0, 0, 2)]:
    run(*args)
print("ALL OK")
