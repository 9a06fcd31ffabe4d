#!/usr/bin/env python3
"""Diagnostic (DGG_X3_STAMP build only): per-workgroup cycle stamps of gemm_x3_kernel.
   DGG_LIB=tools/exp/libdgg_stamp.so python tools/exp/stamps.py [mode M N K splits]"""
import ctypes as C, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dex_gan_grasp_amd import _lib
mode, M, N, K, S = (int(x) for x in (sys.argv[1:6] if len(sys.argv) >= 6 else (0, 1024, 768, 4120, 4)))
dev = "cuda"
g = torch.Generator().manual_seed(0)
A = (torch.randn((M, K), generator=g) if mode != 2 else torch.randn((K, M), generator=g)).to(dev)
W = (torch.randn((N, K), generator=g) if mode == 0 else torch.randn((K, N), generator=g)).to(dev)
slabs = torch.empty((max(S, 1) * M * N,), device=dev)
out = torch.empty((M, N), device=dev)
lib = _lib.lib()
raw = C.CDLL(_lib.LIB_PATH)
def run():
    _lib.check(lib.dgg_gemm_f32(mode, M, N, K, _lib.ptr(A), A.stride(0), _lib.ptr(W), W.stride(0), None if S >= 1 else _lib.ptr(out), N, S, _lib.ptr(slabs), 0, _lib.stream_ptr()))
for _ in range(200):   # long enough for the clock to settle under load
    run()
torch.cuda.synchronize()
nb = ((M + 127) // 128) * ((N + 127) // 128) * max(S, 1)
buf = (C.c_ulonglong * (nb * 8))()
assert raw.dgg_x3_debug_stamps(buf, nb) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 8).astype(np.float64)
cyc_total = a[:, 4] - a[:, 0]
real = (a[:, 5] - a[:, 1]) * 10e-9     # 100 MHz ticks
clk = cyc_total / real / 1e9
nkt = a[:, 6]
loop = (a[:, 3] - a[:, 2]) / nkt
print("blocks %d  nkt %s" % (nb, np.unique(nkt)))
print("clock GHz: median %.3f  min %.3f max %.3f" % (np.median(clk), clk.min(), clk.max()))
print("cycles: prologue %.0f  main loop %.0f (%.0f per k-tile)  epilogue %.0f  total %.0f" % (
    np.median(a[:, 2] - a[:, 0]), np.median(a[:, 3] - a[:, 2]), np.median(loop), np.median(a[:, 4] - a[:, 3]), np.median(cyc_total)))
print("per-WG wall us: median %.2f max %.2f ; launch span us %.2f" % (np.median(real) * 1e6, real.max() * 1e6,
      (a[:, 5].max() - a[:, 1].min()) * 10e-9 * 1e6))
x = a[:, 7].astype(int)
print("blocks per XCC:", np.bincount(x, minlength=8))
