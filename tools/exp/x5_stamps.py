#!/usr/bin/env python3
"""Diagnostic (DGG_X5_STAMP build only): per-wave cycle stamps of gemm_x5_kernel.
   bash tools/exp/x5_stamp_build.sh && DGG_LIB=tools/exp/libdgg_x5stamp.so python3 tools/exp/x5_stamps.py [mode M N K splits]"""
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
for _ in range(200):
    run()
torch.cuda.synchronize()
nb = ((M + 127) // 128) * ((N + 127) // 128) * max(S, 1)
buf = (C.c_ulonglong * (nb * 32))()
assert raw.dgg_x5_debug_stamps(buf, nb) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 4, 8).astype(np.float64)
tot = a[:, :, 6] - a[:, :, 0]
real = (a[:, :, 7] - a[:, :, 1]) * 10e-9
clk = tot / real / 1e9
med = lambda x: float(np.median(x))
print("workgroups %d; clock GHz median %.3f (min %.3f max %.3f)" % (nb, med(clk), clk.min(), clk.max()))
print("cycles (median over waves): setup+prologue %.0f | k loop %.0f | wait at the barrier %.0f | reduction %.0f | epilogue %.0f | total %.0f" % (
    med(a[:, :, 2] - a[:, :, 0]), med(a[:, :, 3] - a[:, :, 2]), med(a[:, :, 4] - a[:, :, 3]), med(a[:, :, 5] - a[:, :, 4]),
    med(a[:, :, 6] - a[:, :, 5]), med(tot)))
for w in range(4):
    print("  wave %d: prologue %.0f loop %.0f barrier wait %.0f reduce %.0f epilogue %.0f" % (
        w, med(a[:, w, 2] - a[:, w, 0]), med(a[:, w, 3] - a[:, w, 2]), med(a[:, w, 4] - a[:, w, 3]), med(a[:, w, 5] - a[:, w, 4]), med(a[:, w, 6] - a[:, w, 5])))
print("per-workgroup wall us: median %.2f max %.2f; launch span us %.2f" % (med(real.max(axis=1)) * 1e6, real.max() * 1e6,
      (a[:, :, 7].max() - a[:, :, 1].min()) * 10e-9 * 1e6))
