#!/usr/bin/env python3
"""Kernel micro-benchmarks used for the rocprofv3 runs under profiles/ (GEMM roofline shape, BPS encode).
    python tools/microbench.py gemm [--iters N]     # the 128x128 NT GEMM kernel (bf16x3 by default, DGG_GEMM_ALGO=f32 for the f32 MFMA) M=1024 N=768 K=4613
    python tools/microbench.py bps  [--iters N]     # bps_mfma_kernel: 1024 clouds x 2048 points x 4096 basis (DGG_BPS=valu: the scan)
"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dex_gan_grasp_amd import _lib, ops, synth  # noqa: E402


def time_it(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        fn()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["gemm", "bps", "bps1"])
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--k", type=int, default=4613)
    ap.add_argument("--m", type=int, default=1024)
    ap.add_argument("--n", type=int, default=768)
    ap.add_argument("--mode", type=int, default=0)
    ap.add_argument("--splits", type=int, default=10)
    a = ap.parse_args()
    dev = "cuda"
    if a.what == "gemm":
        M, N, K, S = a.m, a.n, a.k, a.splits
        g = torch.Generator().manual_seed(0)
        # mode 0: A[M][K] B[N][K];  1: A[M][K] B[K][N];  2: A[K][M] B[K][N]
        A = (torch.randn((M, K), generator=g) if a.mode != 2 else torch.randn((K, M), generator=g)).to(dev)
        W = (torch.randn((N, K), generator=g) if a.mode == 0 else torch.randn((K, N), generator=g)).to(dev)
        slabs = torch.empty((max(S, 1) * M * N,), device=dev)
        out = torch.empty((M, N), device=dev)
        lib = _lib.lib()
        print("occupancy (WG/CU) NT128 %d NT64 %d TN128 %d TN64 %d" % (lib.dgg_gemm_occupancy(0, 128, 128),
              lib.dgg_gemm_occupancy(0, 64, 64), lib.dgg_gemm_occupancy(2, 128, 128), lib.dgg_gemm_occupancy(2, 64, 64)))
        fn = lambda: _lib.check(lib.dgg_gemm_f32(a.mode, M, N, K, _lib.ptr(A), A.stride(0), _lib.ptr(W), W.stride(0),
                                                 None if S >= 1 else _lib.ptr(out), N, S, _lib.ptr(slabs), 0,
                                                 _lib.stream_ptr()))
        ms = time_it(fn, a.iters)
        fl = 2.0 * M * N * K
        print(json.dumps({"kernel": ("gemm_x5_kernel" if lib.dgg_get_gemm_algo() >= 1 else "gemm_f32_kernel<128,128,NT>"), "mode": a.mode, "M": M, "N": N, "K": K, "splits": S, "ms": ms,
                          "tflops": fl / ms / 1e9, "algorithmic_bytes": 4 * (M * K + N * K + S * M * N)}))
    else:
        B, Np = (1024, 2048) if a.what == "bps" else (1, 73000)
        basis = torch.rand((4096, 3)) * 0.2 - 0.1
        clouds = synth.make_clouds(B, Np, 7).to(dev)
        bd = basis.to(dev)
        d = torch.empty((B, 4096), device=dev)
        i = torch.empty((B, 4096), dtype=torch.int32, device=dev)
        lib = _lib.lib()
        fn = lambda: _lib.check(lib.dgg_bps_encode(_lib.ptr(clouds), B, Np, _lib.ptr(bd), 4096, _lib.ptr(d), _lib.ptr(i),
                                                   _lib.stream_ptr()))
        ms = time_it(fn, a.iters)
        pairs = B * Np * 4096.0
        byts = B * (Np * 12 + 4096 * 8) + 4096 * 12
        print(json.dumps({"kernel": "bps_encode", "B": B, "N": Np, "ms": ms, "gpairs_per_s": pairs / ms / 1e6,
                          "clouds_per_s": B / ms * 1e3, "algorithmic_bytes": byts, "gb_per_s": byts / ms / 1e6,
                          "valu_tflops_at_8_flop_per_pair": pairs * 8 / ms / 1e9}))


if __name__ == "__main__":
    main()
