"""gemm_x5_kernel against the f32-MFMA kernel of gemm_f32.hip (DGG_X5=0 in a child process; until round 5 that switch
selected gemm_x3_kernel, which this script was written to beat): error against fp64 and HIP-event time of the train step's
three large shapes (+ ragged ones).  python3 tools/exp/x5_time.py [check|time]"""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dex_gan_grasp_amd import _lib, ops

DEV = "cuda:0"
SHAPES = [  # mode, M, N, K, splits, what
    (0, 1024, 768, 4120, 4, "forward NT"),
    (2, 768, 4120, 1024, 0, "wgrad TN"),
    (1, 1024, 512, 768, 3, "dgrad NN"),
    (0, 1024, 768, 512, 0, "hidden segment NT"),
    (0, 300, 768, 2003, 3, "ragged NT"),
    (2, 768, 4101, 300, 0, "ragged TN"),
    (1, 1300, 1664, 77, 0, "ragged NN"),
    (0, 250, 130, 64, 0, "K = 64"),
    (0, 1024, 768, 4101, 4, "forward NT K=4101"),
]


def run_one(mode, M, N, K, splits, iters):
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn((M, K), generator=g)
    Bm = torch.randn((K, N), generator=g)
    a = (A if mode in (0, 1) else A.t().contiguous()).to(DEV)
    b = (Bm.t().contiguous() if mode == 0 else Bm).to(DEV)
    out = ops.gemm(mode, a, b, M, N, K, splits=splits)
    ref = A.double() @ Bm.double()
    scale = A.abs().double() @ Bm.abs().double()
    err = ((out.cpu().double() - ref).abs() / scale).max().item()
    bad = int(((out.cpu().double() - ref).abs() > 4e-7 * scale).sum())
    t = None
    if iters:
        for _ in range(3):
            ops.gemm(mode, a, b, M, N, K, splits=splits)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(5):
            e0.record()
            for _ in range(iters):
                ops.gemm(mode, a, b, M, N, K, splits=splits)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / iters)
        t = sorted(ts)[len(ts) // 2]
    return err, bad, t


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "time"
    if os.environ.get("X5_CHILD") is None:
        env = dict(os.environ, X5_CHILD="1", DGG_X5="0")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), what], env=env, capture_output=True, text=True)
        print("---- DGG_X5=0 (f32-MFMA kernel, gemm_f32.hip)")
        print(r.stdout, r.stderr[-2000:])
        print("---- default (gemm_x5_kernel)")
    for mode, M, N, K, splits, name in SHAPES:
        err, bad, t = run_one(mode, M, N, K, splits, 20 if what == "time" else 0)
        fl = 2.0 * M * N * K
        print("%-22s mode %d %5d x %5d x %5d splits %d: err %.2e of sum|a||b|, %d beyond 4e-7%s" % (
            name, mode, M, N, K, splits, err, bad,
            "" if t is None else ", %.1f us = %.1f TFLOP/s (%.3f of 416.7)" % (t, fl / t * 1e-6, fl / t * 1e-6 / 416.7)), flush=True)


if __name__ == "__main__":
    main()
