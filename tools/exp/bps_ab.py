"""BPS encode A/B: writes ids / distance bits of one seeded batch to an .npz; run once per arithmetic and compare.

    DGG_BPS=valu python3 tools/exp/bps_ab.py run gpurun_out/bps_valu.npz [B N]
    python3 tools/exp/bps_ab.py run gpurun_out/bps_mfma.npz [B N]
    python3 tools/exp/bps_ab.py cmp gpurun_out/bps_valu.npz gpurun_out/bps_mfma.npz [x B N]
cmp lists the first differing outputs and checks each side's (id, distance) pair against the contract's arithmetic on the
CPU — a distance that is not the distance to its id came from mixed-up data, not from a tie.
Also prints the kernel time (events over 10 launches).  The clouds: synth.make_clouds plus duplicated points, a cloud of
identical points, a near-degenerate cloud (all points within 1e-6 of each other) and one far-away cloud.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def make_clouds(B, N):
    from dex_gan_grasp_amd import synth
    clouds = synth.make_clouds(B, N, 31337)
    clouds[0, 7] = clouds[0, 3]
    clouds[0, N - 1] = clouds[0, 3]
    if B > 4:
        clouds[1, :] = clouds[1, 0]
        clouds[2] = clouds[2, :1] + 1e-6 * torch.randn((N, 3), generator=torch.Generator().manual_seed(5))
        clouds[3, : N // 2] = 3e19
        clouds[4] = clouds[4] + 50.0
    return clouds


def run(out, B=1024, N=2048):
    from dex_gan_grasp_amd import _lib
    dev = torch.device("cuda:0")
    basis = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "bps_real_objects.npz"))["basis"]
    clouds = make_clouds(B, N)
    cl = clouds.to(dev).contiguous()
    bs = torch.from_numpy(basis).to(dev)
    P = basis.shape[0]
    d = torch.empty((B, P), device=dev)
    ids = torch.empty((B, P), device=dev, dtype=torch.int32)
    lib = _lib.lib()
    fn = lambda: _lib.check(lib.dgg_bps_encode(_lib.ptr(cl), B, N, _lib.ptr(bs), P, _lib.ptr(d), _lib.ptr(ids), _lib.stream_ptr()))
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("DGG_BPS=%s  %d x %d: %.4f ms per launch" % (os.environ.get("DGG_BPS", ""), B, N, e0.elapsed_time(e1) / 10))
    np.savez(out, ids=ids.cpu().numpy(), dist=d.cpu().numpy().view(np.int32))


def cmp(a, b):
    A, Bz = np.load(a), np.load(b)
    ni = int((A["ids"] != Bz["ids"]).sum())
    nd = int((A["dist"] != Bz["dist"]).sum())
    print("ids differ: %d   distance bits differ: %d   of %d" % (ni, nd, A["ids"].size))
    if ni or nd:
        w = np.argwhere((A["ids"] != Bz["ids"]) | (A["dist"] != Bz["dist"]))[:10]
        print(w, A["ids"][tuple(w.T)], Bz["ids"][tuple(w.T)])
        # is each side's (id, distance) pair self-consistent?  (the contract's arithmetic on the CPU)
        from dex_gan_grasp_amd import synth
        import torch
        B, N = A["ids"].shape[0], int(sys.argv[5]) if len(sys.argv) > 5 else 2048
        clouds = make_clouds(B, N)
        basis = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "bps_real_objects.npz"))["basis"]
        f = np.float32
        for b, i in w:
            for name, Z in (("a", A), ("b", Bz)):
                j = int(Z["ids"][b, i])
                d = clouds[b, j].numpy().astype(f) - basis[i]
                d2 = f(np.float64(d[2]) * np.float64(d[2]) + np.float64(f(np.float64(d[1]) * np.float64(d[1]) + np.float64(f(d[0] * d[0])))))
                print(name, "cloud", b, "basis", i, "id", j, "dist bits", hex(int(Z["dist"][b, i]) & 0xffffffff), "sqrt(d2(id)) bits",
                      hex(int(np.sqrt(d2).astype(f).view(np.int32)) & 0xffffffff))
        sys.exit(1)


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], *[int(x) for x in sys.argv[3:5]])
    else:
        cmp(sys.argv[2], sys.argv[3])
