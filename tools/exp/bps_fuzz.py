"""Random shapes through dgg_bps_encode against the CPU oracle (bit-identical ids and distances expected):
    python3 tools/exp/bps_fuzz.py [configs] [seed]
Shapes: B 1..300, N 1..2600 (about half of them multiples of 4 -> the matrix-core path when B is large enough), P 1..4200,
cloud scale 1e-4 .. 1e3, offsets from the origin, basis points 2 .. 3000 cloud scales away, duplicated points, sometimes
an unaligned cloud pointer."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dex_gan_grasp_amd import _lib  # noqa: E402
from oracle import bps_oracle  # noqa: E402


def main():
    n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    dev = torch.device("cuda:0")
    lib = _lib.lib()
    bad = 0
    for c in range(n_cfg):
        B = int(rng.choice([1, 2, 7, 48, 64, 100, 192, 300]))
        N = int(rng.integers(1, 2601))
        if rng.random() < 0.6:
            N = max(4, N // 4 * 4)
        P = int(rng.choice([1, 31, 130, 1000, 1500, 4096, 4097, 4200]))
        scale = 10.0 ** rng.uniform(-4, 3)
        off = rng.choice([0.0, 0.0, 0.3, 20.0]) * scale
        clouds = (rng.normal(size=(B, N, 3)) * scale * [1.0, 0.6, 1.5] + off).astype(np.float32)
        if N > 8:
            clouds[0, N - 1] = clouds[0, 2]
            clouds[0, 5] = clouds[0, 2]
        span = float(rng.choice([2.0, 2.0, 40.0, 3000.0]))   # basis points up to thousands of cloud radii away (the all-points path)
        basis = (rng.uniform(-span, span, size=(P, 3)) * scale + off * rng.choice([0.0, 1.0])).astype(np.float32)
        unaligned = rng.random() < 0.2
        flat = torch.empty((B * N * 3 + 1,), device=dev)
        cl = flat[1:] if unaligned else flat[:-1]
        cl.copy_(torch.from_numpy(clouds).reshape(-1))
        bs = torch.from_numpy(basis).to(dev)
        d = torch.empty((B, P), device=dev)
        ids = torch.empty((B, P), device=dev, dtype=torch.int32)
        _lib.check(lib.dgg_bps_encode(_lib.ptr(cl), B, N, _lib.ptr(bs), P, _lib.ptr(d), _lib.ptr(ids), _lib.stream_ptr()))
        dn, idn = d.cpu().numpy(), ids.cpu().numpy()
        for b in sorted(set([0, B - 1, int(rng.integers(0, B))])):
            od, oi = bps_oracle.encode_f32(clouds[b], basis)
            ok = (idn[b] == oi).all() and (dn[b].view(np.int32) == od.view(np.int32)).all()
            if not ok:
                bad += 1
                print("MISMATCH cfg %d: B %d N %d P %d scale %.3g off %.3g unaligned %s cloud %d: %d ids differ" %
                      (c, B, N, P, scale, off, unaligned, b, int((idn[b] != oi).sum())))
        print("cfg %2d  B %3d  N %4d  P %4d  scale %8.3g  off %8.3g  basis span %6g  %s ok" % (c, B, N, P, scale, off, span, "unaligned" if unaligned else "aligned  "), flush=True)
    print("mismatching clouds: %d" % bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
