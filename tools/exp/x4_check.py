"""Where does a GEMM kernel go wrong?  Integer data, mode 0 / 1 / 2, a sweep over K and — for the failing K — the k-tiles
that contribute wrongly (A restricted to one k-tile at a time)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dex_gan_grasp_amd import ops
dev = "cuda"
g = torch.Generator().manual_seed(1)


def run(mode, M, N, K, A, Bm):
    a_dev = (A if mode in (0, 1) else A.t().contiguous()).to(dev)
    b_dev = (Bm.t().contiguous() if mode == 0 else Bm).to(dev)
    return ops.gemm(mode, a_dev, b_dev, M, N, K).cpu().double()


for mode in (0, 1, 2):
    for (M, N, K) in [(256, 768, 64), (256, 768, 96), (256, 768, 128), (256, 768, 160), (256, 768, 1024), (256, 768, 4096), (256, 768, 4101), (256, 768, 4120)]:
        A = torch.randint(-3, 4, (M, K), generator=g).float()
        Bm = torch.randint(-3, 4, (K, N), generator=g).float()
        ref = A.double() @ Bm.double()
        out = run(mode, M, N, K, A, Bm)
        bad = (out != ref)
        msg = "mode %d K %5d: %d wrong of %d" % (mode, K, int(bad.sum()), bad.numel())
        if bad.any():
            rows = bad.any(1).nonzero().flatten()
            cols = bad.any(0).nonzero().flatten()
            msg += "  rows %d..%d (%d)  cols %d..%d (%d)" % (rows[0], rows[-1], len(rows), cols[0], cols[-1], len(cols))
            wrong_tiles = []
            for t in range((K + 31) // 32):
                At = torch.zeros_like(A)
                At[:, 32 * t:32 * t + 32] = A[:, 32 * t:32 * t + 32]
                o = run(mode, M, N, K, At, Bm)
                if (o != At.double() @ Bm.double()).any():
                    wrong_tiles.append(t)
                if len(wrong_tiles) > 12:
                    break
            msg += "  k-tiles wrong alone: %s" % wrong_tiles
        print(msg, flush=True)
