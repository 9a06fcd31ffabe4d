# Round 6, VERDICT item 7: what fraction of (32 basis points x 32 cloud points) tiles could bounding boxes of Morton-sorted
# basis blocks / point tiles skip?  CPU simulation on the bench's clouds and basis (no GPU).  usage: python3 tools/exp/bps_prune_sim.py
import numpy as np, torch, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from dex_gan_grasp_amd import synth
def morton(p, lo, hi):
    q = ((p - lo) / (hi - lo) * 1023).clip(0,1023).astype(np.int64)
    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    return spread(q[:,0]) | (spread(q[:,1])<<1) | (spread(q[:,2])<<2)
g = torch.Generator(); g.manual_seed(7)
basis = (torch.rand((4096,3), generator=g)*0.2-0.1).numpy()
bs = basis[np.argsort(morton(basis, -0.1, 0.1))]
clouds = synth.make_clouds(8, 2048, 3).numpy()
for TP, TB in ((32,32),(16,32),(32,16),(64,32)):
    fr=[]
    for c in clouds:
        lo, hi = c.min(), c.max()
        cs = c[np.argsort(morton(c, lo, hi))]
        pt = cs.reshape(-1, TP, 3); plo, phi = pt.min(1), pt.max(1)
        bb = bs.reshape(-1, TB, 3); blo, bhi = bb.min(1), bb.max(1)
        # mindist between boxes
        gap = np.maximum(0, np.maximum(blo[:,None]-phi[None], plo[None]-bhi[:,None]))
        mind2 = (gap**2).sum(-1)
        far = np.maximum(np.abs(bhi[:,None]-plo[None]), np.abs(phi[None]-blo[:,None]))
        maxd2 = (far**2).sum(-1)
        ub = maxd2.min(1, keepdims=True)
        keep = mind2 <= ub
        # tighter: ub = max over basis pts in block of true NN dist
        d2 = ((bs[:,None,:]-cs[None,:,:])**2).sum(-1).min(1).reshape(-1,TB).max(1)
        keep2 = mind2 <= d2[:,None]
        fr.append((keep.mean(), keep2.mean()))
    print(TP, TB, np.mean(fr,0))
