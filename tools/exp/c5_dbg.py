import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, "tests")
import torch, numpy as np
import test_gpu_fullsize as T
import helpers as H
from dex_gan_grasp_amd.models.dexgangrasp import DexGanGrasp
cfg = dict(T.CFG); cfg["is_train"] = False
m = DexGanGrasp(cfg)
m.DexGANGrasp.load_state_dict(H.gan_sd("wide")); m.DexEvaluator.load_state_dict(H.eva_sd("wide"))
n_obj, n = 1250, 200
g = torch.Generator().manual_seed(3)
rows = (torch.rand((n_obj, 4096), generator=g) * 0.2).numpy().astype(np.float32)
z = torch.randn((n_obj * n, 5), generator=g).to("cuda")
out = m.sweep_objects(rows, n, thresholds=(0.5,), z=z)
for o in (0, 1, 617, n_obj - 1):
    orig = torch.randn
    try:
        torch.randn = lambda *a, **k: z[o * n:(o + 1) * n].to(k.get("device", "cpu"))
        gr = m.generate_grasps(rows[o], n, return_arr=True)
    finally:
        torch.randn = orig
    d = np.abs(out["transl"][o].cpu().numpy() - gr["transl"]).max(1)
    print("object", o, "bad rows", np.where(d > 2e-5)[0], "max", d.max(), flush=True)
dn = m.sweep_objects(rows[:8], n, thresholds=(0.5,), z=z[:8 * n], dense=True)
print("dense vs factorised first 8 objects: max", float((dn["transl"] - out["transl"][:8]).abs().max()))
