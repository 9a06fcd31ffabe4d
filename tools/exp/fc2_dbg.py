import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, "tests")
import torch, numpy as np
import test_gpu_fullsize as T
from dex_gan_grasp_amd import synth, _lib
L, ops, engine = T._mods()
eng, sd, sde = T._engine("wide")
out = {}
for b in (200, 1000, 51200):
    bt = synth.make_batch(b, 7, with_label=True)
    f = lambda t: t.float().to("cuda").contiguous()
    z = synth.make_z(b, 9).to("cuda")
    o = ops.gen_forward(eng.G, z, f(bt["bps_object"]), False)
    out["g%d" % b] = o.cpu().numpy()
    e = ops.eva_forward(eng.E, f(bt["bps_object"]), f(bt["rot_matrix"]).reshape(b, 9), f(bt["transl"]), False)
    out["e%d" % b] = e.cpu().numpy()
np.savez(sys.argv[1], **out)
