"""Kernel-trace timing of the fused fc2 forward GEMM inside a D forward (B = 1024), for ablation builds (DGG_LIB=)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_fullsize as T
from dex_gan_grasp_amd import synth
L, ops, engine = T._mods()
eng, _, _ = T._engine("wide")
b = 1024
bt = synth.make_batch(b, 7)
f = lambda t: t.float().to("cuda").contiguous()
args = (f(bt["bps_object"]), f(bt["rot_matrix"]).reshape(b, 9), f(bt["transl"]), f(bt["joint_conf"]))
for _ in range(30):
    ops.disc_forward(eng.D, *args, True)
torch.cuda.synchronize()
