"""WGAN-GP step (B = 1024) alone, for the profiler.  Run it as the profiler's own child — never through env, a shell or a
shebang (rocprofv3 has initialised the GPU by then; an exec behind it takes the box down):
    cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d <outdir> -o t -- python3 <repo>/tools/exp/wgan_prof.py
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from dex_gan_grasp_amd import _lib, engine, synth
dev = torch.device("cuda", 0)
cfg = dict(synth.DEFAULT_CFG)
B = 1024
fg, fd, fe = (_lib.FlatNet(k, cfg, dev) for k in (_lib.NET_GEN, _lib.NET_DISC, _lib.NET_EVA))
sd = synth.make_state_dict("dexgangrasp", 0)
fg.load_state_dict(sd, "generator."); fd.load_state_dict(sd, "discriminator.")
fe.load_state_dict(synth.make_state_dict("evaluator", 5))
eng = engine.TrainEngine(cfg, fg, fd, fe, dev)
f = lambda t: t.float().to(dev).contiguous()
gb = synth.make_batch(B, 1)
g_in = (f(gb["bps_object"]), f(gb["rot_matrix"]).reshape(B, 9), f(gb["transl"]), f(gb["joint_conf"]))
z = synth.make_z(B, 3).to(dev)
al = torch.rand((B, 3), generator=torch.Generator().manual_seed(5)).to(dev)
for i in range(5):
    eng.wgan_step(*g_in, z, al, True)
torch.cuda.synchronize(); t0 = time.perf_counter()
N = 20
for i in range(N):
    eng.wgan_step(*g_in, z, al, True)
torch.cuda.synchronize(); print("wgan step %.3f ms" % ((time.perf_counter() - t0) / N * 1e3))
