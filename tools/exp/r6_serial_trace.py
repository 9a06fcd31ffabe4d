"""Round 6: kernel trace of the cGAN step ALONE (cgan_step, two_streams) — run under rocprofv3 --kernel-trace; tools/timeline.py
then prints its per-stream launch sequence (compare kernel by kernel with the interleaved step's)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from dex_gan_grasp_amd import _lib, synth, engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = "cuda"
cfg = dict(synth.DEFAULT_CFG)
sd = synth.make_state_dict("dexgangrasp", 0)
fg, fd, fe = (_lib.FlatNet(k, cfg, dev) for k in (_lib.NET_GEN, _lib.NET_DISC, _lib.NET_EVA))
fg.load_state_dict(sd, "generator."); fd.load_state_dict(sd, "discriminator."); fe.load_state_dict(synth.make_state_dict("evaluator", 5))
eng = engine.TrainEngine(cfg, fg, fd, fe, dev)
f = lambda t: t.float().to(dev).contiguous()
gb = [synth.make_batch(B, 1 + 10 * i) for i in range(2)]
g_in = [(f(x["bps_object"]), f(x["rot_matrix"]).reshape(B, 9), f(x["transl"]), f(x["joint_conf"])) for x in gb]
z = synth.make_z(B, 3).to(dev)
for i in range(16):
    eng.cgan_step(*g_in[i % 2], z, True, two_streams=True)
torch.cuda.synchronize()
