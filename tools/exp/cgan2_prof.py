"""cgan_step(two_streams=True) N times, for rocprofv3 (what bench.py's config 2 times): python3 tools/exp/cgan2_prof.py [B] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dex_gan_grasp_amd import _lib, synth, engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = "cuda"
cfg = dict(synth.DEFAULT_CFG)
sd, sde = synth.make_state_dict("dexgangrasp", 0, 0.02, False), synth.make_state_dict("evaluator", 5, 0.02, False)
fg, fd, fe = (_lib.FlatNet(k, cfg, dev) for k in (_lib.NET_GEN, _lib.NET_DISC, _lib.NET_EVA))
fg.load_state_dict(sd, "generator."); fd.load_state_dict(sd, "discriminator."); fe.load_state_dict(sde)
eng = engine.TrainEngine(cfg, fg, fd, fe, dev)
f = lambda t: t.float().to(dev).contiguous()
gb = [synth.make_batch(B, 1 + 10 * i) for i in range(2)]
g_in = [(f(x["bps_object"]), f(x["rot_matrix"]).reshape(B, 9), f(x["transl"]), f(x["joint_conf"])) for x in gb]
z = synth.make_z(B, 3).to(dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(steps + 3):
    if i == 3:
        torch.cuda.synchronize(); e0.record()
    eng.cgan_step(*g_in[i % 2], z, True, two_streams=True)
e1.record(); torch.cuda.synchronize()
print("cgan two_streams B=%d: %.1f us per step" % (B, e0.elapsed_time(e1) / steps * 1e3), flush=True)
