import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from dex_gan_grasp_amd import _lib, synth, engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = "cuda"
cfg = dict(synth.DEFAULT_CFG)
sd, sde = synth.make_state_dict("dexgangrasp", 0, 0.02, False), synth.make_state_dict("evaluator", 5, 0.02, False)
fg, fd, fe = (_lib.FlatNet(k, cfg, dev) for k in (_lib.NET_GEN, _lib.NET_DISC, _lib.NET_EVA))
fg.load_state_dict(sd, "generator."); fd.load_state_dict(sd, "discriminator."); fe.load_state_dict(sde)
eng = engine.TrainEngine(cfg, fg, fd, fe, dev)
f = lambda t: t.float().to(dev).contiguous()
gb = [synth.make_batch(B, 1 + 10 * i) for i in range(2)]
g_in = [(f(x["bps_object"]), f(x["rot_matrix"]).reshape(B, 9), f(x["transl"]), f(x["joint_conf"])) for x in gb]
eb = synth.make_batch(B, 2, with_label=True)
e_in = (f(eb["bps_object"]), f(eb["rot_matrix"]).reshape(B, 9), f(eb["transl"]), f(eb["label"]))
z = synth.make_z(B, 3).to(dev)
def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(steps + 5):
        if i == 5:
            torch.cuda.synchronize(); e0.record()
        fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3
print("cgan two_streams B=%d: %.1f us" % (B, timed(lambda i: eng.cgan_step(*g_in[i % 2], z, True, two_streams=True))), flush=True)
print("eva  two_streams B=%d: %.1f us" % (B, timed(lambda i: eng.eva_step(*e_in, None, 100 + i, two_streams=True))), flush=True)
