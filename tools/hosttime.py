#!/usr/bin/env python3
"""Host-side enqueue time of one joint_step vs its GPU time: is the step launch-bound?"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dex_gan_grasp_amd import _lib, engine, parallel, synth

parallel.init_from_env()   # DGG_FORCE_DISTRIBUTED=1: host cost of the collective path (one-rank RCCL group)
dev = torch.device("cuda", 0)
cfg = dict(synth.DEFAULT_CFG)
B = int(os.environ.get("DGG_HOSTTIME_B", "1024"))
fg, fd, fe = (_lib.FlatNet(k, cfg, dev) for k in (_lib.NET_GEN, _lib.NET_DISC, _lib.NET_EVA))
sd = synth.make_state_dict("dexgangrasp", 0)
fg.load_state_dict(sd, "generator."); fd.load_state_dict(sd, "discriminator.")
fe.load_state_dict(synth.make_state_dict("evaluator", 5))
eng = engine.TrainEngine(cfg, fg, fd, fe, dev)
gb = synth.make_batch(B, 1); eb = synth.make_batch(B, 2, with_label=True)
f = lambda t: t.float().to(dev).contiguous()
g_in = (f(gb["bps_object"]), f(gb["rot_matrix"]).reshape(B, 9), f(gb["transl"]), f(gb["joint_conf"]))
e_in = (f(eb["bps_object"]), f(eb["rot_matrix"]).reshape(B, 9), f(eb["transl"]), f(eb["label"]))
z = synth.make_z(B, 3).to(dev)
for i in range(10):
    eng.joint_step(g_in, z, e_in, True, None, i, next_gan_inputs=g_in)
torch.cuda.synchronize()
N = 40
host = []
t0 = time.perf_counter()
for i in range(N):
    a = time.perf_counter()
    eng.joint_step(g_in, z, e_in, True, None, i, next_gan_inputs=g_in)
    host.append(time.perf_counter() - a)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
host.sort()
print("host enqueue per step: median %.3f ms  min %.3f ms   | loop (no sync) %.3f ms/step | with final sync %.3f ms/step"
      % (host[N // 2] * 1e3, host[0] * 1e3, (t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
