#!/usr/bin/env python3
"""Round 6: is the host ahead of the device in a loop of cgan_step calls (the drop-in order)?  Host time per call against the
device's step time, B = 1024 and 256.    python tools/exp/r6_host_cgan.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from dex_gan_grasp_amd import _lib, engine, synth
dev = torch.device("cuda", 0)
cfg = dict(synth.DEFAULT_CFG)
fg, fd, fe = (_lib.FlatNet(k, cfg, dev) for k in (_lib.NET_GEN, _lib.NET_DISC, _lib.NET_EVA))
sd = synth.make_state_dict("dexgangrasp", 0)
fg.load_state_dict(sd, "generator."); fd.load_state_dict(sd, "discriminator.")
fe.load_state_dict(synth.make_state_dict("evaluator", 5))
eng = engine.TrainEngine(cfg, fg, fd, fe, dev)
f = lambda t: t.float().to(dev).contiguous()
for B in (1024, 256):
    gb = synth.make_batch(B, 1)
    g_in = (f(gb["bps_object"]), f(gb["rot_matrix"]).reshape(B, 9), f(gb["transl"]), f(gb["joint_conf"]))
    z = synth.make_z(B, 3).to(dev)
    for i in range(20):
        eng.cgan_step(*g_in, z, True, two_streams=True)
    torch.cuda.synchronize()
    N = 200
    host = []
    t0 = time.perf_counter()
    for i in range(N):
        a = time.perf_counter()
        eng.cgan_step(*g_in, z, True, two_streams=True)
        host.append(time.perf_counter() - a)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.sort()
    print("B=%d host per call: median %.3f ms min %.3f max %.3f | enqueue loop %.3f ms/step | with final sync %.3f ms/step"
          % (B, host[N // 2] * 1e3, host[0] * 1e3, host[-1] * 1e3, (t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3), flush=True)
