#!/usr/bin/env python3
"""Round 6: the interleaved step by batch size, with / without the look-ahead real pass, by the side-work level of the
evaluator update and of the look-ahead pass (DGG_SIDE_EVA_LEVEL / DGG_SIDE_PRE_LEVEL are read per call).  One process, the
settings alternate; ms per step (median of 3 blocks of 200 steps).
    python tools/exp/r6_joint_levels.py"""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from dex_gan_grasp_amd import _lib, engine, synth
dev = torch.device("cuda", 0)
EVA_LEVELS = tuple(int(x) for x in os.environ.get("LEVELS", "0,1,2").split(","))
cfg = dict(synth.DEFAULT_CFG)
fg, fd, fe = (_lib.FlatNet(k, cfg, dev) for k in (_lib.NET_GEN, _lib.NET_DISC, _lib.NET_EVA))
sd = synth.make_state_dict("dexgangrasp", 0)
fg.load_state_dict(sd, "generator."); fd.load_state_dict(sd, "discriminator.")
fe.load_state_dict(synth.make_state_dict("evaluator", 5))
eng = engine.TrainEngine(cfg, fg, fd, fe, dev)
f = lambda t: t.float().to(dev).contiguous()
step = [0]
for B in (1024, 512, 256):
    def gin(seed):
        gb = synth.make_batch(B, seed)
        return (f(gb["bps_object"]), f(gb["rot_matrix"]).reshape(B, 9), f(gb["transl"]), f(gb["joint_conf"]))
    g_ins = [gin(1), gin(11), gin(21)]
    eb = synth.make_batch(B, 2, with_label=True)
    e_in = (f(eb["bps_object"]), f(eb["rot_matrix"]).reshape(B, 9), f(eb["transl"]), f(eb["label"]))
    zs = [synth.make_z(B, 3 + i).to(dev) for i in range(8)]
    def block(n, ahead):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            i = step[0]; step[0] += 1
            eng.joint_step(g_ins[i % 3], zs[i % 8], e_in, True, None, i,
                           next_gan_inputs=g_ins[(i + 1) % 3] if ahead else None)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    for ahead in (True, False):
        res = {}
        for rep in range(3):
            for eva in EVA_LEVELS:
                for pre in ((1, 2) if ahead else (2,)):
                    os.environ["DGG_SIDE_EVA_LEVEL"], os.environ["DGG_SIDE_PRE_LEVEL"] = str(eva), str(pre)
                    eng.flush()
                    block(30, ahead)
                    res.setdefault((eva, pre), []).append(block(200, ahead))
        print("B=%d look-ahead=%s" % (B, ahead), {k: round(statistics.median(v), 4) for k, v in res.items()}, flush=True)
