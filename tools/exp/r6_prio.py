#!/usr/bin/env python3
"""Round 6: stream priorities once more.  The interleaved B = 1024 step with the chain on the default stream or on a
high-priority stream, the side streams probed (normal priority) or created with the lowest priority (DGG_SIDE_LOW=1, an
experiment switch in parallel.side_streams_or_plain).  One setting per process: ms per step, median of 5 blocks of 200.
    python tools/exp/r6_prio.py {default|high}"""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from dex_gan_grasp_amd import _lib, engine, synth, parallel
dev = torch.device("cuda", 0)
cfg = dict(synth.DEFAULT_CFG)
B = 1024
fg, fd, fe = (_lib.FlatNet(k, cfg, dev) for k in (_lib.NET_GEN, _lib.NET_DISC, _lib.NET_EVA))
sd = synth.make_state_dict("dexgangrasp", 0)
fg.load_state_dict(sd, "generator."); fd.load_state_dict(sd, "discriminator.")
fe.load_state_dict(synth.make_state_dict("evaluator", 5))
eng = engine.TrainEngine(cfg, fg, fd, fe, dev)
f = lambda t: t.float().to(dev).contiguous()
def gin(seed):
    gb = synth.make_batch(B, seed)
    return (f(gb["bps_object"]), f(gb["rot_matrix"]).reshape(B, 9), f(gb["transl"]), f(gb["joint_conf"]))
g_ins = [gin(1), gin(11), gin(21)]
eb = synth.make_batch(B, 2, with_label=True)
e_in = (f(eb["bps_object"]), f(eb["rot_matrix"]).reshape(B, 9), f(eb["transl"]), f(eb["label"]))
zs = [synth.make_z(B, 3 + i).to(dev) for i in range(8)]
step = [0]
def block(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        i = step[0]; step[0] += 1
        eng.joint_step(g_ins[i % 3], zs[i % 8], e_in, True, None, i, next_gan_inputs=g_ins[(i + 1) % 3])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
mode = sys.argv[1] if len(sys.argv) > 1 else "default"
main = torch.cuda.Stream(device=dev, priority=-1) if mode == "high" else torch.cuda.current_stream(dev)
torch.cuda.synchronize()
with torch.cuda.stream(main):
    block(50)
    ts = [block(200) for _ in range(5)]
    eng.flush()
torch.cuda.synchronize()
print("main=%s side_low=%s" % (mode, os.environ.get("DGG_SIDE_LOW", "0")), round(statistics.median(ts), 4), [round(t, 4) for t in ts], parallel.pick_report, flush=True)
