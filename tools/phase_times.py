#!/usr/bin/env python3
"""Where the benched step spends its time, from timed events at the phase boundaries (no profiler attached: the
kernel-trace tools slow the host down enough to starve the streams and show gaps that a normal run does not have).
   python tools/phase_times.py [B]"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dex_gan_grasp_amd import _lib, engine, synth
dev = torch.device("cuda", 0)
cfg = dict(synth.DEFAULT_CFG)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
fg, fd, fe = (_lib.FlatNet(k, cfg, dev) for k in (_lib.NET_GEN, _lib.NET_DISC, _lib.NET_EVA))
sd = synth.make_state_dict("dexgangrasp", 0)
fg.load_state_dict(sd, "generator."); fd.load_state_dict(sd, "discriminator.")
fe.load_state_dict(synth.make_state_dict("evaluator", 5))
eng = engine.TrainEngine(cfg, fg, fd, fe, dev)
f = lambda t: t.float().to(dev).contiguous()
def gin(seed):
    gb = synth.make_batch(B, seed)
    return (f(gb["bps_object"]), f(gb["rot_matrix"]).reshape(B, 9), f(gb["transl"]), f(gb["joint_conf"]))
g_ins = [gin(1), gin(11)]
eb = synth.make_batch(B, 2, with_label=True)
e_in = (f(eb["bps_object"]), f(eb["rot_matrix"]).reshape(B, 9), f(eb["transl"]), f(eb["label"]))
z = synth.make_z(B, 3).to(dev)
PRE = os.environ.get("DGG_PREFETCH", "1") != "0"
nxt = lambda i: g_ins[(i + 1) % 2] if PRE else None
PRIO = int(os.environ.get("DGG_MAIN_PRIO", "0"))
if PRIO != 0:
    print("stream priority range (least, greatest):", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")
    hp = torch.cuda.Stream(device=dev, priority=PRIO)
    torch.cuda.set_stream(hp)
for i in range(20):
    eng.joint_step(g_ins[i % 2], z, e_in, True, None, i, next_gan_inputs=nxt(i))
torch.cuda.synchronize()
eng.trace = []
N = 40
for i in range(N):
    eng.joint_step(g_ins[i % 2], z, e_in, True, None, i, next_gan_inputs=nxt(i))
torch.cuda.synchronize()
tr = eng.trace
per = len(tr) // N
acc = collections.OrderedDict()
for s in range(2, N - 1):
    ev = dict(tr[s * per:(s + 1) * per])
    nxt = dict(tr[(s + 1) * per:(s + 2) * per])
    t0 = ev["step_begin"]
    for name in ("d_phase_end", "adam_d_end", "g_phase_end", "adam_g_end", "e_phase_end", "adam_e_end"):
        acc.setdefault(name, []).append(t0.elapsed_time(ev[name]) * 1e3)
    acc.setdefault("next_step_begin", []).append(t0.elapsed_time(nxt["step_begin"]) * 1e3)
print("B = %d; microseconds after the step's first launch (median of %d steps)" % (B, N - 3))
for k, v in acc.items():
    v.sort()
    print("  %-16s %8.1f" % (k, v[len(v) // 2]))
