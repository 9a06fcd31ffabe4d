#!/usr/bin/env python3
"""Soak run of the benched step: N joint steps (prefetch on, fresh latents), then checks — finite losses and
parameters, device memory unchanged after the first 100 steps, step time of the last block within 10 % of the first.
    python tools/soak.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dex_gan_grasp_amd import _lib, engine, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
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
def block(n, i0):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(i0, i0 + n):
        eng.joint_step(g_ins[i % 3], zs[i % 8], e_in, True, None, i, next_gan_inputs=g_ins[(i + 1) % 3])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
first = block(100, 0)
mem0 = torch.cuda.memory_allocated(dev), torch.cuda.mem_get_info(dev)[0]
done, last = 100, first
while done < N:
    n = min(1000, N - done)
    last = block(n, done)
    done += n
    print("steps %d: %.4f ms/step" % (done, last), flush=True)
mem1 = torch.cuda.memory_allocated(dev), torch.cuda.mem_get_info(dev)[0]
losses = eng.losses.cpu()
assert torch.isfinite(losses[:10]).all(), losses
for nm, fl in (("G", fg), ("D", fd), ("E", fe)):
    assert torch.isfinite(fl.params).all() and torch.isfinite(fl.m).all() and torch.isfinite(fl.v).all(), nm
assert mem1[0] == mem0[0], (mem0, mem1)
assert abs(mem1[1] - mem0[1]) < 64 << 20, (mem0, mem1)
assert last < 1.1 * first, (first, last)
print("soak ok: %d steps, %.4f -> %.4f ms/step, allocated %.1f MB, losses %s" % (N, first, last, mem1[0] / 2**20, [round(float(x), 4) for x in losses[:9]]))
