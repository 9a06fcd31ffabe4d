"""Config 4's per-rank loop (512 clouds x 2048 points -> encode -> joint step, batch 512) with the encode of batch k + 1 on a
side stream: plain / low-priority side stream.   python3 tools/exp/cfg4_prof.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dex_gan_grasp_amd import _lib, synth, engine
from dex_gan_grasp_amd.data.synthetic_data_set import SyntheticGraspBatches
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda", 0)
cfg = dict(synth.DEFAULT_CFG)
b = 512
sd, sde = synth.make_state_dict("dexgangrasp", 0, 0.02, False), synth.make_state_dict("evaluator", 5, 0.02, False)
fg, fd, fe = (_lib.FlatNet(k, cfg, dev) for k in (_lib.NET_GEN, _lib.NET_DISC, _lib.NET_EVA))
fg.load_state_dict(sd, "generator."); fd.load_state_dict(sd, "discriminator."); fe.load_state_dict(sde)
eng = engine.TrainEngine(cfg, fg, fd, fe, dev)
basis = (torch.rand((4096, 3), generator=torch.Generator().manual_seed(0)) * 0.2 - 0.1).to(dev)
ds = SyntheticGraspBatches(cfg, b, 4, basis, n_points=2048, device=dev, with_label=True, device_tensors=True)
clouds = ds.clouds(0)
bt = ds.batch(0)
g_in = (bt["bps_object"], bt["rot_matrix"].reshape(b, 9).contiguous(), bt["transl"], bt["joint_conf"])
e_in = (bt["bps_object"], bt["rot_matrix"].reshape(b, 9).contiguous(), bt["transl"], bt["label"].float())
z = synth.make_z(b, 3).to(dev)

def wall(fn, n):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

k = [0]
def step_only():
    k[0] += 1
    eng.joint_step(g_in, z, e_in, True, None, 500 + k[0])
def serial():
    k[0] += 1
    bps = ds.encode(clouds)
    eng.joint_step((bps,) + g_in[1:], z, (bps,) + e_in[1:], True, None, 500 + k[0])
print("step only            %.1f us" % (wall(step_only, steps) * 1e6))
print("encode then step     %.1f us" % (wall(serial, steps) * 1e6))
for name, prio in (("plain side stream   ", 0), ("low-priority stream ", None)):
    lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
    side = torch.cuda.Stream(device=dev) if prio == 0 else torch.cuda.Stream(device=dev, priority=lo)
    pend = [None]
    def enqueue():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            bps = ds.encode(clouds)
            ev = torch.cuda.Event(); ev.record(side)
        bps.record_stream(torch.cuda.current_stream())
        pend[0] = (bps, ev)
    def ahead():
        k[0] += 1
        bps, ev = pend[0]
        torch.cuda.current_stream().wait_event(ev)
        enqueue()
        eng.joint_step((bps,) + g_in[1:], z, (bps,) + e_in[1:], True, None, 500 + k[0])
    enqueue()
    print("%s %.1f us  (priority range %s)" % (name, wall(ahead, steps) * 1e6, (lo, hi)))
    torch.cuda.synchronize()
