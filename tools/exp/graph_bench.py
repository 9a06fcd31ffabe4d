import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, "tests")
import torch
from dex_gan_grasp_amd import synth
import test_gpu_fullsize as T
dev = torch.device("cuda:0")
b = 1024
for mode in ("eager", "graph"):
    e, _, _ = T._engine("wide")
    gb = T._dev_inputs(synth.make_batch(b, 1))
    eb = T._dev_inputs(synth.make_batch(b, 2, with_label=True), with_label=True)
    z = synth.make_z(b, 3).to(dev)
    if mode == "eager":
        step = lambda: e.joint_step(gb, z, eb, True, None, 1, next_gan_inputs=gb)
    else:
        gs = e.capture_joint_step(gb, z, eb, True, seed=5)
        step = gs.replay
    for _ in range(10): step()
    torch.cuda.synchronize()
    ts = []
    for r in range(10):
        t0 = time.perf_counter()
        for _ in range(50): step()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 50 * 1e3)
    print(mode, "ms/step median %.4f min %.4f" % (sorted(ts)[5], min(ts)), flush=True)
    e.close()
