"""High-priority main stream for joint_step: gain, and what it does to cgan_step / eva_step in the same process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, "tests")
import torch
from dex_gan_grasp_amd import synth
import test_gpu_fullsize as T
dev = torch.device("cuda:0")
def wall(f, n=40, rep=5):
    for _ in range(10): f()
    torch.cuda.synchronize(); ts = []
    for _ in range(rep):
        t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n * 1e3)
    return sorted(ts)[len(ts) // 2]
b = 1024
e, _, _ = T._engine("wide")
gb = T._dev_inputs(synth.make_batch(b, 1)); eb = T._dev_inputs(synth.make_batch(b, 2, with_label=True), with_label=True)
z = synth.make_z(b, 3).to(dev)
print("cgan two_streams (before any joint_step): %.4f" % wall(lambda: e.cgan_step(*gb, z, True, two_streams=True)), flush=True)
print("joint_step: %.4f" % wall(lambda: e.joint_step(gb, z, eb, True, None, 1, next_gan_inputs=gb)), flush=True)
print("cgan two_streams (after): %.4f  eva: %.4f" % (wall(lambda: e.cgan_step(*gb, z, True, two_streams=True)), wall(lambda: e.eva_step(*eb, None, 5, two_streams=True))), flush=True)
os.environ["DGG_WGRAD_STREAMS"] = "0"
e.close()
print("cgan two_streams, no wgrad streams (after): %.4f  eva: %.4f" % (wall(lambda: e.cgan_step(*gb, z, True, two_streams=True)), wall(lambda: e.eva_step(*eb, None, 5, two_streams=True))), flush=True)
