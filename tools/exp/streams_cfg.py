"""cgan_step / eva_step alone under the stream configurations (DGG_WGRAD_STREAMS = 1 | main | 0)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, "tests")
import torch
from dex_gan_grasp_amd import synth
import test_gpu_fullsize as T
dev = torch.device("cuda:0")
def wall(f, n=40, rep=7):
    for _ in range(10): f()
    torch.cuda.synchronize(); ts = []
    for _ in range(rep):
        t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n * 1e3)
    return sorted(ts)[len(ts) // 2]
for b in (1024, 256):
    e, _, _ = T._engine("wide")
    gb = T._dev_inputs(synth.make_batch(b, 1)); eb = T._dev_inputs(synth.make_batch(b, 2, with_label=True), with_label=True)
    z = synth.make_z(b, 3).to(dev)
    for ts in (True, False):
        print("B %d two_streams %s: cgan %.4f ms  eva %.4f ms" % (b, ts, wall(lambda: e.cgan_step(*gb, z, True, two_streams=ts)),
              wall(lambda: e.eva_step(*eb, None, 5, two_streams=ts))), flush=True)
    e.close()
