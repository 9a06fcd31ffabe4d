#!/usr/bin/env python3
"""Which streams share a hardware queue?  (parallel._blocked_behind: a long device-side spin in front of X, a tiny launch
on Y: Y waits = same queue.)  Run with DGG_FORCE_DISTRIBUTED=1 to include RCCL's stream."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
from dex_gan_grasp_amd import parallel
parallel.init_from_env()
dev = torch.device("cuda", 0)
probe = torch.zeros(256, device=dev)
N = 9
S = [torch.cuda.Stream(device=dev) for _ in range(N)]
if dist.is_initialized():
    dist.all_reduce(probe)
spin = 3_000_000
def on(st):
    def run(c):
        with torch.cuda.stream(st):
            torch.cuda._sleep(c)
    return run
def main_block(c):
    torch.cuda._sleep(c)
def rccl_block(c):
    with torch.cuda.stream(S[0]):
        torch.cuda._sleep(c)
        rccl_block.w = dist.all_reduce(probe, async_op=True)
print("rows: candidate stream i; columns: blocker = main, S0..S%d%s; value ms" % (N - 1, ", rccl(behind S0)" if dist.is_initialized() else ""))
for i in range(N):
    row = [parallel._blocked_behind(S[i], main_block, dev, spin)]
    for j in range(N):
        row.append(parallel._blocked_behind(S[i], on(S[j]), dev, spin) if j != i else float("nan"))
    if dist.is_initialized():
        row.append(parallel._blocked_behind(S[i], rccl_block, dev, spin))
    print("S%d: " % i + " ".join("%5.2f" % v for v in row))
