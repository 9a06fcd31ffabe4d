"""Data-parallel plumbing: one process per GPU, batch dimension sharded contiguously across ranks,
one flat gradient bucket per network all-reduced (sum) over RCCL (torch.distributed backend
"nccl" on ROCm) between backward and the fused Adam, which applies the 1/world_size scale
(SURVEY.md 8(e)).  BatchNorm statistics stay per-rank (what DDP does).  The reference has no working
multi-GPU behaviour to match (its nn.DataParallel path dereferences attributes the wrapper does not
proxy, dexgangrasp.py:127-129 vs :770-772), so this defines it.

The same code runs on the CPU with the gloo backend (tests/test_parallel_cpu.py, world_size 2).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's environment (no-op for WORLD_SIZE=1)."""
    rank, world, local = env_rank_world()
    if world > 1 and not (dist.is_available() and dist.is_initialized()):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_size():
    return dist.get_world_size() if is_distributed() else 1


def rank():
    return dist.get_rank() if is_distributed() else 0


def shard_rows(t, r=None, w=None):
    """Contiguous shard r of w along dim 0 (global batch must divide evenly)."""
    r = rank() if r is None else r
    w = world_size() if w is None else w
    n = t.shape[0]
    if n % w != 0:
        raise ValueError("global batch %d is not divisible by world size %d" % (n, w))
    per = n // w
    return t[r * per:(r + 1) * per]


def shard_batch(batch, r=None, w=None):
    return {k: (shard_rows(v, r, w) if torch.is_tensor(v) else v) for k, v in batch.items()}


def allreduce_sum_(flat_grads, async_op=False):
    """In-place sum over ranks of one flat gradient bucket (27.9 / 28.0 / 42.6 MB fp32 for G / D / E).
    Returns the work handle when async_op (so it can overlap with independent compute)."""
    if not is_distributed():
        return None
    return dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, async_op=async_op)


def grad_scale():
    """Factor the fused Adam applies to the summed gradients: mean over the global batch."""
    return 1.0 / float(world_size())
