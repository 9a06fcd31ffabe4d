"""World-size-2 gloo test of the data-parallel path (dex_gan_grasp_amd/parallel.py): batch sharding,
flat-bucket all-reduce and the 1/world gradient scale reproduce, on every rank, the update that the
average of the per-shard gradients gives.  The per-shard compute here is the CPU oracle (the HIP
engine cannot run without a GPU); the plumbing under test is the code the GPU engine calls."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers as H
from dex_gan_grasp_amd import parallel, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _flat(grads, names):
    return torch.cat([grads[n].reshape(-1) for n in names])


def _worker(rank, world, port, out_dir):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank),
                       "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank)})
    torch.set_num_threads(2)
    from oracle import dexgan_oracle as O
    r, w, _ = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and parallel.is_distributed() and parallel.world_size() == world
    batch = synth.make_batch(16, 2, with_label=True)
    shard = parallel.shard_batch(batch)
    assert shard["bps_object"].shape[0] == 8
    assert torch.equal(shard["transl"], batch["transl"][rank * 8:(rank + 1) * 8])
    sde = H.eva_sd("wide")
    names = O.param_names(sde)
    masks = synth.make_dropout_masks(16, 3)[:, rank * 8:(rank + 1) * 8]
    for n in names:
        sde[n].requires_grad_(True)
    p = O.evaluator_forward(sde, shard["bps_object"], shard["rot_matrix"], shard["transl"], True, masks)
    loss, _ = O.loss_evaluator(p, shard["label"].float().unsqueeze(-1))
    gs = dict(zip(names, torch.autograd.grad(loss, [sde[n] for n in names])))
    bucket = _flat(gs, names).clone()
    local = bucket.clone()
    work = parallel.allreduce_sum_(bucket, async_op=True)
    work.wait()
    scaled = bucket * parallel.grad_scale()
    torch.save({"local": local, "reduced": scaled}, os.path.join(out_dir, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = torch.load(os.path.join(str(tmp_path), "r0.pt"))
    b = torch.load(os.path.join(str(tmp_path), "r1.pt"))
    assert torch.equal(a["reduced"], b["reduced"])                      # every rank holds the same bucket
    want = (a["local"] + b["local"]) / 2                                # mean of the per-shard gradients
    np.testing.assert_allclose(a["reduced"].numpy(), want.numpy(), rtol=1e-6, atol=1e-12)
    assert not torch.equal(a["local"], b["local"])


def test_single_process_is_a_noop():
    assert parallel.world_size() == 1 and parallel.rank() == 0 and parallel.grad_scale() == 1.0
    t = torch.ones(4)
    assert parallel.allreduce_sum_(t) is None and torch.equal(t, torch.ones(4))
    with pytest.raises(ValueError):
        parallel.shard_rows(torch.zeros(5, 2), 0, 2)


def _gather_worker(rank, world, port, out_dir):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank),
                       "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank)})
    parallel.init_from_env(backend="gloo")
    t = torch.arange((rank + 2) * 3, dtype=torch.int64).view(rank + 2, 3) + 100 * rank     # ragged shards
    g = parallel.gather_counts(t)
    torch.save(g, os.path.join(out_dir, "g%d.pt" % rank))
    b = torch.full((5,), float(rank))
    parallel.broadcast_(b, src=0)
    assert float(b.sum()) == 0.0
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_of_ragged_counts(tmp_path):
    """The collective of the object-sharded inference sweep: per-rank count blocks of different
    lengths arrive concatenated in rank order on every rank."""
    port = _free_port()
    mp.spawn(_gather_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = torch.load(os.path.join(str(tmp_path), "g0.pt"))
    b = torch.load(os.path.join(str(tmp_path), "g1.pt"))
    want = torch.cat([torch.arange(6).view(2, 3), torch.arange(9).view(3, 3) + 100])
    assert torch.equal(a, want) and torch.equal(b, want)
