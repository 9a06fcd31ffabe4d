#!/usr/bin/env python3
"""Multi-rank check of the data-parallel train step, run as

    DGG_REHEARSE=1 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 \
        --master-port P tools/dp_check.py [engine|wrapper]

DGG_REHEARSE=1: every rank drives cuda:0 and the collectives go over gloo, so the multi-rank code path of
engine.TrainEngine.joint_step (async all-reduce of the three flat buckets, wait() ordering against the library's
side streams, cross-step prefetch) executes on a 1-GPU box.  Without it the same script runs one rank per GPU over
RCCL.

engine : 3 joint_steps (with prefetch) on contiguous shards of a global batch; every rank's parameters are compared
         with the data-parallel oracle (oracle on each shard -> mean of the gradients -> one Adam), and with each
         other.
wrapper: the drop-in DexGanGrasp wrapper under torchrun: replicas are identical after construction although every rank
         seeds its own init, and stay identical over update_dexgangrasp / update_dexevaluator steps on different shards.
DGG_FORCE_DISTRIBUTED=1 python tools/dp_check.py engine   (no launcher): ONE rank with a real RCCL process group — the
collective path as it runs on a multi-GPU node (RCCL's stream, the lent feeder streams, the work handles in front of the
optimiser launches), checked against the same oracle.
Exit code 0 = all checks passed on this rank.
"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

from dex_gan_grasp_amd import _lib, engine, parallel, synth  # noqa: E402


def _all_equal(t, what):
    """Every rank holds the same tensor."""
    w = parallel.world_size()
    t = t.detach().contiguous()
    if dist.get_backend() != "nccl":       # gloo gathers host tensors only; RCCL device tensors only
        t = t.cpu()
    parts = [torch.zeros_like(t) for _ in range(w)]
    dist.all_gather(parts, t)
    for r in range(1, w):
        assert torch.equal(parts[0], parts[r]), "%s differs between rank 0 and rank %d" % (what, r)


def check_engine(per_rank=64, steps=3):
    import helpers as H
    from oracle import dexgan_oracle as O
    rank, world, idx = parallel.init_from_env()
    assert parallel.is_distributed() and (world > 1 or parallel.forced())
    dev = torch.device("cuda", idx)
    cfg = dict(synth.DEFAULT_CFG)
    B = per_rank * world
    sd0 = H.gan_sd("wide")
    sde0 = H.eva_sd("wide")
    fg, fd, fe = (_lib.FlatNet(k, cfg, dev) for k in (_lib.NET_GEN, _lib.NET_DISC, _lib.NET_EVA))
    fg.load_state_dict(sd0, "generator.")
    fd.load_state_dict(sd0, "discriminator.")
    fe.load_state_dict(sde0)
    eng = engine.TrainEngine(cfg, fg, fd, fe, dev)
    # the data-parallel oracle, computed in full on every rank (CPU, small)
    sds = [{k: v.clone() for k, v in sd0.items()} for _ in range(world)]
    sdes = [{k: v.clone() for k, v in sde0.items()} for _ in range(world)]
    ag = O.AdamState([n for n in O.param_names(sd0) if n.startswith("generator.")])
    ad = O.AdamState([n for n in O.param_names(sd0) if n.startswith("discriminator.")])
    ae = O.AdamState(O.param_names(sde0))
    f = lambda t: t.float().to(dev).contiguous()

    def gan_in(seed):
        gb = synth.make_batch(B, seed)
        sh = parallel.shard_batch(gb)
        return gb, (f(sh["bps_object"]), f(sh["rot_matrix"]).reshape(per_rank, 9), f(sh["transl"]), f(sh["joint_conf"]))
    batches = [gan_in(1), gan_in(11)]
    eb = synth.make_batch(B, 2, with_label=True)
    esh = parallel.shard_batch(eb)
    e_in = (f(esh["bps_object"]), f(esh["rot_matrix"]).reshape(per_rank, 9), f(esh["transl"]), f(esh["label"]))
    for i in range(steps):
        gb, g_in = batches[i % 2]
        nxt = batches[(i + 1) % 2][1] if i + 1 < steps else None
        z = synth.make_z(B, 100 + i)
        masks = synth.make_dropout_masks(B, 50 + i)
        zs = [parallel.shard_rows(z, r, world) for r in range(world)]
        ms = [masks[:, r * per_rank:(r + 1) * per_rank].contiguous() for r in range(world)]
        prev, preve = {k: v.clone() for k, v in sds[0].items()}, {k: v.clone() for k, v in sdes[0].items()}
        outs, _, _ = O.cgan_step_dp(sds, [parallel.shard_batch(gb, r, world) for r in range(world)], zs, ag, ad, cfg)
        oute, _ = O.eva_step_dp(sdes, [parallel.shard_batch(eb, r, world) for r in range(world)], ae, cfg, ms)
        got = eng.joint_step(g_in, zs[rank].to(dev), e_in, True, ms[rank].to(dev), 0, next_gan_inputs=nxt)
        torch.cuda.synchronize()
        got = got.cpu().numpy()
        names = ["total_loss_disc", "bce_loss_real", "bce_loss_fake", "gen_loss_fake", "transl_loss", "rot_loss",
                 "conf_loss", "total_loss_gen"]
        for k, n in enumerate(names):   # losses are per-rank (its shard), like the reference's per-replica loss
            want = float(outs[rank][n])
            assert abs(got[k] - want) <= 1e-4 * abs(want) + 1e-6, (i, n, got[k], want)
        want = float(oute[rank]["total_loss_eva"])
        assert abs(got[_lib.L_TOTAL_EVA] - want) <= 1e-4 * abs(want) + 1e-6, (i, got[_lib.L_TOTAL_EVA], want)
        # parameters after the step: this rank vs the data-parallel oracle
        for flat, sd, pv, pre in ((fg, sds[0], prev, "generator."), (fd, sds[0], prev, "discriminator."),
                                  (fe, sdes[0], preve, "")):
            cur = flat.state_dict()
            for n, t in cur.items():
                if "running_" in n:
                    continue
                H.assert_update_close(n, t, sd[pre + n], pv[pre + n], 0)
            # next step starts from identical state on both sides (see tests/test_gpu_steps._sync_from_oracle):
            # here the ORACLE adopts the engine's parameters and moments, so the prefetched pass stays valid
            adam = ag if pre == "generator." else ad if pre == "discriminator." else ae
            pvw, mv, vv = flat.views("params"), flat.views("m"), flat.views("v")
            for n in pvw:
                for s in (sds if pre else sdes):
                    s[pre + n].copy_(pvw[n].cpu())
                adam.m[pre + n] = mv[n].cpu().clone()
                adam.v[pre + n] = vv[n].cpu().clone()
        # replicas identical, bit for bit
        for flat, nm in ((fg, "G"), (fd, "D"), (fe, "E")):
            _all_equal(flat.params, nm + ".params")
            _all_equal(flat.m, nm + ".m")
    # BatchNorm running statistics are per rank (DDP semantics): this rank's equal ITS oracle copy
    for flat, s, pre in ((fg, sds[rank], "generator."), (fd, sds[rank], "discriminator."), (fe, sdes[rank], "")):
        for n, t in flat.state_dict().items():
            if "running_var" in n or (n.endswith("running_mean") and "rb" not in n):
                r = s[pre + n]
                assert bool(((t - r).abs() <= 1e-4 * r.abs() + 1e-5 * float(r.pow(2).mean().sqrt()) + 1e-7).all()), n
    eng.close()
    print("dp_check engine: rank %d ok" % rank, flush=True)


def check_wrapper(per_rank=32, steps=2):
    from dex_gan_grasp_amd.models.dexgangrasp import DexGanGrasp
    rank, world, _ = parallel.env_rank_world()
    torch.manual_seed(1234 + rank)            # every rank draws a DIFFERENT init: the wrapper must reconcile them
    cfg = dict(synth.DEFAULT_CFG)
    cfg["train_ffhevaluator"] = True
    m = DexGanGrasp(cfg)
    assert parallel.is_distributed() and parallel.world_size() == world
    assert m.device.index == parallel.local_device_index() == torch.cuda.current_device()
    mods = (("G", m.DexGANGrasp.generator), ("D", m.DexGANGrasp.discriminator), ("E", m.DexEvaluator))
    for nm, mod in mods:
        _all_equal(mod._require_flat().params, nm + " after init")
    B = per_rank * world
    for i in range(steps):
        gb = parallel.shard_batch(synth.make_batch(B, 20 + i))
        eb = parallel.shard_batch(synth.make_batch(B, 30 + i, with_label=True))
        l1 = m.update_dexgangrasp(gb)
        l2 = m.update_dexevaluator(eb)
        assert all(torch.isfinite(v).item() for v in list(l1.values()) + list(l2.values()))
        for nm, mod in mods:
            _all_equal(mod._require_flat().params, "%s after step %d" % (nm, i))
    # the ranks saw different shards: their losses differ, their parameters do not
    t = torch.stack([l1["total_loss_gen"].detach().float(), l2["total_loss_eva"].detach().float()])
    if dist.get_backend() != "nccl":
        t = t.cpu()
    parts = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    assert not torch.equal(parts[0], parts[1])
    print("dp_check wrapper: rank %d ok" % rank, flush=True)


def check_sweep(n_obj=37, n=64):
    """Config 5 sharded by object: every rank sweeps its contiguous share with its own replica; the gathered counts
    equal a single-rank sweep of all objects (same latents)."""
    import helpers as H
    from dex_gan_grasp_amd.models.dexgangrasp import DexGanGrasp
    rank, world, _ = parallel.env_rank_world()
    cfg = dict(synth.DEFAULT_CFG)
    cfg["is_train"] = False
    m = DexGanGrasp(cfg)
    m.DexGANGrasp.load_state_dict(H.gan_sd("wide"))
    m.DexEvaluator.load_state_dict(H.eva_sd("wide"))
    g = torch.Generator().manual_seed(3)
    rows = torch.rand((n_obj, 4096), generator=g) * 0.2
    z = torch.randn((n_obj * n, 5), generator=g).to(m.device)
    full = m.sweep_objects(rows, n, thresholds=(0.3, 0.5, 0.7), z=z)
    local, (lo, hi), count_all = m.sweep_objects_sharded(rows, n, thresholds=(0.3, 0.5, 0.7), z=z)
    per = (n_obj + world - 1) // world
    assert (lo, hi) == (rank * per, min(n_obj, (rank + 1) * per))
    assert tuple(count_all.shape) == (3, n_obj)
    near = sum(int(((full["score"] > t - 1e-5) & (full["score"] < t + 1e-5)).sum()) for t in (0.3, 0.5, 0.7))
    assert int((count_all.cpu() - full["count"].cpu()).abs().sum()) <= near
    # another batch size picks another split-K plan: same scores up to the summation order
    assert float((local["score"] - full["score"][lo:hi]).abs().max()) <= 2e-6
    assert int(count_all.sum()) > 0
    print("dp_check sweep: rank %d ok" % rank, flush=True)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "engine"
    if mode == "engine":
        check_engine(per_rank=int(os.environ.get("DGG_DP_PER_RANK", "64")))   # (rows per rank: smaller for many ranks)
    else:
        {"wrapper": check_wrapper, "sweep": check_sweep}[mode]()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
