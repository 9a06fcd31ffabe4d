#!/usr/bin/env python3
"""bench.py — grasps/sec of the gen+eval train step (BASELINE.json metric), batch 1024 / BPS-4096 per GPU.

A "step" = one DexGanGrasp.update_dexgangrasp (D phase, Adam(D), G phase, Adam(G)) + one
update_dexevaluator (fwd, BCE, bwd, Adam) on one batch of synthetic inputs already resident in HBM
(dexgangrasp.py:739-823).  value = global_batch * steps / time, max over ranks, weak scaling (batch
1024 per GPU; gradients all-reduced over RCCL when N > 1).

    python bench.py [--gpus N --steps K --warmup W]        # N > 1: launched by torch.distributed.run

Prints ONE JSON line with the contract fields + "roofline" (dominant kernel: the fp32-MFMA GEMM,
timed live with HIP events on the launch stream) + "cpu_baseline" (the CPU oracle = plain-PyTorch
restatement of the reference step, timed on this host's cores, rank 0, N = 1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from dex_gan_grasp_amd import _lib, engine, ops, parallel, synth  # noqa: E402

MFLOP_PER_GRASP_FULL = 231.38   # cGAN 167.51 + evaluator 63.87 as the reference executes it
# What this implementation executes: no input-gradient GEMM of the BN'd BPS columns (the
# input-BN parameter gradients come out of the wgrad epilogue), no discriminator dW in the G phase:
# MACs/sample G 14.6 M + D 36.9 M + E 22.5 M = 74.0 M  ->  148 MFLOP
MFLOP_PER_GRASP_EXEC = 135.4   # 148.0 minus the BPS-column product that D's fake pass takes from the real pass
# the next step's batch is resident (as with a prefetching loader): its D real pass is enqueued one step ahead
PIPELINE = os.environ.get("DGG_PIPELINE", "1") == "1"
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense


def gemm_roofline(dev, iters=20):
    """Dominant kernel: gemm_x3_kernel<NT, 16-byte staging> on the fused fc1||fc3 layer of a ResBlock at the bench
    batch, shaped as the discriminator's second ResBlock runs it: C[1024,768] = X0[1024,4120] W13[768,0:4120]^T with
    the weight rows 4632 floats apart, split-K 4 as the forward plans it (48 tiles x 4 = 192 one-per-CU workgroups; the rest of the chip is left to the other streams of the step).
    achieved = algorithmic FLOPs (2*M*N*K, fp32-equivalent) / mean launch duration (HIP events on the launch stream).
    The kernel computes every fp32 product as six bf16 partial products on v_mfma_f32_32x32x16_bf16, so the matrix-
    pipe bound for it is the dense bf16 peak / 6; the native f32-MFMA peak is reported next to it."""
    M, N, K, LDB = 1024, 768, 4120, 4632
    g = torch.Generator().manual_seed(0)
    A = torch.randn((M, K), generator=g).to(dev)
    W = torch.randn((N, LDB), generator=g).to(dev)
    splits = 4
    slabs = torch.empty((splits * M * N,), device=dev)
    lib = _lib.lib()
    st = torch.cuda.current_stream()
    algo = int(lib.dgg_get_gemm_algo())

    def launch():
        # mode 0 (NT), split-K partial products into slabs; C = NULL leaves out the (separate, HBM-bound)
        # bias/BN-statistics reduce kernel so that the MFMA kernel alone is timed
        _lib.check(lib.dgg_gemm_f32(0, M, N, K, _lib.ptr(A), K, _lib.ptr(W), LDB, None, N, splits, _lib.ptr(slabs), 0,
                                    C.c_void_p(st.cuda_stream)))
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        launch()
    e1.record(st)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * M * N * K
    achieved = flops / (ms * 1e-3) / 1e12
    if algo == 1:
        peak, kern = PEAK_BF16_MFMA_TFLOPS / 6.0, "gemm_x3_kernel<NT,vec16,vec16> (bf16x3: 6 bf16 MFMA products per fp32 product)"
    else:
        peak, kern = PEAK_F32_MFMA_TFLOPS, "gemm_f32_kernel<128,128,NT>"
    # HBM-side bytes per launch of this kernel at this shape come from separate rocprofv3 --pmc passes (FETCH_SIZE,
    # WRITE_SIZE with the gfx950 correction), recorded in profiles/; PMC counters cannot be read from inside a run
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_gemm_x3_pmc.json")))
        if algo == 1 and pmc["shape"] == {"M": M, "N": N, "K": K, "splits": splits}:
            traffic = int(pmc["traffic_bytes"])
    except Exception:
        traffic = None
    return {"bound": "mfma", "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": traffic, "algorithmic_bytes": 4 * (M * K + N * K + splits * M * N),
            "kernel": kern + " M=1024 N=768 K=4120 split-K 4 (fc1||fc3 of a ResBlock)",
            "peak_basis": "dense bf16 MFMA 2500 TFLOP/s / 6 products" if algo == 1 else "dense f32 MFMA",
            "f32_mfma_peak": PEAK_F32_MFMA_TFLOPS, "frac_of_f32_mfma_peak": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
            "ms_per_launch": round(ms, 4)}


def usable_cores():
    """Cores this process may actually use: affinity mask, capped by the cgroup CPU quota (a GPU
    box exposes all host cores through os.cpu_count() but schedules only its share)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(round(int(q) / int(p)))))
    except Exception:
        pass
    if n > 64:  # no quota visible: the documented CPU share of a 1-GPU box is 16 cores
        n = 16
    return n


def cpu_baseline(cfg, budget_s=20.0):
    """The CPU oracle (port of the reference step) on this host's cores, bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import dexgan_oracle as O
    cores = usable_cores()
    torch.set_num_threads(cores)
    b = 256
    sd = synth.make_state_dict("dexgangrasp", 0)
    sde = synth.make_state_dict("evaluator", 5)
    batch = synth.make_batch(b, 1, with_label=True)
    z = synth.make_z(b, 3)
    masks = synth.make_dropout_masks(b, 4)
    ag = O.AdamState([n for n in O.param_names(sd) if n.startswith("generator.")])
    ad = O.AdamState([n for n in O.param_names(sd) if n.startswith("discriminator.")])
    ae = O.AdamState(O.param_names(sde))
    def step():
        O.cgan_step(sd, batch, z, ag, ad, cfg, True)
        O.eva_step(sde, batch, ae, cfg, masks)
    step()
    t0 = time.time()
    n = 0
    while True:
        step()
        n += 1
        if time.time() - t0 > budget_s or n >= 50:
            break
    dt = time.time() - t0
    return {"value": round(b * n / dt, 1), "unit": "grasps/s", "cores": cores, "kind": "port",
            "sample": "%d gen+eval steps at batch %d (same step, smaller batch so the sample stays within ~%ds)" % (n, b, int(budget_s))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1024, help="per-GPU batch (BASELINE: 1024)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    # DGG_REHEARSE=1: rehearsal of the multi-rank code path on a 1-GPU box (gloo backend, all ranks on cuda:0);
    # never used for reported numbers
    rehearse = os.environ.get("DGG_REHEARSE") == "1"
    rank, world, local = parallel.init_from_env(backend="gloo" if rehearse else None)
    assert world == args.gpus or world == 1 and args.gpus == 1, "launch with torch.distributed.run for --gpus > 1"
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cfg = dict(synth.DEFAULT_CFG)
    B = args.batch

    fg, fd, fe = (_lib.FlatNet(k, cfg, dev) for k in (_lib.NET_GEN, _lib.NET_DISC, _lib.NET_EVA))
    sd = synth.make_state_dict("dexgangrasp", 0)       # reference-style init (xavier 0.02), same on every rank
    fg.load_state_dict(sd, "generator.")
    fd.load_state_dict(sd, "discriminator.")
    fe.load_state_dict(synth.make_state_dict("evaluator", 5))
    eng = engine.TrainEngine(cfg, fg, fd, fe, dev)
    # synthetic batches of this rank's shard (seeds offset by rank), resident in HBM before timing
    gb = synth.make_batch(B, 1 + 100 * rank)
    eb = synth.make_batch(B, 2 + 100 * rank, with_label=True)
    f = lambda t: t.float().to(dev).contiguous()
    g_in = (f(gb["bps_object"]), f(gb["rot_matrix"]).reshape(B, 9), f(gb["transl"]), f(gb["joint_conf"]))
    e_in = (f(eb["bps_object"]), f(eb["rot_matrix"]).reshape(B, 9), f(eb["transl"]), f(eb["label"]))
    zs = [synth.make_z(B, 3 + 100 * rank + i).to(dev) for i in range(4)]

    def step(i):
        # one update_dexgangrasp + one update_dexevaluator; gradient all-reduces (N > 1) overlap with the
        # independent phase, see engine.TrainEngine.joint_step
        eng.joint_step(g_in, zs[i % 4], e_in, True, None, 1234 + i, next_gan_inputs=g_in if PIPELINE else None)

    for i in range(args.warmup):
        step(i)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    losses = eng.losses.cpu().numpy()
    assert all(map(lambda x: x == x, losses[:10])), "NaN in losses"

    if rank == 0:
        value = B * world * args.steps / dt
        res = {
            "metric": "grasps/sec (gen+eval train step), batch=1024 BPS-4096", "value": round(value, 1),
            "unit": "grasps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "cGAN step (update_dexgangrasp) + evaluator step (update_dexevaluator), "
                                   "batch %d per GPU, BPS-4096, fp32, Adam included" % B,
                       "global_batch": B * world, "parallelism": "dp%d" % world,
                       "mflop_per_grasp_as_reference": MFLOP_PER_GRASP_FULL,
                       "mflop_per_grasp_executed": MFLOP_PER_GRASP_EXEC,
                       "step_tflops_executed": round(value * MFLOP_PER_GRASP_EXEC * 1e6 / 1e12 / world, 2),
                       "pipelining": ("the next step's D real-batch pass is enqueued one step ahead (its batch is "
                                      "resident, as with a prefetching loader); same results bit for bit"
                                      if PIPELINE else "none"),
                       "gemm_arithmetic": ("fp32 operands split exactly into 3 bf16 pieces, 6 partial products on the "
                                           "bf16 MFMA, fp32 accumulate (fp32-accurate)"
                                           if int(_lib.lib().dgg_get_gemm_algo()) == 1 else "f32 MFMA")},
        }
        res["roofline"] = gemm_roofline(dev)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(res))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
