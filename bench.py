#!/usr/bin/env python3
"""bench.py — grasps/sec of the gen+eval train step (BASELINE.json metric), batch 1024 / BPS-4096 per GPU.

A "step" = one DexGanGrasp.update_dexgangrasp (D phase, Adam(D), G phase, Adam(G)) + one
update_dexevaluator (fwd, BCE, bwd, Adam) on one batch of synthetic inputs already resident in HBM
(dexgangrasp.py:739-823).  value = global_batch * steps / time, max over ranks, weak scaling (batch
1024 per GPU; gradients all-reduced over RCCL when N > 1).

    python bench.py [--gpus N --steps K --warmup W]        # N > 1: launched by torch.distributed.run

The timed region is K steps bracketed by barrier + synchronize; it is repeated --repeats times (default 25, so the
measurement spans > 1 s) and the MEDIAN block gives `value` / `ms_per_step` (min / max are reported next to it).
Prints ONE JSON line: the contract fields + "roofline" (the dominant kernel: the weight-gradient GEMM; the forward
GEMM, BPS encode, Adam and the input-BatchNorm kernel under "roofline_kernels"), "sub_results" (BASELINE configs 2, 3,
5, the BPS encode stage and the drop-in API path) and "cpu_baseline" (the CPU oracle = plain-PyTorch restatement of
the reference step at the SAME batch, timed on this host's cores; rank 0, N = 1 only).
"""
import argparse
import ctypes as C
import json
import os
import statistics
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
# minus what D's two passes share over the 4096 BPS columns (same x_hat in both): the forward product (2 blocks x
# 768 x 4096 MAC = 12.6 MFLOP) and the weight gradient, formed once on dY_real + dY_fake (another 12.6 MFLOP)
MFLOP_PER_GRASP_EXEC = 122.8
# the next step's batch is resident (as with a prefetching loader): its D real pass is enqueued one step ahead
PIPELINE = os.environ.get("DGG_PIPELINE", "1") == "1"
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F32_VALU_TFLOPS = 157.3    # MI355X_MICROARCH.md: fp32 vector peak
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E spec (6.3 TB/s achievable)


def _time_stream(fn, iters, warm=3):
    """Mean duration (ms) of fn() over `iters` back-to-back calls, HIP events on the stream the kernels launch on."""
    st = torch.cuda.current_stream()
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        fn()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def _pmc_traffic(name, shape):
    """HBM-side bytes per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 per the gfx950 correction +
    WRITE_SIZE; separate passes, see profiles/): PMC counters cannot be read from inside a run."""
    for rnd in ("r06", "r05"):   # (rounds 1-4 profiled other kernels: gemm_x3_kernel, deleted in round 5)
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "%s_%s_pmc.json" % (rnd, name))))
            if pmc.get("shape") == shape:
                return int(pmc["traffic_bytes"]), "profiles/%s_%s_pmc.json" % (rnd, name)
        except Exception:
            continue
    return None, None


def large_product_flops(B, nj=12, in_bps=4096, latent=5):
    """FLOPs of the products of ONE gen+eval step that run on gemm_x5_kernel (bf16x3: 128 x 128 tiles), 2*M*N*K each — the
    numerator of roofline.step_mfma_frac.  Per ResBlock r of a network with input width k0: forward [a1|a3] = X W13^T
    (N = 768, K = k0 + 512 r) and its weight gradient (same M*N*K); the fc2 weight gradient (512 x 256 x B) and the
    generator heads' (21 x 512 x B) ride in those launches.  Not counted (f32-MFMA kernels): fc2 forward / dgrad stages,
    the K = 768 dgrads, the pose / latent column segments (K < 64).  D: the BPS columns' forward product is shared by the
    real and fake pass and its weight gradient formed once on dY_real + dY_fake; the G phase's pass through the updated D
    is a full forward, no weight gradient."""
    n13, nn, hid = 768, 512, 256
    pose_d, pose_e = 12 + nj, 12
    f = 0.0
    fc2w = 2.0 * nn * hid * B
    # generator: forward + weight gradient, blocks 0 and 1 (the 5 latent columns run on the f32 kernel)
    for r in range(2):
        k = in_bps + nn * r
        f += 2 * (2.0 * B * n13 * k) + fc2w
    f += 2.0 * (nj + 9) * nn * B
    # discriminator: shared BPS forward (once per D phase) + the previous block's 512 columns in each of the two passes;
    # the G phase's full forward; weight gradients: BPS block once, previous block's columns per pass
    for r in range(2):
        f += 2.0 * B * n13 * in_bps + 2 * (2.0 * B * n13 * nn * r)          # D phase forwards
        f += 2.0 * B * n13 * (in_bps + pose_d + nn * r)                      # G phase forward (pose columns ride: K >= 64 whole)
        f += 2.0 * B * n13 * in_bps + 2 * (2.0 * B * n13 * nn * r) + 2 * fc2w  # weight gradients
    # evaluator: three blocks, forward + weight gradient
    for r in range(3):
        k = in_bps + pose_e + (nn if r else 0)
        f += 2 * (2.0 * B * n13 * k) + fc2w
    return f


def in_step_profile():
    """What the committed kernel trace of THIS command says about the timed region (tools/collect_profiles.sh runs
    `rocprofv3 --kernel-trace --stats -- python3 bench.py ...`; tools/summarize_profiles.py keeps the per-kernel statistics as
    profiles/r06_bench_kernel_stats.csv and, from the trace itself, profiles/r06_bench_in_step.json: per (kernel, grid) mean
    durations and the launches per step and hardware queue).  A trace cannot be taken from inside the run it traces."""
    out = {}
    path = os.path.join(ROOT, "profiles", "r06_bench_in_step.json")
    try:
        out = json.load(open(path))
        out["source"] = "profiles/r06_bench_in_step.json (+ r06_bench_kernel_stats.csv, same trace)"
    except Exception:
        return None
    return out


def gemm_roofline(dev, mode, M, N, K, lda, ldb, splits, what, pmc_name, iters=20):
    """One large fc GEMM as the step runs it, alone on the chip.  achieved = algorithmic FLOPs (2*M*N*K, fp32-
    equivalent) / mean launch duration.  The bf16x3 arithmetic computes every fp32 product as six bf16 partial products
    on v_mfma_f32_32x32x16_bf16, so its matrix-pipe bound is the dense bf16 peak / 6; the native f32-MFMA peak is
    reported next to it.  mode 0 (NT): C = A[M,K] B[N,K]^T;  mode 2 (TN): C = A[K,M]^T B[K,N]."""
    g = torch.Generator().manual_seed(0)
    if mode == 0:
        A = torch.randn((M, lda), generator=g).to(dev)
        W = torch.randn((N, ldb), generator=g).to(dev)
    else:
        A = torch.randn((K, lda), generator=g).to(dev)
        W = torch.randn((K, ldb), generator=g).to(dev)
    slabs = torch.empty((max(splits, 1) * M * N,), device=dev)
    out = torch.empty((M, N), device=dev)
    lib = _lib.lib()
    st = torch.cuda.current_stream()
    algo = int(lib.dgg_get_gemm_algo())

    def launch():
        # split-K partial products into slabs with C = NULL: the (separate, HBM-bound) reduce kernel is left out so
        # that the MFMA kernel alone is timed
        _lib.check(lib.dgg_gemm_f32(mode, M, N, K, _lib.ptr(A), lda, _lib.ptr(W), ldb,
                                    None if splits >= 1 else _lib.ptr(out), N, splits, _lib.ptr(slabs), 0,
                                    C.c_void_p(st.cuda_stream)))
    ms = _time_stream(launch, iters)
    flops = 2.0 * M * N * K
    achieved = flops / (ms * 1e-3) / 1e12
    peak = PEAK_BF16_MFMA_TFLOPS / 6.0 if algo == 1 else PEAK_BF16_MFMA_TFLOPS if algo == 3 else PEAK_F32_MFMA_TFLOPS
    shape = {"M": M, "N": N, "K": K, "splits": splits}
    traffic, src = _pmc_traffic(pmc_name, shape)
    alg_bytes = 4 * (M * K + N * K + max(splits, 1) * M * N)
    return {"bound": "mfma", "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": src,
            "algorithmic_bytes": alg_bytes, "kernel": what, "shape": shape,
            "peak_basis": ("dense bf16 MFMA 2500 TFLOP/s / 6 products" if algo == 1 else
                           "dense bf16 MFMA 2500 TFLOP/s (plain bf16: one product)" if algo == 3 else "dense f32 MFMA"),
            "f32_mfma_peak": PEAK_F32_MFMA_TFLOPS, "frac_of_f32_mfma_peak": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
            "ms_per_launch": round(ms, 4)}


def bps_roofline(dev, B=1024, Np=2048, iters=10):
    """bps_mfma_kernel (A1): 4096 * N (basis point, cloud point) pairs per cloud against N*12 + 4096*8 bytes — arithmetic
    intensity ~1e3 flop/B, HBM cannot bind.  Since round 3 the pairs go through a two-plane fp16 coarse pass
    on the matrix cores: one v_mfma_f32_32x32x16_f16 per 32 x 32 pairs = 32 matrix flop per pair, which is what
    `achieved` counts against the dense fp16 / bf16 peak; the exact re-evaluation of the candidate blocks (~1.6 % of the
    pairs) is the contract's fp32 arithmetic on the vector ALU.  The kernel is bound by vector ISSUE (13 vector
    instructions per 512 pairs of a lane, which the matrix instruction's issue shares), not by the matrix pipe, so `frac`
    is low by construction (the measured port occupancy is in profiles/r04_bps_encode_pmc.json: vector_issue_port_busy_frac);
    `scan_equiv_tflops_cross_round_ruler` prices the same launch as the round-2 scan was priced (8 fp32 flop per pair) —
    a ruler across rounds, not a fraction of anything this kernel executes."""
    g = torch.Generator().manual_seed(0)
    basis = (torch.rand((4096, 3), generator=g) * 0.2 - 0.1).to(dev)
    clouds = synth.make_clouds(B, Np, 7).to(dev)
    d = torch.empty((B, 4096), device=dev)
    ids = torch.empty((B, 4096), dtype=torch.int32, device=dev)
    lib = _lib.lib()
    st = torch.cuda.current_stream()
    fn = lambda: _lib.check(lib.dgg_bps_encode(_lib.ptr(clouds), B, Np, _lib.ptr(basis), 4096, _lib.ptr(d),
                                               _lib.ptr(ids), C.c_void_p(st.cuda_stream)))
    ms = _time_stream(fn, iters)
    pairs = B * Np * 4096.0
    byts = B * (Np * 12 + 4096 * 8) + 4096 * 12
    tf = pairs * 32 / ms / 1e9
    tfv = pairs * 8 / ms / 1e9
    return {"bound": "mfma", "achieved": round(tf, 1), "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(tf / PEAK_BF16_MFMA_TFLOPS, 4), "kernel": "bps_mfma_kernel %d clouds x %d points x 4096 basis" % (B, Np),
            "ms_per_launch": round(ms, 4), "pairs_per_s": round(pairs / ms * 1e3, 1), "clouds_per_s": round(B / ms * 1e3, 1),
            "scan_equiv_tflops_cross_round_ruler": round(tfv, 2),
            "algorithmic_bytes": byts, "hbm_gb_per_s": round(byts / ms / 1e6, 2),
            "hbm_frac": round(byts / ms / 1e6 / PEAK_HBM_GBS, 5),
            "note": "32 matrix flop per pair (K = 16 fp16 slots, 14 used); vector-issue bound beside the matrix pipe "
                    "(DESIGN.md 3.4); round 2's scan: 1.07 ms, DGG_BPS=valu"}


def bps_one_cloud(dev, sizes=(73000, 15009), iters=30):
    """The reference's own call shape (dexgangrasp_real_robot.py:133-141, data/bps_encoder.py:28-55): ONE cloud of 15k-73k
    points.  Round 4: chunk by chunk on the matrix cores (dgg_bps_encode_ws), chunks meeting in a workspace; round 3: ranges
    of the vector-ALU scan meeting in the output arrays (dgg_bps_encode without a workspace), timed next to it.  A launch
    of ~10 us of arithmetic: latency-bound (two dependent launches), reported in us."""
    g = torch.Generator().manual_seed(0)
    basis = (torch.rand((4096, 3), generator=g) * 0.2 - 0.1).to(dev)
    lib = _lib.lib()
    st = torch.cuda.current_stream()
    out = {}
    for n in sizes:
        cl = synth.make_clouds(1, n, 5).to(dev).contiguous()
        d = torch.empty((1, 4096), device=dev)
        ids = torch.empty((1, 4096), dtype=torch.int32, device=dev)
        nws = int(lib.dgg_bps_workspace_bytes(1, n, 4096))
        ws = torch.empty((max(nws, 8),), dtype=torch.uint8, device=dev)
        new = lambda: _lib.check(lib.dgg_bps_encode_ws(_lib.ptr(cl), 1, n, _lib.ptr(basis), 4096, _lib.ptr(d), _lib.ptr(ids),
                                                       _lib.ptr(ws), nws, C.c_void_p(st.cuda_stream)))
        old = lambda: _lib.check(lib.dgg_bps_encode(_lib.ptr(cl), 1, n, _lib.ptr(basis), 4096, _lib.ptr(d), _lib.ptr(ids),
                                                    C.c_void_p(st.cuda_stream)))
        # (median of three blocks: one host hiccup inside a block of 10 launches of ~30 us would otherwise be the result)
        t_new = statistics.median(_time_stream(new, iters // 3) for _ in range(3)) * 1e3
        t_old = statistics.median(_time_stream(old, iters // 3) for _ in range(3)) * 1e3
        out["%d_points" % n] = {"us_matrix_cores_chunked": round(t_new, 1), "us_vector_alu_ranges": round(t_old, 1),
                                "pairs_per_s": round(n * 4096.0 / t_new * 1e6, 1), "workspace_bytes": nws}
    return out


def adam_roofline(dev, n_small=10661657, n_big=32 * 1024 * 1024, iters=20):
    """adam_kernel (A12): 16 B read + 12 B written per parameter.  The entry streams a working set LARGER than the 256 MB
    Infinity Cache (4 x 32 M floats = 512 MB), so that `achieved` is HBM bandwidth; the evaluator's own flat buffer
    (4 x 42.6 MB = 170 MB, launches back to back: served partly from the Infinity Cache) is reported next to it as
    `cache_assisted` — round 3 quoted that figure (6.8 TB/s) as a fraction of the HBM peak, which it is not."""
    lib = _lib.lib()
    st = torch.cuda.current_stream()

    def run(n):
        p, g, m, v = (torch.randn(n, device=dev) * 0.01 for _ in range(4))
        v.abs_()
        k = [0]

        def fn():
            k[0] += 1
            _lib.check(lib.dgg_adam_l2(_lib.ptr(p), _lib.ptr(g), _lib.ptr(m), _lib.ptr(v), n, k[0], 1e-4, 0.9, 0.999,
                                       1e-8, 5e-4, 1.0, C.c_void_p(st.cuda_stream)))
        ms = _time_stream(fn, iters)
        return ms, 28.0 * n / ms / 1e6
    ms, gbs = run(n_big)
    ms_s, gbs_s = run(n_small)
    return {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": round(gbs / PEAK_HBM_GBS, 4), "kernel": "adam_kernel, %d parameters (28 B each; 4 arrays = %d MB)"
                                                            % (n_big, 16 * n_big // 2 ** 20),
            "algorithmic_bytes": int(28.0 * n_big), "ms_per_launch": round(ms, 4),
            "cache_assisted": {"gb_per_s": round(gbs_s, 1), "ms_per_launch": round(ms_s, 4),
                               "kernel": "adam_kernel, %d parameters (the evaluator's flat buffer: 170 MB)" % n_small,
                               "note": "working set < 256 MB Infinity Cache, launches back to back: not an HBM figure"}}


def usable_cores():
    """Cores this process may actually use: the affinity mask, capped by the cgroup CPU quota when one is set."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                q, p = txt[0], txt[1]
            else:
                q, p = txt[0], open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().split()[0]
            if q not in ("max", "-1"):
                n = min(n, max(1, int(round(int(q) / int(p)))))
            break
        except Exception:
            continue
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(cfg, b):
    """The CPU oracle (port of the reference step: plain PyTorch fp32) at the SAME batch as the GPU step, on this
    host's cores: >= 5 timed steps after 2 warm-up with all usable cores, plus a 1-thread figure (2 timed after 1)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import dexgan_oracle as O
    cores = usable_cores()
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

    def run(threads, warm, timed, budget):
        torch.set_num_threads(threads)
        for _ in range(warm):
            step()
        t0, n = time.time(), 0
        while n < timed or (time.time() - t0 < budget and n < 20):
            step()
            n += 1
        return n, time.time() - t0
    n, dt = run(cores, 2, 5, 8.0)
    n1, dt1 = run(1, 1, 2, 6.0)
    torch.set_num_threads(cores)
    return {"value": round(b * n / dt, 1), "unit": "grasps/s", "cores": cores, "kind": "port",
            "cpu_model": cpu_model(), "ms_per_step": round(dt / n * 1e3, 1),
            "single_thread": {"value": round(b * n1 / dt1, 1), "unit": "grasps/s", "cores": 1,
                              "ms_per_step": round(dt1 / n1 * 1e3, 1), "sample": "%d steps after 1 warm-up" % n1},
            "sample": "%d gen+eval steps at batch %d (the GPU step's batch) after 2 warm-up, %d threads" % (n, b, cores)}


def sub_results(cfg, dev, eng):
    """BASELINE configs 2, 3 and 5 and the pieces around the headline step, each timed on its own (1 GPU)."""
    out = {}
    f = lambda t: t.float().to(dev).contiguous()

    def wall(fn, steps, warm=10, blocks=5):
        # median of `blocks` timed blocks: a single short block after host-side set-up catches the clock ramp of an idle
        # GPU now and then (one sub-result per run up to 1.8x off)
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(blocks):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / steps)
        return statistics.median(ts)
    # config 2: cGAN update only, B = 256 (train.py's generator phase)
    b = 256
    gb = synth.make_batch(b, 1)
    g_in = (f(gb["bps_object"]), f(gb["rot_matrix"]).reshape(b, 9), f(gb["transl"]), f(gb["joint_conf"]))
    z = synth.make_z(b, 3).to(dev)
    dt = wall(lambda: eng.cgan_step(*g_in, z, True, two_streams=True), 40)
    out["config2_cgan_step_b256"] = {"ms_per_step": round(dt * 1e3, 4), "grasps_per_s": round(b / dt, 1),
                                     "workload": "update_dexgangrasp only, batch 256, fp32"}
    # config 3: evaluator update only, B = 1024
    b = 1024
    eb = synth.make_batch(b, 2, with_label=True)
    e_in = (f(eb["bps_object"]), f(eb["rot_matrix"]).reshape(b, 9), f(eb["transl"]), f(eb["label"]))
    k = [0]

    def eva():
        k[0] += 1
        eng.eva_step(*e_in, None, 99 + k[0], two_streams=True)
    dt = wall(eva, 40)
    out["config3_eva_step_b1024"] = {"ms_per_step": round(dt * 1e3, 4), "grasps_per_s": round(b / dt, 1),
                                     "workload": "update_dexevaluator only, batch 1024, in-kernel dropout"}
    # config 3 as BASELINE words it ("bf16 MFMA fc stacks"): the same update with the large products in plain bf16
    # (DGG_GEMM_BF16: one product, fp32 accumulate, fp32 master weights; tolerance vs fp32 stated in
    # tests/test_gpu_steps.py::test_bf16_mode_evaluator_step_tolerance_vs_fp32).  Never the default, never the headline dtype.
    lib_ = _lib.lib()
    algo0 = int(lib_.dgg_get_gemm_algo())
    try:
        _lib.check(lib_.dgg_set_gemm_algo(3))
        dt16 = wall(eva, 40)
        nt16 = gemm_roofline(dev, 0, b, 768, 4120, 4120, 4632, 4, "forward of fc1||fc3 (NT), plain bf16: a13[%d,768] = "
                             "X0[%d,4120] W13[768,4120]^T, split-K 4 (DGG_GEMM_BF16)" % (b, b), "gemm_x5_bf16")
    finally:
        _lib.check(lib_.dgg_set_gemm_algo(algo0))
    out["config3_eva_step_b1024_bf16"] = {"ms_per_step": round(dt16 * 1e3, 4), "grasps_per_s": round(b / dt16, 1),
                                          "dtype": "bf16 (large products; fp32 master weights, fp32 accumulate)",
                                          "roofline_forward_gemm": nt16,
                                          "workload": "update_dexevaluator only, batch 1024, DGG_GEMM_BF16 (BASELINE config 3 as "
                                                      "worded); tolerance vs fp32: scores 2e-2, see the test"}
    # the cGAN update alone at the headline batch, serial entry with intra-step side streams
    gb = synth.make_batch(b, 1)
    g_in = (f(gb["bps_object"]), f(gb["rot_matrix"]).reshape(b, 9), f(gb["transl"]), f(gb["joint_conf"]))
    z = synth.make_z(b, 3).to(dev)
    dt = wall(lambda: eng.cgan_step(*g_in, z, True, two_streams=True), 30)
    out["cgan_step_b1024"] = {"ms_per_step": round(dt * 1e3, 4), "grasps_per_s": round(b / dt, 1)}
    # the two engine steps ALTERNATING (cGAN, evaluator, cGAN, ...): what the wrapper's gen+eval pair below does minus the
    # wrapper — value_serial sums the two homogeneous loops above, which is S8(d)'s formula but flatters a caller that
    # alternates (different weights and workspaces every other step); api_path / this = the wrapper's own overhead
    e1024 = (f(eb["bps_object"]), f(eb["rot_matrix"]).reshape(b, 9), f(eb["transl"]), f(eb["label"]))
    kk = [0]

    def pair():
        kk[0] += 1
        eng.cgan_step(*g_in, z, True, two_streams=True)
        eng.eva_step(*e1024, None, 7000 + kk[0], two_streams=True)
    dt = wall(pair, 30)
    out["engine_pair_b1024"] = {"ms_per_step": round(dt * 1e3, 4), "grasps_per_s": round(b / dt, 1),
                                "workload": "engine.cgan_step + engine.eva_step alternating, batch 1024 (no wrapper)"}
    # N3: the WGAN-GP variant of the same update (is_wgan: Wasserstein losses + gradient penalty through D's double backward)
    al = torch.rand((b, 3), generator=torch.Generator().manual_seed(5)).to(dev)
    dt = wall(lambda: eng.wgan_step(*g_in, z, al, True), 20)
    out["wgan_gp_step_b1024"] = {"ms_per_step": round(dt * 1e3, 4), "grasps_per_s": round(b / dt, 1),
                                 "workload": "update_dexgangrasp with is_wgan=True (gradient penalty), batch 1024"}
    # the drop-in API path: DexGanGrasp.update_dexgangrasp + update_dexevaluator through the wrapper, as an unchanged
    # train.py would call them (one after the other; batches resident on the device as float32)
    from dex_gan_grasp_amd.models.dexgangrasp import DexGanGrasp
    wcfg = dict(cfg)
    wcfg["train_ffhevaluator"] = True
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        m = DexGanGrasp(wcfg)
    gd = {"bps_object": g_in[0], "rot_matrix": g_in[1].view(b, 3, 3), "transl": g_in[2], "joint_conf": g_in[3]}
    ed = {"bps_object": e_in[0], "rot_matrix": e_in[1].view(b, 3, 3), "transl": e_in[2], "label": e_in[3]}

    # (a) the reference's own order (train.py:231-285): within an epoch ALL generator batches (update_dexgangrasp), then ALL
    #     evaluator batches (update_dexevaluator) — two homogeneous loops; ms per gen+eval pair = t_gen + t_eva
    dt_g = wall(lambda: m.update_dexgangrasp(dict(gd)), 30)
    dt_e = wall(lambda: m.update_dexevaluator(dict(ed)), 30)
    dt = dt_g + dt_e
    out["api_path_b1024"] = {"ms_per_step": round(dt * 1e3, 4), "grasps_per_s": round(b / dt, 1),
                             "ms_update_dexgangrasp": round(dt_g * 1e3, 4), "ms_update_dexevaluator": round(dt_e * 1e3, 4),
                             "workload": "DexGanGrasp.update_dexgangrasp over a loop of batches, then update_dexevaluator over a "
                                         "loop of batches (the wrapper in train.py's order, :231-285), batch 1024"}

    # (b) a caller that alternates the two updates batch by batch (rounds 1-5 timed this as the drop-in path): the two
    #     networks' weights, Adam state and workspaces (~0.5 GB) evict each other from the Infinity Cache every other step
    def api():
        m.update_dexgangrasp(dict(gd))
        m.update_dexevaluator(dict(ed))
    dt = wall(api, 30)
    out["api_path_alternating_b1024"] = {"ms_per_step": round(dt * 1e3, 4), "grasps_per_s": round(b / dt, 1),
                                         "workload": "update_dexgangrasp + update_dexevaluator alternating per batch (wrapper), batch 1024"}
    # config 4 front end (one rank's share of the global batch 4096 = 512 / GPU): synthetic clouds -> BPS encode ->
    # batch dict -> gen+eval step; the encode is reported as its own stage (it is not part of the reference's step)
    from dex_gan_grasp_amd.data.synthetic_data_set import SyntheticGraspBatches
    b = 512
    g = torch.Generator().manual_seed(0)
    basis = (torch.rand((4096, 3), generator=g) * 0.2 - 0.1).to(dev)
    ds = SyntheticGraspBatches(cfg, b, 4, basis, n_points=2048, device=dev, with_label=True, device_tensors=True)
    clouds = ds.clouds(0)
    enc_ms = _time_stream(lambda: ds.encode(clouds), 10)
    bt = ds.batch(0)
    g_in = (bt["bps_object"], bt["rot_matrix"].reshape(b, 9).contiguous(), bt["transl"], bt["joint_conf"])
    e_in = (bt["bps_object"], bt["rot_matrix"].reshape(b, 9).contiguous(), bt["transl"], bt["label"].float())
    z = synth.make_z(b, 3).to(dev)
    k2 = [0]

    def both():
        k2[0] += 1
        bps = ds.encode(clouds)
        eng.joint_step((bps,) + g_in[1:], z, (bps,) + e_in[1:], True, None, 500 + k2[0])
    dt = wall(both, 30)
    bps0 = ds.encode(clouds)

    def step_only():   # the same step on an already encoded batch
        k2[0] += 1
        eng.joint_step((bps0,) + g_in[1:], z, (bps0,) + e_in[1:], True, None, 500 + k2[0])
    dt_step = wall(step_only, 30)
    # ... and with the next batch's real pass enqueued one step ahead, as in the headline (two resident batches alternate)
    bt1 = ds.batch(1)
    pair = [(bps0,) + g_in[1:],
            (bt1["bps_object"], bt1["rot_matrix"].reshape(b, 9).contiguous(), bt1["transl"], bt1["joint_conf"])]

    def step_ahead():
        k2[0] += 1
        cur, nxt = pair[k2[0] & 1], pair[(k2[0] + 1) & 1]
        eng.joint_step(cur, z, (cur[0],) + e_in[1:], True, None, 500 + k2[0], next_gan_inputs=nxt)
    eng.flush()
    dt_step_ahead = wall(step_ahead, 30)
    eng.flush()
    # the same with the loader one batch ahead: batch k+1's encode on a side stream, enqueued before step k
    side = torch.cuda.Stream(device=dev)
    pend = [None]

    def enqueue_encode():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            bps = ds.encode(clouds)
            ev = torch.cuda.Event()
            ev.record(side)
        bps.record_stream(torch.cuda.current_stream())
        pend[0] = (bps, ev)

    def both_ahead():
        k2[0] += 1
        bps, ev = pend[0]
        torch.cuda.current_stream().wait_event(ev)
        enqueue_encode()
        eng.joint_step((bps,) + g_in[1:], z, (bps,) + e_in[1:], True, None, 500 + k2[0])
    enqueue_encode()
    dt_ahead = wall(both_ahead, 30)
    torch.cuda.synchronize()
    out["config4_front_end_b512"] = {"encode_ms": round(enc_ms, 4), "encode_clouds_per_s": round(b / enc_ms * 1e3, 1),
                                     "ms_per_step_encode_one_batch_ahead": round(dt_ahead * 1e3, 4),
                                     "grasps_per_s_encode_one_batch_ahead": round(b / dt_ahead, 1),
                                     "ms_per_step_with_encode": round(dt * 1e3, 4), "grasps_per_s": round(b / dt, 1),
                                     "ms_per_step_without_encode": round(dt_step * 1e3, 4),
                                     "ms_per_step_without_encode_look_ahead": round(dt_step_ahead * 1e3, 4),
                                     "workload": "512 clouds x 2048 points -> dgg_bps_encode -> gen+eval step, batch 512 "
                                                 "(one rank's share of global batch 4096)"}
    # config 5 (one rank's share): 1250 objects x 200 grasps, encode (N = 2048) + generate + evaluate + 3 thresholds
    n_obj, n_g, n_pts = 1250, 200, 2048
    with contextlib.redirect_stdout(io.StringIO()):
        icfg = dict(cfg)
        icfg["is_train"] = False
        mi = DexGanGrasp(icfg)
    mi.DexGANGrasp.load_state_dict(synth.make_state_dict("dexgangrasp", 10, 1.0, True))
    mi.DexEvaluator.load_state_dict(synth.make_state_dict("evaluator", 15, 1.0, True))
    g = torch.Generator().manual_seed(0)
    basis = (torch.rand((4096, 3), generator=g) * 0.2 - 0.1).to(dev)
    clouds = synth.make_clouds(n_obj, n_pts, 7).to(dev)
    kept = [0]
    stage = {}

    def sweep():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dists, _ = ops.bps_encode(clouds, basis)
        torch.cuda.synchronize()
        stage["encode_s"] = time.perf_counter() - t0
        o = mi.sweep_objects(dists, n_g, thresholds=(0.5, 0.75, 0.90))
        kept[0] = int(o["count"].sum())
    dt = wall(sweep, 2, warm=1)
    out["config5_sweep_1250x200"] = {"seconds": round(dt, 4), "objects_per_s": round(n_obj / dt, 1),
                                     "grasps_per_s": round(n_obj * n_g / dt, 1), "encode_seconds": round(stage["encode_s"], 5),
                                     "kept_total": kept[0],
                                     "workload": "per object: BPS-encode a 2048-point cloud, generate 200 grasps, evaluate, "
                                                 "filter at 0.5/0.75/0.9; one rank's share (1250 of 10k objects)"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (defaults: blocks long enough that the pipeline's fill and drain at the two synchronisation points of a block — the first
    #  step has no prefetched real pass to pick up, the last one's prefetch is wasted — stay under half a percent; with 20
    #  steps per block they were ~2 %)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--repeats", type=int, default=11, help="timed blocks of --steps steps; the median is reported")
    ap.add_argument("--batch", type=int, default=1024, help="per-GPU batch (BASELINE: 1024)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sub", action="store_true")
    ap.add_argument("--same-batch", action="store_true",
                    help="rounds 1-5: ONE resident gan batch and ONE eva batch stepped forever (their BPS blocks and x_hat stay "
                         "L2 / Infinity-Cache warm across steps); default since round 6: --nbatch distinct resident batches "
                         "rotate and the look-ahead real pass gets the NEXT batch")
    ap.add_argument("--nbatch", type=int, default=4, help="distinct resident gan / eva batches that rotate (default 4)")
    ap.add_argument("--no-other-mode", action="store_true", help="skip the short comparison run in the other data mode "
                                                                  "(profiling: the trace then holds one mode only)")
    args = ap.parse_args()

    # DGG_REHEARSE=1: rehearsal of the multi-rank code path on a 1-GPU box (gloo backend, all ranks on cuda:0);
    # never used for reported numbers
    rank, world, local = parallel.init_from_env()
    assert world == args.gpus or world == 1 and args.gpus == 1, "launch with torch.distributed.run for --gpus > 1"
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
    f = lambda t: t.float().to(dev).contiguous()
    nb = 1 if args.same_batch else max(1, args.nbatch)
    g_ins, e_ins = [], []
    for j in range(nb):   # (batch 0 = the batches of rounds 1-5: seeds 1 / 2)
        gb = synth.make_batch(B, 1 + 100 * rank + 1000 * j)
        eb = synth.make_batch(B, 2 + 100 * rank + 1000 * j, with_label=True)
        g_ins.append((f(gb["bps_object"]), f(gb["rot_matrix"]).reshape(B, 9), f(gb["transl"]), f(gb["joint_conf"])))
        e_ins.append((f(eb["bps_object"]), f(eb["rot_matrix"]).reshape(B, 9), f(eb["transl"]), f(eb["label"])))
    zs = [synth.make_z(B, 3 + 100 * rank + i).to(dev) for i in range(4)]
    it = [0]

    def make_step(n_rot):
        def step():
            # one update_dexgangrasp + one update_dexevaluator; gradient all-reduces (N > 1) overlap with the
            # independent phase, see engine.TrainEngine.joint_step.  The look-ahead real pass gets the NEXT batch.
            i = it[0]
            it[0] += 1
            eng.joint_step(g_ins[i % n_rot], zs[i % 4], e_ins[i % n_rot], True, None, 1234 + i,
                           next_gan_inputs=g_ins[(i + 1) % n_rot] if PIPELINE else None)
        return step

    def timed(step, repeats):
        for _ in range(args.warmup):
            step()
        blocks = []
        for _ in range(max(1, repeats)):
            if world > 1:
                torch.distributed.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            if world > 1:
                torch.distributed.barrier()
            blocks.append(time.perf_counter() - t0)
        if world > 1:   # MAX over ranks, block by block
            t = torch.tensor(blocks, device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            blocks = [float(x) for x in t.cpu()]
        return blocks

    step = make_step(nb)
    blocks = timed(step, args.repeats)
    dt = statistics.median(blocks)
    # the other data mode beside it (fewer blocks): how much of the step the warm caches of a never-changing batch were worth
    other_ms = None
    if nb > 1 and not args.no_other_mode:
        it[0] = 0
        ob = timed(make_step(1), min(5, max(1, args.repeats)))
        other_ms = statistics.median(ob) / args.steps * 1e3
        it[0] = 0
        for _ in range(4):   # back to the rotating mode for the trace steps below
            step()
    losses = eng.losses.cpu().numpy()
    assert all(map(lambda x: x == x, losses[:10])), "NaN in losses"

    # where a step spends its time: timed events at the phase boundaries (engine.trace), outside the timed region — every
    # rank runs the same extra steps (their collectives match).  Data parallel this is also what makes the first N > 1 run
    # falsifiable: per-bucket all-reduce durations and, per network, the time its optimiser step
    # waited for the all-reduce after its backward had finished ("exposed": nothing else runs on that stream in between).
    eng.trace = []
    eng.trace_buckets = [] if parallel.is_distributed() else None
    n_tr = 12
    for _ in range(n_tr):
        step()
    torch.cuda.synchronize()
    tr, eng.trace = eng.trace, None
    tb, eng.trace_buckets = eng.trace_buckets, None
    per = len(tr) // n_tr
    acc = {}
    for s_ in range(2, n_tr - 1):
        ev, nxt = dict(tr[s_ * per:(s_ + 1) * per]), dict(tr[(s_ + 1) * per:(s_ + 2) * per])
        for name in ("d_phase_end", "adam_d_end", "g_phase_end", "adam_g_end", "e_phase_end", "adam_e_end",
                     "prefetch_begin", "prefetch_end", "ar_d_waited", "ar_g_waited", "ar_e_waited"):
            if name in ev:
                acc.setdefault(name, []).append(ev["step_begin"].elapsed_time(ev[name]) * 1e3)
        for net, a, b_ in (("D", "d_phase_end", "ar_d_waited"), ("G", "g_phase_end", "ar_g_waited"),
                           ("E", "e_phase_end", "ar_e_waited")):
            if a in ev and b_ in ev:
                acc.setdefault("exposed_" + net, []).append(ev[a].elapsed_time(ev[b_]) * 1e3)
        acc.setdefault("next_step_begin", []).append(ev["step_begin"].elapsed_time(nxt["step_begin"]) * 1e3)
    phase_us = {k: round(statistics.median(v), 1) for k, v in acc.items() if not k.startswith("exposed_")}
    exposed_us = {k[len("exposed_"):]: round(statistics.median(v), 1) for k, v in acc.items() if k.startswith("exposed_")}
    bucket_us = {}
    if tb:
        by = {}
        for tag, i, nbytes, e0, e1 in tb:
            by.setdefault((tag, i, nbytes), []).append(e0.elapsed_time(e1) * 1e3)
        for (tag, i, nbytes), v in sorted(by.items()):
            bucket_us.setdefault(tag, []).append({"bucket": i, "bytes": nbytes, "us": round(statistics.median(v[2:] or v), 1)})

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
                       "data_rotation": ("one resident gan batch and one eva batch, stepped forever (--same-batch)" if nb == 1 else
                                         "%d distinct resident gan batches and %d eva batches rotate; the look-ahead real "
                                         "pass gets the next batch" % (nb, nb)),
                       "ms_per_step_same_batch": round(other_ms, 4) if other_ms is not None else None,
                       "phase_us_after_step_begin": phase_us,
                       "timed_blocks": len(blocks),
                       "block_ms_per_step_min_median_max": [round(min(blocks) / args.steps * 1e3, 4),
                                                            round(dt / args.steps * 1e3, 4),
                                                            round(max(blocks) / args.steps * 1e3, 4)],
                       "mflop_per_grasp_as_reference": MFLOP_PER_GRASP_FULL,
                       "mflop_per_grasp_executed": MFLOP_PER_GRASP_EXEC,
                       "step_tflops_executed": round(value * MFLOP_PER_GRASP_EXEC * 1e6 / 1e12 / world, 2),
                       "pipelining": ("the next step's D real-batch pass is enqueued one step ahead (its batch is "
                                      "resident, as with a prefetching loader); same results bit for bit"
                                      if PIPELINE else "none"),
                       "gemm_arithmetic": ("fp32 operands split exactly into 3 bf16 pieces, 6 partial products on the "
                                           "bf16 MFMA, fp32 accumulate (fp32-accurate)"
                                           if int(_lib.lib().dgg_get_gemm_algo()) >= 1 else "f32 MFMA")},
        }
        # dominant kernel of the step: the weight gradient of a ResBlock's fused fc1||fc3 layer,
        # dW13[768, 4120] = dY13[1024, 768]^T X0[1024, 4120] (TN); the forward of the same layer (NT, split-K 4) next
        tn = gemm_roofline(dev, 2, 768, 4120, B, 768, 4120, 0,
                           "gemm_x5_kernel, weight gradient of fc1||fc3 (TN): dW13[768,4120] = dY13[%d,768]^T X0[%d,4120]" % (B, B),
                           "gemm_x5_tn")
        nt = gemm_roofline(dev, 0, B, 768, 4120, 4120, 4632, 4,
                           "gemm_x5_kernel, forward of fc1||fc3 (NT): a13[%d,768] = X0[%d,4120] W13[768,4120]^T, split-K 4" % (B, B),
                           "gemm_x5")
        res["roofline"] = dict(tn)
        # ... and where the step runs: the same instantiation INSIDE the timed region, from the committed
        # kernel trace of this command; and the whole step against the same bound
        prof = in_step_profile()
        if prof and prof.get("dominant_tn"):
            d = prof["dominant_tn"]
            pb = d.get("per_block_mean_us", d["mean_us"])   # a grouped launch holds 2-3 ResBlocks' products (round 6)
            res["roofline"]["in_step"] = {
                "mean_us": d["mean_us"], "min_us": d["min_us"], "max_us": d["max_us"], "launches": d["launches"],
                "launches_per_step": d.get("launches_per_step"), "mean_workgroups": d.get("mean_workgroups"),
                "per_block_mean_us": pb,
                "achieved": round(2.0 * 768 * 4120 * B / (pb * 1e-6) / 1e12, 2) if B == 1024 else None,
                "frac": round(2.0 * 768 * 4120 * B / (pb * 1e-6) / 1e12 / tn["peak"], 4) if B == 1024 else None,
                "what": d["what"], "source": prof["source"]}
            res["main_stream_launches_per_step"] = prof.get("launches_per_step_by_queue")
        lp = large_product_flops(B)
        res["roofline"]["step_mfma_frac"] = round(lp / (dt / args.steps) / 1e12 / tn["peak"], 4)
        res["roofline"]["step_large_product_gflop"] = round(lp / 1e9, 2)
        res["roofline_kernels"] = {"gemm_wgrad_tn": tn, "gemm_forward_nt": nt, "bps_encode": bps_roofline(dev),
                                   "bps_encode_one_cloud": bps_one_cloud(dev),
                                   "adam": adam_roofline(dev)}
        # what the hardware-queue probe behind the engine's side streams found (parallel.pick_side_streams): a fallback to
        # plain streams costs 15-20 % and would otherwise go unnoticed
        res["side_streams"] = dict(parallel.pick_report) if parallel.pick_report else {"mode": "not created"}
        nbytes = 4 * (fg.n_params + fd.n_params + fe.n_params)
        res["rccl_ranks"] = {"world_size": world,
                             "backend": (torch.distributed.get_backend() if torch.distributed.is_initialized() else None),
                             "allreduce_bytes_per_step_per_rank": nbytes if world > 1 else 0,
                             "buckets": "per ResBlock, behind in-backward events (parallel.BucketReducer)" if world > 1 else None,
                             "evaluator_group": os.environ.get("DGG_EVA_GROUP", "0") == "1",
                             # measured on rank 0 in the untimed trace steps (events; DESIGN.md 6 holds the predictions
                             # these numbers are to be compared with):
                             "exposed_us": exposed_us or None,     # per network: optimiser step waited this long for the all-reduce
                             "bucket_allreduce_us": bucket_us or None,   # per bucket, in issue order, queueing included
                             "rccl_env": {k: os.environ.get(k) for k in ("NCCL_ALGO", "NCCL_PROTO", "NCCL_MIN_NCHANNELS",
                                                                         "NCCL_MAX_NCHANNELS", "RCCL_MSCCL_ENABLE",
                                                                         "DGG_BUCKETED_AR", "DGG_EVA_GROUP")
                                          if os.environ.get(k) is not None}}
        if world == 1 and not args.no_sub:
            res["sub_results"] = sub_results(cfg, dev, eng)
            sr = res["sub_results"]
            # `value` is the interleaved engine step (evaluator update beside the cGAN update, one-batch look-ahead on D's
            # real pass).  What the reference's own call pattern gets:
            #   value_serial = B / (t_cGAN + t_eval), SURVEY 8(d)'s formula on the two engine steps run one after the other
            #   value_dropin = the unchanged train.py order through the drop-in wrapper (update_dexgangrasp, then
            #                  update_dexevaluator, per batch)
            t_serial = (sr["cgan_step_b1024"]["ms_per_step"] + sr["config3_eva_step_b1024"]["ms_per_step"]) * 1e-3
            res["value_serial"] = round(1024 / t_serial, 1) if B == 1024 else None
            res["value_dropin"] = sr["api_path_b1024"]["grasps_per_s"] if B == 1024 else None
            res["config"]["value_serial"] = {"grasps_per_s": res["value_serial"],
                                             "formula": "SURVEY 8(d): B / (t_cGAN_step + t_evaluator_step), the two engine steps "
                                                        "timed one after the other (sub_results.cgan_step_b1024, config3_eva_step_b1024)",
                                             "t_cgan_ms": sr["cgan_step_b1024"]["ms_per_step"],
                                             "t_eval_ms": sr["config3_eva_step_b1024"]["ms_per_step"]}
            res["config"]["value_dropin"] = {"grasps_per_s": res["value_dropin"],
                                             "what": "unchanged train.py order through the drop-in wrapper: a loop of generator "
                                                     "batches, then a loop of evaluator batches (sub_results.api_path_b1024)",
                                             "over_value_serial": round(res["value_dropin"] / res["value_serial"], 4) if B == 1024 else None,
                                             "alternating_per_batch": {
                                                 "wrapper_ms": sr["api_path_alternating_b1024"]["ms_per_step"],
                                                 "engine_ms": sr["engine_pair_b1024"]["ms_per_step"],
                                                 "note": "a caller that alternates the two updates batch by batch: the networks evict "
                                                         "each other from the Infinity Cache; wrapper / engine = the wrapper's own overhead"}}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(cfg, B)
        print(json.dumps(res))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
