"""CPU tests of the boundary: libdgg.so loads, exports every symbol include/dgg.h declares, its
layout tables describe exactly the reference's state_dict, argument validation returns error
codes (no compute is launched without a GPU)."""
import ctypes as C
import os
import re

import pytest
import torch

from dex_gan_grasp_amd import _lib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "dgg.h")).read()
    declared = set(re.findall(r"\b(dgg_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"dgg_status", "dgg_net", "dgg_cfg", "dgg_entry", "dgg_hparams", "dgg_gan_state"}
    assert len(declared) >= 20
    lib = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), "libdgg.so does not export " + name
    assert declared == set(_lib.SIGNATURES.keys()), declared ^ set(_lib.SIGNATURES.keys())
    assert b"gfx950" in _lib.lib().dgg_version()


@pytest.mark.parametrize("net,name", [(_lib.NET_GEN, "generator"), (_lib.NET_DISC, "discriminator"),
                                      (_lib.NET_EVA, "evaluator")])
def test_layout_table_matches_reference_state_dict(net, name):
    c = _lib.make_cfg(synth.DEFAULT_CFG)
    t = _lib.table(net, c)
    spec = {n: sh for n, sh, k in synth.net_spec(name) if k != "bn_nbt"}
    assert set(spec) == {e[0] for e in t}
    used = []
    for nm, off, rows, cols, kind in t:
        got = (rows, cols) if cols > 1 else (rows,)
        assert tuple(spec[nm]) == got, nm
        assert off % 4 == 0 or "fc3" in nm or "bias" in nm or "gen_" in nm
        used.append((kind, off, off + rows * cols))
    # entries of one buffer never overlap
    for kind in (0, 1):
        iv = sorted((a, b) for k, a, b in used if k == kind)
        for (a0, b0), (a1, b1) in zip(iv, iv[1:]):
            assert b0 <= a1
    n_par = sum(r * cc for _, _, r, cc, k in t if k == 0)
    assert n_par == {"generator": 6979093, "discriminator": 6998065, "evaluator": 10661657}[name]
    assert int(_lib.lib().dgg_param_floats(net, C.byref(c))) >= n_par
    # fc1.weight and fc3.weight are adjacent: one (768 x Fin) matrix for the fused fc1||fc3 GEMM
    d = {nm: (off, rows, cols) for nm, off, rows, cols, _ in t}
    for nm in d:
        if nm.endswith("fc1.weight"):
            o1, r1, c1 = d[nm]
            o3, _, c3 = d[nm.replace("fc1", "fc3")]
            assert o3 == o1 + r1 * c1 and c1 == c3


def test_queries_and_argument_validation():
    c = _lib.make_cfg(synth.DEFAULT_CFG)
    lib = _lib.lib()
    for net in (0, 1, 2):
        assert lib.dgg_workspace_bytes(net, C.byref(c), 1024) > 0
        assert lib.dgg_workspace_bytes(net, C.byref(c), 0) < 0
    assert lib.dgg_cgan_workspace_bytes(C.byref(c), 256) > 0
    assert lib.dgg_table(7, C.byref(c), None, 0) < 0
    bad = _lib.make_cfg({"n_neurons": 300})
    assert lib.dgg_param_floats(0, C.byref(bad)) < 0
    # null pointers / bad sizes are rejected before anything is launched
    assert lib.dgg_bps_encode(None, 1, 10, None, 4096, None, None, None) == -1
    assert lib.dgg_adam_l2(None, None, None, None, 10, 1, 1e-4, 0.9, 0.999, 1e-8, 0.0, 1.0, None) == -1
    assert lib.dgg_gemm_f32(5, 1, 1, 1, None, 1, None, 1, None, 1, 0, None, 0, None) == -1
    assert lib.dgg_ortho6d_forward(None, 6, 4, None, None) == -1


def test_modules_construct_and_serialise_without_gpu_but_forward_fails_loudly():
    from dex_gan_grasp_amd.models import networks
    cfg = dict(synth.DEFAULT_CFG)
    gen = networks.Generator(cfg)
    sd = gen.state_dict()
    spec = synth.net_spec("generator")
    assert list(sd.keys()) == [n for n, _, _ in spec]
    gen.load_state_dict({k[len("generator."):]: v for k, v in synth.make_state_dict("dexgangrasp", 0).items()
                         if k.startswith("generator.")})
    if not torch.cuda.is_available():
        with pytest.raises(_lib.DggError):
            gen(torch.zeros(2, 5), torch.zeros(2, 4096))
        from dex_gan_grasp_amd.utils import utils
        with pytest.raises(_lib.DggError):
            utils.rot_matrix_from_ortho6d(torch.zeros(2, 6))
    with pytest.raises(_lib.DggError):
        networks.ResBlock(10, 20)(torch.zeros(2, 10))


def test_wrapper_error_conventions():
    from dex_gan_grasp_amd.models.dexgangrasp import DexGanGrasp
    cfg = dict(synth.DEFAULT_CFG)
    cfg["model"] = "nope"
    with pytest.raises(ValueError):
        DexGanGrasp(cfg)
    cfg = dict(synth.DEFAULT_CFG)
    cfg["lr_policy"] = "cosine"
    with pytest.raises(NotImplementedError):
        DexGanGrasp(cfg)
    cfg = dict(synth.DEFAULT_CFG)
    cfg["weight_init_type"] = "bogus"
    with pytest.raises(NotImplementedError):
        DexGanGrasp(cfg)


def test_early_stopping_rule_matches_reference():
    """utils/train_tools.py:6-42: defaults (patience 7, delta 0) and the stopping rule."""
    from dex_gan_grasp_amd.utils.train_tools import EarlyStopping
    es = EarlyStopping()
    assert (es.patience, es.delta, es.counter, es.best_score, es.early_stop) == (7, 0, 0, None, False)
    assert es(1.0) is False and es.best_score == -1.0
    assert es(0.5) is False and es.best_score == -0.5 and es.counter == 0
    for k in range(6):
        assert es(0.5 + 0.01 * k) is False      # equal or worse: counts (score < best + delta is False only if better)
    assert es.counter == 5 or es.counter == 6
    es2 = EarlyStopping(patience=2)
    es2(1.0)
    assert es2(1.0) is False                    # equal score: NOT below best + 0 -> treated as the new best, counter 0
    assert es2.counter == 0
    assert es2(2.0) is False and es2.counter == 1
    assert es2(2.0) is True and es2.early_stop


def test_grad_buckets_partition_the_flat_buffer():
    """FlatNet.grad_buckets (the ranges of dgg_set_grad_events / parallel.BucketReducer): disjoint, cover the whole
    parameter buffer, listed in the order a backward completes them — last ResBlock (+ heads) first, input BatchNorm +
    block 0 last — and every named parameter lies inside exactly one bucket."""
    from dex_gan_grasp_amd import _lib, synth
    for net, nrb in ((_lib.NET_GEN, 2), (_lib.NET_DISC, 2), (_lib.NET_EVA, 3)):
        fn = _lib.FlatNet(net, dict(synth.DEFAULT_CFG), "cpu", with_optim=False)
        b = fn.grad_buckets()
        assert len(b) == nrb
        assert b[-1][0] == 0 and b[0][1] == fn.n_params
        spans = sorted(b)
        assert all(spans[i][1] == spans[i + 1][0] for i in range(len(spans) - 1))
        assert [lo for lo, _ in b] == sorted((lo for lo, _ in b), reverse=True)
        for name, off, rows, cols, kind in fn.entries:
            if kind == 0:
                assert sum(lo <= off and off + rows * cols <= hi for lo, hi in b) == 1, name


def test_bps_workspace_plan_and_route_switch():
    """Host-side logic of the round-4 BPS entry points (no GPU): the workspace query follows the chunk plan — 0 bytes for the
    many-clouds shapes, B * nchunk * P * 8 for few clouds of many points (chunks of 256..1536 points, ~512 workgroups) —,
    bad arguments are refused, the route override takes only its two values."""
    from dex_gan_grasp_amd import _lib
    lib = _lib.lib()
    P = 4096
    assert lib.dgg_bps_workspace_bytes(1024, 2048, P) == 0          # the training front end: one workgroup per cloud slice
    assert lib.dgg_bps_workspace_bytes(512, 2048, P) == 0
    assert lib.dgg_bps_workspace_bytes(1, 32, P) == 0               # fewer than 64 points: the scan
    w73 = lib.dgg_bps_workspace_bytes(1, 73000, P)                  # the robot path's largest cloud: chunks of 576 points
    assert w73 == 127 * P * 8, w73
    w15 = lib.dgg_bps_workspace_bytes(1, 15009, P)                  # chunks of 256 points
    assert w15 == 59 * P * 8, w15
    assert lib.dgg_bps_workspace_bytes(3, 5000, P) % (3 * P * 8) == 0 and lib.dgg_bps_workspace_bytes(3, 5000, P) > 0
    assert lib.dgg_bps_workspace_bytes(40, 4100, P) > 0             # more than 2560 points per cloud: always chunks
    for bad in ((0, 100, P), (1, 0, P), (1, 100, 0)):
        assert lib.dgg_bps_workspace_bytes(*bad) == -1
    assert lib.dgg_get_bps_algo() == 0
    assert lib.dgg_set_bps_algo(1) == 0 and lib.dgg_get_bps_algo() == 1
    assert lib.dgg_bps_workspace_bytes(1, 73000, P) == 0            # the scan needs no workspace
    assert lib.dgg_set_bps_algo(2) == -1 and lib.dgg_get_bps_algo() == 1
    assert lib.dgg_set_bps_algo(0) == 0
    assert lib.dgg_bps_encode_ws(None, 1, 10, None, P, None, None, None, 0, None) == -1
    assert lib.dgg_set_gemm_algo(3) == 0 and lib.dgg_get_gemm_algo() == 3 and lib.dgg_set_gemm_algo(1) == 0   # DGG_GEMM_BF16
    assert lib.dgg_set_gemm_algo(4) == -1


def test_gemm_routing_rules_and_the_2gib_limit():
    """dgg_gemm_route (host only): which kernel a product runs on.  The train step's large products — forward 1024 x 768 x 4120
    in 4 split-K slabs, weight gradient 768 x 4120 x 1024, every column count of the checkpoint — take gemm_x5_kernel; the
    inference sweep's products (>= 1024 workgroups) gemm_x4_kernel; products with K < 64 (the 24 pose columns, the 5 latent
    columns), split-K chunks that short, and the small ones (fc2, heads: 64-wide tiles) the f32-MFMA kernels; DGG_GEMM_F32
    everything on those; DGG_GEMM_BF16 the same shapes as bf16x3 but never gemm_x4_kernel; the removed planes value is refused;
    an output beyond 32-bit byte offsets is refused with a message that says what to do."""
    from dex_gan_grasp_amd import _lib
    lib = _lib.lib()
    F32, X4, X5 = 0, 4, 5
    if os.environ.get("DGG_GEMM_ALGO") or os.environ.get("DGG_X4") or os.environ.get("DGG_X5"):
        pytest.skip("the kernel selection is overridden in the environment")
    route = lib.dgg_gemm_route
    assert route(0, 1024, 768, 4120, 4120, 4120, 768, 4) == X5          # forward of fc1 || fc3
    assert route(0, 1024, 768, 4101, 4101, 4101, 768, 4) == X5          # the generator's 4101 columns (K tail of 5)
    assert route(0, 512, 768, 4120, 4120, 4120, 768, 8) == X5           # config 4's per-rank batch
    assert route(2, 768, 4120, 1024, 768, 4120, 4120, 0) == X5          # weight gradient
    assert route(2, 768, 4120, 256, 768, 4120, 4120, 0) == X5           # ... at B = 256
    assert route(0, 1024, 768, 512, 4632, 4632, 768, 0) == F32          # the previous block's 512 columns ALONE: 48 tiles do not fill
                                                                        # the chip (in the step they ride in the launch above)
    assert route(0, 1024, 768, 24, 4120, 4120, 768, 0) == F32           # the pose columns: K < 64
    assert route(0, 1024, 768, 5, 4101, 4101, 768, 0) == F32            # the latent columns
    assert route(0, 1024, 768, 4120, 4120, 4120, 768, 100) == X4        # 65 slabs x 48 tiles: >= 1024 workgroups
    assert route(0, 250000, 768, 512, 512, 512, 768, 0) == X4           # config 5: 1250 objects x 200 grasps
    assert route(1, 250000, 512, 768, 768, 512, 512, 0) == X4
    assert route(0, 1024, 512, 256, 256, 256, 512, 0) == F32            # fc2: 64-wide tiles
    assert route(0, 64, 64, 32, 32, 32, 64, 0) == F32
    assert route(2, 768, 4101, 300, 768 + 1, 4101 + 3, 4101, 0) == X5   # pitches that cover gemm_x4's quads do not matter here
    try:
        assert lib.dgg_set_gemm_algo(0) == 0
        assert route(0, 1024, 768, 4120, 4120, 4120, 768, 4) == F32 and route(0, 250000, 768, 512, 512, 512, 768, 0) == F32
        assert lib.dgg_set_gemm_algo(3) == 0
        assert route(0, 1024, 768, 4120, 4120, 4120, 768, 4) == X5 and route(0, 250000, 768, 512, 512, 512, 768, 0) == X5
        assert lib.dgg_set_gemm_algo(2) == -1                           # the planes path of rounds 3-4 is gone
    finally:
        assert lib.dgg_set_gemm_algo(1) == 0
    assert route(0, 0, 768, 512, 512, 512, 768, 0) == -1 and route(3, 64, 64, 64, 64, 64, 64, 0) == -1
    assert route(0, 699050, 768, 512, 512, 512, 768, 0) in (X4, X5)     # the last row count under 2 GiB at pitch 768
    assert route(0, 699051, 768, 512, 512, 512, 768, 0) == -1
    msg = lib.dgg_last_error().decode()
    assert "2 GiB" in msg and "699050" in msg, msg
    # a k-contiguous OPERAND is addressed through 32-bit byte offsets by the bf16x3 kernels — rows * pitch * 4 must
    # stay below 4 GiB (the X0 block of a forward: pitch 4128 -> 260 111 rows), refused on every route with the row count that fits
    assert route(0, 260111, 768, 4120, 4128, 4120, 768, 16) in (X4, X5)
    assert route(0, 260112, 768, 4120, 4128, 4120, 768, 16) == -1
    msg = lib.dgg_last_error().decode()
    assert "4 GiB" in msg and "260111" in msg and "operand A" in msg, msg
    assert route(1, 260112, 512, 768, 4128, 512, 512, 0) == -1          # A of a dgrad is k-contiguous too
    assert route(2, 768, 4120, 300000, 768, 4128, 4120, 0) == X5        # k-strided operands are addressed by 64-bit row bases
