"""GPU parity of the training steps (A10-A12) and of the drop-in Python surface (A13-A15, checkpoints):
HIP path vs the CPU oracle on the same seeded inputs, and vs the fixtures produced by the reference."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import helpers as H
from dex_gan_grasp_amd import synth

pytestmark = pytest.mark.gpu
CFG = dict(synth.DEFAULT_CFG)
DEV = "cuda"


def _mods():
    from dex_gan_grasp_amd import _lib, ops, engine
    return _lib, ops, engine


def _strip(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def _cmp_grads(flat, ref, prefix="", batch=64):
    gv = {k: v.cpu() for k, v in flat.views("grads").items()}
    for n, r in ref.items():
        H.assert_grad_tensor_close(n[len(prefix):], gv[n[len(prefix):]], r, train=True, batch=batch)


def _cmp_state(flat, sd_ref, prev, prefix, step, lr=1e-4):
    """Full-tensor post-step comparison with the oracle (helpers.assert_update_close explains the
    tolerance model); BatchNorm running statistics tightly."""
    sd = flat.state_dict()
    for n, t in sd.items():
        r = sd_ref[prefix + n]
        if "running_" in n:
            d = (t - r).abs()
            if "rb" in n and n.endswith("running_mean"):
                assert float(d.max()) <= 2.5 * lr * (step + 1), n   # integrates the pre-BN bias
            else:
                tol = 1e-4 * r.abs() + 1e-5 * float(r.pow(2).mean().sqrt()) + 1e-7
                assert bool((d <= tol).all()), "%s: max err %.3e" % (n, float(d.max()))
            continue
        H.assert_update_close(n, t, r, prev[prefix + n], step, lr)


def _snap(sd):
    return {k: v.detach().clone() for k, v in sd.items()}


def _sync_from_oracle(flat, sd, adam, prefix=""):
    """Load the oracle's parameters, BN statistics and Adam moments into the flat buffers so that the
    next step starts from IDENTICAL state on both sides (otherwise the noise-driven Adam elements of
    step t perturb the gradients of step t+1 by percents in the eps-dominated reference-init regime)."""
    flat.load_state_dict(sd, prefix)
    mv, vv = flat.views("m"), flat.views("v")
    for n in mv:
        mv[n].copy_(adam.m[prefix + n].to(mv[n].device))
        vv[n].copy_(adam.v[prefix + n].to(vv[n].device))
    flat.step = adam.t


@pytest.mark.parametrize("variant,bname", [("wide", "syn64"), ("init", "real16"), ("init", "syn64")])
def test_cgan_and_eva_steps_vs_oracle_and_goldens(variant, bname):
    L, ops, engine = _mods()
    from oracle import dexgan_oracle as O
    from summ import unpack_mask
    g = H.load("step_goldens.npz")
    has_golden = (variant + "/" + bname + "/cgan0/z") in g.files
    key = "%s/%s/" % (variant, bname)
    batch = H.batches()[bname]
    b = batch["bps_object"].shape[0]
    sd = H.gan_sd(variant)
    fg, fd = L.FlatNet(L.NET_GEN, CFG, DEV), L.FlatNet(L.NET_DISC, CFG, DEV)
    fg.load_state_dict(sd, "generator.")
    fd.load_state_dict(sd, "discriminator.")
    fe = L.FlatNet(L.NET_EVA, CFG, DEV)
    sde = H.eva_sd(variant)
    fe.load_state_dict(sde)
    eng = engine.TrainEngine(CFG, fg, fd, fe, DEV)
    ag = O.AdamState([n for n in O.param_names(sd) if n.startswith("generator.")])
    ad = O.AdamState([n for n in O.param_names(sd) if n.startswith("discriminator.")])
    bps = batch["bps_object"].to(DEV)
    rot9 = batch["rot_matrix"].float().reshape(b, 9).to(DEV)
    tr, jt = batch["transl"].float().to(DEV), batch["joint_conf"].float().to(DEV)
    hp = ops.make_hparams(CFG)
    for step in range(2):
        z = torch.from_numpy(g[key + "cgan%d/z" % step]) if has_golden else synth.make_z(b, 100 + step)
        prev = _snap(sd)
        ref_losses, g_d, g_g, _ = O.cgan_step(sd, H.clone_batch(batch), z, ag, ad, CFG, True, return_grads=True)
        # phases by hand so the gradients can be inspected before each Adam
        ws = eng._cgan_ws(b)
        st = eng._state()
        zd = z.to(DEV)
        args = (L.ptr(bps), L.ptr(rot9), L.ptr(tr), L.ptr(jt), L.ptr(zd))
        L.check(L.lib().dgg_cgan_d_phase(C.byref(eng.ccfg), C.byref(st), C.byref(hp), L.ptr(ws), ws.numel(), b, *args,
                                         L.ptr(eng.losses), L.stream_ptr()))
        _cmp_grads(fd, g_d, "discriminator.", b)
        fd.adam_step(hp.lr_dis)
        L.check(L.lib().dgg_cgan_g_phase(C.byref(eng.ccfg), C.byref(st), C.byref(hp), L.ptr(ws), ws.numel(), b, *args,
                                         L.ptr(eng.losses), L.stream_ptr()))
        _cmp_grads(fg, g_g, "generator.", b)
        fg.adam_step(hp.lr_gen)
        got = eng.losses.cpu().numpy()
        names = ["total_loss_disc", "bce_loss_real", "bce_loss_fake", "gen_loss_fake", "transl_loss", "rot_loss",
                 "conf_loss", "total_loss_gen"]
        for i, n in enumerate(names):
            np.testing.assert_allclose(got[i], float(ref_losses[n]), rtol=1e-4, atol=1e-6, err_msg=n)
            if has_golden:  # the reference's own numbers
                np.testing.assert_allclose(got[i], float(g[key + "cgan%d/loss/%s" % (step, n)]), rtol=1e-4, atol=1e-6,
                                           err_msg="golden " + n)
        _cmp_state(fg, sd, prev, "generator.", step)
        _cmp_state(fd, sd, prev, "discriminator.", step)
        if has_golden:
            cur = fg.state_dict("generator.")
            cur.update(fd.state_dict("discriminator."))
            for n, t in cur.items():
                H.assert_state_close(n, t, g[key + "cgan%d/state/%s" % (step, n)], step)
        _sync_from_oracle(fg, sd, ag, "generator.")
        _sync_from_oracle(fd, sd, ad, "discriminator.")
    # evaluator
    ae = O.AdamState(O.param_names(sde))
    eb = H.eva_batch(bname)
    for step in range(2):
        masks = unpack_mask(g[key + "eva%d/masks" % step], (3, b, 512)) if has_golden else \
            synth.make_dropout_masks(b, 50 + step)
        prev = _snap(sde)
        ref_losses, g_e, _ = O.eva_step(sde, H.clone_batch(eb), ae, CFG, masks, return_grads=True)
        ws = eng._eva_ws(b)
        ebps, erot = eb["bps_object"].to(DEV), eb["rot_matrix"].float().reshape(b, 9).to(DEV)
        etr, elab = eb["transl"].float().to(DEV), eb["label"].float().to(DEV)
        md = masks.to(DEV).contiguous()
        L.check(L.lib().dgg_eva_phase(C.byref(eng.ccfg), L.ptr(fe.params), L.ptr(fe.grads), L.ptr(fe.bn), C.byref(hp),
                                      L.ptr(ws), ws.numel(), b, L.ptr(ebps), L.ptr(erot), L.ptr(etr), L.ptr(elab),
                                      L.ptr(md), 0, L.ptr(eng.losses), L.stream_ptr()))
        _cmp_grads(fe, g_e, "", b)
        fe.adam_step(hp.lr_eva)
        got = eng.losses.cpu().numpy()
        np.testing.assert_allclose(got[L.L_TOTAL_EVA], float(ref_losses["total_loss_eva"]), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(got[L.L_BCE_EVA], float(ref_losses["bce_loss"]), rtol=1e-4, atol=1e-6)
        if has_golden:
            np.testing.assert_allclose(got[L.L_TOTAL_EVA], float(g[key + "eva%d/loss/total_loss_eva" % step]),
                                       rtol=1e-4, atol=1e-6)
            for n, t in fe.state_dict().items():
                H.assert_state_close(n, t, g[key + "eva%d/state/%s" % (step, n)], step)
        _cmp_state(fe, sde, prev, "", step)
        _sync_from_oracle(fe, sde, ae)


def test_step_properties_at_full_size():
    """BASELINE batch 1024: size-independent properties — (i) the fused single-call step equals the
    hand-sequenced phases bit for bit (determinism, no atomics), (ii) a batch made of two identical
    halves gives the same BatchNorm statistics and mean gradients as one half (batch-mean
    linearity of the whole backward), (iii) is_train_gen=False leaves the generator untouched."""
    L, ops, engine = _mods()
    b = 1024
    batch = synth.make_batch(b, 1)
    z = synth.make_z(b, 3)
    sd = H.gan_sd("wide")

    def run(bt, zz, train_gen=True):
        fg, fd = L.FlatNet(L.NET_GEN, CFG, DEV), L.FlatNet(L.NET_DISC, CFG, DEV)
        fg.load_state_dict(sd, "generator.")
        fd.load_state_dict(sd, "discriminator.")
        fe = L.FlatNet(L.NET_EVA, CFG, DEV)
        eng = engine.TrainEngine(CFG, fg, fd, fe, DEV)
        n = bt["bps_object"].shape[0]
        out = eng.cgan_step(bt["bps_object"].to(DEV), bt["rot_matrix"].float().reshape(n, 9).to(DEV),
                            bt["transl"].float().to(DEV), bt["joint_conf"].float().to(DEV), zz.to(DEV), train_gen)
        return fg, fd, out

    fg1, fd1, o1 = run(batch, z)
    fg2, fd2, o2 = run(batch, z)
    assert torch.equal(fg1.params, fg2.params) and torch.equal(fd1.params, fd2.params) and torch.equal(o1, o2)
    half = {k: v[:512] for k, v in batch.items()}
    dup = {k: torch.cat([v[:512], v[:512]]) for k, v in batch.items()}
    fgh, fdh, oh = run(half, z[:512])
    fgd, fdd, od = run(dup, torch.cat([z[:512], z[:512]]))
    np.testing.assert_allclose(od.cpu().numpy()[:8], oh.cpu().numpy()[:8], rtol=2e-5, atol=1e-6)
    # gradients: duplicated batch -> identical mean gradients -> identical first Adam step (tolerance: summation order)
    gd, gh = fgd.grads.cpu(), fgh.grads.cpu()
    assert float((gd - gh).abs().max()) <= 2e-4 * float(gh.abs().max())
    gd, gh = fdd.grads.cpu(), fdh.grads.cpu()
    assert float((gd - gh).abs().max()) <= 2e-4 * float(gh.abs().max())
    fg3, fd3, o3 = run(batch, z, train_gen=False)
    ref = L.FlatNet(L.NET_GEN, CFG, DEV)
    ref.load_state_dict(sd, "generator.")
    assert torch.equal(fg3.params, ref.params)
    assert torch.equal(fd3.params, fd1.params)  # D is only updated in the D phase


def test_joint_step_equals_separate_steps():
    L, ops, engine = _mods()
    b = 64
    gb, eb = synth.make_batch(b, 1), synth.make_batch(b, 2, with_label=True)
    z = synth.make_z(b, 3)

    def fresh():
        fg, fd, fe = (L.FlatNet(k, CFG, DEV) for k in (L.NET_GEN, L.NET_DISC, L.NET_EVA))
        sd = H.gan_sd("wide")
        fg.load_state_dict(sd, "generator.")
        fd.load_state_dict(sd, "discriminator.")
        fe.load_state_dict(H.eva_sd("wide"))
        return engine.TrainEngine(CFG, fg, fd, fe, DEV)
    f = lambda t: t.float().to(DEV)
    g_in = (f(gb["bps_object"]), f(gb["rot_matrix"]).reshape(b, 9), f(gb["transl"]), f(gb["joint_conf"]))
    e_in = (f(eb["bps_object"]), f(eb["rot_matrix"]).reshape(b, 9), f(eb["transl"]), f(eb["label"]))
    e1, e2, e3 = fresh(), fresh(), fresh()
    zd = z.to(DEV)
    # e3: pipelined across steps — the next step's D real pass is prefetched behind this step's third D forward;
    # alternating batches (g_in / g_alt) so that a prefetch of the wrong tensors could not go unnoticed
    gb2 = synth.make_batch(b, 11)
    g_alt = (f(gb2["bps_object"]), f(gb2["rot_matrix"]).reshape(b, 9), f(gb2["transl"]), f(gb2["joint_conf"]))
    seq = [g_in, g_alt, g_in, g_in]
    for i in range(3):
        cur, nxt = seq[i], seq[i + 1]
        e1.cgan_step(*cur, zd, True)
        l1 = e1.eva_step(*e_in, None, 77 + i).clone()
        l2 = e2.joint_step(cur, zd, e_in, True, None, 77 + i).clone()
        l3 = e3.joint_step(cur, zd, e_in, True, None, 77 + i, next_gan_inputs=nxt if i < 2 else None).clone()
        torch.cuda.synchronize()
        assert torch.equal(l1[:10], l2[:10]) and torch.equal(l1[:10], l3[:10])
    for a, c, d in ((e1.G, e2.G, e3.G), (e1.D, e2.D, e3.D), (e1.E, e2.E, e3.E)):
        assert torch.equal(a.params, c.params) and torch.equal(a.m, c.m) and torch.equal(a.bn, c.bn)
        assert torch.equal(a.params, d.params) and torch.equal(a.m, d.m) and torch.equal(a.bn, d.bn)
    # a prefetch that is not picked up (different tensors in the next call) is dropped and redone: only D's running
    # statistics see the extra pass, the update itself is the unpipelined one
    e4, e5 = fresh(), fresh()
    e4.joint_step(g_in, zd, e_in, True, None, 5, next_gan_inputs=g_alt)
    e5.joint_step(g_in, zd, e_in, True, None, 5)
    g_in2 = tuple(t.clone() for t in g_in)
    e4.joint_step(g_in2, zd, e_in, True, None, 6)
    e5.joint_step(g_in2, zd, e_in, True, None, 6)
    torch.cuda.synchronize()
    assert torch.equal(e4.G.params, e5.G.params) and torch.equal(e4.E.params, e5.E.params)


# ------------------------------------------------------------------ drop-in Python surface
def _wrapper(variant, is_train=True):
    from dex_gan_grasp_amd.models.dexgangrasp import DexGanGrasp
    cfg = dict(CFG)
    cfg["is_train"] = is_train
    cfg["train_ffhevaluator"] = True
    m = DexGanGrasp(cfg)
    m.DexGANGrasp.load_state_dict(H.gan_sd(variant))
    m.DexEvaluator.load_state_dict(H.eva_sd(variant))
    return m


def test_wrapper_state_dict_layout():
    m = _wrapper("init")
    sd = m.DexGANGrasp.state_dict()
    spec = synth.net_spec("dexgangrasp")
    assert list(sd.keys()) == [n for n, _, _ in spec]          # 82 entries, reference order
    assert len(sd) == 82 and len(m.DexEvaluator.state_dict()) == 55
    for n, shape, _ in spec:
        assert tuple(sd[n].shape) == tuple(shape), n
    ref = H.gan_sd("init")
    for n in sd:
        assert torch.equal(sd[n].cpu(), ref[n]), n


@pytest.mark.parametrize("variant,bname", [("wide", "syn64"), ("init", "real16")])
def test_wrapper_update_steps_vs_reference_goldens(variant, bname, tmp_path):
    from summ import unpack_mask
    g = H.load("step_goldens.npz")
    key = "%s/%s/" % (variant, bname)
    m = _wrapper(variant)
    batch = H.batches()[bname]
    b = batch["bps_object"].shape[0]
    for step in range(2):
        m._z_override = torch.from_numpy(g[key + "cgan%d/z" % step])
        data = H.clone_batch(batch)
        losses = m.update_dexgangrasp(data, is_train_gen=True)
        assert tuple(data["rot_matrix"].shape) == (b, 9)        # :810 mutates the caller's dict
        for k, v in losses.items():
            assert v.dim() == 0
            np.testing.assert_allclose(v.item(), float(g[key + "cgan%d/loss/%s" % (step, k)]), rtol=1e-4, atol=1e-6,
                                       err_msg=k)
        for n, t in m.DexGANGrasp.state_dict().items():
            H.assert_state_close(n, t.cpu(), g[key + "cgan%d/state/%s" % (step, n)], step)
    sd = m.DexGANGrasp.state_dict()
    assert int(sd["generator.gen_bn1.num_batches_tracked"]) == 2
    assert int(sd["discriminator.disc_bn1.num_batches_tracked"]) == 6
    eb = H.eva_batch(bname)
    for step in range(2):
        m._drop_keep = unpack_mask(g[key + "eva%d/masks" % step], (3, b, 512))
        losses = m.update_dexevaluator(H.clone_batch(eb))
        for k, v in losses.items():
            np.testing.assert_allclose(v.item(), float(g[key + "eva%d/loss/%s" % (step, k)]), rtol=1e-4, atol=1e-6)
        for n, t in m.DexEvaluator.state_dict().items():
            H.assert_state_close(n, t.cpu(), g[key + "eva%d/state/%s" % (step, n)], step)
    # eval-mode losses after the two steps (train.py:65,110-111)
    m._z_override = None
    torch.manual_seed(300)
    m._z_override = torch.randn((b, 5))  # the reference drew Z on the CPU generator after manual_seed(300)
    el = m.eval_dexgangrasp_generator_loss(H.clone_batch(batch))
    for k, v in el.items():
        np.testing.assert_allclose(float(v), float(g[key + "evalloss_gen/" + k]), rtol=2e-3, atol=1e-5, err_msg=k)
    el = m.eval_dexevaluator_loss(H.clone_batch(eb))
    for k, v in el.items():
        np.testing.assert_allclose(float(v), float(g[key + "evalloss_eva/" + k]), rtol=2e-3, atol=1e-5, err_msg=k)
    # checkpoint round trip in the reference's file layout (dexgangrasp.py:662-712)
    m.cfg["save_dir"] = str(tmp_path)
    m.save_dexgenerator("3", 3)
    m.save_dexevaluator("3", 3)
    ck = torch.load(os.path.join(str(tmp_path), "3_gen_net.pt"), map_location="cpu", weights_only=False)
    assert set(ck.keys()) == {"epoch", "ffhgenerator_state_dict", "optim_ffhgenerator_state_dict",
                              "scheduler_ffhgenerator_state_dict", "optim_ffhgan_generator_state_dict",
                              "scheduler_ffhgan_generator_state_dict", "optim_ffhgan_discriminator_state_dict",
                              "scheduler_ffhgan_discriminator_state_dict"}
    assert len(ck["ffhgenerator_state_dict"]) == 82
    st = ck["optim_ffhgan_discriminator_state_dict"]["state"]
    assert len(st) == 24 and float(st[0]["step"]) == 2.0  # 24 trainable tensors in the discriminator
    cke = torch.load(os.path.join(str(tmp_path), "3_eva_net.pt"), map_location="cpu", weights_only=False)
    assert set(cke.keys()) == {"epoch", "ffhevaluator_state_dict", "optim_ffhevaluator_state_dict",
                               "scheduler_ffhevaluator_state_dict"}
    m2 = _wrapper("init")
    m2.load_dexgenerator(3, str(tmp_path))
    m2.load_dexevaluator(3, str(tmp_path))
    for n, t in m.DexGANGrasp.state_dict().items():
        assert torch.equal(t.cpu(), m2.DexGANGrasp.state_dict()[n].cpu()), n
    d1, d2 = m.DexGANGrasp.discriminator._flat, m2.DexGANGrasp.discriminator._flat
    assert torch.equal(d1.m, d2.m) and torch.equal(d1.v, d2.v) and d2.step == 2
    # a third step on both must agree exactly (resume == continue)
    for mm in (m, m2):
        mm._z_override = synth.make_z(b, 9)
    l1 = m.update_dexgangrasp(H.clone_batch(batch))
    l2 = m2.update_dexgangrasp(H.clone_batch(batch))
    for k in l1:
        assert float(l1[k]) == float(l2[k]), k


@pytest.mark.parametrize("variant", ["init", "wide"])
def test_wrapper_inference_vs_reference_goldens(variant):
    g = H.load("inference_goldens.npz")
    bps = H.load("bps_real_objects.npz")["dists_red_mug"]
    key = variant + "/"
    m = _wrapper(variant, is_train=False)
    n = 32
    z = torch.from_numpy(g[key + "z"])
    # generate_grasps draws Z with torch.randn on the device; inject the reference's latent instead
    gan = m.DexGANGrasp
    gan.eval()
    orig = torch.randn
    try:
        torch.randn = lambda *a, **k: z.to(k.get("device", "cpu"))
        grasps = m.generate_grasps(bps, n_samples=n, return_arr=True)
    finally:
        torch.randn = orig
    for k in ["transl", "joint_conf", "z", "rot_matrix"]:
        np.testing.assert_allclose(grasps[k], g[key + "gen/" + k], rtol=1e-4, atol=1e-4, err_msg=k)
    p = m.evaluate_grasps(bps, {k: v.copy() for k, v in grasps.items()})
    np.testing.assert_allclose(p, g[key + "scores"], rtol=1e-4, atol=1e-4)
    thresh = float(g[key + "thresh"])
    filt = m.filter_grasps(bps, {k: v.copy() for k, v in grasps.items()}, thresh=thresh)
    # scores within 1e-4 of the threshold could legitimately flip membership
    margin = np.abs(g[key + "scores"] - thresh).min()
    if margin > 2e-4:
        for k in ["transl", "joint_conf", "z", "rot_matrix"]:
            np.testing.assert_allclose(filt[k], g[key + "filt/" + k], rtol=1e-4, atol=1e-4, err_msg=k)
        assert tuple(filt["bps_object"].shape) == tuple(g[key + "filt/bps_object_shape"])
    with pytest.raises(ValueError):
        m.filter_grasps(bps, {k: v.copy() for k, v in grasps.items()}, thresh=1.1)
    # N1: the three-threshold pipeline with a single evaluator pass == three filter_grasps calls
    ths = [float(np.quantile(p, q)) for q in (0.2, 0.5, 0.8)] + [1.1]
    multi = m.filter_grasps_multi(bps, {k: v.copy() for k, v in grasps.items()}, ths)
    assert multi[3] is None
    for th, fm in zip(ths[:3], multi[:3]):
        single = m.filter_grasps(bps, {k: v.copy() for k, v in grasps.items()}, thresh=th)
        assert set(single.keys()) == set(fm.keys())
        for k in single:
            np.testing.assert_array_equal(single[k], fm[k])
    fd = m.filter_grasps_discriminator(bps, {k: v.copy() for k, v in grasps.items()}, thresh=0.0)
    np.testing.assert_allclose(fd["transl"], g[key + "filt_disc/transl"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("variant", ["init", "wide"])
def test_config1_magd_on_eval_batch_objects(variant):
    """BASELINE config 1 (eval.py:309-344 on data/eval_batch.pth) through the HIP path: per object of the
    batch, generate_grasps on its BPS row with the reference's latent, MAGD (eval.metrics, host NumPy by design: at most
    a few dozen grasps per object) against the object's rows — equal to the reference's own four numbers (fixture
    magd_eval_batch.npz, generated by running the reference) and to the oracle's on the oracle's grasps."""
    from dex_gan_grasp_amd.eval import metrics
    from oracle import dexgan_oracle as O
    g = H.load("magd_eval_batch.npz")
    m = _wrapper(variant, is_train=False)
    m.DexGANGrasp.eval()
    sd = H.gan_sd(variant)
    tot, tot_ref = np.zeros(4), np.zeros(4)
    for oi in range(len(g["objects"])):
        rows = np.nonzero(g["obj_index"] == oi)[0]
        gt = {k: g[k][rows] for k in ("rot_matrix", "transl", "joint_conf")}
        key = "%s/%d/" % (variant, oi)
        z = torch.from_numpy(g[key + "z"])
        orig = torch.randn
        try:
            torch.randn = lambda *a, **k: z.to(k.get("device", "cpu"))
            pred = m.generate_grasps(g["bps_rows"][oi], n_samples=len(rows), return_arr=True)
        finally:
            torch.randn = orig
        for k in ("rot_matrix", "transl", "joint_conf"):
            np.testing.assert_allclose(pred[k], g[key + "pred/" + k], rtol=1e-4, atol=1e-4, err_msg=k)
        got = np.array(metrics.magd_for_grasp_distribution({k: np.asarray(v).copy() for k, v in pred.items()}, gt))
        op = O.generate_grasps(sd, g["bps_rows"][oi], z)
        r = O.magd(gt, op)
        # the sums move with the grasps (1e-4 on poses); the nearest-neighbour assignment — coverage — is the same
        np.testing.assert_allclose(got[:3], g[key + "magd"][:3], rtol=2e-4)
        np.testing.assert_allclose(got[:3], [r["transl"], r["rot"], r["joint"]], rtol=2e-4)
        assert got[3] == g[key + "magd"][3] == r["coverage"]
        tot += got
        tot_ref += g[key + "magd"]
    np.testing.assert_allclose(tot[:3], tot_ref[:3], rtol=2e-4)      # eval.py:346-359 sums these over the objects


def test_refine_grasps_gradient_matches_oracle():
    """N4 (dexgangrasp.py:516-541, 619-660): the gradient-refinement loop, step for step, against the same loop
    written with torch autograd over the oracle's evaluator (gradients accumulate in .grad, as in the reference)."""
    from oracle import dexgan_oracle as O
    m = _wrapper("wide", is_train=False)
    m.DexEvaluator.eval()
    batch = H.batches()["syn8"]
    steps = 3
    data = {"bps_object": batch["bps_object"].clone(), "rot_matrix": batch["rot_matrix"].clone(),
            "transl": batch["transl"].clone(), "joint_conf": batch["joint_conf"].clone()}
    refined_data, refined_success = m.refine_grasps(data, "gradient", num_refine_steps=steps)
    assert len(refined_data) == steps + 1 and len(refined_success) == steps + 1
    # oracle loop
    sde = H.eva_sd("wide")
    rot = batch["rot_matrix"].float().clone().requires_grad_(True)
    tr = batch["transl"].float().clone().requires_grad_(True)
    want_p = []
    for _ in range(steps):
        p = O.evaluator_forward(sde, batch["bps_object"].float(), rot, tr, False)
        p.squeeze(-1).backward(torch.ones(p.shape[0]))
        alpha = torch.min(0.01 / torch.norm(tr.grad, p=2, dim=-1), torch.tensor(1.0))
        tr.data += tr.grad * alpha[:, None]
        rot.data += rot.grad * alpha.view(-1, *([1] * (rot.dim() - 1)))
        want_p.append(p.detach().squeeze().numpy())
    want_p.append(O.evaluator_forward(sde, batch["bps_object"].float(), rot, tr, False).detach().squeeze().numpy())
    for got, want in zip(refined_success, want_p):
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(refined_data[-1]["transl"], tr.detach().numpy(), rtol=0, atol=2e-5)   # steps are <= 1 cm
    np.testing.assert_allclose(refined_data[-1]["rot_matrix"].reshape(8, -1), rot.detach().numpy().reshape(8, -1),
                               rtol=0, atol=2e-4)
    np.testing.assert_array_equal(refined_data[-1]["joint_conf"], refined_data[0]["joint_conf"])
    assert np.abs(refined_data[-1]["transl"] - refined_data[0]["transl"]).max() > 1e-4      # it did move
    with pytest.raises(NotImplementedError):
        m.refine_grasps(dict(data), "sampling")


def test_sweep_objects_matches_per_object_pipeline():
    """Config 5 in one generator + one evaluator pass (sweep_objects) against the reference's per-object sequence
    generate_grasps -> filter_grasps x3 (here filter_grasps_multi), same latents."""
    m = _wrapper("wide", is_train=False)
    n_obj, n = 5, 64
    rows = H.batches()["syn64"]["bps_object"][:n_obj].float().numpy()
    thresholds = (0.3, 0.5, 0.7)
    zs, per_obj = [], []
    for o in range(n_obj):
        torch.manual_seed(100 + o)
        zs.append(torch.randn((n, m.DexGANGrasp.latentD), dtype=torch.float32, device=DEV))
        torch.manual_seed(100 + o)                       # generate_poses draws the same latents
        g = m.generate_grasps(rows[o], n, return_arr=True)
        per_obj.append((g, m.filter_grasps_multi(rows[o], g, thresholds=thresholds)))
    out = m.sweep_objects(rows, n, thresholds=thresholds, z=torch.cat(zs))
    assert out["score"].shape == (n_obj, n) and out["count"].shape == (len(thresholds), n_obj)
    for o in range(n_obj):
        g, filt = per_obj[o]
        np.testing.assert_allclose(out["transl"][o].cpu().numpy(), g["transl"], rtol=0, atol=2e-5)  # other split-K plan
        np.testing.assert_allclose(out["rot_matrix"][o].cpu().numpy(), g["rot_matrix"], rtol=0, atol=2e-5)
        np.testing.assert_allclose(out["joint_conf"][o].cpu().numpy(), g["joint_conf"], rtol=0, atol=2e-5)
        score = out["score"][o].cpu().numpy()
        for t, thr in enumerate(thresholds):
            cnt = int(out["count"][t, o])
            near = int(((score > thr - 1e-5) & (score < thr + 1e-5)).sum())     # scores within rounding of the cut
            ref_cnt = 0 if filt[t] is None else filt[t]["transl"].shape[0]
            assert abs(cnt - ref_cnt) <= near, (o, thr, cnt, ref_cnt)
            if cnt and filt[t] is not None and near == 0:
                sel = out["order"][o, :cnt].cpu().numpy()
                np.testing.assert_allclose(np.sort(out["transl"][o].cpu().numpy()[sel], axis=0),
                                           np.sort(filt[t]["transl"], axis=0), rtol=0, atol=2e-5)
    assert int(out["count"].sum()) > 0          # the thresholds actually select something
    # tensor-returning form of generate_grasps (device-side translate_along_axis on the generator's strided output)
    torch.manual_seed(100)
    gt = m.generate_grasps(rows[0], n, return_arr=False)
    assert gt["transl"].is_cuda
    np.testing.assert_allclose(gt["transl"].cpu().numpy(), per_obj[0][0]["transl"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(gt["joint_conf"].cpu().numpy(), per_obj[0][0]["joint_conf"], rtol=0, atol=2e-6)


def test_autograd_path_matches_oracle():
    """Module forward + torch backward (the non-fused path: e.g. refine_grasps' input gradients)."""
    from oracle import dexgan_oracle as O
    m = _wrapper("wide")
    batch = H.batches()["syn8"]
    b = 8
    eva = m.DexEvaluator
    eva.eval()
    data = {"bps_object": batch["bps_object"].to(DEV), "rot_matrix": batch["rot_matrix"].float().to(DEV).requires_grad_(True),
            "transl": batch["transl"].float().to(DEV).requires_grad_(True)}
    p = eva(data)
    p.squeeze().backward(torch.ones(b, device=DEV))
    sde = H.eva_sd("wide")
    rot = batch["rot_matrix"].float().clone().requires_grad_(True)
    tr = batch["transl"].float().clone().requires_grad_(True)
    pr = O.evaluator_forward(sde, batch["bps_object"], rot, tr, False)
    pr.squeeze().backward(torch.ones(b))
    np.testing.assert_allclose(p.detach().cpu().numpy(), pr.detach().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(data["transl"].grad.cpu().numpy(), tr.grad.numpy(), rtol=2e-3,
                               atol=2e-3 * float(tr.grad.abs().max()))
    np.testing.assert_allclose(data["rot_matrix"].grad.cpu().numpy(), rot.grad.numpy(), rtol=2e-3,
                               atol=2e-3 * float(rot.grad.abs().max()))
    # generator through autograd in train mode: parameter gradients land in Parameter.grad
    gen = m.DexGANGrasp.generator
    gen.train()
    z = synth.make_z(b, 3).to(DEV)
    y = gen(z, batch["bps_object"].to(DEV))
    loss = (y["transl"] ** 2).sum() + y["joint_conf"].sum() + (y["rot_6D"] * 0.5).sum()
    loss.backward()
    sd = _strip(H.gan_sd("wide"), "generator.")
    for n in O.param_names(sd):
        sd[n].requires_grad_(True)
    yr = O.generator_forward(sd, z.cpu(), batch["bps_object"], True)
    lr = (yr["transl"] ** 2).sum() + yr["joint_conf"].sum() + (yr["rot_6D"] * 0.5).sum()
    lr.backward()
    for n, p_ in gen.named_parameters():
        if H.is_pre_bn_bias(n):
            continue
        r = sd[n].grad
        assert float((p_.grad.cpu() - r).abs().max()) <= 2e-3 * float(r.abs().max()) + 1e-12, n


def test_bps_encoder_mirror(tmp_path):
    from dex_gan_grasp_amd.data.bps_encoder import BPSEncoder
    d = H.load("bps_real_objects.npz")
    path = os.path.join(str(tmp_path), "basis_point_set.npy")
    np.save(path, d["basis"])
    enc = BPSEncoder(CFG, path)
    out = enc.encode_pcd_with_bps(d["cloud_red_mug"].astype(np.float64))
    assert out.shape == (4096,) and out.dtype == np.float32
    want = d["dists_red_mug"].reshape(-1)
    assert np.abs(out.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64)).max() <= 2
    with pytest.raises(AssertionError):
        enc.encode_pcd_with_bps(d["cloud_red_mug"] + 1.0)


# ------------------------------------------------------------------ BASELINE config 3: plain-bf16 fc stacks (DGG_GEMM_BF16)
@pytest.mark.parametrize("variant", ["init", "wide"])
def test_bf16_mode_evaluator_step_tolerance_vs_fp32(variant):
    """BASELINE config 3 as worded ("DexEvaluator joint training, batch=1024, bf16 MFMA fc stacks"): DGG_GEMM_BF16 — the large
    products in plain bf16 (fp32 master weights rounded while they are staged, one product, fp32 accumulate).  The reference
    is fp32 only (dexgangrasp.py:70-71), so the tolerance of this mode is DEFINED here, against the fp32-accurate default
    (bf16x3, itself pinned to the oracle and the reference's goldens at 1e-4) on the same inputs, weights and dropout masks:
      forward scores        |p_bf16 - p_fp32| <= 2e-2          (measured ~2e-3 "init", ~1e-2 "wide": 8 significand bits on
                                                                K = 4108 dot products, three blocks deep)
      BCE loss              relative 2e-2
      gradient tensors      cosine >= 0.98 with the fp32 gradients (the direction an optimiser follows), norm within 10 %
    — outside the 1e-4 contract by construction: never the default, never the headline dtype."""
    L, ops, engine = _mods()
    lib = L.lib()
    b = 1024
    sde = H.eva_sd(variant)
    batch = synth.make_batch(b, 2, with_label=True)
    masks = synth.make_dropout_masks(b, 4).to(DEV)
    f = lambda t: t.float().to(DEV).contiguous()
    ins = (f(batch["bps_object"]), f(batch["rot_matrix"]).reshape(b, 9), f(batch["transl"]), f(batch["label"]))
    if os.environ.get("DGG_GEMM_ALGO"):
        pytest.skip("DGG_GEMM_ALGO overrides the default GEMM arithmetic this test compares with")
    assert lib.dgg_get_gemm_algo() == 1
    res = {}
    try:
        for algo in (1, 3):
            assert lib.dgg_set_gemm_algo(algo) == 0
            fe = L.FlatNet(L.NET_EVA, CFG, DEV)
            fe.load_state_dict(sde)
            fg, fd = L.FlatNet(L.NET_GEN, CFG, DEV), L.FlatNet(L.NET_DISC, CFG, DEV)
            eng = engine.TrainEngine(CFG, fg, fd, fe, DEV)
            p = ops.eva_forward(fe, ins[0], ins[1], ins[2], True, masks).cpu()
            losses = eng.eva_step(*ins, masks, 0).cpu().numpy()
            grads = {k: v.cpu().clone() for k, v in fe.views("grads").items()}
            res[algo] = (p, float(losses[L.L_TOTAL_EVA]), grads)
            eng.close()
    finally:
        assert lib.dgg_set_gemm_algo(1) == 0
    p32, l32, g32 = res[1]
    p16, l16, g16 = res[3]
    dp = float((p16 - p32).abs().max())
    assert dp > 1e-6, "the bf16 mode must actually run bf16 products"
    worst_cos, worst_norm = 1.0, 0.0
    for n, g in g32.items():
        if H.is_pre_bn_bias(n) or float(g.abs().max()) == 0.0:
            continue
        a, c = g.double().flatten(), g16[n].double().flatten()
        cos = float((a * c).sum() / (a.norm() * c.norm() + 1e-300))
        worst_cos = min(worst_cos, cos)
        worst_norm = max(worst_norm, abs(float(c.norm() / a.norm()) - 1.0))
    print("bf16 mode vs fp32-accurate default (%s weights, B = %d): max |dp| = %.2e, loss %.6f vs %.6f (rel %.2e), "
          "worst gradient cosine %.5f, worst norm deviation %.2e" % (variant, b, dp, l16, l32, abs(l16 - l32) / abs(l32),
                                                                      worst_cos, worst_norm))
    assert dp <= 2e-2
    assert abs(l16 - l32) <= 2e-2 * abs(l32)
    assert worst_cos >= 0.98 and worst_norm <= 0.10
