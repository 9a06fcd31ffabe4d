"""CPU tests: the oracle (oracle/) against fixtures produced by the reference itself
(tests/golden/make_goldens.py).  This is what pins the oracle; the GPU tests then compare the
HIP path with the oracle and with the same fixtures."""
import numpy as np
import pytest
import torch

import helpers as H
from oracle import bps_oracle, dexgan_oracle as O
from dex_gan_grasp_amd import synth
from summ import unpack_mask

CFG = synth.DEFAULT_CFG


# ------------------------------------------------------------------ A1: BPS known-answer vectors
def _ulp_diff(a, b):
    ai = a.view(np.int32).astype(np.int64)
    bi = b.view(np.int32).astype(np.int64)
    return np.abs(ai - bi)


def test_bps_oracle_reproduces_reference_files():
    d = H.load("bps_real_objects.npz")
    basis = d["basis"]
    assert basis.shape == (4096, 3) and basis.dtype == np.float32
    for name in d["names"]:
        cloud = d["cloud_" + str(name)]
        want = d["dists_" + str(name)].reshape(-1)
        got, ids = bps_oracle.encode_f32(cloud, basis)
        assert got.dtype == np.float32 and want.dtype == np.float32
        assert _ulp_diff(got, want).max() <= 2, name
        # index parity with the fp64 brute force, near-ties (gap < 1e-6) accept either index
        d64, i64, i2, gap = bps_oracle.encode_f64(cloud, basis)
        bad = (ids != i64) & ~((gap < 1e-6) & (ids == i2))
        assert not bad.any(), name
        assert np.abs(got - d64).max() < 3e-8  # 2 ulp at 0.2
        enc = bps_oracle.encode_pcd_with_bps(cloud.astype(np.float64), basis)
        assert enc.shape == (4096,)


def test_bps_oracle_c_matches_numpy_small():
    basis = H.load("bps_real_objects.npz")["basis"]
    cloud = synth.make_clouds(1, 257, 11)[0].numpy()
    dc, ic = bps_oracle.encode_f32(cloud, basis)
    dn, inn = bps_oracle.encode_numpy(cloud, basis)
    assert (ic == inn).mean() > 0.999
    assert _ulp_diff(dc, dn).max() <= 2


def test_bps_oracle_edge_cases():
    basis = H.load("bps_real_objects.npz")["basis"]
    one = np.array([[0.01, -0.02, 0.03]], np.float32)
    d, i = bps_oracle.encode_f32(one, basis)
    assert (i == 0).all()
    np.testing.assert_allclose(d, np.linalg.norm(basis - one, axis=1), rtol=1e-6)
    dup = np.repeat(one, 5, 0)  # exact ties -> lowest index wins
    d2, i2 = bps_oracle.encode_f32(dup, basis)
    assert (i2 == 0).all() and (d2 == d).all()
    far = one + 1.0
    with pytest.raises(AssertionError):
        bps_oracle.encode_pcd_with_bps(far, basis)


# ------------------------------------------------------------------ A2-A6: forwards
@pytest.mark.parametrize("variant", ["init", "wide"])
@pytest.mark.parametrize("bname", ["real16", "syn8", "syn64"])
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_forward_vs_reference(variant, bname, mode):
    g = H.load("forward_goldens.npz")
    key = "%s/%s/%s/" % (variant, bname, mode)
    batch = H.batches()[bname]
    b = batch["bps_object"].shape[0]
    z = synth.make_z(b, 3)
    train = mode == "train"
    sd = H.gan_sd(variant)
    sde = H.eva_sd(variant)
    with torch.no_grad():
        y = O.generator_forward(sd, z, batch["bps_object"], train, "generator.")
        r = O.rot_matrix_from_ortho6d(y["rot_6D"])
        tol = dict(rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(y["rot_6D"].numpy(), g[key + "gen_rot_6D"], **tol)
        np.testing.assert_allclose(y["transl"].numpy(), g[key + "gen_transl"], **tol)
        np.testing.assert_allclose(y["joint_conf"].numpy(), g[key + "gen_joint_conf"], **tol)
        np.testing.assert_allclose(r.numpy(), g[key + "gen_rot_matrix"], **tol)
        dr = O.discriminator_forward(sd, batch["bps_object"], batch["rot_matrix"], batch["transl"],
                                     batch["joint_conf"], train, "discriminator.")
        df = O.discriminator_forward(sd, batch["bps_object"], r, y["transl"], y["joint_conf"], train, "discriminator.")
        np.testing.assert_allclose(dr.numpy(), g[key + "disc_real"], **tol)
        np.testing.assert_allclose(df.numpy(), g[key + "disc_fake"], **tol)
        masks = None
        if train:
            masks = unpack_mask(g[key + "eva_masks"], (3, b, 512))
        pe = O.evaluator_forward(sde, batch["bps_object"], batch["rot_matrix"], batch["transl"], train, masks)
        np.testing.assert_allclose(pe.numpy(), g[key + "eva"], **tol)
    if train:
        for n in ["generator.gen_bn1.running_mean", "generator.gen_bn1.running_var",
                  "discriminator.disc_bn1.running_var", "discriminator.disc_rb2.bn2.running_mean"]:
            H.assert_summary_close(sd[n], g[key + "bn/" + n], 1e-5, 1e-6, n)
        assert int(sd["discriminator.disc_bn1.num_batches_tracked"]) == 2
        assert int(sd["generator.gen_bn1.num_batches_tracked"]) == 1


# ------------------------------------------------------------------ A7-A12: train steps
@pytest.mark.parametrize("variant", ["init", "wide"])
@pytest.mark.parametrize("bname", ["real16", "syn64"])
def test_train_steps_vs_reference(variant, bname):
    g = H.load("step_goldens.npz")
    key = "%s/%s/" % (variant, bname)
    batch = H.batches()[bname]
    sd = H.gan_sd(variant)
    ag = O.AdamState([n for n in O.param_names(sd) if n.startswith("generator.")])
    ad = O.AdamState([n for n in O.param_names(sd) if n.startswith("discriminator.")])
    for step in range(2):
        sk = key + "cgan%d/" % step
        z = torch.from_numpy(g[sk + "z"])
        losses, g_d, g_g, _ = O.cgan_step(sd, H.clone_batch(batch), z, ag, ad, CFG, True, return_grads=True)
        for k, v in losses.items():
            np.testing.assert_allclose(float(v), float(g[sk + "loss/" + k]), rtol=2e-5, atol=1e-7, err_msg=k)
        for n, t in g_d.items():
            sib = H.rms_of_summary(g[sk + "grad_d/" + H.sibling_bn_bias(n)])
            H.assert_grad_close(n, t, g[sk + "grad_d/" + n], sib, 2e-3, 2e-4)
        for n, t in g_g.items():
            sib = H.rms_of_summary(g[sk + "grad_g/" + H.sibling_bn_bias(n)])
            H.assert_grad_close(n, t, g[sk + "grad_g/" + n], sib, 2e-3, 2e-4)
        for n, t in sd.items():
            H.assert_state_close(n, t, g[sk + "state/" + n], step)
    # num_batches_tracked: G 1/step, D 3/step
    assert int(sd["generator.gen_bn1.num_batches_tracked"]) == 2
    assert int(sd["discriminator.disc_bn1.num_batches_tracked"]) == 6

    sde = H.eva_sd(variant)
    ae = O.AdamState(O.param_names(sde))
    eb = H.eva_batch(bname)
    b = eb["bps_object"].shape[0]
    for step in range(2):
        sk = key + "eva%d/" % step
        masks = unpack_mask(g[sk + "masks"], (3, b, 512))
        losses, g_e, _ = O.eva_step(sde, H.clone_batch(eb), ae, CFG, masks, return_grads=True)
        for k, v in losses.items():
            np.testing.assert_allclose(float(v), float(g[sk + "loss/" + k]), rtol=2e-5, atol=1e-7, err_msg=k)
        for n, t in g_e.items():
            sib = H.rms_of_summary(g[sk + "grad_e/" + H.sibling_bn_bias(n)])
            H.assert_grad_close(n, t, g[sk + "grad_e/" + n], sib, 2e-3, 2e-4)
        for n, t in sde.items():
            H.assert_state_close(n, t, g[sk + "state/" + n], step)


# ------------------------------------------------------------------ A13-A14: inference
@pytest.mark.parametrize("variant", ["init", "wide"])
def test_inference_vs_reference(variant):
    g = H.load("inference_goldens.npz")
    bps = H.load("bps_real_objects.npz")["dists_red_mug"]
    key = variant + "/"
    sd = H.gan_sd(variant)
    sde = H.eva_sd(variant)
    z = torch.from_numpy(g[key + "z"])
    grasps = O.generate_grasps(sd, bps, z)
    for k in ["transl", "joint_conf", "z", "rot_matrix"]:
        np.testing.assert_allclose(grasps[k], g[key + "gen/" + k], rtol=1e-5, atol=1e-6, err_msg=k)
    p = O.evaluate_grasps(sde, bps, grasps)
    np.testing.assert_allclose(p, g[key + "scores"], rtol=1e-5, atol=1e-6)
    thresh = float(g[key + "thresh"])
    filt = O.filter_grasps(sde, bps, grasps, thresh)
    for k in ["transl", "joint_conf", "z", "rot_matrix"]:
        np.testing.assert_allclose(filt[k], g[key + "filt/" + k], rtol=1e-5, atol=1e-6, err_msg=k)
    assert tuple(filt["bps_object"].shape) == tuple(g[key + "filt/bps_object_shape"])
    with pytest.raises(ValueError):
        O.filter_grasps(sde, bps, grasps, 1.1)


def test_magd_vs_reference():
    g = H.load("magd_golden.npz")
    pred = {k: g["pred_" + k] for k in ["rot_matrix", "transl", "joint_conf"]}
    gt = {k: g["gt_" + k] for k in ["rot_matrix", "transl", "joint_conf"]}
    r = O.magd(gt, pred)
    np.testing.assert_allclose([r["transl"], r["rot"], r["joint"], r["coverage"]], g["result"], rtol=1e-5)


def test_magd_mirror_vs_reference():
    """dex_gan_grasp_amd.eval.metrics (mirror of eval.py:47-137) against the reference's own output."""
    from dex_gan_grasp_amd.eval import metrics
    g = H.load("magd_golden.npz")
    pred = {k: g["pred_" + k].copy() for k in ["rot_matrix", "transl", "joint_conf"]}
    gt = {k: g["gt_" + k].copy() for k in ["rot_matrix", "transl", "joint_conf"]}
    t, r, j, c = metrics.magd_for_grasp_distribution(pred, gt)
    np.testing.assert_allclose([t, r, j, c], g["result"], rtol=1e-6)
    t2, r2, j2, c2 = metrics.magd_for_grasp_distribution({k: torch.from_numpy(v) for k, v in pred.items()}, gt)
    assert (t2, c2) == (t, c)


@pytest.mark.parametrize("variant", ["init", "wide"])
def test_magd_on_eval_batch_grouped_by_object_vs_reference(variant):
    """BASELINE config 1 / SURVEY 8(f) N2 (eval.py:309-344): data/eval_batch.pth grouped by obj_name — the rows of an
    object are its ground-truth grasps —, the reference's generate_grasps on the object's BPS row with a seeded latent
    and the reference's magd_for_grasp_distribution (tests/golden/make_goldens.py: gen_magd_eval_batch_golden).  The
    oracle's generator reproduces the reference's grasps; the oracle's MAGD and the package's metrics mirror reproduce
    the reference's four numbers, on the reference's grasps and on the oracle's own."""
    from dex_gan_grasp_amd.eval import metrics
    g = H.load("magd_eval_batch.npz")
    sd = H.gan_sd(variant)
    nobj = len(g["objects"])
    assert nobj == 6 and g["rot_matrix"].shape == (64, 3, 3)
    for oi in range(nobj):
        rows = np.nonzero(g["obj_index"] == oi)[0]
        gt = {k: g[k][rows] for k in ("rot_matrix", "transl", "joint_conf")}
        key = "%s/%d/" % (variant, oi)
        ref_pred = {k: g[key + "pred/" + k] for k in ("rot_matrix", "transl", "joint_conf")}
        want = g[key + "magd"]
        mine = O.generate_grasps(sd, g["bps_rows"][oi], torch.from_numpy(g[key + "z"]))
        for k in ref_pred:
            np.testing.assert_allclose(mine[k], ref_pred[k], rtol=1e-5, atol=1e-6, err_msg=k)
        for pred in (ref_pred, mine):
            r = O.magd(gt, pred)
            np.testing.assert_allclose([r["transl"], r["rot"], r["joint"], r["coverage"]], want, rtol=1e-5)
            t, rr, j, c = metrics.magd_for_grasp_distribution({k: np.asarray(v).copy() for k, v in pred.items()},
                                                              {k: v.copy() for k, v in gt.items()})
            np.testing.assert_allclose([t, rr, j, c], want, rtol=1e-5)


# ------------------------------------------------------------------ N3: WGAN-GP
@pytest.mark.parametrize("variant,bname", [("wide", "real16"), ("wide", "syn64"), ("init", "syn64")])
def test_wgan_gp_steps_vs_reference(variant, bname):
    """dexgangrasp.py:149-211, 234-270 (cfg["is_wgan"]): two update_dexgangrasp steps of the reference, with its Z and
    its three interpolation weights injected — losses, every gradient before each optimiser step, post-step state."""
    g = H.load("wgan_goldens.npz")
    key = "%s/%s/" % (variant, bname)
    batch = H.batches()[bname]
    sd = H.gan_sd(variant)
    ag = O.AdamState([n for n in O.param_names(sd) if n.startswith("generator.")])
    ad = O.AdamState([n for n in O.param_names(sd) if n.startswith("discriminator.")])
    for step in range(2):
        sk = key + "wgan%d/" % step
        z = torch.from_numpy(g[sk + "z"])
        alphas = tuple(torch.from_numpy(g[sk + "alpha_" + n]) for n in ("rot", "transl", "joint"))
        losses, g_d, g_g = O.wgan_step(sd, H.clone_batch(batch), z, alphas, ag, ad, CFG, True, return_grads=True)
        for k in ("total_loss_disc", "bce_loss_real", "bce_loss_fake", "gen_loss_fake", "transl_loss", "rot_loss",
                  "conf_loss", "total_loss_gen"):
            np.testing.assert_allclose(float(losses[k]), float(g[sk + "loss/" + k]), rtol=2e-5, atol=1e-7, err_msg=k)
        # second step: starts from post-Adam parameters that two fp32 evaluations only share to ~1e-3 of lr where the
        # gradient is eps-sized (helpers.assert_state_close) — the absolute slack follows that
        atol = 2e-4 if step == 0 else 2e-3
        if step == 1 and variant == "init":
            # 0.02-gain weights put D's Wasserstein gradients (sigmoid' * tiny weights) at Adam's eps: its first update
            # is +-lr noise in both evaluations, and the second step's gradients inherit percent-level differences.
            # Losses and the bounded post-step state are still compared; gradients only where they are meaningful.
            for n, t in sd.items():
                H.assert_state_close(n, t, g[sk + "state/" + n], step)
            continue
        for n, t in g_d.items():
            sib = H.rms_of_summary(g[sk + "grad_d/" + H.sibling_bn_bias(n)])
            H.assert_grad_close(n, t, g[sk + "grad_d/" + n], sib, 2e-3, atol)
        for n, t in g_g.items():
            sib = H.rms_of_summary(g[sk + "grad_g/" + H.sibling_bn_bias(n)])
            H.assert_grad_close(n, t, g[sk + "grad_g/" + n], sib, 2e-3, atol)
        for n, t in sd.items():
            H.assert_state_close(n, t, g[sk + "state/" + n], step)
    # the discriminator runs FOUR train-mode forwards per step here (real, fake, interpolates, generator phase)
    assert int(sd["discriminator.disc_bn1.num_batches_tracked"]) == 8
    assert int(sd["generator.gen_bn1.num_batches_tracked"]) == 2


def test_forced_slopes_reproduce_and_steer_the_oracle():
    """oracle.forced_slopes (the device for strict gradient comparisons, helpers.assert_grad_tensor_close): with the
    sides the oracle itself took, outputs and gradients are unchanged bit for bit; with one side flipped, exactly that
    element's slope changes (its gradient contribution scales by 0.2 or 5)."""
    from oracle import dexgan_oracle as O
    import helpers as H
    sde = H.eva_sd("wide")
    batch = H.eva_batch("syn64")
    masks = synth.make_dropout_masks(64, 3)
    cfg = dict(synth.DEFAULT_CFG)

    def run(slopes=None, rec=False, capped=True):
        sd = {k: v.clone() for k, v in sde.items()}
        st = O.AdamState(O.param_names(sd))
        if rec:
            with O.record_slopes() as r:
                out, g, p = O.eva_step(sd, H.clone_batch(batch), st, cfg, masks, return_grads=True)
            return out, g, p, r.masks
        s = {"e": slopes} if slopes is not None else None
        if s is not None and not capped:
            s["max_flips"] = s["max_ratio"] = None
        out, g, p = O.eva_step(sd, H.clone_batch(batch), st, cfg, masks, return_grads=True, slopes=s)
        return out, g, p, (s or {}).get("flips")

    out0, g0, p0, sides = run(rec=True)
    assert len(sides) == 9 and sides[0].shape == (64, 512) and sides[1].shape == (64, 256)
    out1, g1, p1, flips = run([m.clone() for m in sides])
    assert flips == 0 and torch.equal(p0, p1)
    for n in g0:
        assert torch.equal(g0[n], g1[n]), n
    flipped = [m.clone() for m in sides]
    flipped[1][5, 7] = ~flipped[1][5, 7]          # rb1's bn1 branch, sample 5, unit 7
    out2, g2, p2, flips = run(flipped, capped=False)
    assert flips >= 1      # (a far-from-zero flip also moves downstream pre-activations across their forced sides)
    assert not torch.equal(g0["rb1.fc1.weight"][7], g2["rb1.fc1.weight"][7])
    # ... and the steering is capped: the same far-from-zero flip is REFUSED by default (|x| / rms <= 1e-5, <= 8 elements),
    # so a device that took a wrong side away from a kink fails the comparison instead of being followed
    with pytest.raises(AssertionError, match="forced_slopes: (steered at a pre-activation|the oracle was steered at)"):
        run(flipped)
    many = [m.clone() for m in sides]
    many[1][:, :] = ~many[1]
    with pytest.raises(AssertionError, match="forced_slopes: the oracle was steered at"):
        run(many)
    with pytest.raises(AssertionError):
        with O.forced_slopes(sides[:3]):
            O.evaluator_forward({k: v.clone() for k, v in sde.items()}, batch["bps_object"], batch["rot_matrix"],
                                batch["transl"], True, masks)
