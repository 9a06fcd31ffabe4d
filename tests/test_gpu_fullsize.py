"""GPU parity AT THE BENCHED SIZES (BASELINE configs 2-4): the path bench.py times — engine.joint_step with the
cross-step prefetch and in-kernel Philox dropout at B = 1024 — and the single-network steps at B = 256 / 1024, all
directly against the CPU oracle; in-kernel dropout statistics; ragged batches under pipelining; the multi-rank engine
and wrapper (2 processes, gloo rehearsal on one GPU); init distributions (A15); the stand-alone ResBlock (A2)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

import helpers as H
from dex_gan_grasp_amd import synth

pytestmark = pytest.mark.gpu
CFG = dict(synth.DEFAULT_CFG)
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOSS_NAMES = ["total_loss_disc", "bce_loss_real", "bce_loss_fake", "gen_loss_fake", "transl_loss", "rot_loss",
              "conf_loss", "total_loss_gen"]


def _mods():
    from dex_gan_grasp_amd import _lib, ops, engine
    return _lib, ops, engine


def _engine(variant="wide"):
    L, ops, engine = _mods()
    sd, sde = H.gan_sd(variant), H.eva_sd(variant)
    fg, fd, fe = (L.FlatNet(k, CFG, DEV) for k in (L.NET_GEN, L.NET_DISC, L.NET_EVA))
    fg.load_state_dict(sd, "generator.")
    fd.load_state_dict(sd, "discriminator.")
    fe.load_state_dict(sde)
    return engine.TrainEngine(CFG, fg, fd, fe, DEV), sd, sde


def _dev_inputs(batch, with_label=False):
    b = batch["bps_object"].shape[0]
    f = lambda t: t.float().to(DEV).contiguous()
    last = f(batch["label"]) if with_label else f(batch["joint_conf"])
    return (f(batch["bps_object"]), f(batch["rot_matrix"]).reshape(b, 9), f(batch["transl"]), last)


def _grad_items(flat, ref, prefix, batch):
    gv = {k: v.cpu() for k, v in flat.views("grads").items()}
    return [(n[len(prefix):], gv[n[len(prefix):]], r, True, batch) for n, r in ref.items()]


def _check_strict(items, key):
    """Every tensor within 2e-3 of its maximum, on every data set (helpers.assert_grad_tensor_close)."""
    for name, got, ref, train, batch in items:
        H.assert_grad_tensor_close(name, got, ref, train=train, batch=batch, what=key + " ")


def _sides_hook(store):
    """engine.after_d_phase: the LeakyReLU sides of D's real and fake pass, before the G phase's pass through the updated
    discriminator overwrites the fake pass' workspace."""
    def hook(e, B):
        store["Dreal"] = e.kink_sides("Dreal", B)
        store["Dfake"] = e.kink_sides("Dfake", B)
    return hook


def _slopes_after(eng, store, b=None, be=None):
    """What the oracle needs to take the HIP step's side at every LeakyReLU (oracle.forced_slopes), after the step:
    {"gan": {"d": G fwd + D(real) + D(fake), "g": the G phase's D pass}, "eva": {"e": ...}}."""
    out = {}
    if b is not None:
        out["gan"] = {"d": H.cpu_masks(eng.kink_sides("G", b) + store["Dreal"] + store["Dfake"]),
                      "g": H.cpu_masks(eng.kink_sides("Dfake", b))}
    if be is not None:
        out["eva"] = {"e": H.cpu_masks(eng.kink_sides("E", be))}
    return out


def _adopt(flat, sd, adam, prefix):
    """The oracle continues from the ENGINE's parameters and Adam moments (running statistics stay its own): both
    sides start the next step from identical state while the engine's prefetched pass stays valid."""
    pv, mv, vv = flat.views("params"), flat.views("m"), flat.views("v")
    for n in pv:
        sd[prefix + n].copy_(pv[n].cpu())
        adam.m[prefix + n] = mv[n].cpu().clone()
        adam.v[prefix + n] = vv[n].cpu().clone()


def _snap_flat(flat):
    return {"p": flat.params.clone(), "m": flat.m.clone(), "v": flat.v.clone(), "t": flat.step}


def _cmp_adam_of_own_grads(flat, before, lr=1e-4, b1=0.9, b2=0.999, eps=1e-8, wd=5e-4, what=""):
    """Exact post-step check where the step's own gradient buffer is still intact: the parameters must be
    torch.optim.Adam (L2-coupled, dexgangrasp.py:97-112) applied to the gradients THIS implementation computed —
    evaluated here in float64 from the pre-step snapshot.  Together with the gradient comparison against the oracle
    this pins the update without Adam's amplification of sub-eps gradient differences (helpers.assert_state_close)."""
    t = before["t"] + 1
    assert flat.step == t, what
    p0, m0, v0 = (before[k].double() for k in ("p", "m", "v"))
    g = flat.grads.double() + wd * p0
    m = m0 + (1 - b1) * (g - m0)
    v = b2 * v0 + (1 - b2) * g * g
    want = p0 - (lr / (1 - b1 ** t)) * m / (v.sqrt() / np.sqrt(1 - b2 ** t) + eps)
    err = (flat.params.double() - want).abs()
    # fp32 evaluation of the same formula: a few ulp of the parameter plus 1e-3 of the largest possible step
    tol = 4e-7 * want.abs() + 1e-3 * lr
    assert bool((err <= tol).all()), "%s: Adam(own grads) max err %.3e" % (what, float(err.max()))
    # first moment: fp32 lerp vs float64 (cancellation where m ~ 0: absolute slack relative to the tensor's scale)
    np.testing.assert_allclose(flat.m.double().cpu().numpy(), m.cpu().numpy(), rtol=1e-5,
                               atol=1e-6 * float(m.abs().max()))


def _cmp_update(flat, sd, prev, prefix, skip_running=False, l2_frac=2e-2):
    for n, t in flat.state_dict().items():
        r = sd[prefix + n]
        if "running_" in n:
            if skip_running:
                continue
            d = (t - r).abs()
            if "rb" in n and n.endswith("running_mean"):
                assert float(d.max()) <= 2.5e-4, n          # integrates the (noise-level) pre-BN bias
            else:
                tol = 1e-4 * r.abs() + 1e-5 * float(r.pow(2).mean().sqrt()) + 1e-7
                assert bool((d <= tol).all()), "%s: max err %.3e" % (n, float(d.max()))
            continue
        _update_close(n, t, r, prev[prefix + n], l2_frac)


def _update_close(name, t, r, prev, l2_frac, lr=1e-4):
    """Post-step parameters vs the oracle's.  Adam's first steps are sign-like (update ~ lr * g / |g|): an element
    whose gradient lies within the fp32 summation noise of zero moves by +-lr on EITHER side of the comparison, and at
    B = 1024 every wide tensor has a few of those (its gradients are sums of 1024 cancelling terms) — one such
    element of a 256-wide bias is already 12 % of that tensor's update in L2.  So: every element within the hard
    bound 2.5 lr (Adam cannot move further); at most 2 + 1 % of the elements off by more than 0.05 lr; and the L2
    error of the REST within l2_frac of the update norm.  The exact statements are the gradient comparison and
    _cmp_adam_of_own_grads."""
    d = (t - r).abs().double()
    assert float(d.max()) <= 2.5 * lr, "%s: max err %.3e beyond the hard bound" % (name, float(d.max()))
    if H.is_pre_bn_bias(name):
        return
    off = d > 0.05 * lr
    assert int(off.sum()) <= 2 + 0.01 * d.numel(), "%s: %d of %d elements off by > 0.05 lr" % (name, int(off.sum()), d.numel())
    upd = float((r - prev).double().pow(2).sum().sqrt())
    err = float(d[~off].pow(2).sum().sqrt())
    assert err <= l2_frac * upd + 1e-12, "%s: update L2 error %.3e vs update norm %.3e" % (name, err, upd)


def _config4_batches(b, n_points):
    """BASELINE config 4's per-rank share: synthetic clouds -> dgg_bps_encode on the device -> batch dicts
    (data/synthetic_data_set.py); for the oracle the SAME clouds are encoded by the CPU brute force (oracle/bps_oracle.c),
    which must reproduce the device block bit for bit."""
    from oracle import bps_oracle
    from dex_gan_grasp_amd.data.synthetic_data_set import SyntheticGraspBatches
    basis = H.load("bps_real_objects.npz")["basis"]
    ds = SyntheticGraspBatches(CFG, b, 5, basis, n_points=n_points, seed=21, with_label=True)
    out = []
    for i in range(5):
        dev_b = ds.batch(i)
        clouds = ds.clouds(i).cpu().numpy()
        enc = np.stack([bps_oracle.encode_f32(clouds[j], basis)[0] for j in range(b)])
        assert np.array_equal(dev_b["bps_object"].cpu().numpy(), enc), "batch %d: device encode != oracle encode" % i
        cpu_b = {k: v for k, v in dev_b.items() if k != "obj_name"}
        cpu_b["bps_object"] = torch.from_numpy(enc)
        out.append(cpu_b)
    return out[:2], out[2:]


@pytest.mark.parametrize("b,variant,front", [(1024, "wide", "direct"), (256, "wide", "direct"), (1024, "init", "direct"),
                                             (512, "wide", "config4")])
def test_benched_joint_step_with_prefetch_and_philox_vs_oracle(b, variant, front):
    """What bench.py times, at its sizes: 3 x joint_step(next_gan_inputs=...) with the evaluator's dropout masks drawn
    in-kernel.  Per step: the 10 losses (1e-4), the gradients of G and E (D's are overwritten by the prefetched pass of
    the next step: compared on the last step) strictly, the masks read back from the workspace and fed to the oracle,
    Adam of the engine's own gradients exactly, every parameter after Adam.  BatchNorm running statistics at the end.
    (1024, "init"): the reference-style 0.02-gain weights bench.py runs (the other cases use the "wide" variant, whose
    gradients sit above Adam's eps: with "init" weights whole tensors have |g| ~ 1e-9, where Adam's update
    lr g / (|g| + eps) amplifies rounding differences — helpers.assert_state_close — so the element-wise parameter
    comparison after the step is left to the exact Adam-of-own-gradients statement).
    (512, "config4"): BASELINE config 4's per-rank share, 512 clouds x 2048 points through dgg_bps_encode in front."""
    from oracle import dexgan_oracle as O
    L, ops, engine = _mods()
    eng, sd, sde = _engine(variant)
    ag = O.AdamState([n for n in O.param_names(sd) if n.startswith("generator.")])
    ad = O.AdamState([n for n in O.param_names(sd) if n.startswith("discriminator.")])
    ae = O.AdamState(O.param_names(sde))
    if front == "config4":
        gbs, ebs = _config4_batches(b, 2048)
    else:
        gbs = [synth.make_batch(b, 1), synth.make_batch(b, 11)]
        # a fresh evaluator batch per step, as in training (seed 2 has a pre-activation on a LeakyReLU kink in rb1: the
        # forced-slope comparison covers it)
        ebs = [synth.make_batch(b, 2 + 10 * i, with_label=True) for i in range(3)]
    g_ins = [_dev_inputs(x) for x in gbs]
    steps = 3
    store, flips, worst = {}, 0, 0.0
    eng.after_d_phase = _sides_hook(store)
    for i in range(steps):
        eb = ebs[i]
        e_in = _dev_inputs(eb, with_label=True)
        z = synth.make_z(b, 100 + i)
        nxt = g_ins[(i + 1) % 2] if i + 1 < steps else None
        before = {k: _snap_flat(fl) for k, fl in (("G", eng.G), ("D", eng.D), ("E", eng.E))}
        got = eng.joint_step(g_ins[i % 2], z.to(DEV), e_in, True, None, 4242 + i, next_gan_inputs=nxt)
        torch.cuda.synchronize()
        got = got.cpu().numpy()
        keep = eng.eva_keep_masks(b)                       # (3, b, 512) uint8, what the backward used
        assert keep.dtype == torch.uint8 and int(keep.max()) == 1 and int(keep.min()) == 0
        masks = keep.float().cpu()
        prev, preve = {k: v.clone() for k, v in sd.items()}, {k: v.clone() for k, v in sde.items()}
        sl = _slopes_after(eng, store, b, b)
        # the G phase runs through the discriminator AFTER its Adam step: the oracle takes the engine's post-Adam D there
        # (Adam amplifies 1e-9 gradient noise to +-lr per element — with the reference-init weights, |w| ~ 4e-4, that moved
        # pre-activations by up to 8e-5 of their rms and 12 LeakyReLU sides per G phase); the Adam step itself is checked
        # exactly below (_cmp_adam_of_own_grads) and the oracle adopts the engine's parameters at the end of the step anyway
        d_post = {k: v.cpu().clone() for k, v in eng.D.views("params").items()}

        def take_engine_d(sd_):
            # ... but only a D that IS an Adam step away from the oracle's own post-Adam D is adopted: every element
            # within Adam's hard reach of it (2.5 lr; lr = 1e-4), and — where the gradients sit above Adam's eps ("wide") —
            # all but 1 % + 2 elements of a tensor within 0.05 lr
            with torch.no_grad():
                for k, v in d_post.items():
                    own = sd_["discriminator." + k]
                    diff = (v - own).abs()
                    assert float(diff.max()) <= 2.5e-4, ("post-Adam D beyond Adam's reach of the oracle's", k, float(diff.max()))
                    if variant != "init":
                        assert int((diff > 5e-6).sum()) <= 0.01 * diff.numel() + 2, ("post-Adam D", k, int((diff > 5e-6).sum()))
                    own.copy_(v)
        ref, g_d, g_g, _ = O.cgan_step(sd, H.clone_batch(gbs[i % 2]), z, ag, ad, CFG, True, return_grads=True,
                                       slopes=sl["gan"], after_adam_d=take_engine_d)
        refe, g_e, _ = O.eva_step(sde, H.clone_batch(eb), ae, CFG, masks, return_grads=True, slopes=sl["eva"])
        # the steering is capped inside the oracle (forced_slopes: per phase at most max(8, 4 per million pre-activations)
        # elements, each within 1e-5 of its tensor's rms of zero); per STEP, stated here as well:
        step_flips = sl["gan"]["flips"] + sl["eva"]["flips"]
        step_worst = max(sl["gan"]["worst"], sl["eva"]["worst"])
        step_cap = sl["gan"]["cap"] + sl["eva"]["cap"]
        assert step_flips <= step_cap and step_worst <= O.MAX_FLIP_RATIO, (i, step_flips, step_cap, step_worst)
        flips, worst = flips + step_flips, max(worst, step_worst)
        for k, n in enumerate(LOSS_NAMES):
            np.testing.assert_allclose(got[k], float(ref[n]), rtol=1e-4, atol=1e-6, err_msg="step %d %s" % (i, n))
        np.testing.assert_allclose(got[L.L_TOTAL_EVA], float(refe["total_loss_eva"]), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(got[L.L_BCE_EVA], float(refe["bce_loss"]), rtol=1e-4, atol=1e-6)
        _check_strict(_grad_items(eng.G, g_g, "generator.", b), "G/%d" % i)
        _check_strict(_grad_items(eng.E, g_e, "", b), "E/%d" % i)
        if nxt is None:
            _check_strict(_grad_items(eng.D, g_d, "discriminator.", b), "D/%d" % i)
        _cmp_adam_of_own_grads(eng.G, before["G"], what="G step %d" % i)
        _cmp_adam_of_own_grads(eng.E, before["E"], what="E step %d" % i)
        if nxt is None:
            _cmp_adam_of_own_grads(eng.D, before["D"], what="D step %d" % i)
        if variant != "init":
            _cmp_update(eng.G, sd, prev, "generator.", skip_running=True, l2_frac=0.1)
            _cmp_update(eng.D, sd, prev, "discriminator.", skip_running=True, l2_frac=0.1)
            _cmp_update(eng.E, sde, preve, "", skip_running=True, l2_frac=0.1)
        else:      # every element within Adam's reach all the same
            for fl, s_, pre in ((eng.G, sd, "generator."), (eng.D, sd, "discriminator."), (eng.E, sde, "")):
                for n, t in fl.state_dict().items():
                    if "running_" not in n:
                        assert float((t - s_[pre + n]).abs().max()) <= 2.5e-4, n
        _adopt(eng.G, sd, ag, "generator.")
        _adopt(eng.D, sd, ad, "discriminator.")
        _adopt(eng.E, sde, ae, "")
    print("B=%d %s %s: the oracle took the HIP path's side at %d LeakyReLU kink crossing(s) over %d steps, the farthest "
          "%.2e of its tensor's rms from zero (caps: max(%d, %d per million pre-activations) per phase, %.0e)"
          % (b, variant, front, flips, steps, worst, O.MAX_FLIPS, O.MAX_FLIPS_PER_MILLION, O.MAX_FLIP_RATIO))
    # running statistics integrate three steps of batch statistics (real, fake, fake for D)
    for flat, s, pre in ((eng.G, sd, "generator."), (eng.D, sd, "discriminator."), (eng.E, sde, "")):
        for n, t in flat.state_dict().items():
            if "running_var" in n or (n.endswith("running_mean") and "rb" not in n):
                r = s[pre + n]
                tol = 1e-4 * r.abs() + 1e-5 * float(r.pow(2).mean().sqrt()) + 1e-7
                assert bool(((t - r).abs() <= tol).all()), n
    eng.close()


def test_joint_step_at_1024_vs_unsteered_oracle():
    """the benched step (B = 1024, joint_step, "wide" weights) against a FULLY INDEPENDENT oracle — no
    forced_slopes, no adopted post-Adam discriminator, dropout masks injected into both.  Whether an unsteered comparison
    can be strict is a property of the data: wherever a pre-activation lies within rounding distance of a LeakyReLU kink
    the two evaluations may take different sides.  So a handful of data seeds is searched for one on which
    the oracle's OWN sides (record_slopes) equal the sides the HIP path stored, at every one of the 33 LeakyReLU sites of
    the step (G 6, D real 6, D fake 6, the G phase's pass through the updated D 6, E 9) — then nothing is steered and
    everything is compared as in the steered tests: losses 1e-4, every gradient tensor of G, D and E strictly, Adam of the
    engine's own gradients exactly, the parameters after the step.  Shows that the steering of the other tests is not
    load-bearing: on data without kink crossings the independent oracle agrees by itself."""
    from oracle import dexgan_oracle as O
    L, ops, engine = _mods()
    b = 1024
    tried = []
    for seed in range(40, 48):
        eng, sd, sde = _engine("wide")
        ag = O.AdamState([n for n in O.param_names(sd) if n.startswith("generator.")])
        ad = O.AdamState([n for n in O.param_names(sd) if n.startswith("discriminator.")])
        ae = O.AdamState(O.param_names(sde))
        gb, eb = synth.make_batch(b, seed), synth.make_batch(b, seed + 100, with_label=True)
        z, masks = synth.make_z(b, seed + 200), synth.make_dropout_masks(b, seed + 300)
        store = {}
        eng.after_d_phase = _sides_hook(store)
        before = {k: _snap_flat(fl) for k, fl in (("G", eng.G), ("D", eng.D), ("E", eng.E))}
        got = eng.joint_step(_dev_inputs(gb), z.to(DEV), _dev_inputs(eb, with_label=True), True, masks.to(DEV), 0)
        torch.cuda.synchronize()
        got = got.cpu().numpy()
        sl = _slopes_after(eng, store, b, b)
        hip_sides = sl["gan"]["d"] + sl["gan"]["g"]
        prev, preve = {k: v.clone() for k, v in sd.items()}, {k: v.clone() for k, v in sde.items()}
        with O.record_slopes() as rec_g:     # the oracle on its own: plain F.leaky_relu, its sides only written down
            ref, g_d, g_g, _ = O.cgan_step(sd, H.clone_batch(gb), z, ag, ad, CFG, True, return_grads=True)
        with O.record_slopes() as rec_e:
            refe, g_e, _ = O.eva_step(sde, H.clone_batch(eb), ae, CFG, masks, return_grads=True)
        assert len(rec_g.masks) == len(hip_sides) == 24 and len(rec_e.masks) == len(sl["eva"]["e"]) == 9
        diff = sum(int((torch.as_tensor(h).bool() != o).sum()) for h, o in zip(hip_sides, rec_g.masks))
        for h, o in zip(sl["eva"]["e"], rec_e.masks):   # int8 with -1 = dropped unit: no side on the device
            h = torch.as_tensor(h)
            diff += int(((h > 0) != o)[h >= 0].sum()) if h.dtype != torch.bool else int((h != o).sum())
        tried.append((seed, diff))
        if diff:
            eng.close()
            continue
        for k, n in enumerate(LOSS_NAMES):
            np.testing.assert_allclose(got[k], float(ref[n]), rtol=1e-4, atol=1e-6, err_msg=n)
        np.testing.assert_allclose(got[L.L_TOTAL_EVA], float(refe["total_loss_eva"]), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(got[L.L_BCE_EVA], float(refe["bce_loss"]), rtol=1e-4, atol=1e-6)
        _check_strict(_grad_items(eng.G, g_g, "generator.", b), "G unsteered")
        _check_strict(_grad_items(eng.D, g_d, "discriminator.", b), "D unsteered")
        _check_strict(_grad_items(eng.E, g_e, "", b), "E unsteered")
        for key, fl in (("G", eng.G), ("D", eng.D), ("E", eng.E)):
            _cmp_adam_of_own_grads(fl, before[key], what=key + " unsteered")
        _cmp_update(eng.G, sd, prev, "generator.", skip_running=True, l2_frac=0.1)
        _cmp_update(eng.D, sd, prev, "discriminator.", skip_running=True, l2_frac=0.1)
        _cmp_update(eng.E, sde, preve, "", skip_running=True, l2_frac=0.1)
        eng.close()
        print("unsteered oracle: data seed %d has no LeakyReLU side in dispute (seeds tried, sides in dispute: %s)" % (seed, tried))
        return
    raise AssertionError("no data seed without a kink crossing among %s" % (tried,))


def test_evaluator_step_at_1024_vs_oracle():
    """BASELINE config 3: the evaluator update alone at B = 1024 (injected masks), two steps, through the API-path
    entry (eva_step with side streams)."""
    from oracle import dexgan_oracle as O
    L, ops, engine = _mods()
    b = 1024
    eng, _, sde = _engine("wide")
    ae = O.AdamState(O.param_names(sde))
    eb = synth.make_batch(b, 2, with_label=True)
    e_in = _dev_inputs(eb, with_label=True)
    for i in range(2):
        masks = synth.make_dropout_masks(b, 50 + i)
        preve = {k: v.clone() for k, v in sde.items()}
        before = _snap_flat(eng.E)
        got = eng.eva_step(*e_in, masks.to(DEV), 0, two_streams=True).cpu().numpy()
        refe, g_e, _ = O.eva_step(sde, H.clone_batch(eb), ae, CFG, masks, return_grads=True,
                                  slopes=_slopes_after(eng, {}, None, b)["eva"])
        np.testing.assert_allclose(got[L.L_TOTAL_EVA], float(refe["total_loss_eva"]), rtol=1e-4, atol=1e-6)
        assert torch.equal(eng.eva_keep_masks(b).cpu(), masks.to(torch.uint8))   # given masks land in the workspace
        _check_strict(_grad_items(eng.E, g_e, "", b), "E/%d" % i)
        _cmp_adam_of_own_grads(eng.E, before, what="E step %d" % i)
        _cmp_update(eng.E, sde, preve, "", l2_frac=0.1)
        _adopt(eng.E, sde, ae, "")
    eng.close()


def test_cgan_step_at_256_vs_oracle():
    """BASELINE config 2: the cGAN update alone at B = 256, two steps, through the API-path entry (cgan_step with the
    D-phase fork and the weight-gradient side streams)."""
    _cgan_steps_vs_oracle(256, 2)


def test_cgan_step_ragged_batch_vs_oracle(monkeypatch):
    """B = 100: two workgroups of the generator's loss kernel, the second with 36 of its 64 rows — the form that adds its
    partial sums under a ticket and forms the heads' bias gradient (gen_loss_kernel, GenLossFin); two steps, so the ticket
    is drawn twice.  The workspace starts as 0xFF bytes (DGG_POISON_WS): the ticket word is whatever the caller's memory
    held until the G phase's first launch clears it, and no kernel may read workspace it has not written."""
    monkeypatch.setenv("DGG_POISON_WS", "1")
    _cgan_steps_vs_oracle(100, 2)


def test_cgan_step_above_the_xshare_limit_vs_oracle():
    """B = 1056: one batch size above what the shared BPS rows of round 6 serve (XShare: B <= 1024, 12 hand joints) — every
    pass normalises its own input again (ibn_stats / ibn_apply / fake_pose, the path of rounds 1-5), the weight gradients
    still go out as one grouped launch per pass.  One step through the API-path entry against the oracle."""
    _cgan_steps_vs_oracle(1056, 1)


def _cgan_steps_vs_oracle(b, nsteps):
    from oracle import dexgan_oracle as O
    L, ops, engine = _mods()
    eng, sd, _ = _engine("wide")
    ag = O.AdamState([n for n in O.param_names(sd) if n.startswith("generator.")])
    ad = O.AdamState([n for n in O.param_names(sd) if n.startswith("discriminator.")])
    gb = synth.make_batch(b, 1)
    g_in = _dev_inputs(gb)
    store = {}
    eng.after_d_phase = _sides_hook(store)
    for i in range(nsteps):
        z = synth.make_z(b, 100 + i)
        prev = {k: v.clone() for k, v in sd.items()}
        before = {k: _snap_flat(fl) for k, fl in (("G", eng.G), ("D", eng.D))}
        got = eng.cgan_step(*g_in, z.to(DEV), True, two_streams=True).cpu().numpy()
        ref, g_d, g_g, _ = O.cgan_step(sd, H.clone_batch(gb), z, ag, ad, CFG, True, return_grads=True,
                                       slopes=_slopes_after(eng, store, b)["gan"])
        for k, n in enumerate(LOSS_NAMES):
            np.testing.assert_allclose(got[k], float(ref[n]), rtol=1e-4, atol=1e-6, err_msg=n)
        _check_strict(_grad_items(eng.G, g_g, "generator.", b), "G/%d" % i)
        _check_strict(_grad_items(eng.D, g_d, "discriminator.", b), "D/%d" % i)
        _cmp_adam_of_own_grads(eng.G, before["G"], what="G step %d" % i)
        _cmp_adam_of_own_grads(eng.D, before["D"], what="D step %d" % i)
        _cmp_update(eng.G, sd, prev, "generator.", l2_frac=0.05)
        _cmp_update(eng.D, sd, prev, "discriminator.", l2_frac=0.05)
        _adopt(eng.G, sd, ag, "generator.")
        _adopt(eng.D, sd, ad, "discriminator.")
    eng.close()


def test_in_kernel_dropout_masks():
    """networks.py:294 nn.Dropout(0.3) drawn in-kernel (Philox4x32-10 keyed by seed and element): keep rate 0.7
    within 4 sigma per site, sites independent, masks change with the seed, and the backward uses the very mask the
    forward drew (the gradient equals the oracle's under the read-back masks: previous tests; here: a dropped unit's
    column of d(rb output) is exactly zero, visible in fc-weight gradients only through the mask)."""
    L, ops, engine = _mods()
    b = 1024
    eng, _, _ = _engine("wide")
    eb = synth.make_batch(b, 2, with_label=True)
    e_in = _dev_inputs(eb, with_label=True)
    seen = []
    for seed in (7, 7, 8):
        eng2, _, _ = _engine("wide")
        eng2.eva_step(*e_in, None, seed)
        seen.append((eng2.eva_keep_masks(b).cpu(), eng2.E.grads.clone().cpu()))
        eng2.close()
    k7, k7b, k8 = seen[0][0], seen[1][0], seen[2][0]
    assert torch.equal(k7, k7b) and torch.equal(seen[0][1], seen[1][1])          # same seed: same masks, same grads
    assert not torch.equal(k7, k8)
    n = b * 512
    sigma = np.sqrt(0.3 * 0.7 / n)
    for r in range(3):
        for k in (k7, k8):
            assert abs(float(k[r].float().mean()) - 0.7) <= 4 * sigma, (r, float(k[r].float().mean()))
    # sites / seeds are independent draws: agreement rate of two masks = 0.7^2 + 0.3^2 = 0.58
    for a, c in ((k7[0], k7[1]), (k7[1], k7[2]), (k7[0], k8[0])):
        agree = float((a == c).float().mean())
        assert abs(agree - 0.58) <= 6 * np.sqrt(0.58 * 0.42 / n), agree
    # per-row and per-column keep rates are unstructured
    assert float(k7[0].float().mean(0).std()) < 3 * np.sqrt(0.21 / b) and float(k7[0].float().mean(1).std()) < 3 * np.sqrt(0.21 / 512)
    eng.close()


def test_ragged_last_batch_under_pipelining():
    """seq = [B, B, B-17] with next_gan_inputs passed every time.  The ragged batch is not prefetched
    (the workspace is carved per batch size); results equal the unpipelined engine bit for bit, and the C entry
    refuses a mismatched prefetch outright."""
    import ctypes as C
    L, ops, engine = _mods()
    b = 96
    gb_a, gb_b, gb_c = synth.make_batch(b, 1), synth.make_batch(b, 11), synth.make_batch(b - 17, 12)
    seq = [_dev_inputs(gb_a), _dev_inputs(gb_b), _dev_inputs(gb_c)]
    eb = synth.make_batch(b, 2, with_label=True)
    e_in = _dev_inputs(eb, with_label=True)
    e1, _, _ = _engine("wide")
    e2, _, _ = _engine("wide")
    for i in range(3):
        z = synth.make_z(seq[i][0].shape[0], 100 + i).to(DEV)
        nxt = seq[i + 1] if i < 2 else None
        l1 = e1.joint_step(seq[i], z, e_in, True, None, 9 + i, next_gan_inputs=nxt)
        l2 = e2.joint_step(seq[i], z, e_in, True, None, 9 + i)
        torch.cuda.synchronize()
        assert torch.equal(l1[:10], l2[:10]), i
    for a, c in ((e1.G, e2.G), (e1.D, e2.D), (e1.E, e2.E)):
        assert torch.equal(a.params, c.params) and torch.equal(a.m, c.m) and torch.equal(a.v, c.v)
    assert torch.equal(e1.G.bn, e2.G.bn) and torch.equal(e1.D.bn, e2.D.bn)
    # the C-ABI itself: a prefetch for another batch size than the step in flight is rejected, nothing enqueued
    ws = e1._cgan_ws(b - 17)
    st, hp = e1._state(), e1._hp()
    big = seq[0]
    s2 = C.c_void_p(e1._side2.cuda_stream)
    rc = L.lib().dgg_cgan_d_real_prefetch(C.byref(e1.ccfg), C.byref(st), C.byref(hp), L.ptr(ws), ws.numel(), b,
                                          L.ptr(big[0]), L.ptr(big[1]), L.ptr(big[2]), L.ptr(big[3]),
                                          L.ptr(e1.losses), s2)
    assert rc in (-1, -3), rc        # DGG_ERR_INVALID (mismatch) or DGG_ERR_WORKSPACE (too small), never 0
    # a serial entry point after a pipelined step joins the prefetched pass first (engine.flush)
    e3, _, _ = _engine("wide")
    e4, _, _ = _engine("wide")
    z = synth.make_z(b, 5).to(DEV)
    e3.joint_step(seq[0], z, e_in, True, None, 1, next_gan_inputs=seq[1])
    e4.joint_step(seq[0], z, e_in, True, None, 1)
    e3.cgan_step(*seq[1], z, True)      # the prefetched pass is dropped and redone on the main stream
    e4.cgan_step(*seq[1], z, True)
    torch.cuda.synchronize()
    assert torch.equal(e3.G.params, e4.G.params) and torch.equal(e3.D.params, e4.D.params)
    assert torch.equal(e3.D.bn, e4.D.bn) and torch.equal(e3.G.bn, e4.G.bn)     # the dropped pass left no trace
    # ... nor does a state read between two pipelined steps: flush() drops the pass that is in flight
    e5, _, _ = _engine("wide")
    e6, _, _ = _engine("wide")
    e5.joint_step(seq[0], z, e_in, True, None, 1, next_gan_inputs=seq[1])
    e6.joint_step(seq[0], z, e_in, True, None, 1)
    e5.flush()
    torch.cuda.synchronize()
    assert torch.equal(e5.D.bn, e6.D.bn) and torch.equal(e5.D.params, e6.D.params)
    e5.joint_step(seq[1], z, e_in, True, None, 2, next_gan_inputs=seq[0])     # (the dropped pass is simply redone)
    e6.joint_step(seq[1], z, e_in, True, None, 2)
    e5.flush()
    torch.cuda.synchronize()
    assert torch.equal(e5.D.bn, e6.D.bn) and torch.equal(e5.D.params, e6.D.params) and torch.equal(e5.G.params, e6.G.params)
    for e in (e1, e2, e3, e4, e5, e6):
        e.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _torchrun(nproc, script_args, extra_env=None, timeout=900):
    env = dict(os.environ)
    env.update({"DGG_REHEARSE": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0", "OMP_NUM_THREADS": "3"})
    env.update(extra_env or {})
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port())] + script_args
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    return r.returncode, r.stdout.decode(errors="replace")


@pytest.mark.parametrize("mode,world,env", [("engine", 2, None), ("wrapper", 2, None), ("sweep", 2, None),
                                            ("engine", 2, {"DGG_EVA_GROUP": "1"}),
                                            ("engine", 3, {"DGG_DP_PER_RANK": "32"})])
def test_multi_rank_data_parallel_on_one_gpu(mode, world, env):
    """BASELINE config 4's code path (engine world > 1): fresh processes, all on cuda:0, collectives over gloo
    (DGG_REHEARSE) — tools/dp_check.py compares every rank with the data-parallel oracle and the ranks with each
    other.  The children initialise the GPU themselves, after the launcher started them.  World 3: what a GPU box's
    process guard leaves room for beside this process and the launcher (6 processes on the card at once; 5 ranks were
    killed by it) — sharding, bucket bounds and the collective issue order at a world size that is neither 2 nor a
    power of two; DGG_EVA_GROUP=1: the evaluator's buckets on a
    process group of their own (parallel.evaluator_group)."""
    rc, out = _torchrun(world, [os.path.join(ROOT, "tools", "dp_check.py"), mode], env)
    assert rc == 0, out[-4000:]
    assert out.count("dp_check %s: rank" % mode) == world, out[-2000:]


def test_bench_py_runs_its_multi_rank_branches_in_rehearsal():
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one process per rank, barrier + MAX over
    ranks), two ranks on this one GPU over gloo: the world > 1 branches run before a real multi-GPU node is the first
    to execute them.  One JSON line from rank 0 with the contract's fields; the numbers of a rehearsal mean nothing."""
    import json
    rc, out = _torchrun(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--repeats",
                            "2", "--batch", "64", "--no-sub", "--no-cpu-baseline"])
    assert rc == 0, out[-4000:]
    lines = [ln for ln in out.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, out[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 128
    assert d["rccl_ranks"]["world_size"] == 2 and d["rccl_ranks"]["backend"] == "gloo"
    assert d["rccl_ranks"]["allreduce_bytes_per_step_per_rank"] == 4 * (6979096 + 6998068 + 10661660) or \
        d["rccl_ranks"]["allreduce_bytes_per_step_per_rank"] > 9e7
    assert d["value"] > 0 and "roofline" in d and d["side_streams"]["mode"] in ("probed", "plain")
    # round 4: what makes the first N > 1 run falsifiable — per-bucket all-reduce durations (with their bytes, in issue
    # order) and the time each network's optimiser step waited for its all-reduce after its backward had finished
    rr = d["rccl_ranks"]
    assert set(rr["exposed_us"]) == {"D", "G", "E"} and all(v >= 0 for v in rr["exposed_us"].values()), rr
    assert set(rr["bucket_allreduce_us"]) == {"D", "G", "E"}, rr
    assert [len(rr["bucket_allreduce_us"][k]) for k in "DGE"] == [2, 2, 3]
    for k in "DGE":
        assert all(b["bytes"] > 1e7 and b["us"] > 0 for b in rr["bucket_allreduce_us"][k]), rr["bucket_allreduce_us"][k]
    assert sum(b["bytes"] for k in "DGE" for b in rr["bucket_allreduce_us"][k]) == rr["allreduce_bytes_per_step_per_rank"]


def test_one_rank_rccl_process_group_runs_the_collective_path():
    """The data-parallel step over the REAL backend on this box's one GPU: a one-rank RCCL process group
    (DGG_FORCE_DISTRIBUTED=1) sends every bucket through dist.all_reduce on RCCL's stream, behind the gradient-ready
    events and the lent feeder streams, in front of the optimiser launches — 3 joint_steps with prefetch against the
    oracle (tools/dp_check.py).  What the gloo rehearsal above cannot show: stream-ordered NCCL work handles."""
    env = dict(os.environ)
    env.update({"DGG_FORCE_DISTRIBUTED": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0", "OMP_NUM_THREADS": "4", "RANK": "0",
                "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port())})
    env.pop("DGG_REHEARSE", None)
    cmd = [sys.executable, os.path.join(ROOT, "tools", "dp_check.py"), "engine"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, out[-4000:]
    assert out.count("dp_check engine: rank 0 ok") == 1, out[-2000:]


def test_side_streams_sit_on_hardware_queues_of_their_own():
    """parallel.pick_side_streams: the runtime binds streams to its 4 hardware queues round robin at creation and a queue
    is strictly in order, so where a plain torch.cuda.Stream() lands depends on the process's history (the step ran 2.05 -
    2.14 instead of 1.74 ms after 2 - 3 earlier streams).  The picked pair must not wait behind a long spin on the current
    stream nor on each other, even when the process has already created a few streams."""
    from dex_gan_grasp_amd import parallel
    dev = torch.device(DEV)
    junk = []
    for _ in range(3):
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            junk.append(torch.zeros(8, device=dev))
        junk.append(st)
    a, b = parallel.pick_side_streams(dev, 2)
    spin = 3_000_000

    def on(st):
        def run(c):
            with torch.cuda.stream(st):
                torch.cuda._sleep(c)
        return run
    for cand, blockers in ((a, (lambda c: torch.cuda._sleep(c), on(b))), (b, (lambda c: torch.cuda._sleep(c), on(a)))):
        parallel._blocked_behind(cand, lambda c: None, dev, 0)
        for blk in blockers:
            ms = min(parallel._blocked_behind(cand, blk, dev, spin) for _ in range(2))
            assert ms < 0.5, ms


def test_init_net_distribution_lands_in_flat_buffers():
    """A15 (dexgangrasp.py:28-62): xavier_normal(gain) on every Linear.weight, BatchNorm weight ~ N(1, gain), BatchNorm
    bias 0, Linear.bias left at torch's default; and the values ARE the library's flat buffers (the kernels see them)."""
    from dex_gan_grasp_amd.models.dexgangrasp import DexGanGrasp
    L, _, _ = _mods()
    cfg = dict(CFG)
    cfg["train_ffhevaluator"] = True
    torch.manual_seed(77)
    m = DexGanGrasp(cfg)
    gain = cfg["init_gain"]
    for mod in (m.DexGANGrasp.generator, m.DexGANGrasp.discriminator, m.DexEvaluator):
        flat = mod._require_flat()
        views = flat.views("params")
        named = dict(mod.named_parameters())
        assert set(views) == set(named)
        bn_w = []
        for name, p in named.items():
            assert p.data_ptr() == views[name].data_ptr(), name      # Parameter storage = flat buffer
            t = p.detach().float().cpu()
            if name.endswith("weight") and t.dim() == 2:
                fan_out, fan_in = t.shape
                want = gain * np.sqrt(2.0 / (fan_in + fan_out))
                if t.numel() >= 5000:                                # 2 % needs ~5k samples (sigma of std ~ 1/sqrt(2n))
                    assert abs(float(t.std()) / want - 1.0) < 0.02, (name, float(t.std()), want)
                else:
                    assert abs(float(t.std()) / want - 1.0) < 4.0 / np.sqrt(2.0 * t.numel()) + 0.02, name
                assert abs(float(t.mean())) < 5 * want / np.sqrt(t.numel())
            elif name.endswith("weight"):                            # BatchNorm gamma ~ N(1, gain)
                bn_w.append(t)
            elif "bn" in name and name.endswith("bias"):
                assert float(t.abs().max()) == 0.0, name
            else:                                                    # Linear.bias: U(-1/sqrt(fan_in), 1/sqrt(fan_in))
                fan_in = named[name.replace("bias", "weight")].shape[1]
                assert float(t.abs().max()) <= 1.0 / np.sqrt(fan_in) + 1e-7, name
        g = torch.cat(bn_w)
        assert abs(float(g.mean()) - 1.0) < 5 * gain / np.sqrt(g.numel()) and abs(float(g.std()) / gain - 1.0) < 0.05
    with pytest.raises(NotImplementedError):
        bad = dict(cfg)
        bad["weight_init_type"] = "nope"
        DexGanGrasp(bad)


@pytest.mark.parametrize("train", [True, False])
def test_standalone_resblock_matches_oracle(train):
    """A2 (networks.py:43-56): ResBlock called on its own, forward and autograd backward (dx and all 10 parameter
    tensors), train and eval mode, against the oracle's resblock under torch autograd."""
    from oracle import dexgan_oracle as O
    from dex_gan_grasp_amd.models.networks import ResBlock
    torch.manual_seed(3)
    fin, b = 333, 96
    rb = ResBlock(fin, 512).to(DEV)
    with torch.no_grad():   # non-trivial BatchNorm state
        for bn in (rb.bn1, rb.bn2):
            bn.weight.normal_(1.0, 0.2)
            bn.bias.normal_(0.0, 0.2)
            bn.running_mean.normal_(0.0, 0.3)
            bn.running_var.uniform_(0.5, 1.5)
    rb.train(train)
    sd = {k: v.detach().cpu().clone() for k, v in rb.state_dict().items()}
    x = torch.randn(b, fin)
    dy = torch.randn(b, 512)
    xg = x.to(DEV).requires_grad_(True)
    y = rb(xg)
    y.backward(dy.to(DEV))
    names = [n for n in sd if sd[n].dtype.is_floating_point and "running" not in n]
    for n in names:
        sd[n].requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    yr = O.resblock(sd, "", xr, train)
    yr.backward(dy)
    np.testing.assert_allclose(y.detach().cpu().numpy(), yr.detach().numpy(), rtol=1e-4, atol=1e-4)
    H.assert_grad_tensor_close("dx", xg.grad, xr.grad, train=False, batch=b)
    got = dict(rb.named_parameters())
    for n in names:
        H.assert_grad_tensor_close(n, got[n].grad, sd[n].grad, train=train, batch=b)
    if train:
        assert int(rb.bn1.num_batches_tracked) == 1
        for n in ("bn1.running_var", "bn2.running_var"):
            np.testing.assert_allclose(rb.state_dict()[n].cpu().numpy(), sd[n].numpy(), rtol=1e-4, atol=1e-6)
    with pytest.raises(Exception):
        rb(xg, final_nl=False)


def test_synthetic_front_end_encode_to_update():
    """Config 4 / N4 front end (dexgenerator_data_set.py:124-186 schema; dexgangrasp_real_robot.py:133-141 call
    pattern): clouds -> dgg_bps_encode -> batch dict -> update_dexgangrasp / update_dexevaluator.  The encoded block
    equals the CPU oracle's brute force bit for bit; the dict has the keys, shapes and dtypes of data/eval_batch.pth;
    the wrapper's step on it equals the engine's step on the same tensors."""
    from oracle import bps_oracle
    from dex_gan_grasp_amd.data.synthetic_data_set import SyntheticGraspBatches
    from dex_gan_grasp_amd.models.dexgangrasp import DexGanGrasp
    L, ops, engine = _mods()
    basis = H.load("bps_real_objects.npz")["basis"]
    b = 48
    ds = SyntheticGraspBatches(CFG, b, 3, basis, n_points=700, seed=5, with_label=True)
    batch = ds.batch(1)
    ref = H.eval_rows()
    for k in ("rot_matrix", "transl", "joint_conf", "bps_object"):
        assert batch[k].dtype == ref[k].dtype and tuple(batch[k].shape[1:]) == tuple(ref[k].shape[1:]), k
    assert batch["bps_object"].is_cuda and batch["label"].dtype == torch.int64 and len(batch["obj_name"]) == b
    clouds = ds.clouds(1).cpu().numpy()
    got = batch["bps_object"].cpu().numpy()
    for i in (0, 17, b - 1):
        od, _ = bps_oracle.encode_f32(clouds[i], basis)
        assert np.array_equal(got[i], od), i
    assert float(got.min()) < 0.1                                   # bps_encoder.py:50
    again = ds.batch(1)
    assert torch.equal(again["bps_object"], batch["bps_object"]) and torch.equal(again["transl"], batch["transl"])
    assert not torch.equal(ds.batch(2)["bps_object"], batch["bps_object"])
    # the dict drops into the wrapper unchanged
    cfg = dict(CFG)
    cfg["train_ffhevaluator"] = True
    m = DexGanGrasp(cfg)
    m.DexGANGrasp.load_state_dict(H.gan_sd("wide"))
    m.DexEvaluator.load_state_dict(H.eva_sd("wide"))
    eng, _, _ = _engine("wide")
    z = synth.make_z(b, 3)
    m._z_override = z
    losses = m.update_dexgangrasp(dict(batch))
    f = lambda t: t.float().to(DEV).contiguous()
    want = eng.cgan_step(batch["bps_object"], f(batch["rot_matrix"]).reshape(b, 9), f(batch["transl"]),
                         f(batch["joint_conf"]), z.to(DEV), True)
    assert float(losses["total_loss_gen"]) == float(want[L.L_TOTAL_GEN])
    assert float(losses["total_loss_disc"]) == float(want[L.L_TOTAL_DISC])
    masks = synth.make_dropout_masks(b, 9)
    m._drop_keep = masks
    le = m.update_dexevaluator(dict(batch))
    we = eng.eva_step(batch["bps_object"], f(batch["rot_matrix"]).reshape(b, 9), f(batch["transl"]),
                      f(batch["label"]), masks.to(DEV))
    assert float(le["total_loss_eva"]) == float(we[L.L_TOTAL_EVA])
    assert torch.equal(m.DexEvaluator._require_flat().params, eng.E.params)
    eng.close()


def test_config5_sweep_at_rank_share_size():
    """BASELINE config 5 at one rank's share (1250 objects x 200 grasps): sweep_objects (one generator + one evaluator
    pass per chunk of objects) against the reference's per-object call sequence generate_grasps -> filter x3
    (dexgangrasp_offline.py:27-92,147-162; dexgangrasp.py:393-442) on a sample of the objects, with "wide" weights and
    thresholds that keep a non-empty, non-total subset; and the sharded form's bookkeeping on one rank."""
    from dex_gan_grasp_amd.models.dexgangrasp import DexGanGrasp
    cfg = dict(CFG)
    cfg["is_train"] = False
    m = DexGanGrasp(cfg)
    m.DexGANGrasp.load_state_dict(H.gan_sd("wide"))
    m.DexEvaluator.load_state_dict(H.eva_sd("wide"))
    n_obj, n = 1250, 200
    g = torch.Generator().manual_seed(3)
    rows = (torch.rand((n_obj, 4096), generator=g) * 0.2).numpy().astype(np.float32)
    z = torch.randn((n_obj * n, 5), generator=g).to(DEV)
    probe = m.sweep_objects(rows[:8], n, thresholds=(0.5,), z=z[:8 * n])
    sc = probe["score"].flatten().cpu().numpy()
    thresholds = tuple(float(np.quantile(sc, q)) for q in (0.3, 0.6, 0.9))   # keep ~70 % / 40 % / 10 %
    out = m.sweep_objects(rows, n, thresholds=thresholds, z=z)
    assert out["score"].shape == (n_obj, n) and out["count"].shape == (3, n_obj)
    tot = out["count"].sum(dim=1).cpu().numpy()
    assert 0 < tot[2] < tot[1] < tot[0] < n_obj * n, tot
    for o in (0, 1, 617, n_obj - 1):          # the reference's per-object sequence on a sample of objects
        orig = torch.randn
        try:
            torch.randn = lambda *a, **k: z[o * n:(o + 1) * n].to(k.get("device", "cpu"))
            gr = m.generate_grasps(rows[o], n, return_arr=True)
        finally:
            torch.randn = orig
        np.testing.assert_allclose(out["transl"][o].cpu().numpy(), gr["transl"], rtol=0, atol=2e-5)
        np.testing.assert_allclose(out["joint_conf"][o].cpu().numpy(), gr["joint_conf"], rtol=0, atol=2e-5)
        score = out["score"][o].cpu().numpy()
        for t, thr in enumerate(thresholds):
            near = int(((score > thr - 1e-5) & (score < thr + 1e-5)).sum())
            try:
                f = m.filter_grasps(rows[o], {k: v.copy() for k, v in gr.items()}, thresh=thr)
                ref_cnt = f["transl"].shape[0]
            except ValueError:
                ref_cnt = 0
            cnt = int(out["count"][t, o])
            assert abs(cnt - ref_cnt) <= near, (o, thr, cnt, ref_cnt)
            if cnt and near == 0:   # same SET of grasps (scores 1e-7 apart may swap places in the two orderings)
                sel = out["order"][o, :cnt].cpu().numpy()
                a, c = out["transl"][o].cpu().numpy()[sel], f["transl"]
                # set equality up to 2e-5: every row of one has a partner in the other (sorting the rows instead breaks
                # when two grasps agree in their first coordinate to within the 1e-6 the two paths differ by)
                dist = np.abs(a[:, None, :] - c[None, :, :]).max(-1)
                assert a.shape == c.shape and dist.min(1).max() <= 2e-5 and dist.min(0).max() <= 2e-5, (o, thr)
    # ... and the ORACLE's per-object sequence on the same four objects: until round 5 the sweep at this
    # size was compared with the HIP wrapper only — pinned to the reference transitively, through the n = 32 goldens.
    # oracle.generate_grasps / evaluate_grasps / filter_grasps restate dexgangrasp.py:495-514, :367-391, :393-442 on the
    # CPU: grasps to 2e-5 (as above), scores to 1e-4 (the north star's tolerance), and the SAME kept index set at every
    # threshold no score lies within 1e-5 of.
    from oracle import dexgan_oracle as O
    sd_g, sd_e = H.gan_sd("wide"), H.eva_sd("wide")
    for o in (0, 1, 617, n_obj - 1):
        zo = z[o * n:(o + 1) * n].cpu()
        og = O.generate_grasps(sd_g, rows[o], zo)
        np.testing.assert_allclose(out["transl"][o].cpu().numpy(), og["transl"], rtol=0, atol=2e-5)
        np.testing.assert_allclose(out["rot_matrix"][o].cpu().numpy().reshape(n, 3, 3), og["rot_matrix"].reshape(n, 3, 3),
                                   rtol=0, atol=2e-5)
        np.testing.assert_allclose(out["joint_conf"][o].cpu().numpy(), og["joint_conf"], rtol=0, atol=2e-5)
        osc = O.evaluate_grasps(sd_e, rows[o], og)
        score = out["score"][o].cpu().numpy()
        np.testing.assert_allclose(score, osc, rtol=0, atol=1e-4)
        for t, thr in enumerate(thresholds):
            cnt = int(out["count"][t, o])
            kept = set(out["order"][o, :cnt].cpu().numpy().tolist())
            if ((np.abs(osc - thr) < 1e-5) | (np.abs(score - thr) < 1e-5)).any():
                continue
            assert kept == set(np.nonzero(osc > thr)[0].tolist()), (o, thr)
            try:
                of = O.filter_grasps(sd_e, rows[o], og, thresh=thr)
                assert of["transl"].shape[0] == cnt, (o, thr, cnt, of["transl"].shape[0])
            except ValueError:
                assert cnt == 0, (o, thr, cnt)
    # the factorised passes against the tiled (dense) ones on a slice: same grasps and scores up to the summation order
    dn = m.sweep_objects(rows[:40], n, thresholds=thresholds, z=z[:40 * n], dense=True)
    for k in ("transl", "rot_matrix", "joint_conf", "score"):
        np.testing.assert_allclose(out[k][:40].cpu().numpy(), dn[k].cpu().numpy(), rtol=0, atol=3e-5, err_msg=k)
    local, (lo, hi), count_all = m.sweep_objects_sharded(rows[:300], n, thresholds=thresholds, z=z[:300 * n])
    assert (lo, hi) == (0, 300)
    near = sum(int(((out["score"][:300] > t - 1e-5) & (out["score"][:300] < t + 1e-5)).sum()) for t in thresholds)
    assert int((count_all.cpu() - out["count"][:, :300].cpu()).abs().sum()) <= near     # another chunking, another split-K


# ------------------------------------------------------------------ N3: WGAN-GP
def _alphas(b, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn((b, 1, 1), generator=g), torch.randn((b, 1), generator=g), torch.randn((b, 1), generator=g)


@pytest.mark.parametrize("b,seeds", [(64, (1, 2, 3)), (300, (1, 2, 3)), (1024, (1,)), (1100, (2,))])
def test_wgan_gp_phases_vs_oracle(b, seeds):
    """dexgangrasp.py:149-211, 234-270: the Wasserstein D phase with gradient penalty (second derivative of the
    discriminator) and the G phase, gradients before each Adam against torch's double backward on the oracle.
    The batch sizes walk the variants of the second-order BatchNorm kernels (second_order.hip): 4 and 8 rows per thread
    in registers at 16 columns per workgroup, 8 rows at 8 columns (the benched B = 1024), and the row sweeps beyond 1024."""
    import ctypes as C
    from oracle import dexgan_oracle as O
    L, ops, engine = _mods()

    def run(seed):
        eng, sd, _ = _engine("wide")
        batch = synth.make_batch(b, seed)
        z = synth.make_z(b, 100 + seed)
        al = _alphas(b, 200 + seed)
        f = lambda t: t.float().to(DEV).contiguous()
        ins = (f(batch["bps_object"]), f(batch["rot_matrix"]).reshape(b, 9), f(batch["transl"]), f(batch["joint_conf"]))
        ald = torch.cat([al[0].view(b, 1), al[1], al[2]], 1).to(DEV).contiguous()
        ws = eng._wgan_ws(b)
        st, hp = eng._state(), eng._hp()
        zd = z.to(DEV)
        args = tuple(L.ptr(t) for t in ins) + (L.ptr(zd),)
        L.check(L.lib().dgg_wgan_d_phase(C.byref(eng.ccfg), C.byref(st), C.byref(hp), L.ptr(ws), ws.numel(), b, *args,
                                         L.ptr(ald), 10.0, L.ptr(eng.losses), L.stream_ptr()))
        got = eng.losses.cpu().numpy()
        # the LeakyReLU sides of the four passes (G, D real, D fake, D interpolates: the WGAN workspace is the cGAN step's
        # followed by the interpolates' pass), handed to the oracle (helpers.assert_grad_tensor_close)
        cfg_p = C.byref(eng.ccfg)
        base_i = int(L.lib().dgg_cgan_workspace_bytes(cfg_p, b))
        sides = eng.kink_sides("G", b, ws) + eng.kink_sides("Dreal", b, ws) + eng.kink_sides("Dfake", b, ws) + \
            L.kink_sides(L.NET_DISC, eng.ccfg, ws, b, base_i)
        with O.forced_slopes(H.cpu_masks(sides)):
            ref, g_d, ctx = O.wgan_d_part(sd, H.clone_batch(batch), z, al, CFG)
        np.testing.assert_allclose(got[L.L_BCE_REAL], float(ref["bce_loss_real"]), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(got[L.L_BCE_FAKE], float(ref["bce_loss_fake"]), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(got[L.L_GRAD_PENALTY], float(ref["gradient_penalty"]), rtol=2e-4, atol=1e-6)
        np.testing.assert_allclose(got[L.L_TOTAL_DISC], float(ref["total_loss_disc"]), rtol=2e-4, atol=1e-5)
        items = _grad_items(eng.D, g_d, "discriminator.", b)
        # G phase on the oracle's post-Adam discriminator
        ad = O.AdamState([n for n in O.param_names(sd) if n.startswith("discriminator.")])
        O.adam_step(sd, g_d, ad, lr=CFG["lr_dis"], b1=CFG["beta1"], wd=CFG["weight_decay"])
        eng.D.load_state_dict(sd, "discriminator.")      # identical D on both sides for the G phase
        L.check(L.lib().dgg_wgan_g_phase(C.byref(eng.ccfg), C.byref(st), C.byref(hp), L.ptr(ws), ws.numel(), b, *args,
                                         L.ptr(eng.losses), L.stream_ptr()))
        got = eng.losses.cpu().numpy()
        with O.forced_slopes(H.cpu_masks(eng.kink_sides("Dfake", b, ws))):
            refg, g_g = O.wgan_g_part(sd, ctx, CFG)
        for k, n in ((L.L_GEN_FAKE, "gen_loss_fake"), (L.L_TRANSL, "transl_loss"), (L.L_ROT, "rot_loss"),
                     (L.L_CONF, "conf_loss"), (L.L_TOTAL_GEN, "total_loss_gen")):
            np.testing.assert_allclose(got[k], float(refg[n]), rtol=1e-4, atol=1e-6, err_msg=n)
        items += _grad_items(eng.G, g_g, "generator.", b)
        eng.close()
        return items
    H.check_grads_over_seeds(run, list(seeds))


@pytest.mark.parametrize("variant,bname", [("wide", "real16"), ("wide", "syn64")])
def test_wrapper_wgan_steps_vs_reference_goldens(variant, bname):
    """The drop-in wrapper with cfg["is_wgan"] = True against the reference's own two update_dexgangrasp steps."""
    from dex_gan_grasp_amd.models.dexgangrasp import DexGanGrasp
    g = H.load("wgan_goldens.npz")
    key = "%s/%s/" % (variant, bname)
    cfg = dict(CFG)
    cfg["is_wgan"] = True
    m = DexGanGrasp(cfg)
    m.DexGANGrasp.load_state_dict(H.gan_sd(variant))
    batch = H.batches()[bname]
    b = batch["bps_object"].shape[0]
    for step in range(2):
        sk = key + "wgan%d/" % step
        m._z_override = torch.from_numpy(g[sk + "z"])
        m._alpha_override = torch.cat([torch.from_numpy(g[sk + "alpha_rot"]).view(b, 1),
                                       torch.from_numpy(g[sk + "alpha_transl"]), torch.from_numpy(g[sk + "alpha_joint"])], 1)
        losses = m.update_dexgangrasp(H.clone_batch(batch), is_train_gen=True)
        for k, v in losses.items():
            np.testing.assert_allclose(v.item(), float(g[sk + "loss/" + k]), rtol=2e-4, atol=1e-5, err_msg=k)
        for n, t in m.DexGANGrasp.state_dict().items():
            H.assert_state_close(n, t.cpu(), g[sk + "state/" + n], step)
    sd = m.DexGANGrasp.state_dict()
    assert int(sd["discriminator.disc_bn1.num_batches_tracked"]) == 8
    # the interpolation weights are drawn like the reference's when nothing is injected
    m._z_override = None
    m._alpha_override = None
    out = m.update_dexgangrasp(H.clone_batch(batch), is_train_gen=False)
    assert set(out.keys()) >= {"total_loss_disc", "bce_loss_real", "bce_loss_fake", "total_loss_gen"}
    assert all(torch.isfinite(v).item() for v in out.values())


def test_graph_replay_equals_eager_steps():
    """HIP-graph capture of the gen+eval step (device-resident Adam step numbers and dropout seed): priming step + 3
    replays are bit-identical to 4 eager joint_steps given the same seeds, inputs refreshed in place between replays."""
    from dex_gan_grasp_amd.engine import SEED_STRIDE
    L, ops, engine = _mods()
    b = 128
    gbs = [_dev_inputs(synth.make_batch(b, s)) for s in (1, 11)]
    eb = _dev_inputs(synth.make_batch(b, 2, with_label=True), with_label=True)
    zs = [synth.make_z(b, 100 + i).to(DEV) for i in range(4)]
    e1, _, _ = _engine("wide")
    e2, _, _ = _engine("wide")
    seed0 = 12345
    for i in range(4):      # eager reference: the seeds the graph path will use
        e1.joint_step(gbs[i % 2], zs[i], eb, True, None, (seed0 + i * SEED_STRIDE) % (1 << 64))
    g_in = tuple(t.clone() for t in gbs[0])
    z_in = zs[0].clone()
    gs = e2.capture_joint_step(g_in, z_in, eb, True, seed=seed0)          # runs step 0 eagerly
    for i in range(1, 4):
        for dst, src in zip(g_in, gbs[i % 2]):
            dst.copy_(src)
        z_in.copy_(zs[i])
        gs.replay()
    torch.cuda.synchronize()
    for a, c in ((e1.G, e2.G), (e1.D, e2.D), (e1.E, e2.E)):
        assert torch.equal(a.params, c.params) and torch.equal(a.m, c.m) and torch.equal(a.v, c.v) and torch.equal(a.bn, c.bn)
        assert a.step == c.step == 4
    assert torch.equal(e1.losses[:10], e2.losses[:10])
    # eager steps continue correctly after replays (device and host step counters agree)
    e1.joint_step(gbs[0], zs[0], eb, True, None, 7)
    e2.joint_step(gbs[0], zs[0], eb, True, None, 7)
    torch.cuda.synchronize()
    assert torch.equal(e1.G.params, e2.G.params) and torch.equal(e1.E.params, e2.E.params) and e2.E.step == 5
    e1.close()
    e2.close()


@pytest.mark.parametrize("b", [33, 127, 1000])
def test_odd_batch_sizes_vs_oracle(b):
    """Batches that are no multiple of the 32-row statistics blocks, the 64-row GEMM tiles or the split-K chunks: the
    fused forward stage (partial row blocks in its statistics epilogue), the slab-summing backward kernels, the shared
    weight gradient of D's two passes and the side-work launches of joint_step — two steps with prefetch against the
    oracle: losses to 1e-4, gradients strictly (the oracle takes the HIP path's side at LeakyReLU kink crossings), Adam of
    the engine's own gradients exact."""
    from oracle import dexgan_oracle as O
    L, ops, engine = _mods()
    eng, sd, sde = _engine("wide")
    ag = O.AdamState([n for n in O.param_names(sd) if n.startswith("generator.")])
    ad = O.AdamState([n for n in O.param_names(sd) if n.startswith("discriminator.")])
    ae = O.AdamState(O.param_names(sde))
    gbs = [synth.make_batch(b, 3), synth.make_batch(b, 13)]
    g_ins = [_dev_inputs(x) for x in gbs]
    ebs = [synth.make_batch(b, 4 + 10 * i, with_label=True) for i in range(2)]
    store = {}
    eng.after_d_phase = _sides_hook(store)
    for i in range(2):
        z = synth.make_z(b, 50 + i)
        e_in = _dev_inputs(ebs[i], with_label=True)
        before = {k: _snap_flat(fl) for k, fl in (("G", eng.G), ("D", eng.D), ("E", eng.E))}
        got = eng.joint_step(g_ins[i], z.to(DEV), e_in, True, None, 31 + i, next_gan_inputs=g_ins[1] if i == 0 else None)
        torch.cuda.synchronize()
        got = got.cpu().numpy()
        masks = eng.eva_keep_masks(b).float().cpu()
        sl = _slopes_after(eng, store, b, b)
        ref, g_d, g_g, _ = O.cgan_step(sd, H.clone_batch(gbs[i]), z, ag, ad, CFG, True, return_grads=True, slopes=sl["gan"])
        refe, g_e, _ = O.eva_step(sde, H.clone_batch(ebs[i]), ae, CFG, masks, return_grads=True, slopes=sl["eva"])
        for k, n in enumerate(LOSS_NAMES):
            np.testing.assert_allclose(got[k], float(ref[n]), rtol=1e-4, atol=1e-6, err_msg="step %d %s" % (i, n))
        np.testing.assert_allclose(got[L.L_TOTAL_EVA], float(refe["total_loss_eva"]), rtol=1e-4, atol=1e-6)
        checked = [_grad_items(eng.G, g_g, "generator.", b), _grad_items(eng.E, g_e, "", b)]
        if i == 1:     # D's gradients of step 0 are already overwritten by the prefetched pass of step 1
            checked.append(_grad_items(eng.D, g_d, "discriminator.", b))
        for items in checked:
            for name, g, r, train, batch in items:
                H.assert_grad_tensor_close(name, g, r, train=train, batch=batch, what="B=%d step %d " % (b, i))
        _cmp_adam_of_own_grads(eng.G, before["G"], what="G")
        _cmp_adam_of_own_grads(eng.E, before["E"], what="E")
        if i == 1:
            _cmp_adam_of_own_grads(eng.D, before["D"], what="D")
        _adopt(eng.G, sd, ag, "generator.")
        _adopt(eng.D, sd, ad, "discriminator.")
        _adopt(eng.E, sde, ae, "")
    eng.close()


def test_weight_gradient_streams_opt_in_is_bit_identical(monkeypatch):
    """DGG_WGRAD_STREAMS=1 (library-owned weight-gradient side streams, dgg_enable_wgrad_stream — opt-in since the engine
    keeps to three streams): cgan_step / eva_step with them give the same bits as without."""
    L, ops, engine = _mods()
    b = 128
    gb = _dev_inputs(synth.make_batch(b, 1))
    eb = _dev_inputs(synth.make_batch(b, 2, with_label=True), with_label=True)
    z = synth.make_z(b, 3).to(DEV)
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("DGG_WGRAD_STREAMS", mode)
        eng, _, _ = _engine("wide")
        for i in range(2):
            eng.cgan_step(*gb, z, True, two_streams=True)
            eng.eva_step(*eb, None, 17 + i, two_streams=True)
        torch.cuda.synchronize()
        out[mode] = [x.params.clone() for x in (eng.G, eng.D, eng.E)]
        assert (len(getattr(eng, "_wg_mapped", None) or ()) > 0) == (mode == "1")
        eng.close()
    for a, c in zip(out["0"], out["1"]):
        assert torch.equal(a, c)
