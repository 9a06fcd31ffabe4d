"""Seeded synthetic weights and batches for the DexGANGrasp hot path.

Everything here is plain torch-CPU and deterministic for a given seed on the
same torch build, so the golden-vector generator (tests/golden/make_goldens.py,
run next to the reference), the CPU oracle, the parity tests and bench.py all
regenerate *identical* tensors from nothing but a seed.

Shapes follow the reference's modules (DexGanGrasp/models/networks.py:26-338):
  ResBlock   networks.py:26-41   fc1 (hid,Fin)  bn1(hid)  fc2 (Fout,hid)  bn2(Fout)  fc3 (Fout,Fin)
  Generator  networks.py:71-77   gen_bn1(4096) gen_rb1(5+4096) gen_rb2(512+5+4096) heads 12/6/3
  Discrim.   networks.py:117-121 disc_bn1(4096+24) disc_rb1 disc_rb2 out_success
  Evaluator  networks.py:289-293 bn1(4096+12) rb1 rb2 rb3 out_success
Init distributions follow dexgangrasp.py:39-62 (xavier-normal with `init_gain`
on Linear.weight, BN weight ~ N(1, gain), BN bias 0, Linear.bias torch default).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch

DEFAULT_CFG = {
    # config_dexgangrasp.yaml:36-57 (hot-path keys only)
    "latentD": 5,
    "n_hand_joints": 12,
    "is_train": True,
    "gpu_ids": [0],
    "lr": 1e-4,
    "lr_gen": 1e-4,
    "lr_dis": 1e-4,
    "beta1": 0.9,
    "weight_decay": 5e-4,
    "bce_weight": 1,
    "kl_coef": 0.005,
    "is_wgan": False,
    "gen_train_freq": 1,
    "weight_init_type": "xavier",
    "init_gain": 0.02,
    "lr_policy": "plateau",
    "patience_lr_policy_plateau": 3,
    "threshold_lr_policy_plateau": 0.01,
    "batch_size": 32,
    "train_ffhgenerator": True,
    "train_ffhevaluator": False,
    "model": "ffhnet",
    "save_dir": "/tmp/dgg_ckpt",
    "load_path": None,
}

IN_BPS = 4096
N_NEURONS = 512
HIDDEN = 256


def _resblock_spec(prefix, fin, fout=N_NEURONS, hid=HIDDEN):
    s = []
    s.append((prefix + "fc1.weight", (hid, fin), "linear_w"))
    s.append((prefix + "fc1.bias", (hid,), "linear_b:%d" % fin))
    s += _bn_spec(prefix + "bn1.", hid)
    s.append((prefix + "fc2.weight", (fout, hid), "linear_w"))
    s.append((prefix + "fc2.bias", (fout,), "linear_b:%d" % hid))
    s += _bn_spec(prefix + "bn2.", fout)
    s.append((prefix + "fc3.weight", (fout, fin), "linear_w"))
    s.append((prefix + "fc3.bias", (fout,), "linear_b:%d" % fin))
    return s


def _bn_spec(prefix, n):
    return [
        (prefix + "weight", (n,), "bn_w"),
        (prefix + "bias", (n,), "bn_b"),
        (prefix + "running_mean", (n,), "bn_rm"),
        (prefix + "running_var", (n,), "bn_rv"),
        (prefix + "num_batches_tracked", (), "bn_nbt"),
    ]


def _linear_spec(prefix, fout, fin):
    return [(prefix + "weight", (fout, fin), "linear_w"), (prefix + "bias", (fout,), "linear_b:%d" % fin)]


def net_spec(net, cfg=None):
    """state_dict entries [(name, shape, kind)] in the reference's registration order."""
    cfg = cfg or DEFAULT_CFG
    lat, nj = cfg["latentD"], cfg["n_hand_joints"]
    if net == "generator":
        s = _bn_spec("gen_bn1.", IN_BPS)
        s += _resblock_spec("gen_rb1.", lat + IN_BPS)
        s += _resblock_spec("gen_rb2.", N_NEURONS + lat + IN_BPS)
        s += _linear_spec("gen_joint_conf.", nj, N_NEURONS)
        s += _linear_spec("gen_rot.", 6, N_NEURONS)
        s += _linear_spec("gen_transl.", 3, N_NEURONS)
        return s
    if net == "discriminator":
        fin = IN_BPS + 12 + nj
        s = _bn_spec("disc_bn1.", fin)
        s += _resblock_spec("disc_rb1.", fin)
        s += _resblock_spec("disc_rb2.", N_NEURONS + fin)
        s += _linear_spec("out_success.", 1, N_NEURONS)
        return s
    if net == "evaluator":
        fin = IN_BPS + 12
        s = _bn_spec("bn1.", fin)
        s += _resblock_spec("rb1.", fin)
        s += _resblock_spec("rb2.", fin + N_NEURONS)
        s += _resblock_spec("rb3.", fin + N_NEURONS)
        s += _linear_spec("out_success.", 1, N_NEURONS)
        return s
    if net == "dexgangrasp":  # networks.py:178-191: discriminator registered first
        s = [("discriminator." + n, sh, k) for n, sh, k in net_spec("discriminator", cfg)]
        s += [("generator." + n, sh, k) for n, sh, k in net_spec("generator", cfg)]
        return s
    raise ValueError(net)


def make_state_dict(net, seed, gain=0.02, trained_like=False, cfg=None):
    """Seeded state_dict (fp32 CPU).  `gain` is the xavier gain (reference: 0.02).

    trained_like=True additionally randomises BN biases and running statistics so
    that eval-mode tests do not run on the trivial (0, 1) running stats.
    """
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    sd = OrderedDict()
    for name, shape, kind in net_spec(net, cfg):
        if kind == "linear_w":
            fout, fin = shape
            std = gain * math.sqrt(2.0 / float(fin + fout))
            t = torch.randn(shape, generator=g, dtype=torch.float32) * std
        elif kind.startswith("linear_b"):
            fin = int(kind.split(":")[1])
            bound = 1.0 / math.sqrt(fin)
            t = (torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1) * bound
        elif kind == "bn_w":
            t = 1.0 + 0.02 * torch.randn(shape, generator=g, dtype=torch.float32)
        elif kind == "bn_b":
            t = torch.zeros(shape, dtype=torch.float32)
            if trained_like:
                t = 0.1 * torch.randn(shape, generator=g, dtype=torch.float32)
        elif kind == "bn_rm":
            t = torch.zeros(shape, dtype=torch.float32)
            if trained_like:
                t = 0.05 * torch.randn(shape, generator=g, dtype=torch.float32)
        elif kind == "bn_rv":
            t = torch.ones(shape, dtype=torch.float32)
            if trained_like:
                t = 0.5 + torch.rand(shape, generator=g, dtype=torch.float32)
        elif kind == "bn_nbt":
            t = torch.tensor(0, dtype=torch.int64)
        else:
            raise ValueError(kind)
        sd[name] = t
    if trained_like:
        # running statistics that resemble the data: the BPS columns have mean ~0.1, var ~3e-3
        for name in list(sd.keys()):
            if name.endswith("bn1.running_mean") and sd[name].numel() >= IN_BPS:
                rm = sd[name]
                first = 0
                rm[first:first + IN_BPS] += 0.1
            if name.endswith("bn1.running_var") and sd[name].numel() >= IN_BPS:
                sd[name][:IN_BPS] *= 3e-3
    return sd


def random_rotations(n, g, dtype=torch.float64):
    """QR-orthonormalised N(0,1) 3x3 with det fixed to +1."""
    a = torch.randn((n, 3, 3), generator=g, dtype=torch.float64)
    q, r = torch.linalg.qr(a)
    d = torch.sign(torch.diagonal(r, dim1=1, dim2=2))
    d = torch.where(d == 0, torch.ones_like(d), d)
    q = q * d[:, None, :]
    det = torch.linalg.det(q)
    q[:, :, 2] = q[:, :, 2] * det[:, None]
    return q.contiguous().to(dtype)


def make_batch(batch_size, seed, with_label=False, cfg=None):
    """Synthetic batch in the DataLoader schema of data/eval_batch.pth
    (dexgenerator_data_set.py:124-186): rot_matrix (B,3,3) f64, transl (B,3) f64,
    joint_conf (B,nj) f64, bps_object (B,4096) f32 [, label (B,) int64]."""
    cfg = cfg or DEFAULT_CFG
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    nj = cfg["n_hand_joints"]
    bps = torch.rand((batch_size, IN_BPS), generator=g, dtype=torch.float32) * 0.2
    rot = random_rotations(batch_size, g)
    transl = torch.rand((batch_size, 3), generator=g, dtype=torch.float64) * 0.34 - 0.2
    joint = torch.rand((batch_size, nj), generator=g, dtype=torch.float64) * 0.5
    out = {"rot_matrix": rot, "transl": transl, "joint_conf": joint, "bps_object": bps}
    if with_label:
        out["label"] = (torch.rand((batch_size,), generator=g) < 0.3).to(torch.int64)
    return out


def make_z(batch_size, seed, latent=5):
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    return torch.randn((batch_size, latent), generator=g, dtype=torch.float32)


def make_dropout_masks(batch_size, seed, n_blocks=3, p=0.3, width=N_NEURONS):
    """Bernoulli keep-masks (1 = keep) for the evaluator's three dropout sites
    (networks.py:329,331,333).  Passed explicitly so CPU oracle and HIP path use
    the same mask (torch's Philox stream is not reproducible in a hand kernel)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    return (torch.rand((n_blocks, batch_size, width), generator=g) >= p).to(torch.float32)


def make_clouds(n_clouds, n_points, seed):
    """Centred synthetic object clouds: points on a random-axis ellipsoid surface with
    semi-axes U(0.03, 0.12) m; f32 (n_clouds, n_points, 3)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    v = torch.randn((n_clouds, n_points, 3), generator=g, dtype=torch.float64)
    v = v / v.norm(dim=2, keepdim=True).clamp_min(1e-12)
    axes = torch.rand((n_clouds, 1, 3), generator=g, dtype=torch.float64) * 0.09 + 0.03
    r = random_rotations(n_clouds, g)
    pts = torch.einsum("bij,bnj->bni", r, v * axes)
    pts = pts - pts.mean(dim=1, keepdim=True)
    return pts.to(torch.float32).contiguous()
