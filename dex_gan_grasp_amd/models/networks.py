"""Drop-in mirror of DexGanGrasp/models/networks.py (reference lines cited per class).

Same class names, constructor/forward signatures, side-effect attributes and state_dict keys as
the reference, but every forward/backward runs in libdgg.so (hand-written HIP for gfx950) through
the C-ABI of include/dgg.h.  torch supplies device memory, autograd plumbing and state_dict only.

Storage: all Parameters and BatchNorm running statistics of one network are views into flat
device buffers laid out by the library (dgg_table): fc1/fc3 of a ResBlock form one 768xFin
matrix, one fused Adam launch and one all-reduce bucket per network.  The views keep the
reference's names and shapes, so checkpoints round-trip unchanged.

There is no PyTorch fallback: on a machine without the library / without a GPU the modules can be
constructed and (de)serialised, but forward() raises.
"""
from __future__ import annotations

import time

import numpy as np
import torch
from torch import nn
from torch.optim import lr_scheduler

from .. import _lib, ops, parallel
from ..utils import utils


# lr_policy -> scheduler factory (the two policies of networks.py:12-23, same hyper-parameters)
_SCHEDULERS = {
    "step": lambda opt, cfg: lr_scheduler.StepLR(opt, step_size=cfg["lr_decay_iters"], gamma=0.1),
    "plateau": lambda opt, cfg: lr_scheduler.ReduceLROnPlateau(
        opt, mode="min", factor=0.2, threshold=cfg["threshold_lr_policy_plateau"],
        patience=cfg["patience_lr_policy_plateau"]),
}


def get_scheduler(optimizer, cfg):
    """The learning-rate scheduler cfg["lr_policy"] names; anything else: NotImplementedError (networks.py:12-23)."""
    make = _SCHEDULERS.get(cfg["lr_policy"])
    if make is None:
        raise NotImplementedError('Scheduler not implemented.')
    return make(optimizer, cfg)


def _default_device(cfg):
    """cfg["gpu_ids"][0] as in the reference (networks.py:170); one process per GPU under torchrun: this rank's GPU."""
    if not torch.cuda.is_available():
        return torch.device('cpu')
    if parallel.env_rank_world()[1] > 1:
        return torch.device('cuda:{}'.format(parallel.local_device_index()))
    return torch.device('cuda:{}'.format(cfg["gpu_ids"][0]))


class _ResBlockFn(torch.autograd.Function):
    """Stand-alone ResBlock through dgg_resblock_forward / dgg_resblock_backward.  The block's tensors are packed
    into the library's flat layout for the call ([fc1.weight ; fc3.weight | biases | bn1 | fc2 | bn2]) — inside a
    network they already live that way; on its own the module keeps ordinary Parameters."""

    @staticmethod
    def forward(ctx, mod, train, x, *params):
        import ctypes as C
        dev = x.device
        fin = mod.Fin
        lib = _lib.lib()
        n = int(lib.dgg_resblock_table(fin, None, 0))
        arr = (_lib.Entry * n)()
        lib.dgg_resblock_table(fin, arr, n)
        entries = [(e.name.decode(), int(e.offset), int(e.rows) * int(e.cols), int(e.kind)) for e in arr]
        named = dict(mod.named_parameters())
        bufs = dict(mod.named_buffers())
        flat = torch.zeros(int(lib.dgg_resblock_param_floats(fin)), dtype=torch.float32, device=dev)
        bn = torch.zeros(int(lib.dgg_resblock_bnstat_floats(fin)), dtype=torch.float32, device=dev)
        for name, off, cnt, kind in entries:
            src = named[name] if kind == 0 else bufs[name]
            (flat if kind == 0 else bn)[off:off + cnt].copy_(src.detach().reshape(-1))
        xc = _lib.f32c(x, dev)
        B = xc.shape[0]
        ws = _lib.new_workspace(int(lib.dgg_resblock_workspace_bytes(fin, B)), dev)
        y = torch.empty((B, mod.Fout), dtype=torch.float32, device=dev)
        _lib.check(lib.dgg_resblock_forward(fin, _lib.ptr(flat), _lib.ptr(bn), _lib.ptr(ws), ws.numel(), B,
                                            _lib.TRAIN if train else 0, _lib.ptr(xc), xc.stride(0), _lib.ptr(y),
                                            _lib.stream_ptr(dev)), "dgg_resblock_forward")
        if train:  # running statistics advanced in the packed copy: write them back
            with torch.no_grad():
                for name, off, cnt, kind in entries:
                    if kind == 1:
                        bufs[name].copy_(bn[off:off + cnt].view_as(bufs[name]))
        ctx.mod, ctx.train, ctx.entries, ctx.ws, ctx.flat = mod, train, entries, ws, flat
        ctx.save_for_backward(xc)
        ctx.need_x = x.requires_grad
        ctx.pnames = [nm for nm, _ in mod.named_parameters()]
        return y

    @staticmethod
    def backward(ctx, dy):
        (xc,) = ctx.saved_tensors
        mod, dev = ctx.mod, xc.device
        lib = _lib.lib()
        B = xc.shape[0]
        grads = torch.zeros_like(ctx.flat)
        dx = torch.empty((B, mod.Fin), dtype=torch.float32, device=dev) if ctx.need_x else None
        dyc = _lib.f32c(dy, dev)
        _lib.check(lib.dgg_resblock_backward(mod.Fin, _lib.ptr(ctx.flat), _lib.ptr(grads), _lib.ptr(ctx.ws),
                                             ctx.ws.numel(), B, _lib.TRAIN if ctx.train else 0, _lib.ptr(xc),
                                             xc.stride(0), _lib.ptr(dyc), _lib.ptr(dx), _lib.stream_ptr(dev)),
                   "dgg_resblock_backward")
        named = dict(mod.named_parameters())
        off = {name: (o, c) for name, o, c, kind in ctx.entries if kind == 0}
        pg = [grads[off[nm][0]:off[nm][0] + off[nm][1]].view_as(named[nm]) for nm in ctx.pnames]
        return (None, None, dx, *pg)


class ResBlock(nn.Module):
    """networks.py:26-56.  Holds the parameters (fc1, bn1, fc2, bn2, fc3) under the reference's names.  Inside
    Generator / Discriminator / DexEvaluator the block is executed by the fused network kernels; called on its own
    (networks.py:43-56) it runs through dgg_resblock_forward / _backward with full autograd support."""

    def __init__(self, Fin, Fout, n_neurons=256):
        super(ResBlock, self).__init__()
        self.Fin = Fin
        self.Fout = Fout
        self.fc1 = nn.Linear(Fin, n_neurons)
        self.bn1 = nn.BatchNorm1d(n_neurons)
        self.fc2 = nn.Linear(n_neurons, Fout)
        self.bn2 = nn.BatchNorm1d(Fout)
        if Fin != Fout:
            self.fc3 = nn.Linear(Fin, Fout)
        self.ll = nn.LeakyReLU(negative_slope=0.2)

    def _apply(self, fn, *args, **kwargs):
        super()._apply(fn, *args, **kwargs)
        _pin_bn_counters(self)
        return self

    def forward(self, x, final_nl=True):
        if not x.is_cuda:
            raise _lib.DggError("ResBlock.forward: GPU tensor required (this implementation has no CPU path)")
        if self.Fin == self.Fout or self.fc1.out_features != 256 or self.Fout != 512 or not final_nl:
            # every ResBlock the reference builds is (Fin != 512) -> 512 with 256 hidden units and final_nl=True
            # (networks.py:73-74,119-120,290-292); other shapes have no kernel
            raise _lib.DggError("stand-alone ResBlock: only Fin != Fout = 512, n_neurons = 256, final_nl=True")
        y = _ResBlockFn.apply(self, self.training, x, *[p for _, p in self.named_parameters()])
        if self.training:
            self.bn1.num_batches_tracked += 1
            self.bn2.num_batches_tracked += 1
        return y


def _pin_bn_counters(module):
    """BatchNorm's `num_batches_tracked` buffers stay HOST tensors whatever device the module moves to.  Nothing on the
    device reads them (momentum is 0.1: the reference only increments them, networks.py / nn.BatchNorm1d), and the train
    steps advance them by `+=`: on device tensors that is one kernel launch per BatchNorm layer and step — 17 tiny
    launches on the step's stream for G + D + E, the largest part of the wrapper's overhead over the engine calls (round
    6: 1.82 against 1.74 ms per gen+eval step).  state_dict() / load_state_dict() / torch.save work as before (the key
    holds a CPU int64 scalar, as after the reference's own `.cpu()` before a save)."""
    for m in module.modules():
        if isinstance(m, nn.BatchNorm1d) and m.num_batches_tracked is not None and m.num_batches_tracked.device.type != "cpu":
            m._buffers["num_batches_tracked"] = m.num_batches_tracked.cpu()


class _FlatBacked(nn.Module):
    """Common machinery: bind nn.Parameters / BN buffers to the library's flat buffers."""

    _net_kind = None

    def _init_flat(self):
        self._flat = None
        self._drop_keep = None  # evaluator only: explicit dropout keep-masks (3,B,512) for parity tests

    def _apply(self, fn, *args, **kwargs):
        super()._apply(fn, *args, **kwargs)
        self._bind_flat()
        _pin_bn_counters(self)
        return self

    def _bind_flat(self):
        """(Re)create the flat buffers on the parameters' current device and alias every
        Parameter / running statistic to its view.  Values are preserved."""
        p0 = next(self.parameters())
        if not p0.is_cuda:
            return  # e.g. .cpu() before torch.save: keep the flat buffers (Adam moments) for the way back
        flat = getattr(self, "_flat", None)
        if flat is not None and flat.device == p0.device:
            if self._is_bound(flat):
                return
        else:
            flat = _lib.FlatNet(self._net_kind, self.cfg, p0.device)
        pv, bv = flat.views("params"), flat.views("bn")
        gv = flat.views("grads")
        named = dict(self.named_parameters())
        bufs = dict(self.named_buffers())
        with torch.no_grad():
            for name, view in pv.items():
                view.copy_(named[name].data)
                named[name].data = view
                named[name].grad = None
            for name, view in bv.items():
                view.copy_(bufs[name].data)
                bufs[name].data = view
        object.__setattr__(self, "_flat", flat)
        object.__setattr__(self, "_grad_views", gv)

    def _is_bound(self, flat):
        p0 = next(self.parameters())
        lo = flat.params.data_ptr()
        return lo <= p0.data_ptr() < lo + flat.params.numel() * 4

    def _require_flat(self):
        p0 = next(self.parameters())
        if not p0.is_cuda:
            raise _lib.DggError("%s: parameters are not on a GPU; this implementation has no CPU path "
                                "(call .to('cuda') on a machine with an MI355X)" % type(self).__name__)
        if getattr(self, "_flat", None) is None or not self._is_bound(self._flat):
            self._bind_flat()
        return self._flat

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        return out

    def _bump_bn_counters(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm1d):
                m.num_batches_tracked += 1

    def bind_param_grads(self):
        """Point every Parameter's .grad at its view of the flat gradient buffer (what the fused
        training phases write), so optimisers / loggers that read .grad see the fused gradients."""
        flat = self._require_flat()
        gv = flat.views("grads")
        for name, p in self.named_parameters():
            p.grad = gv[name]


def _param_list(mod, flat):
    named = dict(mod.named_parameters())
    return [named[e[0]] for e in flat.entries if e[4] == 0]


class _NetFn(torch.autograd.Function):
    """autograd bridge: forward = dgg_*_forward, backward = dgg_*_backward on the workspace saved
    by this very forward call (each call owns its workspace, so a module may be called several
    times before backward, as the discriminator is: dexgangrasp.py:788-808)."""

    @staticmethod
    def forward(ctx, mod, kind, train, drop_keep, seed, n_in, *tensors):
        ins = tensors[:n_in]
        flat = mod._flat
        dev = flat.device
        B = ins[-1].shape[0] if kind == "gen" else ins[0].shape[0]
        ws = _lib.new_workspace(flat.workspace(B).numel(), dev)
        ctx.mod, ctx.kind, ctx.train, ctx.ws, ctx.n_in = mod, kind, train, ws, n_in
        ins = [_lib.f32c(t, dev) for t in ins]
        if kind == "gen":
            out = ops.gen_forward(flat, ins[0], ins[1], train, ws=ws)
        elif kind == "disc":
            out = ops.disc_forward(flat, ins[0], ins[1], ins[2], ins[3], train, ws=ws)
        else:
            out = ops.eva_forward(flat, ins[0], ins[1], ins[2], train, drop_keep, seed, ws=ws)
        ctx.save_for_backward(*ins, out)
        ctx.needs = [t.requires_grad for t in tensors[:n_in]]
        ctx.param_needs = any(t.requires_grad for t in tensors[n_in:])
        return out

    @staticmethod
    def backward(ctx, d_out):
        mod, kind, train, ws = ctx.mod, ctx.kind, ctx.train, ctx.ws
        flat = mod._flat
        saved = ctx.saved_tensors
        ins, out = saved[:-1], saved[-1]
        dev = flat.device
        d_out = _lib.f32c(d_out, dev)
        # fresh gradient buffer for this backward; autograd accumulates the returned views into .grad
        gbuf = torch.zeros_like(flat.params)
        saved_grads = flat.grads
        flat.grads = gbuf
        try:
            skip = not ctx.param_needs
            if kind == "gen":
                dz, dbps = ops.gen_backward(flat, ins[0], ins[1], d_out, train, want_dz=ctx.needs[0],
                                            want_dbps=ctx.needs[1], ws=ws)
                din = [dz, dbps]
            else:
                d_logit = d_out * out * (1.0 - out)  # sigmoid backward (tiny elementwise, B values)
                if kind == "disc":
                    need = any(ctx.needs[1:])
                    dp = ops.disc_backward(flat, ins[0], ins[1], ins[2], ins[3], d_logit, train, need_input=need,
                                           skip_param=skip, ws=ws)
                    din = [None, dp[:, :9] if need else None, dp[:, 9:12] if need else None,
                           dp[:, 12:] if need else None]
                    if ctx.needs[0]:
                        raise _lib.DggError("gradient w.r.t. bps_object through the discriminator is not provided")
                else:
                    need = any(ctx.needs[1:])
                    dp = ops.eva_backward(flat, ins[0], ins[1], ins[2], d_logit, train, need_input=need,
                                          skip_param=skip, ws=ws)
                    din = [None, dp[:, :9] if need else None, dp[:, 9:12] if need else None]
        finally:
            flat.grads = saved_grads
        pg = []
        if ctx.param_needs:
            views = {}
            for e in flat.entries:
                if e[4] == 0:
                    views[e[0]] = flat._shape_view(gbuf, e)
            pg = [views[e[0]] for e in flat.entries if e[4] == 0]
        else:
            pg = [None] * sum(1 for e in flat.entries if e[4] == 0)
        din = [d if n else None for d, n in zip(din, ctx.needs)]
        return (None, None, None, None, None, None, *din, *pg)


class Generator(_FlatBacked):
    """networks.py:58-103."""
    _net_kind = _lib.NET_GEN

    def __init__(self, cfg, n_neurons=512, in_bps=4096, dtype=torch.float32, **kwargs):
        super(Generator, self).__init__()
        self._init_flat()
        self.cfg = cfg
        self.latentD = cfg["latentD"]
        self.gen_bn1 = nn.BatchNorm1d(in_bps)
        self.gen_rb1 = ResBlock(self.latentD + in_bps, n_neurons)
        self.gen_rb2 = ResBlock(n_neurons + self.latentD + in_bps, n_neurons)
        self.gen_joint_conf = nn.Linear(n_neurons, cfg["n_hand_joints"])
        self.gen_rot = nn.Linear(n_neurons, 6)
        self.gen_transl = nn.Linear(n_neurons, 3)
        if self.cfg["is_train"]:
            print("Generator currently in TRAIN mode!")
            self.train()
        else:
            print("Generator currently in EVAL mode!")
            self.eval()
        self.dtype = dtype

    def forward(self, Zin, bps_object):
        flat = self._require_flat()
        nj = self.cfg["n_hand_joints"]
        out = _NetFn.apply(self, "gen", self.training, None, 0, 2, Zin, bps_object, *_param_list(self, flat))
        if self.training:
            self._bump_bn_counters()
        return {"rot_6D": out[:, nj:nj + 6], "transl": out[:, nj + 6:nj + 9], "joint_conf": out[:, :nj], "z": Zin}


class Discriminator(_FlatBacked):
    """networks.py:105-159."""
    _net_kind = _lib.NET_DISC

    def __init__(self, cfg, in_pose, n_neurons=512, in_bps=4096, dtype=torch.float32, **kwargs):
        super(Discriminator, self).__init__()
        self._init_flat()
        self.cfg = cfg
        self.disc_bn1 = nn.BatchNorm1d(in_bps + in_pose)
        self.disc_rb1 = ResBlock(in_bps + in_pose, n_neurons)
        self.disc_rb2 = ResBlock(n_neurons + in_bps + in_pose, n_neurons)
        self.out_success = nn.Linear(n_neurons, 1)
        self.sigmoid = nn.Sigmoid()
        if self.cfg["is_train"]:
            print("Discriminator currently in TRAIN mode!")
            self.train()
        else:
            print("Discriminator currently in EVAL mode!")
            self.eval()
        self.dtype = dtype

    def set_input(self, data):
        """networks.py:132-146."""
        rot_matrix = data["rot_matrix"].to(dtype=self.dtype, device=self.device)
        transl = data["transl"].to(dtype=self.dtype, device=self.device)
        joint_conf = data["joint_conf"].to(dtype=self.dtype, device=self.device)
        bps_object = data["bps_object"].to(dtype=self.dtype, device=self.device).contiguous()
        self.rot_matrix = rot_matrix.requires_grad_(self.cfg["is_train"]) if rot_matrix.is_leaf else rot_matrix
        self.transl = transl.requires_grad_(self.cfg["is_train"]) if transl.is_leaf else transl
        self.joint_conf = joint_conf.requires_grad_(self.cfg["is_train"]) if joint_conf.is_leaf else joint_conf
        self.bps_object = bps_object  # gradient w.r.t. the BPS encoding is never consumed
        self.rot_matrix = self.rot_matrix.view(self.bps_object.shape[0], -1)

    def forward(self, data):
        flat = self._require_flat()
        self.set_input(data)
        p = _NetFn.apply(self, "disc", self.training, None, 0, 4, self.bps_object, self.rot_matrix, self.transl,
                         self.joint_conf, *_param_list(self, flat))
        if self.training:
            self._bump_bn_counters()
        return p


class DexGANGrasp(nn.Module):
    """networks.py:161-276."""

    def __init__(self, cfg, n_neurons=512, in_bps=4096, in_pose=9 + 3, dtype=torch.float32, **kwargs):
        super(DexGANGrasp, self).__init__()
        self.cfg = cfg
        in_pose += cfg["n_hand_joints"]
        self.device = _default_device(cfg)
        self.latentD = cfg["latentD"]
        self.discriminator = Discriminator(cfg, in_pose, n_neurons, in_bps, dtype)
        self.discriminator.to(self.device)
        self.generator = Generator(cfg, n_neurons, in_bps, dtype)
        self.generator.to(self.device)
        self.discriminator.device = self.device
        self.generator.device = self.device
        if self.cfg["is_train"]:
            print("DexGANGrasp currently in TRAIN mode!")
            self.train()
        else:
            print("DexGANGrasp currently in EVAL mode!")
            self.eval()
        self.dtype = dtype

    def forward(self, Zin, bps_object, real_data=None):
        return self.generator(Zin, bps_object)

    def generate_poses(self, bps_object, return_arr=False, seed=None, sample_uniform=False, z_offset=0.0):
        """networks.py:223-255."""
        start = time.time()
        n_samples = bps_object.shape[0]
        self.eval()
        with torch.no_grad():
            if not sample_uniform:
                Zgen = torch.randn((n_samples, self.latentD), dtype=self.dtype, device=self.device)
            else:
                Zgen = 8 * torch.rand((n_samples, self.latentD), dtype=self.dtype, device=self.device) - 4
        results = self.generator(Zgen, bps_object)
        results['rot_matrix'] = utils.rot_matrix_from_ortho6d(results.pop('rot_6D'))
        if return_arr:
            for k, v in results.items():
                results[k] = v.cpu().detach().numpy()
        print("Sampling took: %.3f" % (time.time() - start))
        results = utils.translate_along_axis(results, 0, z_offset)
        return results

    def generate_grasps(self, bps, n_samples, return_arr=True, z_offset=0.025):
        """networks.py:257-276."""
        if len(bps.shape) > 1:
            bps = bps.squeeze()
        bps = np.tile(bps, (n_samples, 1))
        bps_tensor = torch.tensor(bps, dtype=self.dtype, device=self.device)
        return self.generate_poses(bps_tensor, return_arr=return_arr, z_offset=z_offset)


class DexEvaluator(_FlatBacked):
    """networks.py:278-338."""
    _net_kind = _lib.NET_EVA

    def __init__(self, cfg, n_neurons=512, in_bps=4096, in_pose=9 + 3, dtype=torch.float32, **kwargs):
        super(DexEvaluator, self).__init__()
        self._init_flat()
        self.cfg = cfg
        self.bn1 = nn.BatchNorm1d(in_bps + in_pose)
        self.rb1 = ResBlock(in_bps + in_pose, n_neurons)
        self.rb2 = ResBlock(in_bps + in_pose + n_neurons, n_neurons)
        self.rb3 = ResBlock(in_bps + in_pose + n_neurons, n_neurons)
        self.out_success = nn.Linear(n_neurons, 1)
        self.dout = nn.Dropout(0.3)
        self.sigmoid = nn.Sigmoid()
        if self.cfg["is_train"]:
            print("DexEvaluator currently in TRAIN mode!")
            self.train()
        else:
            print("DexEvaluator currently in EVAL mode!")
            self.eval()
        self.dtype = torch.float32
        self.device = _default_device(cfg)

    def set_input(self, data):
        """networks.py:307-313."""
        self.rot_matrix = data["rot_matrix"].to(dtype=self.dtype, device=self.device)
        self.transl = data["transl"].to(dtype=self.dtype, device=self.device)
        self.bps_object = data["bps_object"].to(dtype=self.dtype, device=self.device).contiguous()
        if 'label' in data.keys():
            self.gt_label = data["label"].to(dtype=self.dtype, device=self.device).unsqueeze(-1)
        self.rot_matrix = self.rot_matrix.view(self.bps_object.shape[0], -1)

    def forward(self, data):
        """networks.py:315-338.  Train-mode dropout masks come from Philox4x32-10 in the kernel,
        keyed by a seed drawn from torch's global generator (torch.manual_seed reproduces them);
        `self._drop_keep` (3,B,512 in {0,1}) overrides them for parity tests."""
        flat = self._require_flat()
        self.set_input(data)
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if (self.training and self._drop_keep is None) else 0
        p = _NetFn.apply(self, "eva", self.training, self._drop_keep if self.training else None, seed, 3,
                         self.bps_object, self.rot_matrix, self.transl, *_param_list(self, flat))
        if self.training:
            self._bump_bn_counters()
        return p
