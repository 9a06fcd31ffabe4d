"""ctypes binding of libdgg.so (include/dgg.h) + flat-buffer helpers.

PyTorch is used only for device memory and streams: every call passes raw device pointers
(`tensor.data_ptr()`) and the current HIP stream.  There is NO fallback: if the shared library is
missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os
from collections import OrderedDict

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DGG_LIB", os.path.join(_HERE, "libdgg.so"))  # DGG_LIB: diagnostic builds only

NET_GEN, NET_DISC, NET_EVA = 0, 1, 2
TRAIN, ACCUM_GRADS, SKIP_PARAM_GRADS, NEED_INPUT_GRADS = 1, 2, 4, 8
# loss slots (include/dgg.h)
L_TOTAL_DISC, L_BCE_REAL, L_BCE_FAKE, L_GEN_FAKE, L_TRANSL, L_ROT, L_CONF, L_TOTAL_GEN, L_TOTAL_EVA, L_BCE_EVA = range(10)
L_GRAD_PENALTY = 10


class DggError(RuntimeError):
    pass


class Cfg(C.Structure):
    _fields_ = [("latent_d", C.c_int32), ("n_hand_joints", C.c_int32), ("in_bps", C.c_int32),
                ("n_neurons", C.c_int32), ("hidden", C.c_int32)]


class Entry(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("offset", C.c_int64), ("rows", C.c_int32), ("cols", C.c_int32),
                ("kind", C.c_int32), ("pad_", C.c_int32)]


class HParams(C.Structure):
    _fields_ = [("lr_gen", C.c_double), ("lr_dis", C.c_double), ("lr_eva", C.c_double), ("beta1", C.c_double),
                ("beta2", C.c_double), ("eps", C.c_double), ("weight_decay", C.c_double), ("bce_weight", C.c_double),
                ("transl_coef", C.c_double), ("rot_coef", C.c_double), ("conf_coef", C.c_double),
                ("grad_scale", C.c_double)]


class GanState(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("g_params", "g_grads", "g_m", "g_v", "g_bn", "d_params", "d_grads", "d_m", "d_v", "d_bn")]


_lib = None

_VP, _I, _U, _F, _SZ, _I64, _U64 = C.c_void_p, C.c_int, C.c_uint, C.c_float, C.c_size_t, C.c_int64, C.c_uint64
_D = C.c_double
_CFG = C.POINTER(Cfg)

# name -> (restype, argtypes); must list every symbol include/dgg.h declares (tests check it)
SIGNATURES = {
    "dgg_version": (C.c_char_p, []),
    "dgg_last_error": (C.c_char_p, []),
    "dgg_table": (_I, [_I, _CFG, C.POINTER(Entry), _I]),
    "dgg_param_floats": (_I64, [_I, _CFG]),
    "dgg_bnstat_floats": (_I64, [_I, _CFG]),
    "dgg_workspace_bytes": (_I64, [_I, _CFG, _I]),
    "dgg_bps_encode": (_I, [_VP, _I, _I, _VP, _I, _VP, _VP, _VP]),
    "dgg_bps_workspace_bytes": (_I64, [_I, _I, _I]),
    "dgg_bps_encode_ws": (_I, [_VP, _I, _I, _VP, _I, _VP, _VP, _VP, _SZ, _VP]),
    "dgg_set_bps_algo": (_I, [_I]),
    "dgg_get_bps_algo": (_I, []),
    "dgg_gen_forward": (_I, [_CFG, _VP, _VP, _VP, _SZ, _I, _U, _VP, _VP, _VP, _VP]),
    "dgg_gen_backward": (_I, [_CFG, _VP, _VP, _VP, _SZ, _I, _U, _VP, _VP, _VP, _VP, _VP, _VP]),
    "dgg_disc_forward": (_I, [_CFG, _VP, _VP, _VP, _SZ, _I, _U, _VP, _VP, _VP, _VP, _VP, _VP]),
    "dgg_disc_backward": (_I, [_CFG, _VP, _VP, _VP, _SZ, _I, _U, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "dgg_eva_forward": (_I, [_CFG, _VP, _VP, _VP, _SZ, _I, _U, _VP, _VP, _VP, _VP, _U64, _VP, _VP]),
    "dgg_eva_backward": (_I, [_CFG, _VP, _VP, _VP, _SZ, _I, _U, _VP, _VP, _VP, _VP, _VP, _VP]),
    "dgg_ortho6d_forward": (_I, [_VP, _I, _I, _VP, _VP]),
    "dgg_ortho6d_backward": (_I, [_VP, _I, _I, _VP, _VP, _I, _VP]),
    "dgg_bce": (_I, [_VP, _VP, _F, _I, _F, _F, _VP, _VP, _VP]),
    "dgg_adam_l2": (_I, [_VP, _VP, _VP, _VP, _I64, _I, _D, _D, _D, _D, _D, _D, _VP]),
    "dgg_cgan_workspace_bytes": (_I64, [_CFG, _I]),
    "dgg_cgan_d_phase": (_I, [_CFG, C.POINTER(GanState), C.POINTER(HParams), _VP, _SZ, _I, _VP, _VP, _VP, _VP, _VP,
                              _VP, _VP]),
    "dgg_set_wgrad_stream": (_I, [_VP, _VP]),
    "dgg_enable_wgrad_stream": (_I, [_VP, _I]),
    "dgg_set_gemm_algo": (_I, [_I]),
    "dgg_get_gemm_algo": (_I, []),
    "dgg_cgan_d_phase2": (_I, [_CFG, C.POINTER(GanState), C.POINTER(HParams), _VP, _SZ, _I, _VP, _VP, _VP, _VP, _VP,
                               _VP, _VP, _VP]),
    "dgg_cgan_d_real_prefetch": (_I, [_CFG, C.POINTER(GanState), C.POINTER(HParams), _VP, _SZ, _I, _VP, _VP, _VP, _VP,
                                      _VP, _VP]),
    "dgg_cgan_g_phase": (_I, [_CFG, C.POINTER(GanState), C.POINTER(HParams), _VP, _SZ, _I, _VP, _VP, _VP, _VP, _VP,
                              _VP, _VP]),
    "dgg_cgan_step": (_I, [_CFG, C.POINTER(GanState), C.POINTER(HParams), _VP, _SZ, _I, _VP, _VP, _VP, _VP, _VP, _I,
                           _I, _I, _VP, _VP]),
    "dgg_eva_phase": (_I, [_CFG, _VP, _VP, _VP, C.POINTER(HParams), _VP, _SZ, _I, _VP, _VP, _VP, _VP, _VP, _U64, _VP,
                           _VP]),
    "dgg_gen_loss": (_I, [_CFG, C.POINTER(HParams), _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "dgg_translate_along_axis": (_I, [_VP, _VP, _I, _I, _F, _VP]),
    "dgg_gemm_occupancy": (_I, [_I, _I, _I]),
    "dgg_objects_workspace_bytes": (_I64, [_I, _CFG, _I, _I]),
    "dgg_gen_forward_objects": (_I, [_CFG, _VP, _VP, _VP, _SZ, _I, _I, _VP, _VP, _VP, _VP]),
    "dgg_eva_forward_objects": (_I, [_CFG, _VP, _VP, _VP, _SZ, _I, _I, _VP, _VP, _VP, _VP, _VP]),
    "dgg_wgan_workspace_bytes": (_I64, [_CFG, _I]),
    "dgg_wgan_d_phase": (_I, [_CFG, C.POINTER(GanState), C.POINTER(HParams), _VP, _SZ, _I, _VP, _VP, _VP, _VP, _VP, _VP,
                              _D, _VP, _VP]),
    "dgg_wgan_g_phase": (_I, [_CFG, C.POINTER(GanState), C.POINTER(HParams), _VP, _SZ, _I, _VP, _VP, _VP, _VP, _VP, _VP,
                              _VP]),
    "dgg_adam_l2_graph": (_I, [_VP, _VP, _VP, _VP, _I64, _VP, _VP, _D, _D, _D, _D, _D, _D, _VP]),
    "dgg_eva_phase_graph": (_I, [_CFG, _VP, _VP, _VP, C.POINTER(HParams), _VP, _SZ, _I, _VP, _VP, _VP, _VP, _VP, _VP,
                                 _VP]),
    "dgg_cgan_release": (_I, [_VP]),
    "dgg_cgan_drop_prefetch": (_I, [_VP, _VP]),
    "dgg_reset": (_I, []),
    "dgg_eva_keep_offset": (_I64, [_CFG, _I, _I]),
    "dgg_ws_offset": (_I64, [_I, _CFG, _I, _I, _I]),
    "dgg_cgan_ws_offset": (_I64, [_CFG, _I, _I]),
    "dgg_resblock_param_floats": (_I64, [_I]),
    "dgg_resblock_bnstat_floats": (_I64, [_I]),
    "dgg_resblock_table": (_I, [_I, C.POINTER(Entry), _I]),
    "dgg_resblock_workspace_bytes": (_I64, [_I, _I]),
    "dgg_resblock_forward": (_I, [_I, _VP, _VP, _VP, _SZ, _I, _U, _VP, _I, _VP, _VP]),
    "dgg_resblock_backward": (_I, [_I, _VP, _VP, _VP, _SZ, _I, _U, _VP, _I, _VP, _VP, _VP]),
    "dgg_gemm_f32": (_I, [_I, _I, _I, _I, _VP, _I, _VP, _I, _VP, _I, _I, _VP, _I, _VP]),
    "dgg_gemm_route": (_I, [_I, _I, _I, _I, _I, _I, _I, _I]),
    "dgg_set_grad_events": (_I, [_VP, _VP, _I, _I]),
    "dgg_set_side_work": (_I, [_I]),
}


def lib():
    """The loaded libdgg.so.  Raises if it has not been built (no CPU / eager fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DggError("libdgg.so not found at %s — build it with dex_gan_grasp_amd/csrc/build.sh "
                           "(or python -c 'import __graft_entry__ as g; g.build()')" % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc, what=""):
    if rc != 0:
        raise DggError("%s failed: status %d (%s)" % (what or "dgg call", rc, lib().dgg_last_error().decode()))


def make_cfg(cfg=None):
    cfg = cfg or {}
    return Cfg(int(cfg.get("latentD", 5)), int(cfg.get("n_hand_joints", 12)), int(cfg.get("in_bps", 4096)),
               int(cfg.get("n_neurons", 512)), int(cfg.get("hidden", 256)))


def ptr(t):
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    """The current HIP stream of `device` (default: the current device) as a void*."""
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def table(net, ccfg):
    n = lib().dgg_table(net, C.byref(ccfg), None, 0)
    if n < 0:
        raise DggError("dgg_table: %d" % n)
    arr = (Entry * n)()
    lib().dgg_table(net, C.byref(ccfg), arr, n)
    return [(e.name.decode(), int(e.offset), int(e.rows), int(e.cols), int(e.kind)) for e in arr]


def f32c(t, device):
    """float32 contiguous tensor on `device` (no copy if already so)."""
    return torch.as_tensor(t).to(device=device, dtype=torch.float32).contiguous()


def new_workspace(nbytes, device):
    """Caller-owned scratch for the C-ABI.  DGG_POISON_WS=1 (tests): fill with 0xFF bytes (= NaN floats) so that
    any kernel reading workspace memory it has not written shows up as NaN instead of depending on leftovers."""
    if os.environ.get("DGG_POISON_WS") == "1":
        return torch.full((nbytes,), 255, dtype=torch.uint8, device=device)
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def kink_sides(net, ccfg, ws, B, base=0):
    """Which side of every LeakyReLU kink (networks.py:41) the last train-mode pass of network `net` took, read back from
    the activations it left in its workspace `ws` (uint8 tensor; `base`: byte offset of the network workspace inside it,
    dgg_cgan_ws_offset): per ResBlock three bool tensors [fc3 branch (B,512), bn1 branch (B,256), block output (B,512)],
    True = positive side, in the order a CPU evaluation of the same pass meets its LeakyReLUs.  The evaluator's block
    outputs are int8 instead: 1 / 0 = positive / negative side, -1 = the unit was DROPPED (the workspace holds 0 there and
    the sign is gone; its gradient is zero either way) — a checker that follows these sides must leave such elements alone
    (oracle.dexgan_oracle.forced_slopes does, and does not count them as steering)."""
    nrb = 3 if net == NET_EVA else 2
    out = []
    for r in range(nrb):
        offs = [int(lib().dgg_ws_offset(net, C.byref(ccfg), B, r, w)) for w in (0, 1, 2)]
        if base < 0 or min(offs) < 0:
            raise DggError("dgg_ws_offset / dgg_cgan_ws_offset: bad arguments")
        f = lambda off, n: ws[base + off: base + off + 4 * B * n].view(torch.float32).view(B, n)
        y = f(offs[2], 512) > 0
        if net == NET_EVA:
            ko = int(lib().dgg_eva_keep_offset(C.byref(ccfg), B, r))
            if ko < 0:
                raise DggError("dgg_eva_keep_offset: %d" % ko)
            keep = ws[base + ko: base + ko + B * 512].view(B, 512)
            y = torch.where(keep != 0, y.to(torch.int8), torch.full_like(keep, -1, dtype=torch.int8))
        out += [f(offs[0], 768)[:, 256:] > 0, f(offs[1], 256) > 0, y]
    return out


class FlatNet:
    """Flat device buffers of one network in the library's layout (dgg_table) with named views.

    params / grads / adam m,v share one layout; bn holds the running statistics.  The views are
    what the nn.Module mirror registers as Parameters / buffers, so state_dict() keys and shapes are
    the reference's while the kernels see one contiguous buffer per network (single fused Adam
    launch, single all-reduce bucket)."""

    def __init__(self, net, cfg=None, device="cuda", with_optim=True):
        self.net = net
        self.ccfg = make_cfg(cfg)
        self.device = torch.device(device)
        self.entries = table(net, self.ccfg)
        self.n_params = int(lib().dgg_param_floats(net, C.byref(self.ccfg)))
        self.n_bn = int(lib().dgg_bnstat_floats(net, C.byref(self.ccfg)))
        z = lambda n: torch.zeros(n, dtype=torch.float32, device=self.device)
        self.params = z(self.n_params)
        self.bn = z(self.n_bn)
        self.grads = z(self.n_params)
        self.m = z(self.n_params) if with_optim else None
        self.v = z(self.n_params) if with_optim else None
        self._ws = {}
        self._step_dev = None   # graph mode: the step number in device memory (int64) + 2 floats of scratch
        self._dyn = None
        self._step = 0

    @property
    def step(self):
        """Adam's step number.  Assigning it (optimiser state loaded from a checkpoint, replicas synchronised) also
        refreshes the device-resident copy the graph path reads (dgg_adam_l2_graph)."""
        return self._step

    @step.setter
    def step(self, v):
        self._step = int(v)
        if self._step_dev is not None:
            self._step_dev.fill_(self._step)

    def _shape_view(self, buf, e):
        name, off, rows, cols, kind = e
        v = buf[off:off + rows * cols]
        # the table gives vectors as (n, 1) and matrices as (rows, cols>1); a (1,512) head stays 2-D
        return v.view(rows, cols) if cols > 1 else v.view(rows)

    def views(self, which="params"):
        buf_p = {"params": self.params, "grads": self.grads, "m": self.m, "v": self.v}.get(which)
        out = OrderedDict()
        for e in self.entries:
            if e[4] == 0 and which != "bn":
                out[e[0]] = self._shape_view(buf_p, e)
            elif e[4] == 1 and which == "bn":
                out[e[0]] = self._shape_view(self.bn, e)
        return out

    def load_state_dict(self, sd, prefix=""):
        pv, bv = self.views("params"), self.views("bn")
        with torch.no_grad():
            for name, v in list(pv.items()) + list(bv.items()):
                v.copy_(sd[prefix + name].to(self.device, torch.float32))

    def state_dict(self, prefix=""):
        out = OrderedDict()
        for name, v in list(self.views("params").items()) + list(self.views("bn").items()):
            out[prefix + name] = v.detach().clone().cpu()
        return out

    def grad_buckets(self):
        """[(lo, hi)] float ranges of the flat gradient buffer in the order a backward completes them
        (dgg_set_grad_events): [last block + heads], .., [block 1], [input BatchNorm + block 0]."""
        starts = sorted(e[1] for e in self.entries if e[4] == 0 and e[0].endswith("fc1.weight"))
        bounds = [(starts[r], starts[r + 1] if r + 1 < len(starts) else self.n_params) for r in range(len(starts))]
        bounds[0] = (0, bounds[0][1])
        return [bounds[r] for r in range(len(starts) - 1, -1, -1)]

    def workspace(self, batch):
        """Cached byte workspace for a batch size (caller-owned memory of the C-ABI)."""
        ws = self._ws.get(batch)
        if ws is None:
            n = int(lib().dgg_workspace_bytes(self.net, C.byref(self.ccfg), batch))
            if n < 0:
                raise DggError("dgg_workspace_bytes: %d" % n)
            ws = new_workspace(n + 4096 + 8 * batch, self.device)
            self._ws[batch] = ws
        return ws

    def adam_step_graph(self, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=5e-4, grad_scale=1.0):
        """dgg_adam_l2_graph: the step number lives on the device (capturable).  The host counter is NOT advanced here:
        the caller adds the number of replays (TrainEngine.GraphStep)."""
        if self._step_dev is None:
            self._step_dev = torch.tensor([self.step], dtype=torch.int64, device=self.device)
            self._dyn = torch.zeros(2, dtype=torch.float32, device=self.device)
        check(lib().dgg_adam_l2_graph(ptr(self.params), ptr(self.grads), ptr(self.m), ptr(self.v), self.n_params,
                                      ptr(self._step_dev), ptr(self._dyn), lr, beta1, beta2, eps, weight_decay,
                                      grad_scale, stream_ptr(self.device)), "dgg_adam_l2_graph")

    def adam_begin(self):
        """Start an optimiser step that is applied range by range (adam_range): advances the step number once."""
        self.step += 1

    def adam_range(self, lo, hi, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=5e-4, grad_scale=1.0):
        """dgg_adam_l2 on the float range [lo, hi) of the flat buffers (bucket bounds are multiples of 4), on the current
        stream, with the step number adam_begin set."""
        check(lib().dgg_adam_l2(ptr(self.params[lo:hi]), ptr(self.grads[lo:hi]), ptr(self.m[lo:hi]), ptr(self.v[lo:hi]),
                                hi - lo, self.step, lr, beta1, beta2, eps, weight_decay, grad_scale,
                                stream_ptr(self.device)), "dgg_adam_l2")

    def adam_step(self, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=5e-4, grad_scale=1.0):
        self.step += 1      # (the setter keeps the device-resident counter of the graph path in step)
        check(lib().dgg_adam_l2(ptr(self.params), ptr(self.grads), ptr(self.m), ptr(self.v), self.n_params, self.step,
                                lr, beta1, beta2, eps, weight_decay, grad_scale, stream_ptr(self.device)), "dgg_adam_l2")
