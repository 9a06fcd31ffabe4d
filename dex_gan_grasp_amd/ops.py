"""Thin Python wrappers over the C-ABI (include/dgg.h).  Inputs are torch CUDA tensors (fp32,
contiguous — converted if not); outputs are freshly allocated torch tensors.  No arithmetic happens
here: every function is one or two library calls on the current HIP stream."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import check, lib, ptr, stream_ptr, f32c


def _dev(t):
    if not t.is_cuda:
        raise _lib.DggError("dex_gan_grasp_amd needs CUDA(HIP) tensors; got a %s tensor" % t.device)
    return t.device


# ------------------------------------------------------------------------------------------ A1
def bps_encode(clouds, basis, return_ids=True):
    """clouds (B,N,3) or (N,3), basis (P,3) -> dists (B,P) f32, ids (B,P) int32."""
    dev = _dev(clouds)
    x = f32c(clouds, dev)
    if x.dim() == 2:
        x = x.unsqueeze(0)
    b = f32c(basis, dev)
    B, N, _ = x.shape
    P = b.shape[0]
    dists = torch.empty((B, P), dtype=torch.float32, device=dev)
    ids = torch.empty((B, P), dtype=torch.int32, device=dev) if return_ids else None
    # few clouds of many points (the reference's call shape: one cloud of 15k-73k points) run chunk by chunk on the matrix
    # cores and meet in a workspace; the many-clouds shapes need none (0 bytes)
    nws = int(lib().dgg_bps_workspace_bytes(B, N, P))
    ws = torch.empty((nws,), dtype=torch.uint8, device=dev) if nws > 0 else None
    check(lib().dgg_bps_encode_ws(ptr(x), B, N, ptr(b), P, ptr(dists), ptr(ids), ptr(ws), nws, stream_ptr(dev)),
          "dgg_bps_encode_ws")
    return (dists, ids) if return_ids else dists


# ------------------------------------------------------------------------------------------ GEMM (tests)
def gemm(mode, A, Bm, M, N, K, splits=0, C_out=None, accumulate=False):
    dev = _dev(A)
    out = C_out if C_out is not None else torch.empty((M, N), dtype=torch.float32, device=dev)
    slabs = torch.empty((max(splits, 1) * M * N,), dtype=torch.float32, device=dev) if splits >= 1 else None
    check(lib().dgg_gemm_f32(mode, M, N, K, ptr(A), A.stride(0), ptr(Bm), Bm.stride(0), ptr(out), out.stride(0),
                             splits, ptr(slabs), 1 if accumulate else 0, stream_ptr(dev)), "dgg_gemm_f32")
    return out


# ------------------------------------------------------------------------------------------ networks
def gen_forward(fn, z, bps, train, ws=None):
    dev = fn.device
    z, bps = f32c(z, dev), f32c(bps, dev)
    B = bps.shape[0]
    ws = ws if ws is not None else fn.workspace(B)
    out = torch.empty((B, fn.ccfg.n_hand_joints + 9), dtype=torch.float32, device=dev)
    check(lib().dgg_gen_forward(C.byref(fn.ccfg), ptr(fn.params), ptr(fn.bn), ptr(ws), ws.numel(), B,
                                _lib.TRAIN if train else 0, ptr(z), ptr(bps), ptr(out), stream_ptr(dev)), "dgg_gen_forward")
    return out


def gen_backward(fn, z, bps, d_heads, train, want_dz=False, want_dbps=False, accumulate=False, ws=None):
    dev = fn.device
    z, bps, d_heads = f32c(z, dev), f32c(bps, dev), f32c(d_heads, dev)
    B = bps.shape[0]
    ws = ws if ws is not None else fn.workspace(B)
    dz = torch.empty((B, fn.ccfg.latent_d), dtype=torch.float32, device=dev) if want_dz else None
    dbps = torch.empty((B, fn.ccfg.in_bps), dtype=torch.float32, device=dev) if want_dbps else None
    flags = (_lib.TRAIN if train else 0) | (_lib.ACCUM_GRADS if accumulate else 0)
    check(lib().dgg_gen_backward(C.byref(fn.ccfg), ptr(fn.params), ptr(fn.grads), ptr(ws), ws.numel(), B, flags,
                                 ptr(z), ptr(bps), ptr(d_heads), ptr(dz), ptr(dbps), stream_ptr(dev)), "dgg_gen_backward")
    return dz, dbps


def disc_forward(fn, bps, rot9, transl, joint, train, ws=None):
    dev = fn.device
    bps, rot9, transl, joint = f32c(bps, dev), f32c(rot9, dev).reshape(-1, 9), f32c(transl, dev), f32c(joint, dev)
    B = bps.shape[0]
    ws = ws if ws is not None else fn.workspace(B)
    out = torch.empty((B, 1), dtype=torch.float32, device=dev)
    check(lib().dgg_disc_forward(C.byref(fn.ccfg), ptr(fn.params), ptr(fn.bn), ptr(ws), ws.numel(), B,
                                 _lib.TRAIN if train else 0, ptr(bps), ptr(rot9), ptr(transl), ptr(joint), ptr(out),
                                 stream_ptr(dev)), "dgg_disc_forward")
    return out


def disc_backward(fn, bps, rot9, transl, joint, d_logit, train, need_input=False, skip_param=False, accumulate=False,
                  ws=None):
    dev = fn.device
    bps, rot9, transl, joint = f32c(bps, dev), f32c(rot9, dev).reshape(-1, 9), f32c(transl, dev), f32c(joint, dev)
    d_logit = f32c(d_logit, dev)
    B = bps.shape[0]
    ws = ws if ws is not None else fn.workspace(B)
    d_pose = torch.empty((B, 12 + fn.ccfg.n_hand_joints), dtype=torch.float32, device=dev) if need_input else None
    flags = (_lib.TRAIN if train else 0) | (_lib.ACCUM_GRADS if accumulate else 0) | \
            (_lib.SKIP_PARAM_GRADS if skip_param else 0) | (_lib.NEED_INPUT_GRADS if need_input else 0)
    check(lib().dgg_disc_backward(C.byref(fn.ccfg), ptr(fn.params), None if skip_param else ptr(fn.grads), ptr(ws),
                                  ws.numel(), B, flags, ptr(bps), ptr(rot9), ptr(transl), ptr(joint), ptr(d_logit),
                                  ptr(d_pose), stream_ptr(dev)), "dgg_disc_backward")
    return d_pose


def eva_forward(fn, bps, rot9, transl, train, drop_keep=None, seed=0, ws=None):
    dev = fn.device
    bps, rot9, transl = f32c(bps, dev), f32c(rot9, dev).reshape(-1, 9), f32c(transl, dev)
    B = bps.shape[0]
    ws = ws if ws is not None else fn.workspace(B)
    dk = f32c(drop_keep, dev) if drop_keep is not None else None
    out = torch.empty((B, 1), dtype=torch.float32, device=dev)
    check(lib().dgg_eva_forward(C.byref(fn.ccfg), ptr(fn.params), ptr(fn.bn), ptr(ws), ws.numel(), B,
                                _lib.TRAIN if train else 0, ptr(bps), ptr(rot9), ptr(transl), ptr(dk), int(seed),
                                ptr(out), stream_ptr(dev)), "dgg_eva_forward")
    return out


def eva_backward(fn, bps, rot9, transl, d_logit, train, need_input=False, skip_param=False, accumulate=False, ws=None):
    dev = fn.device
    bps, rot9, transl = f32c(bps, dev), f32c(rot9, dev).reshape(-1, 9), f32c(transl, dev)
    d_logit = f32c(d_logit, dev)
    B = bps.shape[0]
    ws = ws if ws is not None else fn.workspace(B)
    d_pose = torch.empty((B, 12), dtype=torch.float32, device=dev) if need_input else None
    flags = (_lib.TRAIN if train else 0) | (_lib.ACCUM_GRADS if accumulate else 0) | \
            (_lib.SKIP_PARAM_GRADS if skip_param else 0) | (_lib.NEED_INPUT_GRADS if need_input else 0)
    check(lib().dgg_eva_backward(C.byref(fn.ccfg), ptr(fn.params), None if skip_param else ptr(fn.grads), ptr(ws),
                                 ws.numel(), B, flags, ptr(bps), ptr(rot9), ptr(transl), ptr(d_logit), ptr(d_pose),
                                 stream_ptr(dev)), "dgg_eva_backward")
    return d_pose


# ------------------------------------------------------------------------------------------ config 5
def _objects_ws(fn, n_obj, n_per):
    key = ("objects", n_obj, n_per)
    ws = fn._ws.get(key)
    if ws is None:
        nb = int(lib().dgg_objects_workspace_bytes(fn.net, C.byref(fn.ccfg), n_obj, n_per))
        if nb < 0:
            raise _lib.DggError("dgg_objects_workspace_bytes: %d" % nb)
        ws = _lib.new_workspace(nb + 4096, fn.device)
        fn._ws[key] = ws
    return ws


def gen_forward_objects(fn, z, bps_obj, n_per):
    """Eval-mode generator over O objects x n_per samples, factorised per object (dgg_gen_forward_objects).
    z (O*n_per, latent), bps_obj (O, 4096) -> heads (O*n_per, nj+9)."""
    dev = fn.device
    z, bps_obj = f32c(z, dev), f32c(bps_obj, dev)
    n_obj = bps_obj.shape[0]
    if z.shape[0] != n_obj * n_per:
        raise _lib.DggError("gen_forward_objects: z has %d rows, expected %d" % (z.shape[0], n_obj * n_per))
    ws = _objects_ws(fn, n_obj, n_per)
    out = torch.empty((n_obj * n_per, fn.ccfg.n_hand_joints + 9), dtype=torch.float32, device=dev)
    check(lib().dgg_gen_forward_objects(C.byref(fn.ccfg), ptr(fn.params), ptr(fn.bn), ptr(ws), ws.numel(), n_obj, n_per,
                                        ptr(z), ptr(bps_obj), ptr(out), stream_ptr(dev)), "dgg_gen_forward_objects")
    return out


def eva_forward_objects(fn, bps_obj, rot9, transl, n_per):
    """Eval-mode evaluator over O objects x n_per samples, factorised per object -> p (O*n_per, 1)."""
    dev = fn.device
    bps_obj, rot9, transl = f32c(bps_obj, dev), f32c(rot9, dev).reshape(-1, 9), f32c(transl, dev)
    n_obj = bps_obj.shape[0]
    if rot9.shape[0] != n_obj * n_per or transl.shape[0] != n_obj * n_per:
        raise _lib.DggError("eva_forward_objects: per-sample inputs need %d rows" % (n_obj * n_per))
    ws = _objects_ws(fn, n_obj, n_per)
    out = torch.empty((n_obj * n_per, 1), dtype=torch.float32, device=dev)
    check(lib().dgg_eva_forward_objects(C.byref(fn.ccfg), ptr(fn.params), ptr(fn.bn), ptr(ws), ws.numel(), n_obj, n_per,
                                        ptr(bps_obj), ptr(rot9), ptr(transl), ptr(out), stream_ptr(dev)),
          "dgg_eva_forward_objects")
    return out


# ------------------------------------------------------------------------------------------ small ops
def ortho6d_forward(o6):
    dev = _dev(o6)
    o6 = f32c(o6, dev)
    B = o6.shape[0]
    r = torch.empty((B, 9), dtype=torch.float32, device=dev)
    check(lib().dgg_ortho6d_forward(ptr(o6), o6.stride(0), B, ptr(r), stream_ptr(dev)), "dgg_ortho6d_forward")
    return r.view(B, 3, 3)


def ortho6d_backward(o6, d_rot):
    dev = _dev(o6)
    o6, d_rot = f32c(o6, dev), f32c(d_rot, dev).reshape(-1, 9)
    B = o6.shape[0]
    d = torch.empty((B, 6), dtype=torch.float32, device=dev)
    check(lib().dgg_ortho6d_backward(ptr(o6), o6.stride(0), B, ptr(d_rot), ptr(d), 6, stream_ptr(dev)),
          "dgg_ortho6d_backward")
    return d


def bce(p, target, weight=1.0, loss_scale=1.0, want_grad=True):
    """p (B,1) probabilities; target: tensor (B,) / (B,1) of labels or a python float constant.
    Returns (loss 0-d tensor, d_logit (B,1) or None)."""
    dev = _dev(p)
    p = f32c(p, dev)
    B = p.numel()
    labels, const = (f32c(target, dev).reshape(-1), 0.0) if torch.is_tensor(target) else (None, float(target))
    loss = torch.empty((1,), dtype=torch.float32, device=dev)
    dl = torch.empty((B, 1), dtype=torch.float32, device=dev) if want_grad else None
    check(lib().dgg_bce(ptr(p), ptr(labels), const, B, float(weight), float(loss_scale), ptr(loss), ptr(dl),
                        stream_ptr(dev)), "dgg_bce")
    return loss[0], dl


def translate_along_axis(transl, rot9, axis, offset):
    dev = _dev(transl)
    if not transl.is_contiguous() or transl.dtype != torch.float32:
        raise _lib.DggError("translate_along_axis works in place on a contiguous float32 (N, 3) tensor")
    check(lib().dgg_translate_along_axis(ptr(transl), ptr(f32c(rot9, dev)), transl.shape[0], int(axis), float(offset),
                                         stream_ptr(dev)), "dgg_translate_along_axis")
    return transl


def make_hparams(cfg, grad_scale=1.0):
    return _lib.HParams(float(cfg.get("lr_gen", 1e-4)), float(cfg.get("lr_dis", 1e-4)), float(cfg.get("lr", 1e-4)),
                        float(cfg.get("beta1", 0.9)), 0.999, 1e-8, float(cfg.get("weight_decay", 5e-4)),
                        float(cfg.get("bce_weight", 1)), 100.0, 1.0, 10.0, float(grad_scale))
