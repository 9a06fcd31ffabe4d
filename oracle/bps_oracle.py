"""CPU ORACLE — test infrastructure, NOT product code.

Python face of oracle/bps_oracle.c plus a numpy restatement for small cases, and the
pre/post-processing of BPSEncoder.encode_pcd_with_bps (DexGanGrasp/data/bps_encoder.py:28-55):
float32 cast, distances only, `min < 0.1` centring assert, squeeze to (4096,).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "libbps_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
    return _LIB


def _c(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def encode_f32(cloud, basis):
    """(N,3) f32 cloud, (P,3) f32 basis -> dists (P,) f32, ids (P,) int32 (C, fixed fp32 contract)."""
    cloud = np.ascontiguousarray(cloud, np.float32)
    basis = np.ascontiguousarray(basis, np.float32)
    p = basis.shape[0]
    d = np.empty(p, np.float32)
    i = np.empty(p, np.int32)
    _lib().bps_encode_f32(_c(cloud, ctypes.c_float), cloud.shape[0], _c(basis, ctypes.c_float), p,
                          _c(d, ctypes.c_float), _c(i, ctypes.c_int32))
    return d, i


def encode_f64(cloud, basis):
    """fp64 brute force: dists f64, ids, second-best ids, relative d^2 gap best vs second."""
    cloud = np.ascontiguousarray(cloud, np.float32)
    basis = np.ascontiguousarray(basis, np.float32)
    p = basis.shape[0]
    d = np.empty(p, np.float64)
    i = np.empty(p, np.int32)
    i2 = np.empty(p, np.int32)
    gap = np.empty(p, np.float64)
    _lib().bps_encode_f64(_c(cloud, ctypes.c_float), cloud.shape[0], _c(basis, ctypes.c_float), p,
                          _c(d, ctypes.c_double), _c(i, ctypes.c_int32), _c(i2, ctypes.c_int32),
                          _c(gap, ctypes.c_double))
    return d, i, i2, gap


def encode_numpy(cloud, basis):
    """Small-case numpy restatement (O(P*N) memory): same formula, numpy fp32 (no fma)."""
    cloud = np.asarray(cloud, np.float32)
    basis = np.asarray(basis, np.float32)
    delta = cloud[None, :, :] - basis[:, None, :]
    d2 = (delta * delta).sum(-1)
    ids = d2.argmin(1).astype(np.int32)
    sel = cloud[ids] - basis
    return np.sqrt((sel * sel).sum(-1)).astype(np.float32), ids


def encode_pcd_with_bps(points, basis):
    """DexGanGrasp/data/bps_encoder.py:28-55: points (N,3) any float dtype (Open3D gives f64)."""
    d, _ = encode_f32(np.asarray(points, np.float32), basis)
    enc = d[None, :]
    assert enc.min() < 0.1, "The pcd might not be in centered in origin !"
    return np.squeeze(enc)
