"""Mirror of the math subset of DexGanGrasp/utils/utils.py that lies on the hot path
(SURVEY.md 2, row 4): rot_matrix_from_ortho6d (:380-394; its helpers normalize_vector :323-332 and
cross_product :68-78 live inside the dgg_ortho6d_* kernels), translate_along_axis (:19-48),
class_labels_from_logits (:60-64), dict_to_tensor (:114-121), data_dict_to_dtype (:106-111).
Device math runs in libdgg.so."""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib, ops


class _Ortho6dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, o6):
        o6c = _lib.f32c(o6, o6.device)
        ctx.save_for_backward(o6c)
        return ops.ortho6d_forward(o6c)

    @staticmethod
    def backward(ctx, g):
        (o6c,) = ctx.saved_tensors
        return ops.ortho6d_backward(o6c, g.contiguous())


def rot_matrix_from_ortho6d(ortho6d, dtype=torch.float32):
    """utils/utils.py:380-394: (B,6) -> (B,3,3), columns x, y, z (Gram-Schmidt).  HIP kernel
    dgg_ortho6d_forward/backward; differentiable."""
    if not ortho6d.is_cuda:
        raise _lib.DggError("rot_matrix_from_ortho6d: GPU tensor required (no CPU path)")
    return _Ortho6dFn.apply(ortho6d)


def translate_along_axis(out, axis, translation):
    """utils/utils.py:19-48: transl[i] <- transl[i] - R[i] @ e_axis * translation.
    numpy arrays: vectorised float64 evaluation stored back into the array (what the reference's
    per-sample loop computes); CUDA tensors: dgg_translate_along_axis in place."""
    t, r = out['transl'], out['rot_matrix']
    if torch.is_tensor(t):
        if t.is_cuda:
            # the kernel works on a dense (N, 3) array; the generator's `transl` is a strided view of its heads
            tc = t if t.is_contiguous() else t.contiguous()
            ops.translate_along_axis(tc, r.reshape(-1, 9), axis, translation)
            out['transl'] = tc
        else:
            t -= (r[:, :, axis] * translation).to(t.dtype)
        return out
    off = np.zeros(3)
    off[axis] = translation
    t[...] = (np.asarray(t, np.float64) - np.einsum('nij,j->ni', np.asarray(r, np.float64), off)).astype(t.dtype)
    return out


def class_labels_from_logits(logits, threshold):
    """utils/utils.py:60-64 (in place, like the reference)."""
    logits[logits >= threshold] = 1.
    logits[logits < threshold] = 0.
    return logits


def data_dict_to_dtype(data, dtype):
    """utils/utils.py:106-111."""
    for k, v in data.items():
        if not isinstance(v, list):
            data[k] = v.to(dtype)
    return data


def dict_to_tensor(in_dict, device, dtype):
    """utils/utils.py:114-121."""
    out_dict = {}
    for k, v in in_dict.items():
        if torch.is_tensor(v):
            out_dict[k] = v.clone().detach()
        else:
            out_dict[k] = torch.tensor(v, dtype=dtype, device=device)
    return out_dict


def grasp_numpy_from_data_dict(data):
    """utils.py:199-204."""
    return {
        "rot_matrix": data["rot_matrix"].cpu().data.numpy(),
        "transl": data["transl"].cpu().data.numpy(),
        "joint_conf": data["joint_conf"].cpu().data.numpy(),
    }
