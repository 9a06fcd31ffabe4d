"""Compact tensor summaries shared by make_goldens.py and the tests: (sum, sum of squares,
numel, 16 evenly spaced elements) as float64 — pins a 7M-element tensor in 19 numbers."""
import numpy as np
import torch

N_SLICE = 16


def summarize(t):
    t = torch.as_tensor(t).detach().to(torch.float64).reshape(-1)
    n = t.numel()
    if n == 0:
        return np.zeros(3 + N_SLICE)
    idx = torch.linspace(0, n - 1, N_SLICE).round().long()
    return np.concatenate([[t.sum().item(), (t * t).sum().item(), float(n)], t[idx].numpy()])


def pack_mask(m):
    """float/bool mask (any shape) -> uint8 bit-packed 1-D array (shape is known to the test)."""
    return np.packbits(np.asarray(m, dtype=bool).reshape(-1))


def unpack_mask(bits, shape):
    n = int(np.prod(shape))
    return torch.from_numpy(np.unpackbits(bits)[:n].reshape(shape).astype(np.float32))
