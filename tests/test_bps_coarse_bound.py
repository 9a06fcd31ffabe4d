"""The error bound the matrix-core BPS path relies on (csrc/bps_encode.hip), checked in numpy.

bps_mfma_kernel SELECTS candidate blocks with a coarse value v_ij = sum over 23 bf16 plane products (fp32 accumulate) and
DECIDES with the contract's arithmetic; its correctness rests on
    |v_ij - c_i + |b_i|^2 - d2_ij|  <=  eps_i = 35 u S_i,   S_i = (R + |b_i|)^2,  u = 2^-24
(+ 2^-17 v for the tile index written over the key's low mantissa bits).  This test restates the kernel's operand
construction — truncation split into three bf16 planes, the six plane products kept per coordinate, |x|^2 in three
planes, c_i = |b|^2 + bias in two — accumulates the 23 products in fp32 in k-slot order (one admissible order; the
bound holds for any) and measures the error against the contract's d2 on clouds near and far from the origin, tiny and
large.  CPU only; the GPU tests compare the kernel's ids and distance bits with the oracle.
"""
import numpy as np
import pytest

f32 = np.float32


def split3(x):
    """x == h + m + l exactly, each a bf16 value (fp32 with the low 16 bits clear), by truncation."""
    x = np.asarray(x, dtype=f32)
    h = (x.view(np.uint32) & np.uint32(0xFFFF0000)).view(f32)
    r1 = (x - h).astype(f32)
    m = (r1.view(np.uint32) & np.uint32(0xFFFF0000)).view(f32)
    r2 = (r1 - m).astype(f32)
    l = (r2.view(np.uint32) & np.uint32(0xFFFF0000)).view(f32)
    assert np.array_equal((h.astype(np.float64) + m + l).astype(f32), x)      # exact
    assert np.array_equal(r2, l)                                               # nothing left below the third plane
    return h, m, l


def contract_d2(cloud, b):
    d = (cloud - b[None, :]).astype(f32)
    t = (d[:, 0] * d[:, 0]).astype(f32)
    t = (d[:, 1].astype(np.float64) * d[:, 1] + t).astype(f32)                 # fmaf
    return (d[:, 2].astype(np.float64) * d[:, 2] + t).astype(f32)


def coarse(cloud, b, R):
    """The kernel's K = 23 dot product for one basis point against every cloud point; fp32 accumulation in slot order."""
    x, y, z = cloud[:, 0], cloud[:, 1], cloud[:, 2]
    n2 = (z.astype(np.float64) * z + (y.astype(np.float64) * y + (x * x).astype(f32)).astype(f32)).astype(f32)
    bb = f32(np.float64(b[2]) * b[2] + f32(np.float64(b[1]) * b[1] + f32(b[0] * b[0])))
    S = f32((R + np.sqrt(bb)) ** 2)
    c = f32(bb + S * f32(2.0 ** -14))
    ch, cm, _ = split3(c)
    prods = []
    for a, q in ((x, f32(-2) * b[0]), (y, f32(-2) * b[1]), (z, f32(-2) * b[2])):
        ah, am, al = split3(a)
        qh, qm, ql = split3(q)
        prods += [ah * np.float64(qh), am * np.float64(qh), al * np.float64(qh), ah * np.float64(ql),
                  ah * np.float64(qm), am * np.float64(qm)]                    # exact in fp64: 8 x 8 bit mantissas
    nh, nm, nl = split3(n2)
    prods += [nh.astype(np.float64), nm.astype(np.float64), nl.astype(np.float64) + ch, np.full(len(x), np.float64(cm))]
    acc = np.zeros(len(x), dtype=f32)
    for p in prods:
        acc = (acc.astype(np.float64) + p).astype(f32)
    return acc, np.float64(ch) + np.float64(cm), S, bb


def clouds():
    g = np.random.default_rng(11)
    base = (g.normal(size=(2048, 3)) * [0.05, 0.03, 0.08]).astype(f32)
    yield "object at the origin", base
    yield "object 0.4 away", (base + f32(0.4)).astype(f32)
    yield "object 50 away", (base + f32(50.0)).astype(f32)
    yield "tiny object", (base * f32(1e-3)).astype(f32)
    yield "points within 1e-6", (base[:1] + g.normal(size=(2048, 3)) * 1e-6).astype(f32)
    yield "large coordinates", (base * f32(4000.0)).astype(f32)               # |x| up to ~1e3, the path's limit


@pytest.mark.parametrize("name,cloud", list(clouds()), ids=[n for n, _ in clouds()])
def test_coarse_value_is_within_its_bound(name, cloud):
    g = np.random.default_rng(5)
    basis = (g.uniform(-0.2, 0.2, size=(64, 3))).astype(f32)
    basis[0] = cloud[17]                                                       # a basis point ON a cloud point: d2 = 0
    x = cloud.astype(np.float64)
    n2 = (cloud[:, 2].astype(np.float64) * cloud[:, 2] + (cloud[:, 1].astype(np.float64) * cloud[:, 1]
          + (cloud[:, 0] * cloud[:, 0]).astype(f32)).astype(f32)).astype(f32)
    R = f32(np.sqrt(n2.max()))
    worst = 0.0
    for b in basis:
        v, c, S, bb = coarse(cloud, b, R)
        d2 = contract_d2(cloud, b)
        bias = c - (np.float64(b[0]) ** 2 + np.float64(b[1]) ** 2 + np.float64(b[2]) ** 2)   # what v carries on top of d2
        eps = 35 * 2.0 ** -24 * np.float64(S)
        err = np.abs(v.astype(np.float64) - bias - d2.astype(np.float64))
        worst = max(worst, float((err / eps).max()))
        assert (v > 0).all()                                                   # keys order like values only if positive
        # the kernel's threshold: 2 eps (rounded up) + the key's overwritten bits; the true minimum must be a candidate
        k = (v.view(np.uint32) & np.uint32(0xFFFFFF80)).view(f32)
        lim = np.float64(k.min()) + np.float64(S) * 1.25 * 2.0 ** -18 + np.float64(k.min()) * 2.0 ** -16
        winners = np.flatnonzero(d2 == d2.min())
        assert (k[winners] <= lim).all(), (name, b)
    print("%-22s  worst |v - bias - d2| / eps = %.3f" % (name, worst))
    assert worst < 1.0, (name, worst)
