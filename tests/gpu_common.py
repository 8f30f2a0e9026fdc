"""Shared helpers of the GPU parity tests."""
import numpy as np


def ocol(oracle, block):
    """product Block -> oracle column (same flat arrays)."""
    b = block.flatten()
    return oracle.Col(b.type, b.values, b.nulls, b.offsets)


def rand_block(pkg, rng, type_id, n, null_frac=0.0, domain=None):
    nulls = (rng.random(n) < null_frac).astype(np.uint8) if null_frac > 0 else None
    if type_id == pkg.BIGINT:
        v = rng.integers(-(2**62), 2**62, n) if domain is None else rng.integers(domain[0], domain[1], n)
        return pkg.Block(pkg.BIGINT, v.astype(np.int64), nulls)
    if type_id in (pkg.INTEGER, pkg.DATE):
        lo, hi = domain or (-(2**31), 2**31 - 1)
        return pkg.Block(type_id, rng.integers(lo, hi, n).astype(np.int32), nulls)
    if type_id == pkg.DOUBLE:
        if domain is None:
            v = rng.standard_normal(n) * 10.0 ** rng.integers(-5, 6, n)
        else:
            v = rng.integers(domain[0], domain[1], n).astype(np.float64)
        return pkg.Block(pkg.DOUBLE, v, nulls)
    if type_id == pkg.BOOLEAN:
        return pkg.Block(pkg.BOOLEAN, rng.integers(0, 2, n).astype(np.uint8), nulls)
    if type_id == pkg.VARCHAR:
        lo, hi = domain or (0, 1000)
        ks = rng.integers(lo, hi, n)
        vals = [None if (nulls is not None and nulls[i]) else ("k%d" % k) * (1 + k % 3) for i, k in enumerate(ks)]
        return pkg.Block(pkg.VARCHAR, vals)
    raise ValueError(type_id)


def ulp_diff(a, b):
    """distance in units in the last place between two float64 arrays (0 for identical bits, NaN == NaN)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    ia = a.view(np.int64).copy()
    ib = b.view(np.int64).copy()
    ia = np.where(ia < 0, np.int64(-(2**63)) - ia, ia)
    ib = np.where(ib < 0, np.int64(-(2**63)) - ib, ib)
    d = np.abs(ia - ib)
    both_nan = np.isnan(a) & np.isnan(b)
    return np.where(both_nan, 0, d)
