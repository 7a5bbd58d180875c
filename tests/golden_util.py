"""Helpers shared by the golden-vector generator and the tests that replay them."""
from __future__ import annotations

import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def hashed_features(n, F, seed=0):
    """Deterministic fp32 features in [-1, 1) from integer arithmetic only.

    The golden files store outputs, not the dense inputs; this regenerates the
    inputs bit-for-bit on any machine (no RNG stream involved).
    """
    i = np.arange(n, dtype=np.uint64)[:, None]
    f = np.arange(F, dtype=np.uint64)[None, :]
    h = (i * np.uint64(0x9E3779B97F4A7C15) + f * np.uint64(0xC2B2AE3D27D4EB4F)
         + np.uint64((int(seed) * 0x165667B19E3779F9) & 0xFFFFFFFFFFFFFFFF))
    h ^= h >> np.uint64(29)
    h *= np.uint64(0xBF58476D1CE4E5B9)
    h ^= h >> np.uint64(32)
    u = (h & np.uint64(0xFFFFFF)).astype(np.float32)          # 24 bits, exact in fp32
    return (u / np.float32(1 << 23) - np.float32(1.0)).astype(np.float32)


def load_golden(name):
    with np.load(os.path.join(GOLDEN_DIR, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def rel_err(a, b):
    """Normwise relative error ||a-b||_F / ||b||_F (b = expected)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / (den if den > 0 else 1.0))


def abs_scale(rowptr, colidx, vals, dinv_row, dinv_col, X, order=1):
    """``(|A_hat|^order) |X|`` in float64: the magnitude each output element's
    terms sum to.  An fp32 sum in any order differs from the exact value by a
    small multiple of 2^-24 times this, so it is the yardstick the 1e-5
    relative bound of the propagation path is applied to element by element
    (a plain |a-b|/|b| is unbounded wherever the exact sum cancels to ~0).
    """
    import scipy.sparse as sp
    n = len(rowptr) - 1
    A = sp.csr_matrix((np.abs(np.asarray(vals, dtype=np.float64)), colidx, rowptr), shape=(n, X.shape[0]))
    if dinv_col is not None:
        A = A @ sp.diags(np.abs(np.asarray(dinv_col, dtype=np.float64)))
    if dinv_row is not None:
        A = sp.diags(np.abs(np.asarray(dinv_row, dtype=np.float64))) @ A
    S = np.abs(np.asarray(X, dtype=np.float64))
    for _ in range(order):
        S = A @ S
    return S


def scaled_err(a, b, scale):
    """max |a-b| / scale, elements with scale 0 must match exactly."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    d = np.abs(a - b)
    if not b.size:
        return 0.0
    zero = scale == 0
    if zero.any() and d[zero].max() > 0:
        return float("inf")
    return float((d[~zero] / scale[~zero]).max()) if (~zero).any() else 0.0
