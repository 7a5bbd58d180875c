"""Randomised sweep of the adjacency build and the propagation against the oracle:
small ragged graphs (empty rows, isolated nodes, duplicates, self-loops, negative and
fractional weights, n down to 1) under every combination of build flags and self-loop
modes.  CSR and degrees bit for bit; propagation within 1e-5 (normwise and against
|A_hat|^k |X| element-wise)."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from golden_util import abs_scale, rel_err, scaled_err
from oracle import sgnn_oracle as orc

pytestmark = pytest.mark.gpu
EXTRA = int(os.environ.get("SGNN_SWEEP_EXTRA", "0"))      # more seeds for a soak run


def _graph(rng):
    n = int(rng.choice([1, 2, 3, 17, 64, 65, 200, 513]))
    m = int(rng.integers(0, 6 * n + 1))
    r = rng.integers(0, n, m)
    c = rng.integers(0, n, m)
    if m and rng.random() < 0.5:                       # a hub row and some guaranteed duplicates / loops
        r[: m // 4] = r[0]
        c[m // 2: m // 2 + m // 8] = r[m // 2: m // 2 + m // 8]
    kind = rng.integers(0, 4)
    if kind == 0:
        v = None
    elif kind == 1:
        v = rng.integers(1, 4, m).astype(np.float32)
    elif kind == 2:
        v = rng.integers(-2, 3, m).astype(np.float32)   # cancellations -> explicit zeros
    else:
        v = rng.random(m).astype(np.float32) + np.float32(0.05)
    return n, r.astype(np.int64), c.astype(np.int64), v


@pytest.mark.parametrize("seed", range(96 + EXTRA))
def test_build_flags_and_loop_modes_bit_exact(gpu_device, seed):
    from sgnn_amd.norm_adj import NormAdj
    rng = np.random.default_rng(1000 + seed)
    n, r, c, v = _graph(rng)
    vv = np.ones(r.size, np.float32) if v is None else v
    for strip in (False, True):
        for drop in (False, True):
            for mode in (orc.SELF_LOOP_NONE, orc.SELF_LOOP_ADD, orc.SELF_LOOP_REMAINING):
                exp = orc.norm_csr(r, c, vv, n, strip_diag=strip, self_loops=mode, drop_zeros=drop)
                adj = NormAdj.from_coo(r, c, v, n, gpu_device, strip_diag=strip, drop_zeros=drop, self_loops=mode)
                tag = (seed, n, strip, drop, mode)
                assert np.array_equal(adj.rowptr.cpu().numpy(), exp[0]), tag
                assert np.array_equal(adj.colidx.cpu().numpy(), exp[1]), tag
                assert np.array_equal(adj.vals.cpu().numpy(), exp[2]), tag
                assert np.array_equal(adj.deg.cpu().numpy(), exp[3], equal_nan=True), tag
    # symmetrise (+ clamp) against the scipy statement of the loaders
    if v is None or (v > 0).all():
        A = orc.symmetrize_binarize(r, c, v, n).tocoo()
        exp = orc.norm_csr(A.row, A.col, A.data, n, drop_zeros=True)
        adj = NormAdj.from_coo(r, c, v, n, gpu_device, symmetrize=True, clamp_one=True, drop_zeros=True)
        assert np.array_equal(adj.rowptr.cpu().numpy(), exp[0]) and np.array_equal(adj.colidx.cpu().numpy(), exp[1])
        if v is None or (v == np.round(v)).all():
            # the loaders' case: 0/1 (or small integer) entries, every sum exact in fp32
            assert np.array_equal(adj.vals.cpu().numpy(), exp[2]) and np.array_equal(adj.deg.cpu().numpy(), exp[3])
        else:
            # scipy sorts a row's entries with an unstable sort before it sums duplicates, so
            # three or more fractional duplicates of one entry have no defined summation order
            np.testing.assert_allclose(adj.vals.cpu().numpy(), exp[2], rtol=3e-7, atol=0)
            np.testing.assert_allclose(adj.deg.cpu().numpy(), exp[3], rtol=1e-6, atol=0)


@pytest.mark.parametrize("seed", range(64 + EXTRA))
def test_propagation_random_graphs_and_widths(gpu_device, seed):
    from sgnn_amd.norm_adj import NormAdj
    rng = np.random.default_rng(2000 + seed)
    n, r, c, v = _graph(rng)
    if v is not None:
        v = np.abs(v) + np.float32(0.25)                 # positive degrees, as every adjacency of the reference
    F = int(rng.choice([1, 5, 31, 33, 70, 129, 300, 601, 700]))
    order = int(rng.integers(1, 4))
    X = rng.standard_normal((n, F)).astype(np.float32)
    vv = np.ones(r.size, np.float32) if v is None else v
    rowptr, colidx, vals, deg, dinv = orc.norm_csr(r, c, vv, n)
    exp = X
    for _ in range(order):
        exp = orc.spmm_csr(rowptr, colidx, vals, dinv, dinv, exp)
    adj = NormAdj.from_coo(r, c, v, n, gpu_device)
    for use_rg in (True, False):
        adj.use_rowgroup = use_rg
        Y = adj.propagate(torch.from_numpy(X).to(gpu_device), order).cpu().numpy()
        scale = abs_scale(rowptr, colidx, vals, dinv, dinv, X, order)
        assert rel_err(Y, exp) <= 1e-5, (seed, n, F, order, use_rg)
        assert scaled_err(Y, exp, scale) <= 1e-5, (seed, n, F, order, use_rg)


def _clustered_graph(rng):
    """Heavy-tailed degrees with neighbourhoods concentrated in communities, so that rows of
    one group share columns (the case the merged column lists and the per-XCD dealing of
    the row-group kernel are built for), plus a few hubs longer than one segment."""
    n = int(rng.integers(2000, 40000))
    size = int(rng.choice([50, 300, 2000]))
    deg = np.minimum((rng.pareto(1.6, n) * rng.uniform(2, 12)).astype(np.int64) + 1, n // 2)
    hubs = rng.integers(0, n, 3)
    deg[hubs] = rng.integers(1500, 6000, 3).clip(max=n - 1)
    r = np.repeat(np.arange(n), deg)
    home = (r // size) * size
    local = home + rng.integers(0, size, r.size)
    c = np.where(rng.random(r.size) < 0.8, local, rng.integers(0, n, r.size)).clip(max=n - 1)
    return n, r.astype(np.int64), c.astype(np.int64)


@pytest.mark.parametrize("seed", range(20 + EXTRA // 20))
def test_propagation_clustered_graphs(gpu_device, seed):
    """Mid-sized graphs through every feature layout of the row-group kernel (16-, 8- and
    4-byte chunks, sliced widths), uniform and per-entry weights, symmetric or not."""
    from sgnn_amd.norm_adj import NormAdj
    rng = np.random.default_rng(3000 + seed)
    n, r, c = _clustered_graph(rng)
    F = int(rng.choice([33, 64, 100, 128, 130, 200, 256, 300, 512, 602, 640, 644, 700, 1000, 1433]))
    F = min(F, max(33, int(4e8 // max(r.size, 1))))         # keeps the float64 oracle to about a second
    weighted = bool(rng.integers(0, 2))
    v = (rng.random(r.size).astype(np.float32) + np.float32(0.1)) if weighted else None
    sym = bool(rng.integers(0, 2))
    order = int(rng.integers(1, 3))
    if sym:
        A = orc.symmetrize_binarize(r, c, None, n).tocoo()
        r, c, v = A.row.astype(np.int64), A.col.astype(np.int64), None
    vv = np.ones(r.size, np.float32) if v is None else v
    rowptr, colidx, vals, deg, dinv = orc.norm_csr(r, c, vv, n)
    X = rng.standard_normal((n, F)).astype(np.float32)
    exp = X
    for _ in range(order):
        exp = orc.spmm_csr(rowptr, colidx, vals, dinv, dinv, exp)
    scale = abs_scale(rowptr, colidx, vals, dinv, dinv, X, order)
    adj = NormAdj.from_coo(r, c, v, n, gpu_device)
    assert np.array_equal(adj.rowptr.cpu().numpy(), rowptr) and np.array_equal(adj.colidx.cpu().numpy(), colidx)
    assert np.array_equal(adj.vals.cpu().numpy(), vals) and np.array_equal(adj.deg.cpu().numpy(), deg)
    Y = adj.propagate(torch.from_numpy(X).to(gpu_device), order).cpu().numpy()
    tag = (seed, n, r.size, F, weighted, sym, order)
    assert rel_err(Y, exp) <= 1e-5, tag
    assert scaled_err(Y, exp, scale) <= 1e-5, tag
