"""The CPU oracle must reproduce every golden vector captured from the reference.

Golden vectors: tests/golden/*.npz, written by tests/golden/make_golden.py from
the reference's own functions (utils.py:54-62, utils.py:81-96,
model/SGNN.py:497-501).

Bar for the fp32 propagation: 1e-5 relative, applied (a) normwise and (b)
element by element against ``|A_hat|^k |X|`` (see golden_util.abs_scale); both
sides are fp32 sums in unspecified order.  CSR structure and degrees are
compared bit for bit.
"""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from golden_util import abs_scale, hashed_features, load_golden, rel_err, scaled_err
from oracle import sgnn_oracle as orc

TOL = 1e-5
HAND = ["directed", "duplicates", "path_isolated", "selfloop", "star", "weight2"]


def _check(Y, exp, r, c, v, n, X, order, strip=False):
    rowptr, colidx, vals, deg, dinv = orc.norm_csr(r, c, v, n, strip_diag=strip, drop_zeros=strip)
    scale = abs_scale(rowptr, colidx, vals, dinv, dinv, X, order)
    Y = np.asarray(Y)
    assert Y.shape == exp.shape and Y.dtype == np.float32
    assert rel_err(Y, exp) <= TOL
    assert scaled_err(Y, exp, scale) <= TOL


@pytest.mark.parametrize("name", HAND)
def test_hand_graphs_raw(name):
    g = load_golden("hand_graphs.npz")
    n = int(g[f"{name}__n"])
    r, c, v = g[f"{name}__row"], g[f"{name}__col"], g[f"{name}__val"]
    X = hashed_features(n, 5, seed=int(g[f"{name}__xseed"]))
    A = sp.coo_matrix((v, (r, c)), shape=(n, n))
    adj = orc.coo_to_sparse_tensor(A)
    for order in (1, 2, 3):
        exp = g[f"{name}__Y_raw_o{order}"]
        _check(orc.propagate_high_order(adj, torch.from_numpy(X), order).numpy(), exp, r, c, v, n, X, order)
        _check(orc.propagate_scipy(A, X, order), exp, r, c, v, n, X, order)


@pytest.mark.parametrize("name", HAND)
def test_hand_graphs_strip_and_csr(name):
    g = load_golden("hand_graphs.npz")
    n = int(g[f"{name}__n"])
    r, c, v = g[f"{name}__row"], g[f"{name}__col"], g[f"{name}__val"]
    A = sp.coo_matrix((v, (r, c)), shape=(n, n)).tocsr()
    S = orc.remove_self_loop(A).tocoo()
    got = sorted(zip(S.row.tolist(), S.col.tolist(), S.data.astype(np.float32).tolist()))
    exp = sorted(zip(g[f"{name}__strip_row"].tolist(), g[f"{name}__strip_col"].tolist(),
                     g[f"{name}__strip_val"].tolist()))
    assert got == exp
    rowptr, colidx, vals, deg, dinv = orc.norm_csr(r, c, v, n, strip_diag=True, drop_zeros=True)
    assert np.array_equal(rowptr, g[f"{name}__csr_rowptr"])
    assert np.array_equal(colidx, g[f"{name}__csr_colidx"])
    assert np.array_equal(vals, g[f"{name}__csr_vals"])
    assert np.array_equal(deg, g[f"{name}__deg"])
    X = hashed_features(n, 5, seed=int(g[f"{name}__xseed"]))
    Y = X
    for order in (1, 2):
        Y = orc.spmm_csr(rowptr, colidx, vals, dinv, dinv, Y)
        _check(Y, g[f"{name}__Y_strip_o{order}"], r, c, v, n, X, order, strip=True)


def test_cora_structure():
    g = load_golden("cora_cites.npz")
    n = int(g["n"])
    assert n == 2708 and g["row"].size == 10556
    r, c = g["row"], g["col"]
    v = np.ones(r.size, dtype=np.float32)
    rowptr, colidx, vals, deg, dinv = orc.norm_csr(r, c, v, n)
    assert np.array_equal(rowptr, g["csr_rowptr"])
    assert np.array_equal(colidx, g["csr_colidx"])
    assert np.array_equal(vals, g["csr_vals"])
    assert np.array_equal(deg, g["deg"])
    X = hashed_features(n, int(g["F"]), seed=int(g["xseed"]))
    A = sp.coo_matrix((v, (r, c)), shape=(n, n))
    adj = orc.coo_to_sparse_tensor(A)
    for order in (1, 2, 3):
        _check(orc.propagate_high_order(adj, torch.from_numpy(X), order).numpy(), g[f"Y_o{order}"],
               r, c, v, n, X, order)
    _check(orc.propagate_scipy(A, X, 3), g["Y_o3"], r, c, v, n, X, 3)


def test_rmat2048_widths():
    g = load_golden("rmat2048.npz")
    n = int(g["n"])
    r, c = g["row"], g["col"]
    v = np.ones(r.size, dtype=np.float32)
    rowptr, colidx, vals, deg, dinv = orc.norm_csr(r, c, v, n)
    assert np.array_equal(rowptr, g["csr_rowptr"])
    assert np.array_equal(colidx, g["csr_colidx"])
    assert np.array_equal(deg, g["deg"])
    adj = orc.coo_to_sparse_tensor(sp.coo_matrix((v, (r, c)), shape=(n, n)))
    for F in (1, 7, 64, 128):
        X = hashed_features(n, F, seed=200 + F)
        _check(orc.propagate_high_order(adj, torch.from_numpy(X), 1).numpy(), g[f"Y_F{F}_o1"], r, c, v, n, X, 1)
        _check(orc.spmm_csr(rowptr, colidx, vals, dinv, dinv, X), g[f"Y_F{F}_o1"], r, c, v, n, X, 1)
    X = hashed_features(n, 64, seed=264)
    _check(orc.propagate_high_order(adj, torch.from_numpy(X), 2).numpy(), g["Y_F64_o2"], r, c, v, n, X, 2)


def test_rmat512_wide_weight2():
    g = load_golden("rmat512_f602_w2.npz")
    n = int(g["n"])
    r, c, v = g["row"], g["col"], g["val"]
    X = hashed_features(n, 602, seed=602)
    _check(orc.propagate_scipy(sp.coo_matrix((v, (r, c)), shape=(n, n)), X, 1), g["Y_F602_o1"], r, c, v, n, X, 1)


def test_directed_raw_with_loops_and_duplicates():
    g = load_golden("rmat1024_directed_raw.npz")
    n = int(g["n"])
    r, c, v = g["row"], g["col"], g["val"]
    X = hashed_features(n, 16, seed=16)
    A = sp.coo_matrix((v, (r, c)), shape=(n, n))
    adj = orc.coo_to_sparse_tensor(A)
    for order in (1, 2):
        _check(orc.propagate_high_order(adj, torch.from_numpy(X), order).numpy(), g[f"Y_raw_o{order}"],
               r, c, v, n, X, order)
        _check(orc.propagate_scipy(A, X, order), g[f"Y_raw_o{order}"], r, c, v, n, X, order)
    rowptr, colidx, vals, deg, dinv = orc.norm_csr(r, c, v, n, strip_diag=True, drop_zeros=True)
    assert np.array_equal(rowptr, g["csr_rowptr"])
    assert np.array_equal(colidx, g["csr_colidx"])
    assert np.array_equal(vals, g["csr_vals"])
    assert np.array_equal(deg, g["deg"])
    _check(orc.spmm_csr(rowptr, colidx, vals, dinv, dinv, X), g["Y_strip_o1"], r, c, v, n, X, 1, strip=True)


def test_pyg_semantics_match_sgnn_on_symmetric_unit_graph():
    """On a symmetric, loop-free, unit-weight graph PyG's gcn_norm and the SGNN
    normalisation are the same matrix (the only cross-check available for the
    un-vendored PyG operators)."""
    g = load_golden("cora_cites.npz")
    n = int(g["n"])
    r, c = g["row"].astype(np.int64), g["col"].astype(np.int64)
    v = np.ones(r.size, dtype=np.float32)
    X = hashed_features(n, 32, seed=11)
    W = np.eye(32, dtype=np.float32)
    _check(orc.pyg_sgconv(X, np.stack((r, c)), W, None, K=2), g["Y_o2"], r, c, v, n, X, 2)
    _check(orc.pyg_gcnconv(X, np.stack((r, c)), W, None), g["Y_o1"], r, c, v, n, X, 1)
