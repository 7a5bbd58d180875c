"""CPU oracle for the sparse-propagation path.  TEST INFRASTRUCTURE ONLY.

This module restates, on the CPU, what the reference computes for
``A_hat^order . X`` and for the adjacency ingestion around it.  It exists so
that tests, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` can check (and time) the HIP path against an independent
implementation.  Nothing under ``sgnn_amd/`` may import it.

Pinning: the reference holds no tests or golden vectors for this path.  The
functions below were checked in the build container against the reference's
own Python functions imported from its source tree (``tests/golden/make_golden.py``
is the generating script, the resulting vectors live in ``tests/golden/*.npz``
and ``tests/test_oracle_golden.py`` replays them).  The PyG-operator
restatements (``pyg_*``) follow torch_geometric's documented semantics; that
package is not vendored by the reference and is not installed here, so those
three functions are "parity unpinned" beyond their agreement with the SGNN
formula on symmetric, diagonal-free, unit-weight graphs.

Reference locations restated here (paths relative to the reference tree):
  * ``model/SGNN.py:497-501``   StackedGNN._remove_self_loop
  * ``utils.py:54-62``          csr_to_sparse_Tensor / coo_to_sparse_Tensor
  * ``utils.py:77-96``          process_data_with_adjacency(_high_order)
  * ``reddit_utils.py:49-57``, ``reddit_utils.py:281-288``,
    ``data_loader.py:673-686``  scipy statements of D^-1/2 (A+I) D^-1/2
  * ``model/SGC.py:11,14``, ``model/GCN.py:9-16``  PyG SGConv / GCNConv call sites
  * ``data_loader.py:451-452``  adj.T + adj, minimum(1) of the loaders
  * ``model/SGNN.py:167-170``   dense sub-adjacency of a sampled batch
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import torch


# --------------------------------------------------------------------------
# adjacency ingestion
# --------------------------------------------------------------------------
def remove_self_loop(adjacency):
    """``A <- A - diag(A)`` exactly as model/SGNN.py:497-501 does it in scipy.

    The sparse subtraction drops the entries that become zero, so the result
    has no stored diagonal.
    """
    size = adjacency.shape[0]
    idx = list(range(size))
    self_loop = sp.coo_matrix((adjacency.diagonal(), (idx, idx)), shape=(size, size))
    return adjacency - self_loop


def coo_to_sparse_tensor(coo_mat):
    """scipy COO -> torch sparse COO (int64 indices, fp32 values), utils.py:59-62."""
    idx = torch.from_numpy(np.vstack((coo_mat.row, coo_mat.col)).astype(np.int64))
    val = torch.from_numpy(np.asarray(coo_mat.data, dtype=np.float32))
    return torch.sparse_coo_tensor(idx, val, torch.Size(coo_mat.shape))


def csr_to_sparse_tensor(csr_mat):
    """scipy (any format) -> torch sparse COO, utils.py:54-56."""
    return coo_to_sparse_tensor(csr_mat.tocoo())


# --------------------------------------------------------------------------
# the hot path, op for op (utils.py:81-96)
# --------------------------------------------------------------------------
def propagate_high_order(adjacency, X, order=1):
    """``(D^-1/2 (A+I) D^-1/2)^order X`` with the reference's op sequence.

    adjacency: torch sparse COO fp32 [n, n] on the CPU (not necessarily
    coalesced); X: dense fp32 [n, F].  Steps, in the reference's order:
    identity COO, ``A + I``, weighted row sum, ``sqrt``, ``1/x``, then per hop
    ``row-scale -> sparse mm -> row-scale`` (utils.py:82-95).
    """
    size = X.shape[0]
    idx = torch.arange(size, dtype=torch.int64)
    self_loop = torch.sparse_coo_tensor(torch.stack((idx, idx)), torch.ones(size), (size, size))
    adj = adjacency + self_loop
    degree = torch.sparse.sum(adj, dim=1).to_dense().sqrt()
    degree = 1 / degree
    processed_X = X
    for _ in range(order):
        processed_X = (processed_X.t() * degree).t()
        processed_X = adj.mm(processed_X)
        processed_X = (processed_X.t() * degree).t()
    return processed_X


def propagate(adjacency, X):
    """order=1 wrapper, utils.py:77-78."""
    return propagate_high_order(adjacency, X, order=1)


# --------------------------------------------------------------------------
# closed form in scipy: canonical CSR of A(+I), degree, dinv
# --------------------------------------------------------------------------
SELF_LOOP_NONE = 0      # keep the diagonal as given
SELF_LOOP_ADD = 1       # diag += 1   (utils.py:83-86: adjacency + self_loop)
SELF_LOOP_REMAINING = 2  # diag = weight of the last existing loop entry, else 1 (PyG add_remaining_self_loops)


def degree_wave_order(rowptr, vals):
    """fp32 row sums in the order the device kernel uses (it is a parallel
    reduction, so "column order" is not available there): 64 strided partial sums
    — partial l adds entries l, l+64, ... of the row in that order — folded in
    halves (32, 16, ..., 1).  For integer-valued weights every order gives the
    same fp32 result; for arbitrary weights this is the order that is bit-exact.
    """
    n = len(rowptr) - 1
    lens = np.diff(rowptr)
    deg = np.zeros(n, dtype=np.float32)
    exact = vals.size == 0 or (np.all(vals == np.round(vals)) and np.abs(vals).sum(dtype=np.float64) < 2 ** 24)
    if exact:                                    # order-independent: take the fast path
        np.add.at(deg, np.repeat(np.arange(n), lens), vals)
        return deg
    for r in range(n):
        v = vals[rowptr[r]:rowptr[r + 1]]
        steps = -(-v.size // 64)
        p = np.zeros(steps * 64, dtype=np.float32)
        p[:v.size] = v
        lanes = np.zeros(64, dtype=np.float32)
        for row64 in p.reshape(steps, 64):       # sequential per lane
            lanes = lanes + row64
        while lanes.size > 1:
            h = lanes.size // 2
            lanes = lanes[:h] + lanes[h:]
        deg[r] = lanes[0]
    return deg


def norm_csr(row, col, val, n, strip_diag=False, self_loops=SELF_LOOP_ADD, drop_zeros=False):
    """Canonical CSR of the matrix the hot path multiplies by, plus its degree.

    Input is a COO triple (duplicates allowed; they are summed, as both
    ``scipy.sparse`` and ``torch.sparse`` coalescing do).  Output follows the
    same conventions the reference's scipy statements use
    (reddit_utils.py:49-57, data_loader.py:673-686): rows sorted, columns sorted
    within a row, one entry per (row, col).

    Returns ``rowptr[int32 n+1], colidx[int32 nnz], vals[fp32 nnz],
    deg[fp32 n], dinv[fp32 n]`` where ``deg`` is the weighted ROW sum
    (utils.py:88) accumulated in fp32 (order: ``degree_wave_order``), and
    ``dinv = 1/sqrt(deg)`` in two roundings (utils.py:88-89).
    """
    row = np.asarray(row, dtype=np.int64)
    col = np.asarray(col, dtype=np.int64)
    val = np.asarray(val, dtype=np.float32)
    if strip_diag:
        keep = row != col
        row, col, val = row[keep], col[keep], val[keep]
    if self_loops == SELF_LOOP_REMAINING:
        # torch_geometric's add_remaining_self_loops: every diagonal entry is taken
        # out, the loop weight of a node is that of its LAST diagonal entry (an
        # index assignment), 1 where it has none
        d = row == col
        loop_w = np.ones(n, dtype=np.float32)
        loop_w[row[d]] = val[d]
        row = np.concatenate((row[~d], np.arange(n, dtype=np.int64)))
        col = np.concatenate((col[~d], np.arange(n, dtype=np.int64)))
        val = np.concatenate((val[~d], loop_w))
        self_loops = SELF_LOOP_NONE
    # stable sort by (row, col); duplicates summed in fp32 in input order
    key = row * np.int64(n) + col
    order = np.argsort(key, kind="stable")
    key, val = key[order], val[order]
    if key.size:
        first = np.concatenate(([True], key[1:] != key[:-1]))
    else:
        first = np.zeros(0, dtype=bool)
    ukey = key[first]
    starts = np.flatnonzero(first)
    uval = np.zeros(ukey.size, dtype=np.float32)
    # sequential fp32 accumulation per run (np.add.reduceat pairs differently)
    seg = np.cumsum(first) - 1
    np.add.at(uval, seg, val)
    if drop_zeros:
        nz = uval != 0
        ukey, uval = ukey[nz], uval[nz]
    urow, ucol = ukey // n, ukey % n
    if self_loops != SELF_LOOP_NONE:
        has_diag = np.zeros(n, dtype=bool)
        d = urow == ucol
        has_diag[urow[d]] = True
        if self_loops == SELF_LOOP_ADD:
            uval = uval.copy()
            uval[d] = uval[d] + np.float32(1.0)
        missing = np.flatnonzero(~has_diag)
        urow = np.concatenate((urow, missing))
        ucol = np.concatenate((ucol, missing))
        uval = np.concatenate((uval, np.ones(missing.size, dtype=np.float32)))
        k2 = urow * np.int64(n) + ucol
        o2 = np.argsort(k2, kind="stable")
        urow, ucol, uval = urow[o2], ucol[o2], uval[o2]
    counts = np.bincount(urow, minlength=n)
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(counts, out=rowptr[1:])
    deg = degree_wave_order(rowptr, uval.astype(np.float32))
    with np.errstate(divide="ignore", invalid="ignore"):
        dinv = (np.float32(1.0) / np.sqrt(deg)).astype(np.float32)
    return (rowptr.astype(np.int32), ucol.astype(np.int32), uval.astype(np.float32), deg, dinv)


def spmm_csr(rowptr, colidx, vals, dinv_row, dinv_col, X):
    """``Y = diag(dinv_row) . CSR . diag(dinv_col) . X`` with the reference's
    rounding structure: scale (fp32), sum, scale (fp32) — utils.py:93-95.
    Sum in float64 so that this variant is an order-independent yardstick.
    """
    n = len(rowptr) - 1
    A = sp.csr_matrix((vals.astype(np.float64), colidx, rowptr), shape=(n, X.shape[0]))
    Xs = X.astype(np.float32)
    if dinv_col is not None:
        Xs = (Xs * dinv_col[:, None].astype(np.float32)).astype(np.float32)
    Y = (A @ Xs.astype(np.float64)).astype(np.float32)
    if dinv_row is not None:
        Y = (Y * dinv_row[:, None].astype(np.float32)).astype(np.float32)
    return Y


def propagate_scipy(A, X, order=1, strip_diag=False):
    """scipy-CSR evaluation of the hot path (float64 sums), for cross-checks."""
    coo = sp.coo_matrix(A)
    rowptr, colidx, vals, deg, dinv = norm_csr(coo.row, coo.col, coo.data, A.shape[0],
                                               strip_diag=strip_diag)
    Y = np.asarray(X, dtype=np.float32)
    for _ in range(order):
        Y = spmm_csr(rowptr, colidx, vals, dinv, dinv, Y)
    return Y


# --------------------------------------------------------------------------
# PyG operator semantics (parity unpinned: torch_geometric is third-party,
# un-vendored and version-unpinned in the reference; README.md:34)
# --------------------------------------------------------------------------
def pyg_gcn_norm(edge_index, edge_weight, n, add_self_loops=True):
    """torch_geometric ``gcn_norm`` as documented: remaining self-loops with
    weight 1, ``deg = scatter_add(w, target)``, ``deg^-1/2`` with inf -> 0,
    ``norm_e = dinv[src] * w_e * dinv[dst]``.
    Returns (src, dst, norm) as numpy arrays.
    """
    src = np.asarray(edge_index[0], dtype=np.int64)
    dst = np.asarray(edge_index[1], dtype=np.int64)
    w = np.ones(src.size, dtype=np.float32) if edge_weight is None else np.asarray(edge_weight, np.float32)
    if add_self_loops:
        loop = src == dst
        loop_w = np.ones(n, dtype=np.float32)
        loop_w[src[loop]] = w[loop]
        src = np.concatenate((src[~loop], np.arange(n)))
        dst = np.concatenate((dst[~loop], np.arange(n)))
        w = np.concatenate((w[~loop], loop_w))
    deg = np.zeros(n, dtype=np.float32)
    np.add.at(deg, dst, w)
    with np.errstate(divide="ignore"):
        dinv = deg.astype(np.float32) ** np.float32(-0.5)
    dinv[np.isinf(dinv)] = 0
    norm = dinv[src] * w * dinv[dst]
    return src, dst, norm.astype(np.float32)


def pyg_propagate(src, dst, norm, x):
    """``out[dst] += norm_e * x[src]`` (MessagePassing.propagate, aggr='add')."""
    out = np.zeros_like(x, dtype=np.float64)
    np.add.at(out, dst, norm[:, None].astype(np.float64) * x[src].astype(np.float64))
    return out.astype(np.float32)


def pyg_sgconv(x, edge_index, weight, bias, K=2, edge_weight=None):
    """SGConv(K): ``lin(A_hat^K x)`` (model/SGC.py:11,14)."""
    n = x.shape[0]
    src, dst, norm = pyg_gcn_norm(edge_index, edge_weight, n)
    h = np.asarray(x, dtype=np.float32)
    for _ in range(K):
        h = pyg_propagate(src, dst, norm, h)
    out = h @ np.asarray(weight, np.float32).T
    if bias is not None:
        out = out + np.asarray(bias, np.float32)
    return out.astype(np.float32)


def pyg_gcnconv(x, edge_index, weight, bias, edge_weight=None):
    """GCNConv: ``A_hat (x W^T) + b`` (model/GCN.py:9-16)."""
    n = x.shape[0]
    src, dst, norm = pyg_gcn_norm(edge_index, edge_weight, n)
    h = np.asarray(x, np.float32) @ np.asarray(weight, np.float32).T
    out = pyg_propagate(src, dst, norm, h)
    if bias is not None:
        out = out + np.asarray(bias, np.float32)
    return out.astype(np.float32)


# --------------------------------------------------------------------------
# ingestion and sampling around the path
# --------------------------------------------------------------------------
def symmetrize_binarize(row, col, val, n):
    """The loaders' ``adj = adj.T + adj; adj = adj.minimum(1)``
    (data_loader.py:451-452, 476-477, 490-491, 507-508) on a COO edge list;
    returns a scipy CSR in fp32."""
    val = np.ones(len(row), np.float32) if val is None else np.asarray(val, np.float32)
    adj = sp.coo_matrix((val, (np.asarray(row), np.asarray(col))), shape=(n, n), dtype=np.float32)
    adj = adj.T + adj
    adj = adj.minimum(1)
    return adj.tocsr()


def sample_block(adjacency, idx):
    """Dense ``adjacency[idx, :][:, idx]`` as fp32 (model/SGNN.py:167-170)."""
    sub = adjacency.tocsr()[idx, :]
    sub = sub[:, idx]
    return np.asarray(sub.todense(), dtype=np.float32)
