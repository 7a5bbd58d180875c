#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE itself.

Run in the build container only (it needs the reference tree, which never
travels to the GPU box):

    python tests/golden/make_golden.py [/root/reference]

The reference's own functions are imported from its source tree and executed on
the CPU:
  * ``utils.csr_to_sparse_Tensor`` / ``coo_to_sparse_Tensor``  (utils.py:54-62)
  * ``utils.process_data_with_adjacency_high_order``           (utils.py:81-96)
  * ``model.SGNN.StackedGNN._remove_self_loop``                (model/SGNN.py:497-501)
``utils`` imports ``metric`` which imports the third-party package ``munkres``
(metric.py:6), absent here and unrelated to this path (clustering accuracy
only); an empty module object is registered under that name before the import.

Only inputs and outputs are written: edge lists, the canonical scipy CSR of
``A - diag(A) + I`` (the matrix the path multiplies by), and the reference's
outputs.  Dense inputs are regenerated from ``golden_util.hashed_features``.
The Cora structure comes from the reference's text file ``data/cora/cora.cites``;
the reference's pickled Planetoid files (``data/node/ind.*``) are deliberately
not opened.
"""
from __future__ import annotations

import os
import sys
import types
import warnings

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from golden_util import hashed_features  # noqa: E402
from sgnn_amd.graphgen import rmat_edges_numpy, symmetric_simple_numpy  # noqa: E402

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"


def import_reference():
    sys.dont_write_bytecode = True
    stub = types.ModuleType("munkres")
    stub.Munkres = object
    sys.modules.setdefault("munkres", stub)
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir("/tmp")            # the reference reads ./global_settings.json lazily; keep cwd harmless
    try:
        import utils as ref_utils
        from model import SGNN as ref_sgnn
    finally:
        os.chdir(cwd)
    return ref_utils, ref_sgnn


def canonical_csr_plus_I(A):
    """scipy statement of the multiplied matrix: canonical CSR of A + I."""
    n = A.shape[0]
    M = (sp.csr_matrix(A) + sp.eye(n, dtype=A.dtype, format="csr")).tocsr()
    M.sum_duplicates()
    M.sort_indices()
    deg = np.asarray(M.sum(axis=1)).ravel()
    return (M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data.astype(np.float32),
            deg.astype(np.float32))


def run_reference(ref_utils, coo, X, order, via_csr):
    import torch
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if via_csr:
            adj = ref_utils.csr_to_sparse_Tensor(coo.tocsr(), "cpu")
        else:
            adj = ref_utils.coo_to_sparse_Tensor(coo, "cpu")
        Y = ref_utils.process_data_with_adjacency_high_order(adj, torch.from_numpy(X), "cpu", order=order)
    return Y.numpy().astype(np.float32)


def sym(pairs, w=1.0):
    r = [a for a, b in pairs] + [b for a, b in pairs]
    c = [b for a, b in pairs] + [a for a, b in pairs]
    return np.array(r), np.array(c), np.full(len(r), w, dtype=np.float32)


def hand_graphs():
    g = {}
    r, c, v = sym([(0, 1), (1, 2), (2, 3)])
    g["path_isolated"] = (6, r, c, v)
    g["directed"] = (5, np.array([0, 0, 1, 3, 4, 4]), np.array([1, 2, 2, 0, 3, 1]),
                     np.array([1.0, 0.5, 2.0, 1.0, 3.0, 1.0], dtype=np.float32))
    r, c, v = sym([(0, 1), (1, 2), (2, 4), (3, 4)])
    g["selfloop"] = (5, np.concatenate((r, [2, 4])), np.concatenate((c, [2, 4])),
                     np.concatenate((v, np.array([1.0, 3.0], dtype=np.float32))))
    g["duplicates"] = (4, np.array([0, 1, 0, 1, 2, 3, 0]), np.array([1, 0, 1, 0, 3, 2, 1]),
                       np.ones(7, dtype=np.float32))
    r, c, v = sym([(0, 1), (0, 2), (1, 2), (3, 4), (4, 5), (5, 6), (6, 3)], w=2.0)
    g["weight2"] = (7, r, c, v)
    r, c, v = sym([(0, k) for k in range(1, 8)])
    g["star"] = (8, r, c, v)
    return g


def main():
    ref_utils, ref_sgnn = import_reference()
    out = {}

    # ---- 1. hand graphs ------------------------------------------------
    for k, (name, (n, r, c, v)) in enumerate(sorted(hand_graphs().items())):
        coo = sp.coo_matrix((v, (r, c)), shape=(n, n))
        X = hashed_features(n, 5, seed=100 + k)
        out[f"{name}__n"] = np.int64(n)
        out[f"{name}__row"] = r.astype(np.int32)
        out[f"{name}__col"] = c.astype(np.int32)
        out[f"{name}__val"] = v
        out[f"{name}__xseed"] = np.int64(100 + k)
        for order in (1, 2, 3):
            # raw path: whatever diagonal A holds stays, then + I (utils.py:86)
            out[f"{name}__Y_raw_o{order}"] = run_reference(ref_utils, coo, X, order, via_csr=False)
        # caller-side strip (model/SGNN.py:497-501) then the same function
        holder = types.SimpleNamespace(adjacency=coo.tocsr())
        ref_sgnn.StackedGNN._remove_self_loop(holder)
        stripped = holder.adjacency.tocsr()
        sc = stripped.tocoo()
        out[f"{name}__strip_row"] = sc.row.astype(np.int32)
        out[f"{name}__strip_col"] = sc.col.astype(np.int32)
        out[f"{name}__strip_val"] = sc.data.astype(np.float32)
        ip, ix, dv, deg = canonical_csr_plus_I(stripped)
        out[f"{name}__csr_rowptr"], out[f"{name}__csr_colidx"] = ip, ix
        out[f"{name}__csr_vals"], out[f"{name}__deg"] = dv, deg
        for order in (1, 2):
            out[f"{name}__Y_strip_o{order}"] = run_reference(ref_utils, stripped.tocoo(), X, order, via_csr=True)
    np.savez_compressed(os.path.join(HERE, "hand_graphs.npz"), **out)

    # ---- 2. Cora structure (text edge list shipped with the reference) ---
    cites = np.loadtxt(os.path.join(REF, "data", "cora", "cora.cites"), dtype=np.int64)
    ids = np.unique(cites)
    a = np.searchsorted(ids, cites[:, 0])
    b = np.searchsorted(ids, cites[:, 1])
    n = ids.size
    keep = a != b
    lo, hi = np.minimum(a[keep], b[keep]), np.maximum(a[keep], b[keep])
    key = np.unique(lo * n + hi)
    lo, hi = key // n, key % n
    r, c = np.concatenate((lo, hi)), np.concatenate((hi, lo))
    A = sp.coo_matrix((np.ones(r.size, dtype=np.float32), (r, c)), shape=(n, n))
    X = hashed_features(n, 32, seed=11)
    ip, ix, dv, deg = canonical_csr_plus_I(A.tocsr())
    cora = dict(n=np.int64(n), row=r.astype(np.int32), col=c.astype(np.int32), xseed=np.int64(11),
                F=np.int64(32), csr_rowptr=ip, csr_colidx=ix, csr_vals=dv, deg=deg)
    for order in (1, 2, 3):
        cora[f"Y_o{order}"] = run_reference(ref_utils, A, X, order, via_csr=True)
    np.savez_compressed(os.path.join(HERE, "cora_cites.npz"), **cora)
    print("cora:", n, "nodes", r.size, "entries")

    # ---- 3. R-MAT, symmetric simple, several feature widths ---------------
    n = 2048
    r, c = symmetric_simple_numpy(n, 16_000, seed=5)
    A = sp.coo_matrix((np.ones(r.size, dtype=np.float32), (r, c)), shape=(n, n))
    ip, ix, dv, deg = canonical_csr_plus_I(A.tocsr())
    rm = dict(n=np.int64(n), row=r.astype(np.int32), col=c.astype(np.int32),
              csr_rowptr=ip, csr_colidx=ix, csr_vals=dv, deg=deg)
    for F in (1, 7, 64, 128):
        X = hashed_features(n, F, seed=200 + F)
        rm[f"Y_F{F}_o1"] = run_reference(ref_utils, A, X, 1, via_csr=True)
    rm["Y_F64_o2"] = run_reference(ref_utils, A, hashed_features(n, 64, seed=264), 2, via_csr=True)
    np.savez_compressed(os.path.join(HERE, "rmat2048.npz"), **rm)
    print("rmat2048:", r.size, "entries, max degree", int(np.diff(ip).max()))

    n = 512
    r, c = symmetric_simple_numpy(n, 4_000, seed=6)
    A = sp.coo_matrix((np.full(r.size, 2.0, dtype=np.float32), (r, c)), shape=(n, n))   # Reddit-style weight 2
    X = hashed_features(n, 602, seed=602)
    wide = dict(n=np.int64(n), row=r.astype(np.int32), col=c.astype(np.int32),
                val=np.full(r.size, 2.0, dtype=np.float32),
                Y_F602_o1=run_reference(ref_utils, A, X, 1, via_csr=True))
    np.savez_compressed(os.path.join(HERE, "rmat512_f602_w2.npz"), **wide)

    # ---- 4. raw directed R-MAT with self-loops and duplicates -------------
    n = 1024
    r, c = rmat_edges_numpy(n, 20_000, seed=9)
    v = (1.0 + (np.arange(r.size) % 3)).astype(np.float32)            # weights 1, 2, 3
    coo = sp.coo_matrix((v, (r, c)), shape=(n, n))
    X = hashed_features(n, 16, seed=16)
    raw = dict(n=np.int64(n), row=r.astype(np.int32), col=c.astype(np.int32), val=v)
    raw["Y_raw_o1"] = run_reference(ref_utils, coo, X, 1, via_csr=False)
    raw["Y_raw_o2"] = run_reference(ref_utils, coo, X, 2, via_csr=False)
    holder = types.SimpleNamespace(adjacency=coo.tocsr())
    ref_sgnn.StackedGNN._remove_self_loop(holder)
    stripped = holder.adjacency.tocsr()
    ip, ix, dv, deg = canonical_csr_plus_I(stripped)
    raw.update(csr_rowptr=ip, csr_colidx=ix, csr_vals=dv, deg=deg)
    raw["Y_strip_o1"] = run_reference(ref_utils, stripped.tocoo(), X, 1, via_csr=True)
    np.savez_compressed(os.path.join(HERE, "rmat1024_directed_raw.npz"), **raw)
    print("done")


if __name__ == "__main__":
    main()
