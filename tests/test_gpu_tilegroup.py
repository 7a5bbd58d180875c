"""Tile-group (matrix-core) SpMM against the CPU oracle and against the plain CSR kernel."""
import numpy as np
import pytest
import torch

from golden_util import abs_scale, rel_err, scaled_err
from oracle import sgnn_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _graph(n, und_edges, seed, dev, weight=None):
    from sgnn_amd import NormAdj
    from sgnn_amd.graphgen import symmetric_simple_numpy
    r, c = symmetric_simple_numpy(n, und_edges, seed=seed)
    v = None if weight is None else np.full(r.size, weight, np.float32)
    adj = NormAdj.from_coo(r, c, v, n, dev)
    adj.use_tilegroup = True
    return adj, r, c, (np.ones(r.size, np.float32) if v is None else v)


@pytest.mark.parametrize("n,edges,F,order", [(3000, 40_000, 128, 1), (3000, 40_000, 602, 2), (700, 9_000, 100, 1),
                                             (5000, 200_000, 37, 3), (513, 3_000, 256, 1), (40, 100, 130, 2),
                                             # narrow widths and last slices of 1, 2 and 3 N-tiles of 32 features
                                             (3000, 40_000, 16, 2), (2000, 30_000, 8, 1), (3000, 40_000, 65, 1),
                                             (2000, 30_000, 96, 2), (2000, 30_000, 193, 1)])
def test_tilegroup_matches_oracle(gpu_device, n, edges, F, order):
    adj, r, c, v = _graph(n, edges, seed=n + F, dev=gpu_device)
    rowptr, colidx, vals, deg, dinv = orc.norm_csr(r, c, v, n)
    X = np.random.default_rng(F).standard_normal((n, F)).astype(np.float32)
    exp = X
    for _ in range(order):
        exp = orc.spmm_csr(rowptr, colidx, vals, dinv, dinv, exp)
    Y = adj.propagate(torch.from_numpy(X).to(gpu_device), order)
    torch.cuda.synchronize()
    assert adj._tg, "tile-group plan was not built"
    assert int(adj._tg_bufs[F]["state"][1].item()) == 0, "fall-back flag raised on ordinary data"
    Y = Y.cpu().numpy()
    scale = abs_scale(rowptr, colidx, vals, dinv, dinv, X, order)
    assert rel_err(Y, exp) <= TOL
    assert scaled_err(Y, exp, scale) <= TOL


def test_tilegroup_weight2_and_single_hop_entry(gpu_device):
    adj, r, c, v = _graph(2500, 60_000, seed=5, dev=gpu_device, weight=2.0)
    assert adj.w_off == 2.0 and adj.corr is not None      # A + A^T then A + I: 2.0 off the diagonal, 1.0 on it
    rowptr, colidx, vals, deg, dinv = orc.norm_csr(r, c, v, 2500)
    Xn = np.random.default_rng(3).standard_normal((2500, 130)).astype(np.float32)
    exp = orc.spmm_csr(rowptr, colidx, vals, dinv, dinv, orc.spmm_csr(rowptr, colidx, vals, dinv, dinv, Xn))
    Yc = adj.propagate(torch.from_numpy(Xn).to(gpu_device), 2).cpu().numpy()
    assert adj._tg and int(adj._tg_bufs[130]["state"][1].item()) == 0
    assert rel_err(Yc, exp) <= TOL
    assert scaled_err(Yc, exp, abs_scale(rowptr, colidx, vals, dinv, dinv, Xn, 2)) <= TOL
    from sgnn_amd import NormAdj
    adj2 = NormAdj.from_coo(r, c, v, 2500, gpu_device, self_loops=0, zero_inf=True)
    adj2.use_tilegroup = True
    assert adj2.w_off == 2.0 and adj2.corr is None
    X = torch.randn(2500, 200, device=gpu_device)
    Y = adj2.spmm(X)
    adj2.use_tilegroup = False
    adj2.use_rowgroup = False
    Y_ref = adj2.spmm(X)
    assert float((Y - Y_ref).norm() / Y_ref.norm()) <= TOL


@pytest.mark.parametrize("F", [16, 37, 48, 96, 601, 602])
def test_tilegroup_weight2_every_last_slice_width(gpu_device, F):
    """Weight 2 off the diagonal, 1 on it: the kernel instances with the diagonal correction, with
    last slices of one to four N-tiles and widths that end inside a tile (the correction's reads
    of X are clamped to the row there), on a node count that leaves a group partly empty."""
    n = 2300
    adj, r, c, v = _graph(n, 50_000, seed=F, dev=gpu_device, weight=2.0)
    assert adj.w_off == 2.0 and adj.corr is not None
    rowptr, colidx, vals, deg, dinv = orc.norm_csr(r, c, v, n)
    X = np.random.default_rng(F + 1).standard_normal((n, F)).astype(np.float32)
    exp = orc.spmm_csr(rowptr, colidx, vals, dinv, dinv, X)
    # X in a buffer of its own that ends with its last row: the clamped reads stay inside it
    Xd = torch.empty(n * F, device=gpu_device)
    Xd.copy_(torch.from_numpy(X).reshape(-1))
    Y = adj.propagate(Xd.view(n, F), 1).cpu().numpy()
    assert adj._tg and int(adj._tg_bufs[F]["state"][1].item()) == 0
    assert rel_err(Y, exp) <= TOL
    assert scaled_err(Y, exp, abs_scale(rowptr, colidx, vals, dinv, dinv, X, 1)) <= TOL


def test_tilegroup_wide_range_and_nonfinite_fall_back(gpu_device):
    adj, r, c, v = _graph(3000, 50_000, seed=9, dev=gpu_device)
    F = 128
    X = torch.randn(3000, F, device=gpu_device)
    X[5, 7] = 1e30
    X[6, 8] = 1e-30                        # 2^200 apart: the planes cannot hold both
    Y = adj.propagate(X, 1)
    torch.cuda.synchronize()
    assert int(adj._tg_bufs[F]["state"][1].item()) != 0
    adj.use_tilegroup = False
    adj.use_rowgroup = False
    Y_ref = adj.propagate(X, 1)
    assert torch.equal(Y, Y_ref)           # the fall-back is the plain CSR kernel itself
    adj.use_tilegroup, adj.use_rowgroup = True, True
    X[5, 7] = float("inf")
    X[6, 8] = 0.0
    Y = adj.propagate(X, 1)
    Y_ref2 = None
    adj.use_tilegroup = False
    adj.use_rowgroup = False
    Y_ref2 = adj.propagate(X, 1)
    assert torch.equal(torch.isfinite(Y), torch.isfinite(Y_ref2))
    assert torch.equal(Y[torch.isfinite(Y)], Y_ref2[torch.isfinite(Y_ref2)])
    # rows scaled over 2^60: thousands of elements lose bits in the planes -> fall-back, exact
    adj.use_tilegroup, adj.use_rowgroup = True, True
    X = torch.randn(3000, F, device=gpu_device) * torch.exp2(torch.randint(-30, 30, (3000, 1), device=gpu_device).float())
    Y = adj.propagate(X, 1)
    torch.cuda.synchronize()
    assert int(adj._tg_bufs[F]["state"][1].item()) != 0
    # rows scaled over 2^12 stay on the matrix cores and stay accurate element by element
    g = torch.Generator(device=gpu_device); g.manual_seed(11)
    X = (torch.randn(3000, F, device=gpu_device, generator=g)
         * torch.exp2(torch.randint(-6, 6, (3000, 1), device=gpu_device, generator=g).float()))
    Y = adj.propagate(X, 1)
    torch.cuda.synchronize()
    assert int(adj._tg_bufs[F]["state"][1].item()) == 0
    adj.use_tilegroup = False
    adj.use_rowgroup = False
    Y_ref = adj.propagate(X, 1)
    scale = adj.propagate(X.abs(), 1)
    assert float(((Y - Y_ref).abs() / scale.clamp_min(1e-30)).max()) <= TOL


@pytest.mark.parametrize("weight", [None, 2.0])
def test_tilegroup_hub_graph_with_split_groups(gpu_device, weight):
    """A graph whose top group's merged list is cut into several work items (partial rows + combine)."""
    from sgnn_amd import NormAdj, _lib
    from sgnn_amd.graphgen import rmat_undirected_device
    dev = gpu_device
    lib = _lib.load()
    lib.sgnn_tune_set(b"tg_seg", 16)
    try:
        n, F = 60_000, 200
        row, col = rmat_undirected_device(n, 6_000_000, seed=2, device=dev)
        val = None if weight is None else torch.full((row.numel(),), weight, device=dev)
        adj = NormAdj.from_coo(row, col, val, n, dev)
        assert (adj.corr is not None) == (weight is not None)
        adj.use_tilegroup = True
        X = torch.randn(n, F, device=dev)
        Y = adj.propagate(X, 1)
        assert adj._tg["n_long"] > 0 and adj._tg["slots"] > 0
    finally:
        lib.sgnn_tune_set(b"tg_seg", 0)
    adj.use_tilegroup = False
    adj.use_rowgroup = False
    Y_ref = adj.propagate(X, 1)
    scale = adj.propagate(X.abs(), 1)
    assert float((Y - Y_ref).norm() / Y_ref.norm()) <= TOL
    assert float(((Y - Y_ref).abs() / scale.clamp_min(1e-30)).max()) <= TOL


def test_kernel_timer_counts_the_main_kernel_of_every_path(gpu_device):
    """sgnn_kernel_time_ms: one timed launch per hop whichever kernel runs, nothing while switched off."""
    from sgnn_amd import _lib
    lib = _lib.load()
    adj, r, c, v = _graph(3000, 40_000, seed=21, dev=gpu_device)
    X = torch.randn(3000, 128, device=gpu_device)
    adj.propagate(X, 1)                                   # plans built outside the count
    assert _lib.kernel_time_ms() == (0.0, 0)
    lib.sgnn_tune_set(b"time_kernel", 1)
    try:
        adj.propagate(X, 2)                               # tile-group: two hops
        adj.use_tilegroup = False
        adj.propagate(X, 1)                               # row-group
        adj.use_rowgroup = False
        adj.propagate(X, 3)                               # plain CSR chain: three hops
        total, launches = _lib.kernel_time_ms()
        assert launches == 6 and 0.0 < total < 1000.0
        assert _lib.kernel_time_ms() == (0.0, 0)          # reading starts a new count
    finally:
        lib.sgnn_tune_set(b"time_kernel", 0)
    adj.propagate(X, 1)
    assert _lib.kernel_time_ms() == (0.0, 0)
