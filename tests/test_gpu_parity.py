"""GPU parity tests: the HIP path, called through the C ABI (ctypes), against the
golden vectors captured from the reference and against the CPU oracle.

Bars: CSR row pointers, column indices, values and degrees bit for bit; dinv bit
for bit against numpy's correctly rounded 1/sqrt; propagation within 1e-5
relative — normwise and element-wise against ``|A_hat|^k |X|`` (golden_util).
"""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from golden_util import abs_scale, hashed_features, load_golden, rel_err, scaled_err
from oracle import sgnn_oracle as orc

pytestmark = pytest.mark.gpu

TOL = 1e-5
HAND = ["directed", "duplicates", "path_isolated", "selfloop", "star", "weight2"]


@pytest.fixture(scope="module")
def dev(gpu_device):
    return gpu_device


def _na():
    from sgnn_amd.norm_adj import NormAdj
    return NormAdj


def _check_Y(Y, exp, rowptr, colidx, vals, dinv, X, order):
    Y = Y.detach().cpu().numpy() if isinstance(Y, torch.Tensor) else np.asarray(Y)
    assert Y.shape == exp.shape and Y.dtype == np.float32
    scale = abs_scale(rowptr, colidx, vals, dinv, dinv, X, order)
    assert rel_err(Y, exp) <= TOL
    assert scaled_err(Y, exp, scale) <= TOL


def _assert_csr_equal(adj, rowptr, colidx, vals, deg):
    assert np.array_equal(adj.rowptr.cpu().numpy(), rowptr)
    assert np.array_equal(adj.colidx.cpu().numpy(), colidx)
    assert np.array_equal(adj.vals.cpu().numpy(), vals)
    assert np.array_equal(adj.deg.cpu().numpy(), deg)
    with np.errstate(divide="ignore"):
        dinv = (np.float32(1.0) / np.sqrt(deg.astype(np.float32))).astype(np.float32)
    assert np.array_equal(adj.dinv.cpu().numpy(), dinv)


# --------------------------------------------------------------------- build
@pytest.mark.parametrize("name", HAND)
def test_build_hand_graphs_bit_exact(dev, name):
    g = load_golden("hand_graphs.npz")
    n = int(g[f"{name}__n"])
    r, c, v = g[f"{name}__row"], g[f"{name}__col"], g[f"{name}__val"]
    adj = _na().from_coo(r, c, v, n, dev, strip_diag=True, drop_zeros=True)
    _assert_csr_equal(adj, g[f"{name}__csr_rowptr"], g[f"{name}__csr_colidx"], g[f"{name}__csr_vals"],
                      g[f"{name}__deg"])
    # raw variant (diagonal kept, += 1) against the oracle's closed form
    adj = _na().from_coo(r, c, v, n, dev)
    rowptr, colidx, vals, deg, dinv = orc.norm_csr(r, c, v, n)
    _assert_csr_equal(adj, rowptr, colidx, vals, deg)


def test_build_real_and_rmat_structures_bit_exact(dev):
    for fname, weighted in (("cora_cites.npz", False), ("rmat2048.npz", False), ("rmat1024_directed_raw.npz", True)):
        g = load_golden(fname)
        n = int(g["n"])
        r, c = g["row"], g["col"]
        v = g["val"] if weighted else None
        adj = _na().from_coo(r, c, v, n, dev, strip_diag=weighted, drop_zeros=weighted)
        _assert_csr_equal(adj, g["csr_rowptr"], g["csr_colidx"], g["csr_vals"], g["deg"])
        assert adj.nnz == g["csr_colidx"].size


def test_build_modes_and_errors(dev):
    NormAdj = _na()
    r = np.array([0, 1, 2, 2, 3]); c = np.array([1, 0, 2, 3, 2]); v = np.array([1, 1, 5, 1, 1], np.float32)
    for mode in (orc.SELF_LOOP_NONE, orc.SELF_LOOP_ADD, orc.SELF_LOOP_REMAINING):
        adj = NormAdj.from_coo(r, c, v, 5, dev, self_loops=mode, zero_inf=(mode != orc.SELF_LOOP_ADD))
        rowptr, colidx, vals, deg, dinv = orc.norm_csr(r, c, v, 5, self_loops=mode)
        assert np.array_equal(adj.rowptr.cpu().numpy(), rowptr)
        assert np.array_equal(adj.colidx.cpu().numpy(), colidx)
        assert np.array_equal(adj.vals.cpu().numpy(), vals)
        assert np.array_equal(adj.deg.cpu().numpy(), deg)
        got = adj.dinv.cpu().numpy()
        if mode == orc.SELF_LOOP_NONE:        # node 4 isolated: 1/sqrt(0) = inf -> 0 (gcn_norm)
            assert got[4] == 0.0
    # empty graph: A + I = I, propagation is the identity
    adj = NormAdj.from_coo(np.zeros(0, np.int64), np.zeros(0, np.int64), None, 7, dev)
    assert adj.nnz == 7 and adj.unit_weights
    X = torch.randn(7, 5, device=dev)
    assert torch.equal(adj.propagate(X, 2), X)
    with pytest.raises(ValueError):
        NormAdj.from_coo(np.array([0, 9]), np.array([1, 0]), None, 5, dev)
    with pytest.raises(ValueError):
        NormAdj.from_coo(np.array([0, 1]), np.array([1]), None, 5, dev)
    from sgnn_amd._lib import SgnnHipError
    with pytest.raises(SgnnHipError):
        NormAdj.from_coo(r, c, v, 5, "cpu")


# --------------------------------------------------------------- propagation
@pytest.mark.parametrize("name", HAND)
def test_propagate_hand_graphs_vs_reference(dev, name):
    from sgnn_amd import utils as su
    g = load_golden("hand_graphs.npz")
    n = int(g[f"{name}__n"])
    r, c, v = g[f"{name}__row"], g[f"{name}__col"], g[f"{name}__val"]
    X = hashed_features(n, 5, seed=int(g[f"{name}__xseed"]))
    Xd = torch.from_numpy(X).to(dev)
    # raw: handle built straight from the COO, as utils.coo_to_sparse_Tensor would
    adj = su.coo_to_sparse_Tensor(sp.coo_matrix((v, (r, c)), shape=(n, n)), dev)
    rowptr, colidx, vals, deg, dinv = orc.norm_csr(r, c, v, n)
    for order in (1, 2, 3):
        Y = su.process_data_with_adjacency_high_order(adj, Xd, dev, order=order)
        assert Y.is_contiguous() and not Y.requires_grad
        _check_Y(Y, g[f"{name}__Y_raw_o{order}"], rowptr, colidx, vals, dinv, X, order)
    assert torch.equal(Xd, torch.from_numpy(X).to(dev))          # input untouched
    # stripped: the caller-side strip of model/SGNN.py:497-501 done in scipy, then the handle
    S = orc.remove_self_loop(sp.coo_matrix((v, (r, c)), shape=(n, n)).tocsr())
    adj = su.csr_to_sparse_Tensor(S, dev)
    assert adj.shape[0] == n
    rowptr, colidx, vals, deg, dinv = orc.norm_csr(r, c, v, n, strip_diag=True, drop_zeros=True)
    for order in (1, 2):
        Y = su.process_data_with_adjacency_high_order(adj, Xd, dev, order=order)
        _check_Y(Y, g[f"{name}__Y_strip_o{order}"], rowptr, colidx, vals, dinv, X, order)
    Y1 = su.process_data_with_adjacency(adj, Xd, dev)
    _check_Y(Y1, g[f"{name}__Y_strip_o1"], rowptr, colidx, vals, dinv, X, 1)


def test_propagate_cora_vs_reference(dev):
    from sgnn_amd import utils as su
    g = load_golden("cora_cites.npz")
    n = int(g["n"])
    r, c = g["row"], g["col"]
    A = sp.coo_matrix((np.ones(r.size, np.float32), (r, c)), shape=(n, n)).tocsr()
    adj = su.csr_to_sparse_Tensor(A, dev)
    X = hashed_features(n, int(g["F"]), seed=int(g["xseed"]))
    rowptr, colidx, vals, dinv = g["csr_rowptr"], g["csr_colidx"], g["csr_vals"], adj.dinv.cpu().numpy()
    for order in (1, 2, 3):
        Y = su.process_data_with_adjacency_high_order(adj, torch.from_numpy(X), dev, order=order)   # X from the CPU
        assert Y.device.type == "cuda"
        _check_Y(Y, g[f"Y_o{order}"], rowptr, colidx, vals, dinv, X, order)


def test_propagate_rmat_widths_vs_reference(dev):
    g = load_golden("rmat2048.npz")
    n = int(g["n"])
    adj = _na().from_coo(g["row"], g["col"], None, n, dev)
    dinv = adj.dinv.cpu().numpy()
    for F in (1, 7, 64, 128):
        X = hashed_features(n, F, seed=200 + F)
        Y = adj.propagate(torch.from_numpy(X).to(dev), 1)
        _check_Y(Y, g[f"Y_F{F}_o1"], g["csr_rowptr"], g["csr_colidx"], g["csr_vals"], dinv, X, 1)
    X = hashed_features(n, 64, seed=264)
    Y = adj.propagate(torch.from_numpy(X).to(dev), 2)
    _check_Y(Y, g["Y_F64_o2"], g["csr_rowptr"], g["csr_colidx"], g["csr_vals"], dinv, X, 2)


def test_propagate_wide_weight2_vs_reference(dev):
    g = load_golden("rmat512_f602_w2.npz")
    n = int(g["n"])
    adj = _na().from_coo(g["row"], g["col"], g["val"], n, dev)
    assert not adj.unit_weights and adj.w_off == 2.0 and adj.corr is not None      # off-diagonal 2, diagonal 1
    X = hashed_features(n, 602, seed=602)
    rowptr, colidx, vals, deg, dinv = orc.norm_csr(g["row"], g["col"], g["val"], n)
    for flag in (True, False):                      # pattern-only row-group kernel, then plain CSR with values
        adj.use_rowgroup = flag
        assert (adj._rowgroup_plan(602) is not None) == flag
        Y = adj.propagate(torch.from_numpy(X).to(dev), 1)
        _check_Y(Y, g["Y_F602_o1"], rowptr, colidx, vals, dinv, X, 1)


def test_propagate_directed_raw_vs_reference(dev):
    from sgnn_amd import utils as su
    g = load_golden("rmat1024_directed_raw.npz")
    n = int(g["n"])
    r, c, v = g["row"], g["col"], g["val"]
    X = hashed_features(n, 16, seed=16)
    # through a genuine torch sparse COO tensor, as the reference's callers hold it
    t = orc.coo_to_sparse_tensor(sp.coo_matrix((v, (r, c)), shape=(n, n))).to(dev)
    rowptr, colidx, vals, deg, dinv = orc.norm_csr(r, c, v, n)
    for order in (1, 2):
        Y = su.process_data_with_adjacency_high_order(t, torch.from_numpy(X).to(dev), dev, order=order)
        _check_Y(Y, g[f"Y_raw_o{order}"], rowptr, colidx, vals, dinv, X, order)
    assert getattr(t, "_sgnn_norm_adj", None) is not None         # converted once, then reused
    adj = _na().from_coo(r, c, v, n, dev, strip_diag=True, drop_zeros=True)
    rowptr, colidx, vals, deg, dinv = orc.norm_csr(r, c, v, n, strip_diag=True, drop_zeros=True)
    _check_Y(adj.propagate(torch.from_numpy(X).to(dev), 1), g["Y_strip_o1"], rowptr, colidx, vals, dinv, X, 1)


@pytest.mark.parametrize("w", [2.0, 0.3])
def test_uniform_weight_graph_with_stored_diagonal(dev, w):
    """Off-diagonal entries all w, some rows with their own diagonal entry (value 5),
    an isolated node, no self-loops added: the pattern kernel's diagonal correction."""
    from sgnn_amd import _lib
    from sgnn_amd.graphgen import symmetric_simple_numpy
    n, F = 3000, 96
    r, c = symmetric_simple_numpy(n - 1, 20_000, seed=12)          # node n-1 stays isolated
    d = np.arange(0, n, 7)
    r2, c2 = np.concatenate((r, d)), np.concatenate((c, d))
    v = np.concatenate((np.full(r.size, w, np.float32), np.full(d.size, 5.0, np.float32)))
    X = np.random.default_rng(3).standard_normal((n, F)).astype(np.float32)
    for loops in (_lib.LOOPS_ADD, _lib.LOOPS_NONE):
        adj = _na().from_coo(r2, c2, v, n, dev, self_loops=loops, zero_inf=True)
        assert adj.w_off == np.float32(w) and adj.corr is not None
        rowptr, colidx, vals, deg, dinv = orc.norm_csr(r2, c2, v, n, self_loops=loops)
        dinv = np.where(np.isinf(dinv), np.float32(0), dinv)
        assert np.array_equal(adj.dinv.cpu().numpy(), dinv)
        exp = orc.spmm_csr(rowptr, colidx, vals, dinv, dinv, orc.spmm_csr(rowptr, colidx, vals, dinv, dinv, X))
        for flag in (True, False):
            adj.use_rowgroup = flag
            _check_Y(adj.propagate(torch.from_numpy(X).to(dev), 2), exp, rowptr, colidx, vals, dinv, X, 2)


def test_degree_bit_exact_for_fractional_weights(dev):
    """The degree is a parallel fp32 reduction in a fixed order; the oracle follows
    the same order, so even non-integer weights give identical bits."""
    n = 2000
    r, c = _hub_graph(n, seed=5)
    key = np.unique(r.astype(np.int64) * n + c)
    r, c = key // n, key % n
    v = np.random.default_rng(9).uniform(0.1, 3.0, r.size).astype(np.float32)
    adj = _na().from_coo(r, c, v, n, dev)
    assert adj.w_off is None
    rowptr, colidx, vals, deg, dinv = orc.norm_csr(r, c, v, n)
    assert np.array_equal(adj.vals.cpu().numpy(), vals)
    assert np.array_equal(adj.deg.cpu().numpy(), deg)
    assert np.array_equal(adj.dinv.cpu().numpy(), dinv)
    X = np.random.default_rng(1).standard_normal((n, 40)).astype(np.float32)
    _check_Y(adj.propagate(torch.from_numpy(X).to(dev), 1), orc.spmm_csr(rowptr, colidx, vals, dinv, dinv, X),
             rowptr, colidx, vals, dinv, X, 1)


def _hub_graph(n, seed):
    """R-MAT plus two hubs whose rows are longer than one wavefront's segment."""
    from sgnn_amd.graphgen import symmetric_simple_numpy
    r, c = symmetric_simple_numpy(n, 6 * n, seed=seed)
    rng = np.random.default_rng(seed)
    hubs = []
    for h, deg in ((3, min(n - 1, 5000)), (n // 2, min(n - 1, 1500))):
        nb = rng.choice(np.setdiff1d(np.arange(n), [h]), size=deg, replace=False)
        hubs.append((np.full(deg, h), nb))
    hr = np.concatenate([a for a, b in hubs] + [b for a, b in hubs])
    hc = np.concatenate([b for a, b in hubs] + [a for a, b in hubs])
    return np.concatenate((r, hr)), np.concatenate((c, hc))


@pytest.mark.parametrize("F", [1, 2, 3, 4, 7, 8, 30, 64, 100, 128, 130, 256, 500, 602, 1433])
def test_propagate_random_vs_oracle_all_widths(dev, F):
    n = 6000
    r, c = _hub_graph(n, seed=F)
    v = np.ones(r.size, np.float32)                       # duplicates (hub edges that already existed) sum to 2
    adj = _na().from_coo(r, c, v, n, dev)
    assert not adj.unit_weights                           # weighted: always the plain CSR kernel
    rowptr, colidx, vals, deg, dinv = orc.norm_csr(r, c, v, n)
    assert np.array_equal(adj.rowptr.cpu().numpy(), rowptr) and np.array_equal(adj.colidx.cpu().numpy(), colidx)
    assert int(np.diff(rowptr).max()) > 1024              # exercises the split-row path
    X = np.random.default_rng(F).standard_normal((n, F)).astype(np.float32)
    Y = adj.propagate(torch.from_numpy(X).to(dev), 1)
    exp = orc.spmm_csr(rowptr, colidx, vals, dinv, dinv, X)
    _check_Y(Y, exp, rowptr, colidx, vals, dinv, X, 1)
    # bitwise reproducible run to run
    Y2 = adj.propagate(torch.from_numpy(X).to(dev), 1)
    assert torch.equal(Y, Y2)


@pytest.mark.parametrize("F", [1, 7, 33, 34, 64, 100, 128, 130, 256, 500, 602, 640, 1280, 1433])
@pytest.mark.parametrize("order", [1, 3])
def test_unit_weight_graph_rowgroup_and_plain_kernels(dev, F, order):
    """Unit-weight graphs take the row-group kernel where it supports F; it and the
    plain CSR kernel must both meet the oracle (and each is bitwise reproducible)."""
    from sgnn_amd import _lib
    n = 5000
    r, c = _hub_graph(n, seed=1000 + F)
    key = np.unique(r.astype(np.int64) * n + c)
    r, c = key // n, key % n
    adj = _na().from_coo(r, c, None, n, dev)
    assert adj.unit_weights
    rowptr, colidx, vals, deg, dinv = orc.norm_csr(r, c, np.ones(r.size, np.float32), n)
    X = np.random.default_rng(F).standard_normal((n, F)).astype(np.float32)
    exp = X
    for _ in range(order):
        exp = orc.spmm_csr(rowptr, colidx, vals, dinv, dinv, exp)
    Xd = torch.from_numpy(X).to(dev)
    lib = _lib.load()
    supported = lib.sgnn_rowgroup_config(F) != 0
    try:
        # both group sizes (a graph this small takes 8 by default), then the plain kernel
        for flag, rows in ((True, 0), (True, 16), (True, 8), (False, 0)):
            lib.sgnn_tune_set(b"rg_R", rows)
            adj.use_rowgroup = flag
            Y = adj.propagate(Xd, order)
            _check_Y(Y, exp, rowptr, colidx, vals, dinv, X, order)
            assert torch.equal(Y, adj.propagate(Xd, order))
            plan = adj._rowgroup_plan(F)
            assert (plan is not None) == (flag and supported)
            if plan is not None:
                assert plan["R"] == (rows or 8)
                assert 0 < plan["merged"] <= adj.nnz and plan["n_long"] >= 1     # hubs: split groups exercised
    finally:
        lib.sgnn_tune_set(b"rg_R", 0)


def test_column_scaling_inside_the_kernel_equals_a_scaled_copy(dev):
    """A small graph's row-group plan scales the fetched rows of X inside the kernel when X
    is 16-byte aligned, and otherwise works from a scaled, aligned copy: the same bits."""
    n, F = 5000, 96
    r, c = _hub_graph(n, seed=77)
    key = np.unique(r.astype(np.int64) * n + c)
    r, c = key // n, key % n
    for val in (None, np.full(r.size, 2.0, np.float32)):          # pattern sum; diagonal correction
        adj = _na().from_coo(r, c, val, n, dev)
        plan = adj._rowgroup_plan(F)
        assert plan is not None and adj._folds_col_scale(plan, F)
        g = torch.Generator(device=dev); g.manual_seed(8)
        X = torch.randn(n, F, device=dev, generator=g)
        shifted = torch.empty(n * F + 1, device=dev)[1:].view(n, F)   # rows off the 16-byte grid
        shifted.copy_(X)
        assert X.data_ptr() % 16 == 0 and shifted.data_ptr() % 16 != 0
        assert torch.equal(adj.spmm(X), adj.spmm(shifted))
        for order in (1, 3):
            assert torch.equal(adj.propagate(X, order), adj.propagate(shifted, order))
        rowptr, colidx, vals, deg, dinv = orc.norm_csr(r, c, np.ones(r.size, np.float32) if val is None else val, n)
        Xh = X.cpu().numpy()
        _check_Y(adj.spmm(X), orc.spmm_csr(rowptr, colidx, vals, dinv, dinv, Xh), rowptr, colidx, vals, dinv, Xh, 1)


def test_strided_input_and_order_zero(dev):
    n = 3000
    r, c = _hub_graph(n, seed=77)
    adj = _na().from_coo(r, c, None, n, dev)
    rowptr, colidx, vals, deg, dinv = orc.norm_csr(r, c, np.ones(r.size, np.float32), n)
    big = torch.randn(n, 200, device=dev)
    Xv = big[:, 8:136]                                    # ld = 200, F = 128, 32-byte offset
    Y = adj.propagate(Xv, 2)
    Xh = Xv.cpu().numpy()
    exp = orc.spmm_csr(rowptr, colidx, vals, dinv, dinv, orc.spmm_csr(rowptr, colidx, vals, dinv, dinv, Xh))
    _check_Y(Y, exp, rowptr, colidx, vals, dinv, Xh, 2)
    Xo = big[:, 3:10]                                     # 4-byte aligned only
    Y = adj.propagate(Xo, 1)
    _check_Y(Y, orc.spmm_csr(rowptr, colidx, vals, dinv, dinv, Xo.cpu().numpy()), rowptr, colidx, vals, dinv,
             Xo.cpu().numpy(), 1)
    assert adj.propagate(big, 0) is big
    with pytest.raises(ValueError):
        adj.propagate(big[:10], 1)
    with pytest.raises(ValueError):
        adj.propagate(big.double(), 1)


def test_size_independent_properties_mid_scale(dev):
    """Identities that hold at any size: A_hat . D^1/2 1 = D^1/2 1, linearity, and
    the weighted checksum (D^1/2 1)^T A_hat X = (D^1/2 1)^T X for symmetric A."""
    from sgnn_amd.graphgen import rmat_undirected_device
    n, F = 200_000, 96
    row, col = rmat_undirected_device(n, 8_000_000, seed=5, device=dev)
    adj = _na().from_coo(row, col, None, n, dev)
    assert adj.symmetric
    s = adj.deg.sqrt()
    Y = adj.propagate(s[:, None].expand(n, 4).contiguous(), 1)
    assert float(((Y - s[:, None]).abs() / s[:, None]).max()) <= TOL
    g = torch.Generator(device=dev); g.manual_seed(1)
    X = torch.randn(n, F, device=dev, generator=g)
    Z = torch.randn(n, F, device=dev, generator=g)
    lhs = adj.propagate(2.0 * X - 0.5 * Z, 1)
    rhs = 2.0 * adj.propagate(X, 1) - 0.5 * adj.propagate(Z, 1)
    assert float((lhs - rhs).norm() / rhs.norm()) <= TOL
    YX = adj.propagate(X, 1)
    chk_l = (s.double()[:, None] * YX.double()).sum(0)
    chk_r = (s.double()[:, None] * X.double()).sum(0)
    absum = (s.double()[:, None] * X.double().abs()).sum(0)
    assert float(((chk_l - chk_r).abs() / absum).max()) <= TOL


# ----------------------------------------------------------------- operators
def test_transpose_matches_scipy(dev):
    g = load_golden("rmat1024_directed_raw.npz")
    n = int(g["n"])
    adj = _na().from_coo(g["row"], g["col"], g["val"], n, dev)
    assert not adj.symmetric
    T = adj.transpose()
    M = adj.to_scipy().T.tocsr()
    M.sort_indices()
    assert np.array_equal(T.rowptr.cpu().numpy(), M.indptr)
    assert np.array_equal(T.colidx.cpu().numpy(), M.indices)
    assert np.array_equal(T.vals.cpu().numpy(), M.data)
    assert T.transpose() is adj


def test_sgconv_gcnconv_vs_pyg_semantics(dev):
    from sgnn_amd.nn import GCNConv, SGConv
    g = load_golden("rmat1024_directed_raw.npz")
    n = int(g["n"])
    ei = torch.from_numpy(np.stack((g["row"], g["col"])).astype(np.int64)).to(dev)
    X = hashed_features(n, 24, seed=3)
    Xd = torch.from_numpy(X).to(dev)
    torch.manual_seed(0)
    conv = SGConv(24, 5, K=2).to(dev)
    out = conv(Xd, ei)
    exp = orc.pyg_sgconv(X, ei.cpu().numpy(), conv.lin.weight.detach().cpu().numpy(),
                         conv.lin.bias.detach().cpu().numpy(), K=2)
    assert rel_err(out.detach().cpu().numpy(), exp) <= TOL
    gcn = GCNConv(24, 6).to(dev)
    with torch.no_grad():
        gcn.bias.uniform_(-1, 1)
    out = gcn(Xd, ei)
    exp = orc.pyg_gcnconv(X, ei.cpu().numpy(), gcn.lin.weight.detach().cpu().numpy(), gcn.bias.detach().cpu().numpy())
    assert rel_err(out.detach().cpu().numpy(), exp) <= TOL
    assert set(dict(gcn.named_parameters())) == {"lin.weight", "bias"}
    assert set(dict(conv.named_parameters())) == {"lin.weight", "lin.bias"}


def test_gcnconv_backward_vs_dense_autograd(dev):
    """A_hat^T . dY: the gradient w.r.t. the layer input and weights equals dense
    autograd through the same normalised matrix (directed graph, so A_hat^T != A_hat)."""
    from sgnn_amd.nn import GCNConv
    g = load_golden("rmat1024_directed_raw.npz")
    n = int(g["n"])
    ei_np = np.stack((g["row"], g["col"])).astype(np.int64)
    ei = torch.from_numpy(ei_np).to(dev)
    src, dst, norm = orc.pyg_gcn_norm(ei_np, None, n)
    M = torch.zeros(n, n, dtype=torch.float64)
    M.index_put_((torch.from_numpy(dst), torch.from_numpy(src)), torch.from_numpy(norm).double(), accumulate=True)
    torch.manual_seed(1)
    gcn = GCNConv(12, 4).to(dev)
    x = torch.randn(n, 12, device=dev, requires_grad=True)
    w_out = torch.randn(n, 4, device=dev)
    (gcn(x, ei) * w_out).sum().backward()
    xd = x.detach().cpu().double().requires_grad_(True)
    W = gcn.lin.weight.detach().cpu().double().requires_grad_(True)
    b = gcn.bias.detach().cpu().double().requires_grad_(True)
    ((M @ (xd @ W.t()) + b) * w_out.cpu().double()).sum().backward()
    assert rel_err(x.grad.cpu().numpy(), xd.grad.numpy()) <= TOL
    assert rel_err(gcn.lin.weight.grad.cpu().numpy(), W.grad.numpy()) <= 1e-4     # dense fp32 GEMM on top
    assert rel_err(gcn.bias.grad.cpu().numpy(), b.grad.numpy()) <= 1e-4


def test_rowgroup_entry_point_rejects_what_it_cannot_run(dev):
    """Direct C-ABI calls: unaligned X, in-kernel column scaling, short workspace and
    mismatched item counts are refused with a message, nothing is launched."""
    from sgnn_amd import _lib
    from sgnn_amd.graphgen import symmetric_simple_numpy
    lib = _lib.load()
    n, F = 2000, 200
    r, c = symmetric_simple_numpy(n, 30_000, seed=12)
    adj = _na().from_coo(r, c, None, n, dev)
    plan = adj._rowgroup_plan(F)
    assert plan is not None
    X = torch.randn(n, F + 4, device=dev)
    Y = torch.empty(n, F, device=dev)
    ws = adj._rowgroup_ws(plan, F)
    st = _lib.current_stream(dev)

    def call(x_ptr, ldx, dinv_col=None, items=plan["items"], ws_bytes=ws.numel()):
        return lib.sgnn_spmm_rowgroup_f32(n, n, adj.nnz, F, plan["R"], _lib.ptr(plan["buf"]), plan["bytes"], None,
                                          items, plan["slots"], plan["n_long"], _lib.ptr(adj.dinv), dinv_col,
                                          1.0, None, 0, x_ptr, ldx, _lib.ptr(Y), F, 0, _lib.ptr(ws), ws_bytes, st)
    assert call(X.data_ptr(), F + 4) == 0
    assert call(X.data_ptr() + 4, F + 4) == -4 and b"16-byte" in lib.sgnn_last_error()       # base off by one float
    assert call(X.data_ptr(), F + 3) == -4                                                     # odd row stride
    # column scaling inside the kernel: groups of 8 only (what a graph this small is planned with)
    assert plan["R"] == 8 and call(X.data_ptr(), F + 4, dinv_col=_lib.ptr(adj.dinv)) == 0
    lib.sgnn_tune_set(b"rg_R", 16)
    try:
        p16 = adj._rowgroup_plan(F)
        ws16 = adj._rowgroup_ws(p16, F)
        assert p16["R"] == 16
        rc = lib.sgnn_spmm_rowgroup_f32(n, n, adj.nnz, F, 16, _lib.ptr(p16["buf"]), p16["bytes"], None, p16["items"],
                                        p16["slots"], p16["n_long"], _lib.ptr(adj.dinv), _lib.ptr(adj.dinv), 1.0, None, 0,
                                        X.data_ptr(), F + 4, _lib.ptr(Y), F, 0, _lib.ptr(ws16), ws16.numel(), st)
        assert rc == -4 and b"only R = 8" in lib.sgnn_last_error()
    finally:
        lib.sgnn_tune_set(b"rg_R", 0)
    assert call(X.data_ptr(), F + 4, items=plan["items"] * 100 + 7) == -1
    if plan["slots"] > 0:
        assert call(X.data_ptr(), F + 4, ws_bytes=16) == -2
    assert call(X.data_ptr(), F - 1) == -1                                                     # ldx < F
    torch.cuda.synchronize()
    # the Python layer stages such inputs itself: an unaligned view gives the same result as a copy
    Xu = X[:, 1:F + 1]
    assert Xu.data_ptr() % 16 != 0
    a = adj.spmm(Xu)
    b = adj.spmm(Xu.contiguous())
    assert torch.equal(a, b)


def test_sgconv_reordered_matches_literal_order(dev):
    """A_hat^K (x W^T) against (A_hat^K x) W^T: outputs and parameter gradients of the two
    SGConv evaluation orders agree to fp32 rounding; the cached variant too."""
    from sgnn_amd.nn import SGConv
    from sgnn_amd.graphgen import symmetric_simple_numpy
    n, Fin, Fout = 1500, 96, 7
    r, c = symmetric_simple_numpy(n, 12_000, seed=31)
    ei = torch.from_numpy(np.stack((r, c))).to(dev)
    x = torch.randn(n, Fin, device=dev)
    tgt = torch.randn(n, Fout, device=dev)
    torch.manual_seed(0)
    a = SGConv(Fin, Fout, K=2, reorder=True).to(dev)
    b = SGConv(Fin, Fout, K=2, reorder=False).to(dev)
    cch = SGConv(Fin, Fout, K=2, cached=True).to(dev)
    for m in (b, cch):
        m.load_state_dict(a.state_dict())
    outs, grads = [], []
    for m in (a, b, cch):
        out = m(x, ei)
        ((out - tgt) ** 2).mean().backward()
        outs.append(out.detach())
        grads.append((m.lin.weight.grad.clone(), m.lin.bias.grad.clone()))
    for o in outs[1:]:
        assert float((o - outs[0]).norm() / outs[0].norm()) <= 1e-5
    for gw, gb in grads[1:]:
        assert float((gw - grads[0][0]).norm() / grads[0][0].norm()) <= 1e-5
        assert float((gb - grads[0][1]).norm() / grads[0][1].norm()) <= 1e-5
