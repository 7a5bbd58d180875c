"""The benchmark configurations that have no golden vector of their own, at their full size.

* PubMed shape (n = 19 717, 88 648 stored entries, F = 500; config.json:254-279, 617-622,
  model/SGC.py:11,14): small enough for the CPU oracle, so everything is compared with it —
  CSR / degrees bit for bit, A_hat^2 X through both kernels, SGConv(500, 3, K=2) in its three
  evaluation orders, and the layer widths 500 -> 256 (order 1) -> 128 (order 2) of the SGNN
  configuration through the drop-in function.  No PubMed data is readable offline (the
  reference ships pickles only), so the structure is a seeded R-MAT of the same size.
* R-MAT 10 M nodes / 200 M entries, F = 128: size-independent properties (two independent
  kernels agree, fixed vector, checksum, linearity, chain = hops).  X is 5.1 GB, i.e. the
  benchmark kernel on the 64-bit addressing path at full scale.
* X between 2 GiB and 4 GiB: a row's byte offset needs all 32 bits.
* X ending exactly where its allocation ends, with the last node as a neighbour.
"""
import numpy as np
import pytest
import torch

from golden_util import abs_scale, rel_err, scaled_err
from oracle import sgnn_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def pubmed(gpu_device):
    from sgnn_amd.graphgen import SHAPES, symmetric_simple_numpy
    n, E, F = SHAPES["pubmed"]
    r, c = symmetric_simple_numpy(n, E // 2, seed=11)
    assert r.size == E
    X = np.random.default_rng(12).standard_normal((n, F)).astype(np.float32)
    return n, F, r, c, X


def test_pubmed_shape_build_and_two_hops_vs_oracle(gpu_device, pubmed):
    from sgnn_amd import NormAdj
    n, F, r, c, X = pubmed
    adj = NormAdj.from_coo(r, c, None, n, gpu_device)
    rowptr, colidx, vals, deg, dinv = orc.norm_csr(r, c, np.ones(r.size, np.float32), n)
    assert np.array_equal(adj.rowptr.cpu().numpy(), rowptr)
    assert np.array_equal(adj.colidx.cpu().numpy(), colidx)
    assert np.array_equal(adj.deg.cpu().numpy(), deg)
    assert np.array_equal(adj.dinv.cpu().numpy(), dinv)
    Xd = torch.from_numpy(X).to(gpu_device)
    exp = X
    for _ in range(2):
        exp = orc.spmm_csr(rowptr, colidx, vals, dinv, dinv, exp)
    scale = abs_scale(rowptr, colidx, vals, dinv, dinv, X, 2)
    for rowgroup in (True, False):
        adj.use_rowgroup = rowgroup
        Y = adj.propagate(Xd, 2).cpu().numpy()
        assert rel_err(Y, exp) <= TOL and scaled_err(Y, exp, scale) <= TOL
    assert adj._plans, "the row-group kernel did not run"


def test_pubmed_shape_sgnn_layer_widths_through_the_drop_in(gpu_device, pubmed):
    """config.json:254-279: layer 1 sees F = 500 at order 1, layer 2 sees 256 at order 2."""
    import scipy.sparse as sp
    from sgnn_amd import utils as su
    n, F, r, c, X = pubmed
    A = sp.coo_matrix((np.ones(r.size, np.float32), (r, c)), shape=(n, n)).tocsr()
    adj = su.csr_to_sparse_Tensor(A, gpu_device)
    ref_adj = orc.csr_to_sparse_tensor(A)
    H = np.random.default_rng(13).standard_normal((n, 256)).astype(np.float32)
    for feats, order in ((X, 1), (H, 2)):
        Y = su.process_data_with_adjacency_high_order(adj, torch.from_numpy(feats).to(gpu_device), gpu_device,
                                                      order=order)
        exp = orc.propagate_high_order(ref_adj, torch.from_numpy(feats), order).numpy()
        assert rel_err(Y.cpu().numpy(), exp) <= TOL


def test_pubmed_shape_sgconv_three_orders_vs_pyg_semantics(gpu_device, pubmed):
    """SGConv(500, 3, K=2) (model/SGC.py:11): linear map first, literal order, cached."""
    from sgnn_amd.nn import SGConv
    n, F, r, c, X = pubmed
    ei_np = np.stack((r, c)).astype(np.int64)
    ei = torch.from_numpy(ei_np).to(gpu_device)
    Xd = torch.from_numpy(X).to(gpu_device)
    torch.manual_seed(0)
    convs = [SGConv(F, 3, K=2, reorder=True), SGConv(F, 3, K=2, reorder=False), SGConv(F, 3, K=2, cached=True)]
    convs = [m.to(gpu_device) for m in convs]
    for m in convs[1:]:
        m.load_state_dict(convs[0].state_dict())
    exp = orc.pyg_sgconv(X, ei_np, convs[0].lin.weight.detach().cpu().numpy(),
                         convs[0].lin.bias.detach().cpu().numpy(), K=2)
    for m in convs:
        out = m(Xd, ei)
        assert out.shape == (n, 3)
        assert rel_err(out.detach().cpu().numpy(), exp) <= TOL
    out2 = convs[2](Xd, ei)                       # second call of the cached operator reuses A_hat^2 X
    assert rel_err(out2.detach().cpu().numpy(), exp) <= TOL


def test_rmat10m_shape_full_size_properties(gpu_device):
    from sgnn_amd import NormAdj
    from sgnn_amd.graphgen import SHAPES, rmat_undirected_device
    dev = gpu_device
    n, E, F = SHAPES["rmat10m"]
    row, col = rmat_undirected_device(n, E, seed=3, device=dev)
    adj = NormAdj.from_coo(row, col, None, n, dev)
    del row, col
    assert adj.nnz >= 200_000_000 and adj.unit_weights
    counts = adj.rowptr[1:] - adj.rowptr[:-1]
    assert int(adj.rowptr[-1]) == adj.nnz and int(counts.min()) >= 1
    assert torch.equal(adj.deg, counts.float())
    g = torch.Generator(device=dev); g.manual_seed(4)
    X = torch.randn(n, F, device=dev, generator=g)
    assert X.numel() * 4 > (1 << 32)              # the operand needs 64-bit row addressing
    adj.use_rowgroup, adj.use_tilegroup = True, False     # (left to itself the handle takes the tile-group kernel: below)
    Y_rg = adj.propagate(X, 1)
    assert adj._plans, "row-group plan was not used"
    adj.use_rowgroup = False
    Y_csr = adj.propagate(X, 1)
    assert float((Y_rg - Y_csr).norm() / Y_csr.norm()) <= TOL
    scale = adj.propagate(X.abs(), 1)
    diff = (Y_rg - Y_csr).abs()
    some = scale > 0                  # 1.3e9 normal draws hold a few exact zeros (isolated nodes keep them)
    assert float(diff[~some].sum()) == 0.0
    assert float((diff[some] / scale[some]).max()) <= TOL
    # The tile-group kernel, which the handle picks for this graph by itself (1.6 stored entries
    # per fetched row): with 10^7 nodes a feature slice of the operand planes spans 5.1 GB, the
    # only case in which that kernel addresses the rows it fetches with 64-bit offsets.
    adj.use_rowgroup, adj.use_tilegroup = True, None
    Y_tg = adj.propagate(X, 1)
    assert adj._tg, "tile-group plan was not used"
    assert n * 512 > (1 << 32)
    assert float((Y_tg - Y_csr).norm() / Y_csr.norm()) <= TOL
    diff = (Y_tg - Y_csr).abs()
    assert float(diff[~some].sum()) == 0.0
    assert float((diff[some] / scale[some]).max()) <= TOL
    del Y_csr, Y_tg, scale, diff, some
    # the identities below on the kernel the handle picks
    s = adj.deg.sqrt()
    fixed = adj.propagate(s[:, None].expand(n, 64).contiguous(), 1)
    assert float(((fixed - s[:, None]).abs() / s[:, None]).max()) <= TOL
    del fixed
    chk_l = (s.double()[:, None] * Y_rg.double()).sum(0)
    chk_r = (s.double()[:, None] * X.double()).sum(0)
    absum = (s.double()[:, None] * X.double().abs()).sum(0)
    assert float(((chk_l - chk_r).abs() / absum).max()) <= TOL
    chain = adj.propagate(X, 2)
    hops = adj.propagate(Y_rg, 1)
    assert float((chain - hops).norm() / hops.norm()) <= TOL


@pytest.mark.parametrize("rowgroup", [True, False])
def test_x_between_2_and_4_gib(gpu_device, rowgroup):
    """Row stride 2 600 floats on 300 000 nodes: X spans 3.1 GB, so the byte offset of a
    fetched row exceeds 2^31 and must be formed without sign."""
    from sgnn_amd import NormAdj, _lib
    from sgnn_amd.graphgen import rmat_undirected_device
    dev = gpu_device
    n, F, ld = 300_000, 96, 2600
    row, col = rmat_undirected_device(n, 6_000_000, seed=4, device=dev)
    adj = NormAdj.from_coo(row, col, None, n, dev)
    adj.use_rowgroup = rowgroup
    big = torch.zeros(n, ld, device=dev)
    assert (1 << 31) < big.numel() * 4 < (1 << 32)
    g = torch.Generator(device=dev); g.manual_seed(6)
    Xc = torch.randn(n, F, device=dev, generator=g)
    big[:, 128:128 + F] = Xc
    view = big[:, 128:128 + F]
    Y_big = torch.empty(n, F, device=dev)
    Y_ref = torch.empty(n, F, device=dev)
    adj._hop(view, ld, Y_big, F, F, True, False, _lib.EPI_SCALE_ROW)     # in place, no staging copy
    adj._hop(Xc, F, Y_ref, F, F, True, False, _lib.EPI_SCALE_ROW)
    assert torch.equal(Y_big, Y_ref)
    assert float(Y_ref.abs().sum()) > 0


@pytest.mark.parametrize("F", [602, 601, 100, 37])
@pytest.mark.parametrize("rowgroup", [True, False])
def test_x_ending_at_the_end_of_its_allocation(gpu_device, F, rowgroup):
    """Every node is a neighbour of the last node, X is a buffer of its own that ends with
    the last row: lanes past the end of a row must not read (their result is checked to be
    unaffected by what follows X: the run is repeated with X placed in front of NaNs)."""
    from sgnn_amd import NormAdj, _lib
    dev = gpu_device
    n = 640
    rng = np.random.default_rng(F)
    r = np.concatenate((rng.integers(0, n, 20 * n), np.arange(n - 1)))
    c = np.concatenate((rng.integers(0, n, 20 * n), np.full(n - 1, n - 1)))
    adj = NormAdj.from_coo(np.concatenate((r, c)), np.concatenate((c, r)), None, n, dev, clamp_one=True)
    adj.use_rowgroup = rowgroup
    ld = (F + 3) // 4 * 4
    X = torch.randn(n, ld, device=dev)
    X[:, F:] = 0
    Y_a = torch.empty(n, F, device=dev)
    adj._hop(X, ld, Y_a, F, F, True, False, _lib.EPI_SCALE_ROW)
    # the same rows in front of a field of NaNs: a lane that read past the end of a row
    # (into the next row's head or past X) would pick them up
    wide = torch.full((n + 8, 2 * ld), float("nan"), device=dev)
    wide[:n, :F] = X[:, :F]
    Y_b = torch.empty(n, F, device=dev)
    adj._hop(wide[:n], 2 * ld, Y_b, F, F, True, False, _lib.EPI_SCALE_ROW)
    assert bool(torch.isfinite(Y_b).all())
    assert torch.equal(Y_a, Y_b)
    ref = orc.spmm_csr(adj.rowptr.cpu().numpy(), adj.colidx.cpu().numpy(), adj.vals.cpu().numpy(),
                       adj.dinv.cpu().numpy(), None, X[:, :F].cpu().numpy())
    assert rel_err(Y_a.cpu().numpy(), ref) <= TOL
