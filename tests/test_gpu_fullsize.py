"""Full-size checks on the benchmark shapes, through properties that need no CPU
reference of that size:

* the three independent kernels (plain CSR, row-group, tile-group on the matrix cores) agree
  to fp32 rounding;
* A_hat . D^1/2 1 = D^1/2 1;  linearity;  the weighted checksum
  (D^1/2 1)^T A_hat X = (D^1/2 1)^T X  (symmetric A);
* K hops as one chain equal K single hops;
* X spanning more than 4 GiB (64-bit addressing path) gives the same rows as a
  compact copy of the same data.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _graph(shape, dev, seed=1):
    from sgnn_amd import NormAdj
    from sgnn_amd.graphgen import SHAPES, rmat_undirected_device
    n, E, F = SHAPES[shape]
    row, col = rmat_undirected_device(n, E, seed=seed, device=dev)
    adj = NormAdj.from_coo(row, col, None, n, dev)
    return adj, n, F


def test_reddit_shape_full_size_properties(gpu_device):
    dev = gpu_device
    adj, n, F = _graph("reddit", dev)
    assert adj.nnz > 114_000_000 and adj.unit_weights and adj.symmetric
    # rowptr / degree invariants of the build at this size
    counts = (adj.rowptr[1:] - adj.rowptr[:-1])
    assert int(adj.rowptr[0]) == 0 and int(adj.rowptr[-1]) == adj.nnz and int(counts.min()) >= 1
    assert torch.equal(adj.deg, counts.float())                      # unit weights: degree = row length
    assert torch.equal(adj.dinv, 1.0 / counts.float().sqrt())
    g = torch.Generator(device=dev); g.manual_seed(3)
    X = torch.randn(n, F, device=dev, generator=g)
    adj.use_rowgroup = True
    Y_tg = adj.propagate(X, 1)                   # this shape is what the tile-group kernel is chosen for
    assert adj._tg, "tile-group plan was not used"
    assert int(adj._tg_bufs[F]["state"][1].item()) == 0, "the fp16 planes refused ordinary data"
    adj.use_tilegroup = False
    Y_rg = adj.propagate(X, 1)
    assert adj._plans, "row-group plan was not used"
    adj.use_rowgroup = False
    Y_csr = adj.propagate(X, 1)
    # element-wise against |A_hat| |X|
    scale = adj.propagate(X.abs(), 1)
    for Y in (Y_rg, Y_tg):
        assert float((Y - Y_csr).norm() / Y_csr.norm()) <= TOL
        assert float(((Y - Y_csr).abs() / scale.clamp_min(1e-30)).max()) <= TOL
    adj.use_rowgroup = True
    adj.use_tilegroup = None
    # narrow widths (SGC's 41 classes, GCN's hidden 16) take the tile-group kernel with fewer
    # N-tiles per slice; a two-hop chain against the plain kernel
    for Fn in (16, 41):
        Xn = X[:, :Fn].contiguous()
        Y_n = adj.propagate(Xn, 2)
        assert int(adj._tg_bufs[Fn]["state"][1].item()) == 0
        adj.use_rowgroup = False
        Y_p = adj.propagate(Xn, 2)
        sc = adj.propagate(adj.propagate(Xn.abs(), 1), 1)
        adj.use_rowgroup = True
        assert float((Y_n - Y_p).norm() / Y_p.norm()) <= TOL
        assert float(((Y_n - Y_p).abs() / sc.clamp_min(1e-30)).max()) <= TOL
    Y_rg = Y_tg                                  # the properties below are checked on the kernel the benchmark runs
    s = adj.deg.sqrt()
    fixed = adj.propagate(s[:, None].expand(n, 64).contiguous(), 1)
    assert float(((fixed - s[:, None]).abs() / s[:, None]).max()) <= TOL
    chk_l = (s.double()[:, None] * Y_rg.double()).sum(0)
    chk_r = (s.double()[:, None] * X.double()).sum(0)
    absum = (s.double()[:, None] * X.double().abs()).sum(0)
    assert float(((chk_l - chk_r).abs() / absum).max()) <= TOL
    Z = torch.randn(n, F, device=dev, generator=g)
    lhs = adj.propagate(1.5 * X - 0.25 * Z, 1)
    rhs = 1.5 * Y_rg - 0.25 * adj.propagate(Z, 1)
    assert float((lhs - rhs).norm() / rhs.norm()) <= TOL
    chain = adj.propagate(X, 2)
    hops = adj.propagate(Y_rg, 1)
    assert float((chain - hops).norm() / hops.norm()) <= TOL
    # three hops: the operands of the later hops hold a hundred-odd elements at zero crossings that
    # the planes keep only to 2^-50 of their column's largest magnitude; that is no reason to leave
    # the matrix cores (the flag words of all three operands stay 0)
    chain3 = adj.propagate(X, 3)
    assert adj._tg_bufs[F]["states"][:3, 1].tolist() == [0, 0, 0]
    hops3 = adj.propagate(hops, 1)
    assert float((chain3 - hops3).norm() / hops3.norm()) <= TOL


def test_reddit_shape_weight2_full_size_three_kernels_agree(gpu_device):
    """The Reddit configuration as the reference loads it: ``adj + adj.T`` makes every stored
    entry 2.0 (data_loader.py:75), ``A + I`` then puts 1.0 on the diagonal (utils.py:86) -- the
    kernel instances with the diagonal correction, at the benchmark's size, widths 602 and 41:
    tile-group, row-group and plain CSR agree normwise and element by element against
    ``|A_hat| |X|``, and the weighted fixed point ``A_hat D^1/2 1 = D^1/2 1`` holds."""
    from sgnn_amd import NormAdj
    from sgnn_amd.graphgen import SHAPES, rmat_undirected_device
    dev = gpu_device
    n, E, _ = SHAPES["reddit"]
    row, col = rmat_undirected_device(n, E, seed=1, device=dev)
    val = torch.full((row.numel(),), 2.0, device=dev)
    adj = NormAdj.from_coo(row, col, val, n, dev)
    del row, col, val
    assert adj.w_off == 2.0 and adj.corr is not None and not adj.unit_weights
    counts = (adj.rowptr[1:] - adj.rowptr[:-1]).float()
    assert torch.equal(adj.deg, 2.0 * (counts - 1.0) + 1.0)          # weighted row sum, exact in fp32
    g = torch.Generator(device=dev); g.manual_seed(4)
    for F in (602, 41):
        X = torch.randn(n, F, device=dev, generator=g)
        adj.use_tilegroup, adj.use_rowgroup = None, True
        Y_tg = adj.propagate(X, 1)
        assert adj._tg, "tile-group plan was not used"
        assert int(adj._tg_bufs[F]["state"][1].item()) == 0
        adj.use_tilegroup = False
        Y_rg = adj.propagate(X, 1)
        adj.use_rowgroup = False
        Y_csr = adj.propagate(X, 1)
        scale = adj.propagate(X.abs(), 1)
        for Y in (Y_rg, Y_tg):
            assert float((Y - Y_csr).norm() / Y_csr.norm()) <= TOL
            assert float(((Y - Y_csr).abs() / scale.clamp_min(1e-30)).max()) <= TOL
        del X, Y_tg, Y_rg, Y_csr, scale
    adj.use_tilegroup, adj.use_rowgroup = None, True
    s = adj.deg.sqrt()
    fixed = adj.propagate(s[:, None].expand(n, 64).contiguous(), 1)
    assert float(((fixed - s[:, None]).abs() / s[:, None]).max()) <= TOL


@pytest.mark.parametrize("rowgroup", [True, False])
def test_x_larger_than_4gib(gpu_device, rowgroup):
    """Rows of X 16 KiB apart: the matrix spans 4.9 GB, so a row's byte offset does
    not fit 32 bits and the kernels re-base their buffer descriptor per fetched row."""
    from sgnn_amd import NormAdj
    from sgnn_amd.graphgen import rmat_undirected_device
    dev = gpu_device
    n, F, ld = 300_000, 64, 4096
    row, col = rmat_undirected_device(n, 6_000_000, seed=2, device=dev)
    adj = NormAdj.from_coo(row, col, None, n, dev)
    adj.use_rowgroup, adj.use_tilegroup = rowgroup, False      # the two kernels that read X in place
    big = torch.zeros(n, ld, device=dev)
    assert big.numel() * 4 > (1 << 32)
    g = torch.Generator(device=dev); g.manual_seed(5)
    Xc = torch.randn(n, F, device=dev, generator=g)
    big[:, 100:100 + F] = Xc
    Y_big = adj.spmm(big[:, 100:100 + F])
    Y_ref = adj.spmm(Xc)
    assert float((Y_big - Y_ref).norm() / Y_ref.norm()) <= 1e-6
    assert torch.equal(Y_big, Y_ref)            # same kernel arithmetic, only the addressing differs


@pytest.mark.parametrize("weighted", [False, True])
def test_rowgroup_kernel_reads_x_larger_than_4gib_in_place(gpu_device, weighted):
    """The row-group kernel itself on an operand that spans more than 4 GiB (already
    scaled and 16-byte aligned, so nothing is staged into a compact copy): same result as
    on a compact copy of the same rows."""
    from sgnn_amd import NormAdj, _lib
    from sgnn_amd.graphgen import rmat_undirected_device
    dev = gpu_device
    n, F, ld = 300_000, 96, 4096
    row, col = rmat_undirected_device(n, 6_000_000, seed=4, device=dev)
    g = torch.Generator(device=dev); g.manual_seed(6)
    weights = None if not weighted else torch.rand(row.numel(), device=dev, generator=g) + 0.5
    adj = NormAdj.from_coo(row, col, weights, n, dev)
    assert (adj.w_off is None) == weighted
    big = torch.zeros(n, ld, device=dev)
    assert big.numel() * 4 > (1 << 32)
    Xc = torch.randn(n, F, device=dev, generator=g)
    big[:, 128:128 + F] = Xc
    view = big[:, 128:128 + F]
    assert view.data_ptr() % 16 == 0 and adj._rowgroup_plan(F) is not None
    Y_big = torch.empty(n, F, device=dev)
    Y_ref = torch.empty(n, F, device=dev)
    adj._hop(view, ld, Y_big, F, F, True, False, _lib.EPI_SCALE_ROW)
    adj._hop(Xc, F, Y_ref, F, F, True, False, _lib.EPI_SCALE_ROW)
    assert torch.equal(Y_big, Y_ref)
    # and it is the scaled-input product: dinv * (A+I) * X
    adj.use_rowgroup = False
    Y_csr = adj.spmm(Xc, scale_rows=True, scale_cols=False)
    assert float((Y_ref - Y_csr).norm() / Y_csr.norm()) <= TOL
