"""How much the tile-group kernel's 16-bit operand planes may change a result, pinned.

The kernel feeds fp32 operands to the matrix cores as two fp16 planes and accumulates in fp32
(reference arithmetic: the fp32 ``adj.mm`` of utils.py:94).  Every operand is either carried
accurately -- each element to 2^-21 relative -- or refused: the device flag goes up and the
plain CSR kernel computes the result, bit for bit.  Nothing in between.

"Accurately" is measured against the exact (float64) product, element by element in units of
``|A_hat| |X|``, and next to the plain fp32 CSR kernel's own distance from it (an fp32 sum of k
terms in any order is off by up to ~sqrt(k) 2^-24 of that yardstick, more where one term
dominates): the tile-group result may be worse than the plain kernel's by the planes' input
rounding, 2^-21 per hop, and no more.  On ordinary data that keeps the two kernels within 2^-20
of each other, which is asserted as well."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BOUND = 2.0 ** -20


def _adj(dev, n=20_000, edges=2_000_000, seed=3):
    from sgnn_amd import NormAdj
    from sgnn_amd.graphgen import rmat_undirected_device
    row, col = rmat_undirected_device(n, edges, seed=seed, device=dev)
    adj = NormAdj.from_coo(row, col, None, n, dev)
    return adj


def _exact(adj, X, order):
    """A_hat^order . X in float64 on the host (scipy), with the fp32 dinv the kernels use."""
    import numpy as np
    import scipy.sparse as sp
    A = adj.to_scipy().astype(np.float64)
    d = sp.diags(adj.dinv.double().cpu().numpy())
    Ah = (d @ A @ d).tocsr()
    Y = X.double().cpu().numpy()
    for _ in range(order):
        Y = Ah @ Y
    return torch.from_numpy(Y).to(X.device)


def _three(adj, X, order=1):
    """(tile-group result, fall-back flag, plain CSR result, |A_hat|^order |X|, exact result, hops)."""
    F = X.shape[1]
    adj.use_tilegroup, adj.use_rowgroup = True, True
    Y = adj.propagate(X, order)
    torch.cuda.synchronize()
    flag = int(adj._tg_bufs[F]["state"][1].item())
    adj.use_tilegroup, adj.use_rowgroup = False, False
    Y_ref = adj.propagate(X, order)
    scale = adj.propagate(X.abs(), order)
    adj.use_tilegroup, adj.use_rowgroup = True, True
    return Y, flag, Y_ref, scale, _exact(adj, X, order), order


def _accurate_or_refused(Y, flag, Y_ref, scale, exact, order=1, close=False):
    if flag != 0:
        assert torch.equal(torch.isfinite(Y), torch.isfinite(Y_ref))
        fin = torch.isfinite(Y_ref)
        assert torch.equal(Y[fin], Y_ref[fin]), "a refused operand must be served by the plain kernel, bit for bit"
        return "refused"
    assert bool(torch.isfinite(Y).all())
    sc = scale.double()
    live = sc > 0
    # elements whose terms are all zero must be exactly zero
    assert not bool((Y[~live] != 0).any())
    if not bool(live.any()):
        return "accurate"
    e_tg = float(((Y.double() - exact).abs() / sc.clamp_min(1e-300))[live].max())
    e_pl = float(((Y_ref.double() - exact).abs() / sc.clamp_min(1e-300))[live].max())
    assert e_tg <= e_pl + order * 2.0 ** -21, (
        f"tile-group result is {e_tg:.3e} of |A_hat||X| from the exact product, the plain fp32 kernel {e_pl:.3e}")
    if close:
        both = float(((Y.double() - Y_ref.double()).abs() / sc.clamp_min(1e-300))[live].max())
        assert both <= order * BOUND, f"the two kernels differ by {both:.3e} of |A_hat||X|"
    return "accurate"


@pytest.mark.parametrize("F,order", [(256, 1), (602, 1), (41, 2), (128, 3)])
def test_planes_bound_on_normal_data(gpu_device, F, order):
    adj = _adj(gpu_device)
    g = torch.Generator(device=gpu_device); g.manual_seed(F)
    X = torch.randn(adj.n, F, device=gpu_device, generator=g)
    Y, flag, Y_ref, scale, exact, _ = _three(adj, X, order)
    assert flag == 0, "N(0,1) data must stay on the matrix cores"
    assert _accurate_or_refused(Y, flag, Y_ref, scale, exact, order, close=True) == "accurate"


def test_all_zero_operand(gpu_device):
    adj = _adj(gpu_device, n=6000, edges=300_000)
    X = torch.zeros(adj.n, 130, device=gpu_device)
    Y, flag, Y_ref, scale, exact, _ = _three(adj, X)
    assert flag == 0
    assert torch.equal(Y, torch.zeros_like(Y)) and torch.equal(Y_ref, Y)


def test_one_nonzero_element(gpu_device):
    adj = _adj(gpu_device, n=6000, edges=300_000)
    X = torch.zeros(adj.n, 96, device=gpu_device)
    X[17, 5] = 3.25
    Y, flag, Y_ref, scale, exact, _ = _three(adj, X, order=2)
    assert flag == 0
    assert _accurate_or_refused(Y, flag, Y_ref, scale, exact, 2) == "accurate"
    assert bool((Y[:, :5] == 0).all()) and bool((Y[:, 6:] == 0).all())


@pytest.mark.parametrize("spread", [8, 14, 28, 60])
def test_columns_of_very_different_scale(gpu_device, spread):
    """Feature columns scaled by powers of two over 2^-spread .. 2^spread (un-normalised
    feature matrices look like this).  Columns never mix in A_hat . X, so each column's result
    must be as accurate relative to that column's own magnitudes as if it were alone."""
    adj = _adj(gpu_device)
    F = 200
    g = torch.Generator(device=gpu_device); g.manual_seed(spread)
    e = torch.randint(-spread, spread + 1, (1, F), device=gpu_device, generator=g).float()
    X = torch.randn(adj.n, F, device=gpu_device, generator=g) * torch.exp2(e)
    Y, flag, Y_ref, scale, exact, _ = _three(adj, X)
    _accurate_or_refused(Y, flag, Y_ref, scale, exact)   # per element, i.e. per column's own scale
    if spread <= 28:
        assert flag == 0, "column scaling inside fp32's comfortable range must not cost the fast path"


def test_rows_of_very_different_scale_are_accurate_or_refused(gpu_device):
    adj = _adj(gpu_device)
    g = torch.Generator(device=gpu_device); g.manual_seed(5)
    for spread in (6, 14, 30):
        e = torch.randint(-spread, spread + 1, (adj.n, 1), device=gpu_device, generator=g).float()
        X = torch.randn(adj.n, 128, device=gpu_device, generator=g) * torch.exp2(e)
        _accurate_or_refused(*_three(adj, X))


def test_subnormal_operand(gpu_device):
    adj = _adj(gpu_device, n=6000, edges=300_000)
    g = torch.Generator(device=gpu_device); g.manual_seed(7)
    X = torch.randn(adj.n, 128, device=gpu_device, generator=g) * 1e-40
    assert float(X.abs().max()) < 1.2e-38            # every element is an fp32 subnormal
    _accurate_or_refused(*_three(adj, X))


def test_operand_near_the_top_of_the_fp32_range(gpu_device):
    """Largest magnitude above 2^126 (the scale that brings it into the planes' range is 2^-112 and
    its inverse 2^112), the bulk of the operand at 2^110 so that the fp32 sums themselves stay finite."""
    adj = _adj(gpu_device, n=6000, edges=300_000)
    g = torch.Generator(device=gpu_device); g.manual_seed(8)
    X = torch.randn(adj.n, 128, device=gpu_device, generator=g) * 2.0 ** 110
    X[5, 7] = 1.5 * 2.0 ** 126
    X[4000, 100] = -1.25 * 2.0 ** 126
    assert float(X.abs().max()) >= 2.0 ** 126
    Y, flag, Y_ref, scale, exact, _ = _three(adj, X)
    assert bool(torch.isfinite(Y_ref).all())
    _accurate_or_refused(Y, flag, Y_ref, scale, exact)
