"""End-to-end, clustering branch: the stacked auto-encoder model trained above the HIP
propagation and block-sampling path reaches the clustering quality the reference
reaches on the same graph.

Golden: tests/golden/e2e_gae_planted.npz, produced by tests/golden/make_golden_gae.py
from the reference's own StackedGNN on the CPU (seeds 0..2: k-means NMI 0.85-0.93,
ACC 0.87-0.98; k-means on the raw features gives NMI 0.59).  Training is stochastic
and the random streams differ, so the comparison is statistical: mean NMI within
0.06 of the reference's mean and clearly above the raw-feature NMI.
"""
import json

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from golden_util import load_golden

pytestmark = pytest.mark.gpu


def _layers(cfg):
    from sgnn_amd.stacked import LayerParam, get_activation
    return [LayerParam(l["neurons"], inner_act=get_activation(l["inner_act"]), act=get_activation(l["activation"]),
                       gnn_type=LayerParam.GAE, mask_rate=cfg["mask_rate"], lam=cfg["lam"], max_iter=l["max_iter"],
                       learning_rate=l["learning_rate"], batch_size=l["batch_size"]) for l in cfg["layers"]]


def test_stacked_gae_matches_reference_clustering_quality(gpu_device):
    from sgnn_amd.stacked import StackedGNN, k_means
    graph = load_golden("e2e_sgnn_planted.npz")
    g = load_golden("e2e_gae_planted.npz")
    cfg = json.loads(str(g["config"]))
    X, y = graph["X"], graph["y"]
    n = X.shape[0]
    A = sp.coo_matrix((np.ones(graph["row"].size, np.float32), (graph["row"], graph["col"])), shape=(n, n)).tocsr()
    # the metric code reproduces the stored numbers from the stored reference embedding
    acc, nmi = k_means(g["emb_seed0"], 4, y, replicates=3, seed=0)
    assert abs(nmi - float(g["ref_nmi"][0])) < 1e-6 and abs(acc - float(g["ref_acc"][0])) < 1e-6
    seen = []
    nmis, accs = [], []
    for seed in range(3):
        torch.manual_seed(seed)
        model = StackedGNN(X, A, _layers(cfg), overlooked_rates=None, BP_count=cfg["BP_count"], eta=cfg["eta"],
                           device=gpu_device, labels=y, metric_func=lambda e, l: seen.append(e.shape))
        emb = model.run()
        assert emb.is_cuda and emb.shape == (n, cfg["layers"][-1]["neurons"]) and bool(torch.isfinite(emb).all())
        a, m = k_means(emb.cpu().numpy(), 4, y, replicates=3, seed=0)
        accs.append(a); nmis.append(m)
    assert len(seen) == 3 * (cfg["BP_count"] + 1)            # metric hook after every forward sweep
    ref = float(g["ref_nmi"].mean())
    assert np.mean(nmis) >= ref - 0.06, (nmis, g["ref_nmi"])
    assert np.mean(nmis) >= float(g["raw_nmi"]) + 0.15, (nmis, float(g["raw_nmi"]))
    assert np.mean(accs) >= float(g["ref_acc"].mean()) - 0.08, (accs, g["ref_acc"])


def test_gae_with_overlooked_edges_and_loss_pieces(gpu_device):
    from sgnn_amd.stacked import GAELayer, StackedGNN
    graph = load_golden("e2e_sgnn_planted.npz")
    cfg = json.loads(str(load_golden("e2e_gae_planted.npz")["config"]))
    for l in cfg["layers"]:
        l["max_iter"] = 5
    X = graph["X"]
    n = X.shape[0]
    A = sp.coo_matrix((np.ones(graph["row"].size, np.float32), (graph["row"], graph["col"])), shape=(n, n)).tocsr()
    torch.manual_seed(0)
    model = StackedGNN(X, A, _layers(cfg), overlooked_rates=[0.3, 0.0], BP_count=1, eta=cfg["eta"],
                       device=gpu_device)
    look = model.gnns[0].overlook
    assert look is not None and model.gnns[1].overlook is None
    L = look.to_scipy()
    # symmetric 0/1, unit diagonal, off-diagonal entries are edges of A (utils.py:33-51)
    assert (abs(L - L.T)).nnz == 0 and set(np.unique(L.data)) == {1.0} and np.all(L.diagonal() == 1.0)
    off = L - sp.identity(n, format="csr")
    off.eliminate_zeros()
    assert off.nnz > 0 and (off.multiply(A) - off).nnz == 0
    emb = model.run()
    assert bool(torch.isfinite(emb).all())
    # loss restatement (model/SGNN.py:682-702) against the formula evaluated literally in fp64
    gen = torch.Generator().manual_seed(3)
    rec = torch.rand(40, 40, generator=gen).clamp(1e-3, 1 - 1e-3)
    tgt = (torch.rand(40, 40, generator=gen) < 0.1).float()
    keep = (torch.rand(40, 40, generator=gen) < 0.8).float()
    got = GAELayer.reconstruction_loss(rec.to(gpu_device), tgt.to(gpu_device), keep.to(gpu_device)).item()
    r, t, k = rec.double(), tgt.double(), keep.double()
    ratio = k.sum() / (t * k).sum() - 1
    exp = (-ratio * t * torch.log(r) * k).mean() + (-(1 - t) * torch.log(1 - r) * k).mean()
    assert abs(got - exp.item()) <= 1e-5 * abs(exp.item())
    zero = GAELayer.reconstruction_loss(rec.to(gpu_device), torch.zeros(40, 40, device=gpu_device),
                                        keep.to(gpu_device)).item()
    exp0 = (-torch.log(1 - r) * k).mean().item()
    assert abs(zero - exp0) <= 1e-5 * abs(exp0)


def test_fused_reconstruction_matches_autograd(gpu_device):
    """The fused loss kernels against the plain-torch statement of the same loss
    (GAELayer.reconstruction_loss under autograd), fed the target and retain blocks the
    kernels produced: loss and dloss/dZ within 1e-5 relative; the blocks themselves are
    checked against the oracle's sub-adjacency and the reference's mask construction."""
    import scipy.sparse as sp
    from oracle import sgnn_oracle as orc
    from sgnn_amd.norm_adj import NormAdj
    from sgnn_amd.stacked import GAELayer, _FusedReconstruction
    graph = load_golden("e2e_sgnn_planted.npz")
    n = graph["X"].shape[0]
    A = sp.coo_matrix((np.ones(graph["row"].size, np.float32), (graph["row"], graph["col"])), shape=(n, n)).tocsr()
    adj = NormAdj.from_scipy(A, gpu_device)
    rng = np.random.default_rng(0)
    pick = rng.choice(A.nnz, 800, replace=False)
    coo = A.tocoo()
    look = NormAdj.from_coo(coo.row[pick], coo.col[pick], None, n, gpu_device, symmetrize=True, clamp_one=True,
                            strip_diag=True)
    L = look.to_scipy()
    for b, d, rate, overlook in [(256, 32, 0.2, None), (200, 16, 0.0, look), (64, 8, 1.0, None), (300, 64, 0.35, look)]:
        gen = torch.Generator().manual_seed(b)
        idx = torch.randperm(n, generator=gen)[:b].to(gpu_device)
        Z = (torch.randn(b, d, generator=gen) * (3.0 if b == 256 else 0.7)).to(gpu_device).requires_grad_(True)
        state = torch.tensor([99, 5, 0], dtype=torch.int64, device=gpu_device)
        loss, target, retain = _FusedReconstruction.apply(Z, adj, overlook, idx, rate, state)
        loss.backward()
        got_grad = Z.grad.clone()
        assert int(state[1]) == 6
        i = idx.cpu().numpy()
        assert np.array_equal(target.cpu().numpy(), orc.sample_block(A, i))            # diagonal-free A
        R = retain.cpu().numpy()
        assert np.array_equal(R, R.T) and not R.diagonal().any() and set(np.unique(R)) <= {0.0, 1.0}
        off = ~np.eye(b, dtype=bool)
        if overlook is not None:
            Ls = orc.sample_block(L, i)
            assert not R[(Ls > 0) & off].any()
            free = off & (Ls == 0)
        else:
            free = off
        expect_keep = (1 - rate) ** 2                                                  # neither draw masks the pair
        assert abs(R[free].mean() - expect_keep) <= 0.02, (R[free].mean(), expect_keep)
        # same loss under autograd
        Z2 = Z.detach().clone().requires_grad_(True)
        rec = torch.sigmoid(Z2.matmul(Z2.t()).clamp(-40.0, 40.0))
        ref = GAELayer.reconstruction_loss(rec, target, retain)
        ref.backward()
        assert abs(loss.item() - ref.item()) <= 1e-5 * abs(ref.item()) + 1e-7
        err = (got_grad - Z2.grad).norm() / Z2.grad.norm().clamp_min(1e-12)
        assert float(err) <= 1e-5, float(err)
    # a directed graph: the target block is not symmetric and the general (G + G^T) Z route is taken
    rngd = np.random.default_rng(5)
    dadj = NormAdj.from_coo(rngd.integers(0, n, 40 * n), rngd.integers(0, n, 40 * n), None, n, gpu_device)
    assert not dadj.symmetric
    gen = torch.Generator().manual_seed(11)
    idx = torch.randperm(n, generator=gen)[:192].to(gpu_device)
    Z = (torch.randn(192, 24, generator=gen) * 0.8).to(gpu_device).requires_grad_(True)
    state = torch.tensor([3, 0, 0], dtype=torch.int64, device=gpu_device)
    loss, target, retain = _FusedReconstruction.apply(Z, dadj, None, idx, 0.25, state)
    loss.backward()
    assert not torch.equal(target, target.t())
    Z2 = Z.detach().clone().requires_grad_(True)
    ref = GAELayer.reconstruction_loss(torch.sigmoid(Z2.matmul(Z2.t()).clamp(-40.0, 40.0)), target, retain)
    ref.backward()
    assert abs(loss.item() - ref.item()) <= 1e-5 * abs(ref.item()) + 1e-7
    assert float((Z.grad - Z2.grad).norm() / Z2.grad.norm()) <= 1e-5
    # same (seed, counter) -> same mask; next counter -> another one
    s1 = torch.tensor([1, 0, 0], dtype=torch.int64, device=gpu_device)
    s2 = torch.tensor([1, 0, 0], dtype=torch.int64, device=gpu_device)
    Z = torch.randn(128, 8, device=gpu_device)
    idx = torch.arange(128, device=gpu_device)
    r1 = _FusedReconstruction.apply(Z, adj, None, idx, 0.3, s1)[2]
    r2 = _FusedReconstruction.apply(Z, adj, None, idx, 0.3, s2)[2]
    r3 = _FusedReconstruction.apply(Z, adj, None, idx, 0.3, s2)[2]
    assert torch.equal(r1, r2) and not torch.equal(r2, r3)
