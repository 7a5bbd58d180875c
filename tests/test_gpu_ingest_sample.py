"""GPU parity of the two callers' steps either side of the propagation: the
loaders' symmetrise-and-binarise of a raw edge list (data_loader.py:451-452) done
inside the CSR build, and the dense sub-adjacency block the auto-encoder layer
samples per iteration (model/SGNN.py:167-170).  Both are integer/structure work:
the bar is bit-exact against the scipy restatement in the oracle.
"""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from golden_util import load_golden
from oracle import sgnn_oracle as orc
from sgnn_amd.graphgen import rmat_edges_numpy

pytestmark = pytest.mark.gpu


def _na():
    from sgnn_amd.norm_adj import NormAdj
    return NormAdj


def _assert_same_csr(adj, rowptr, colidx, vals, deg):
    assert np.array_equal(adj.rowptr.cpu().numpy(), rowptr)
    assert np.array_equal(adj.colidx.cpu().numpy(), colidx)
    assert np.array_equal(adj.vals.cpu().numpy(), vals)
    assert np.array_equal(adj.deg.cpu().numpy(), deg)


def _raw_cases():
    g = load_golden("rmat1024_directed_raw.npz")          # directed, duplicates, self-loops, weights
    yield "rmat1024_raw", int(g["n"]), g["row"].astype(np.int64), g["col"].astype(np.int64), g["val"]
    r, c = rmat_edges_numpy(4096, 60_000, seed=21)          # directed multigraph, unit weights
    yield "rmat4096_unit", 4096, r, c, None
    yield "tiny", 5, np.array([0, 0, 1, 3, 3, 2]), np.array([1, 1, 0, 3, 4, 2]), None
    yield "empty", 4, np.zeros(0, np.int64), np.zeros(0, np.int64), None


@pytest.mark.parametrize("strip", [True, False])
def test_symmetrize_binarize_in_build_bit_exact(gpu_device, strip):
    for name, n, r, c, v in _raw_cases():
        A = orc.symmetrize_binarize(r, c, v, n)
        if strip:
            A = orc.remove_self_loop(A).tocsr()
        coo = A.tocoo()
        rowptr, colidx, vals, deg, dinv = orc.norm_csr(coo.row, coo.col, coo.data, n, drop_zeros=True)
        adj = _na().from_coo(r, c, v, n, gpu_device, symmetrize=True, clamp_one=True, strip_diag=strip,
                             drop_zeros=True)
        _assert_same_csr(adj, rowptr, colidx, vals, deg)
        assert adj.symmetric, name


def test_symmetrize_without_clamp_adds_weights(gpu_device):
    g = load_golden("rmat1024_directed_raw.npz")
    n, r, c = int(g["n"]), g["row"].astype(np.int64), g["col"].astype(np.int64)
    v = np.round(g["val"] * 4).astype(np.float32)            # integer weights: sums exact in any order
    A = sp.coo_matrix((v, (r, c)), shape=(n, n), dtype=np.float32)
    S = (A.T + A).tocoo()
    rowptr, colidx, vals, deg, dinv = orc.norm_csr(S.row, S.col, S.data, n)
    adj = _na().from_coo(r, c, v, n, gpu_device, symmetrize=True)
    _assert_same_csr(adj, rowptr, colidx, vals, deg)


@pytest.mark.parametrize("b", [1, 7, 64, 300, 512])
def test_sample_block_bit_exact(gpu_device, b):
    n = 3000
    r, c = rmat_edges_numpy(n, 40_000, seed=5)
    A = orc.remove_self_loop(orc.symmetrize_binarize(r, c, None, n)).tocsr()
    A.eliminate_zeros()
    adj = _na().from_scipy(A, gpu_device)
    gen = torch.Generator().manual_seed(b)
    idx = torch.randperm(n, generator=gen)[:b]
    got = adj.sample_block(idx.to(gpu_device))
    assert got.dtype == torch.float32 and got.shape == (b, b)
    assert np.array_equal(got.cpu().numpy(), orc.sample_block(A, idx.numpy()))
    # stored matrix (A + I) as is: the diagonal of the block is the self-loop weight
    full = adj.sample_block(idx.to(gpu_device), skip_self=False).cpu().numpy()
    assert np.array_equal(full, orc.sample_block(A + sp.identity(n, dtype=np.float32, format="csr"), idx.numpy()))


def test_sample_block_repeats_weights_and_strided_out(gpu_device):
    g = load_golden("rmat1024_directed_raw.npz")
    n = int(g["n"])
    A = sp.coo_matrix((g["val"], (g["row"], g["col"])), shape=(n, n), dtype=np.float32).tocsr()
    A = orc.remove_self_loop(A).tocsr()
    adj = _na().from_scipy(A, gpu_device)
    idx = np.array([5, 17, 5, 900, 17, 0, 1023, 3], dtype=np.int64)      # repeated ids
    exp = orc.sample_block(A, idx)
    # fancy indexing returns A[v, v] = 0 for a repeated id as well (the diagonal is stripped)
    got = adj.sample_block(torch.from_numpy(idx).to(gpu_device))
    assert np.array_equal(got.cpu().numpy(), exp)
    big = torch.full((8, 20), -1.0, device=gpu_device)
    adj.sample_block(torch.from_numpy(idx).to(gpu_device), out=big[:, 4:12])
    assert np.array_equal(big[:, 4:12].cpu().numpy(), exp)
    assert float(big[:, :4].max()) == -1.0 and float(big[:, 12:].max()) == -1.0
    # empty batch
    assert adj.sample_block(torch.zeros(0, dtype=torch.int64, device=gpu_device)).shape == (0, 0)


def test_sample_block_reddit_batch_density(gpu_device):
    """Batch of the Reddit clustering configuration (512 nodes) on a mid-size
    graph: symmetric block, zero diagonal, entry count equals the scipy count."""
    n = 100_000
    r, c = rmat_edges_numpy(n, 2_000_000, seed=9)
    adj = _na().from_coo(r, c, None, n, gpu_device, symmetrize=True, clamp_one=True, strip_diag=True)
    idx = torch.randperm(n, device=gpu_device)[:512]
    blk = adj.sample_block(idx)
    assert torch.equal(blk, blk.t()) and float(blk.diagonal().abs().sum()) == 0.0
    A = adj.to_scipy()
    i = idx.cpu().numpy()
    exp = orc.sample_block(orc.remove_self_loop(A), i)
    assert np.array_equal(blk.cpu().numpy(), exp)


def test_batch_indices_are_a_permutation_head(gpu_device):
    """sgnn_batch_indices stands in for torch.randperm(n)[:b] (model/SGNN.py:115-117):
    distinct ids in range, a full permutation when b = n, a new draw per call, the same
    draw for the same (seed, counter), and no position or id favoured."""
    from sgnn_amd import _lib
    lib = _lib.load()

    def draw(n, b, state):
        out = torch.empty(b, dtype=torch.int64, device=gpu_device)
        _lib.check(lib.sgnn_batch_indices(n, b, _lib.ptr(state), None, _lib.ptr(out), _lib.current_stream(gpu_device)))
        return out.cpu().numpy()

    for n in (1, 2, 5, 64, 1000, 4097, 232_965):
        state = torch.tensor([12345 + n, 0, 0], dtype=torch.int64, device=gpu_device)
        full = draw(n, n, state)
        assert np.array_equal(np.sort(full), np.arange(n)), n
        assert int(state[1]) == 1
        again = draw(n, n, state)
        if n > 64:
            assert not np.array_equal(full, again)
        state2 = torch.tensor([12345 + n, 0, 0], dtype=torch.int64, device=gpu_device)
        assert np.array_equal(draw(n, min(n, 17), state2), full[:min(n, 17)])
    # uniformity: 2000 batches of 512 from 10 000 ids, every id expected 102.4 times
    n, b, reps = 10_000, 512, 2000
    state = torch.tensor([7, 0, 0], dtype=torch.int64, device=gpu_device)
    counts = np.zeros(n, np.int64)
    first = np.zeros(n, np.int64)
    for _ in range(reps):
        d = draw(n, b, state)
        assert np.unique(d).size == b
        counts[d] += 1
        first[d[0]] += 1
    expected = reps * b / n
    chi2 = float(((counts - expected) ** 2 / expected).sum())
    assert abs(chi2 - n) < 6 * np.sqrt(2 * n), chi2          # chi-square with ~n degrees of freedom
    assert first.max() <= 6                                    # no id owns the first slot
    with pytest.raises(_lib.SgnnHipError):
        draw(5, 6, state)
    # with a map the drawn positions come back as the ids they stand for, into a slice of a longer vector
    ids = torch.arange(0, 3 * 50, 3, dtype=torch.int64, device=gpu_device) + 7
    s1 = torch.tensor([5, 0, 0], dtype=torch.int64, device=gpu_device)
    s2 = torch.tensor([5, 0, 0], dtype=torch.int64, device=gpu_device)
    plain = draw(50, 20, s1)
    both = torch.full((30,), -1, dtype=torch.int64, device=gpu_device)
    _lib.check(lib.sgnn_batch_indices(50, 20, _lib.ptr(s2), _lib.ptr(ids), _lib.ptr(both[10:]),
                                      _lib.current_stream(gpu_device)))
    got = both.cpu().numpy()
    assert (got[:10] == -1).all() and np.array_equal(got[10:], ids.cpu().numpy()[plain])
    assert int(s2[1]) == 1 and int(s2[2]) == 0
