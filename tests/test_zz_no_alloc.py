"""The SpMM entry point that allocates nothing (scratch in a device array inside the
library) against the entry point it is a variant of: same kernels, so the outputs must be
equal bit for bit — on two calls in a row, the second on a smaller problem, so that what
the first call left in the shared scratch is seen not to matter."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _hub_graph(n, n_edges, hubs, seed):
    """Symmetric graph with a few rows far longer than the 1024-entry segment length, so
    that the scratch (segment lists, partial rows) is really used."""
    rng = np.random.default_rng(seed)
    r = rng.integers(0, n, n_edges)
    c = rng.integers(0, n, n_edges)
    hr = np.repeat(np.arange(hubs), n // 2)
    hc = rng.integers(0, n, hr.size)
    r, c = np.concatenate((r, hr)), np.concatenate((c, hc))
    return np.concatenate((r, c)), np.concatenate((c, r))


def _both(adj, X, lib, _lib, dev):
    n, F = X.shape
    st = _lib.current_stream(dev)
    Y_ws = torch.empty(n, F, device=dev)
    Y_na = torch.full((n, F), float("nan"), device=dev)
    ws_bytes = lib.sgnn_spmm_workspace_bytes(n, adj.nnz, F)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    common = (n, n, adj.nnz, F, _lib.ptr(adj.rowptr), _lib.ptr(adj.colidx), None, _lib.ptr(adj.dinv),
              _lib.ptr(adj.dinv), _lib.ptr(X), F)
    _lib.check(lib.sgnn_spmm_csr_f32(*common, _lib.ptr(Y_ws), F, _lib.EPI_SCALE_ROW, _lib.ptr(ws), ws_bytes, st))
    before = torch.cuda.memory_allocated(dev)
    _lib.check(lib.sgnn_spmm_csr_noalloc_f32(*common, _lib.ptr(Y_na), F, _lib.EPI_SCALE_ROW, st))
    assert torch.cuda.memory_allocated(dev) == before
    torch.cuda.synchronize()
    return Y_ws, Y_na


def test_noalloc_spmm_equals_workspace_spmm_twice_in_a_row(gpu_device):
    from sgnn_amd import NormAdj, _lib
    dev = gpu_device
    lib = _lib.load()
    g = torch.Generator(device=dev); g.manual_seed(7)
    # first call: 40 000 nodes, 6 hub rows of ~20 000 entries, F = 96
    r, c = _hub_graph(40_000, 400_000, 6, seed=1)
    big = NormAdj.from_coo(r, c, None, 40_000, dev, clamp_one=True)
    assert int((big.rowptr[1:] - big.rowptr[:-1]).max()) > 4 * 1024
    Xb = torch.randn(40_000, 96, device=dev, generator=g)
    Y_ws, Y_na = _both(big, Xb, lib, _lib, dev)
    assert torch.equal(Y_ws, Y_na)
    # second call, smaller in every dimension, different width: leftovers in the scratch are ignored
    r, c = _hub_graph(9_000, 50_000, 2, seed=2)
    small = NormAdj.from_coo(r, c, None, 9_000, dev, clamp_one=True)
    assert int((small.rowptr[1:] - small.rowptr[:-1]).max()) > 2 * 1024
    Xs = torch.randn(9_000, 37, device=dev, generator=g)
    Y_ws, Y_na = _both(small, Xs, lib, _lib, dev)
    assert torch.equal(Y_ws, Y_na)
    # and the first problem again after the smaller one
    Y_ws, Y_na = _both(big, Xb, lib, _lib, dev)
    assert torch.equal(Y_ws, Y_na)


def test_noalloc_spmm_refuses_what_its_scratch_cannot_hold(gpu_device):
    from sgnn_amd import NormAdj, _lib
    dev = gpu_device
    lib = _lib.load()
    r, c = _hub_graph(2_000, 10_000, 1, seed=3)
    adj = NormAdj.from_coo(r, c, None, 2_000, dev, clamp_one=True)
    X = torch.randn(2_000, 8, device=dev)
    Y = torch.zeros(2_000, 8, device=dev)
    st = _lib.current_stream(dev)
    # claim a problem whose scratch would exceed the library's array: refused before any launch
    huge_nnz, huge_F = 120_000_000, 602
    assert lib.sgnn_spmm_workspace_bytes(2_000, huge_nnz, huge_F) > (32 << 20)
    rc = lib.sgnn_spmm_csr_noalloc_f32(2_000, 2_000, huge_nnz, huge_F, _lib.ptr(adj.rowptr), _lib.ptr(adj.colidx), None,
                                       _lib.ptr(adj.dinv), _lib.ptr(adj.dinv), _lib.ptr(X), huge_F, _lib.ptr(Y),
                                       huge_F, _lib.EPI_SCALE_ROW, st)
    assert rc == -2 and b"built-in scratch" in lib.sgnn_last_error()
    torch.cuda.synchronize()
    assert float(Y.abs().sum()) == 0.0
