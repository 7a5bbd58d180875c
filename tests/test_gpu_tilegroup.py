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


@pytest.mark.parametrize("weight", [None, 2.0])
def test_chain_writes_the_next_operand_from_the_epilogue(gpu_device, weight):
    """K = 3 as one chain (utils.py:91-96): X is split once and the two intermediates exist only as
    the fp16 planes the hop before wrote from its epilogue, at scales derived from bounds (a row
    of D^-1 (A+I) sums to one).  It must equal three single hops -- each of which measures and
    splits its fp32 operand -- element by element, and ordinary data must never raise the flag."""
    from sgnn_amd import NormAdj
    from sgnn_amd.graphgen import rmat_undirected_device
    dev = gpu_device
    n, F = 30_000, 200
    row, col = rmat_undirected_device(n, 3_000_000, seed=12, device=dev)
    val = None if weight is None else torch.full((row.numel(),), weight, device=dev)
    adj = NormAdj.from_coo(row, col, val, n, dev)
    adj.use_tilegroup = True
    g = torch.Generator(device=dev); g.manual_seed(13)
    X = torch.randn(n, F, device=dev, generator=g)
    chain = adj.propagate(X, 3)
    torch.cuda.synchronize()
    states = adj._tg_bufs[F]["states"]
    assert states.shape[0] >= 3 and [int(states[k, 1].item()) for k in range(3)] == [0, 0, 0]
    hops = X
    for _ in range(3):
        hops = adj.spmm(hops)
        assert int(adj._tg_bufs[F]["state"][1].item()) == 0
    adj.use_tilegroup, adj.use_rowgroup = False, False
    plain = adj.propagate(X, 3)
    scale = adj.propagate(X.abs(), 3)
    for other in (hops, plain):
        assert float((chain - other).norm() / other.norm()) <= TOL
        assert float(((chain - other).abs() / scale.clamp_min(1e-30)).max()) <= TOL
    # the same chain twice: same bits (the per-feature maxima are gathered with atomics, the sums are not)
    adj.use_tilegroup, adj.use_rowgroup = True, True
    assert torch.equal(adj.propagate(X, 3), chain)


def test_chain_refused_anywhere_equals_the_plain_chain(gpu_device):
    """A chain whose first operand the planes cannot hold (an infinite element), and one that
    only a later hop refuses (the first operand's columns are fine, but one row is so large that
    after a hop most of every column sits 2^40 below its bound): the flag travels down the chain
    and the result is the plain CSR chain's, bit for bit."""
    adj, r, c, v = _graph(4000, 80_000, seed=31, dev=gpu_device)
    F = 96
    g = torch.Generator(device=gpu_device); g.manual_seed(32)
    base = torch.randn(4000, F, device=gpu_device, generator=g)
    for kind in ("inf", "late"):
        X = base.clone()
        if kind == "inf":
            X[7, 3] = float("inf")
        else:
            X[11] *= 2.0 ** 45
        adj.use_tilegroup, adj.use_rowgroup = True, True
        Y = adj.propagate(X, 3)
        torch.cuda.synchronize()
        states = adj._tg_bufs[F]["states"]
        flags = [int(states[k, 1].item()) for k in range(3)]
        assert flags[2] != 0, f"{kind}: the last operand's flag must be up"
        if kind == "inf":
            assert flags[0] != 0
        adj.use_tilegroup, adj.use_rowgroup = False, False
        Y_ref = adj.propagate(X, 3)
        assert torch.equal(torch.isfinite(Y), torch.isfinite(Y_ref))
        fin = torch.isfinite(Y_ref)
        assert torch.equal(Y[fin], Y_ref[fin])


def test_launch_counts_that_do_not_match_the_plan_raise_the_flag(gpu_device):
    """The C ABI takes the plan's counts from the caller.  A tuple that differs from what the plan
    was filled with must not be used to carve the plan: the kernel compares it with the counts the
    plan records, raises the flag (bit 4) and leaves the result to the fall-back."""
    from sgnn_amd import _lib
    from sgnn_amd.norm_adj import tilegroup_buffers
    lib = _lib.load()
    adj, r, c, v = _graph(3000, 40_000, seed=41, dev=gpu_device)
    F = 128
    X = torch.randn(3000, F, device=gpu_device)
    Y_good = adj.spmm(X)
    plan = adj._tg
    bufs = tilegroup_buffers(plan, adj.n, F, gpu_device)
    st = _lib.current_stream(gpu_device)
    state = bufs["states"][0]
    _lib.check(lib.sgnn_tilegroup_split_f32(adj.n, F, _lib.ptr(adj.dinv), _lib.ptr(X), F, _lib.ptr(bufs["planes"]),
                                            bufs["pl_bytes"], _lib.ptr(state), st))
    Y = torch.full((3000, F), 7.0, device=gpu_device)
    ws = adj._workspace(F)
    # one step fewer than the plan holds: every later sub-array would be read at a shifted offset
    _lib.check(lib.sgnn_spmm_tilegroup_f32(
        adj.n, adj.n, F, _lib.ptr(plan["buf"]), plan["bytes"], plan["stages"], plan["steps"] - 1, plan["items"],
        plan["slots"], plan["n_long"], _lib.ptr(adj.dinv), 1.0, None, _lib.ptr(bufs["planes"]), bufs["pl_bytes"],
        _lib.ptr(state), _lib.ptr(Y), F, _lib.EPI_SCALE_ROW, _lib.ptr(bufs["ws"]), bufs["ws"].numel(), adj.nnz,
        _lib.ptr(adj.rowptr), _lib.ptr(adj.colidx), None, _lib.ptr(adj.dinv), _lib.ptr(X), F, _lib.ptr(ws), ws.numel(),
        0, None, 0, None, 0, 0, None, 0, st))
    torch.cuda.synchronize()
    assert int(state[1].item()) & 4
    adj.use_tilegroup, adj.use_rowgroup = False, False
    assert torch.equal(Y, adj.spmm(X))                   # the fall-back computed it
    assert float((Y - Y_good).norm() / Y_good.norm()) <= TOL


@pytest.mark.parametrize("weight,to_planes", [(None, False), (2.0, False), (2.0, True)])
def test_slice_ranges_equal_the_whole_launch_bit_for_bit(gpu_device, weight, to_planes):
    """sgnn_spmm_tilegroup_slices_f32 over ranges of the operand's 128-feature slices, in ascending
    order on one stream, against the one launch over all of them: a slice's rows are computed by
    the same units of work either way, so Y (or the next operand's planes, maxima, bounds and tiny
    count) must come out bit-identical.  F = 602: five slices, the last of 90 features; ranges of
    2 + 1 + 2 slices; the hub rows of the graph split groups, so k_tg_combine runs per range too."""
    from sgnn_amd import _lib
    from sgnn_amd.norm_adj import tilegroup_buffers
    lib = _lib.load()
    n, F = 6000, 602
    adj, r, c, v = _graph(n, 700_000, seed=21, dev=gpu_device, weight=weight)
    X = torch.randn(n, F, device=gpu_device)
    adj.propagate(X, 1)                                    # builds the plan
    plan = adj._tg
    assert plan and plan["n_long"] > 0, "the graph should split some groups"
    st = _lib.current_stream(gpu_device)
    sb = lib.sgnn_tilegroup_state_bytes(F)
    pl_bytes = lib.sgnn_tilegroup_planes_bytes(n, F)
    ws = torch.empty(max(lib.sgnn_spmm_tilegroup_workspace_bytes(plan["slots"], F), 256), dtype=torch.uint8, device=gpu_device)
    planes = torch.empty(pl_bytes, dtype=torch.uint8, device=gpu_device)
    state = torch.zeros(sb // 4, dtype=torch.int32, device=gpu_device)
    _lib.check(lib.sgnn_tilegroup_split_f32(n, F, _lib.ptr(adj.dinv), _lib.ptr(X), F, _lib.ptr(planes), pl_bytes,
                                            _lib.ptr(state), st))
    corr = getattr(adj, "corr", None)

    def run(ranges, reserve=0):
        Y = torch.zeros(n, F, device=gpu_device)
        out_planes = torch.zeros(pl_bytes, dtype=torch.uint8, device=gpu_device) if to_planes else None
        out_state = torch.full((sb // 4,), -1, dtype=torch.int32, device=gpu_device) if to_planes else None
        for s0, cnt in ranges:
            _lib.check(lib.sgnn_spmm_tilegroup_slices_f32(
                n, n, F, _lib.ptr(plan["buf"]), plan["bytes"], plan["stages"], plan["steps"], plan["items"], plan["slots"],
                plan["n_long"], _lib.ptr(adj.dinv), float(adj.w_off), _lib.ptr(corr), _lib.ptr(planes), pl_bytes,
                _lib.ptr(state), None if to_planes else _lib.ptr(Y), F,
                _lib.EPI_SCALE_ROW2 if to_planes else _lib.EPI_SCALE_ROW, _lib.ptr(ws), ws.numel(), adj.nnz,
                None, None, None, _lib.ptr(adj.dinv), _lib.ptr(X), F, None, 0, 0, None, 0,
                _lib.ptr(out_planes) if to_planes else None, n, 0, _lib.ptr(out_state) if to_planes else None, 1,
                s0, cnt, reserve, st))
        torch.cuda.synchronize()
        assert int(state[_lib.TG_STATE_FLAG].item()) == 0
        if to_planes:
            Fpad = (F + 127) // 128 * 128
            live = 16 + 2 * Fpad                            # header, maxima, bounds (the scratch behind them is not written)
            head = out_state[:live].clone()
            head[7:16] = 0                                  # the work counters of the next operand's launches
            return out_planes, head
        return Y, None

    whole, whole_state = run([(0, 0)])
    parts, parts_state = run([(0, 2), (2, 1), (3, 2)])
    assert torch.equal(whole, parts)
    # CUs left free for a collective's kernels: fewer workgroups take the same units off the counter
    spared, spared_state = run([(0, 2), (2, 1), (3, 2)], reserve=32)
    assert torch.equal(whole, spared)
    if to_planes:
        assert torch.equal(whole_state, spared_state)
        assert torch.equal(whole_state, parts_state)
        assert int(whole_state[_lib.TG_STATE_FLAG].item()) == 0
    # a range that is not the whole operand has no fall-back of its own, and ranges must lie inside the operand
    with pytest.raises(_lib.SgnnHipError):
        _lib.check(lib.sgnn_spmm_tilegroup_slices_f32(
            n, n, F, _lib.ptr(plan["buf"]), plan["bytes"], plan["stages"], plan["steps"], plan["items"], plan["slots"],
            plan["n_long"], _lib.ptr(adj.dinv), float(adj.w_off), _lib.ptr(corr), _lib.ptr(planes), pl_bytes,
            _lib.ptr(state), _lib.ptr(whole if not to_planes else torch.zeros(n, F, device=gpu_device)), F, _lib.EPI_SCALE_ROW,
            _lib.ptr(ws), ws.numel(), adj.nnz, None, None, None, None, None, 0, None, 0, 0, None, 0, None, 0, 0, None, 1,
            4, 2, 0, st))


def test_provisional_scales_from_a_sample_and_the_exact_passes_behind_them(gpu_device):
    """Operands of 65536 rows and more are split at scales taken from every 32nd row (times 8).
    Ordinary data: the exact passes behind the split do nothing (redo word 0) and the result
    meets the same bound.  An outlier the sample missed, a column that is zero in every sampled
    row, a value between the provisional bound and twice it: the exact passes run (redo word 1)
    and the result is bit for bit the one of a library told never to sample.  A non-finite
    element in an unsampled row still ends in the plain kernel's result."""
    from sgnn_amd import _lib
    lib = _lib.load()
    n, F = 100_000, 96
    adj, r, c, v = _graph(n, 1_500_000, seed=31, dev=gpu_device)
    g = torch.Generator(device=gpu_device); g.manual_seed(7)
    X = torch.randn(n, F, device=gpu_device, generator=g)

    def run(Xd, order=2):
        Y = adj.propagate(Xd, order)
        torch.cuda.synchronize()
        st = adj._tg_bufs[F]["states"]
        return Y, [int(x) for x in st[0, :8].tolist()], int(adj._tg_bufs[F]["state"][_lib.TG_STATE_FLAG].item())

    adj.use_tilegroup = False
    ref = adj.propagate(X, 2)
    scale = adj.propagate(adj.propagate(X.abs(), 1), 1)
    adj.use_tilegroup = True
    Y, head, flag = run(X)
    assert adj._tg, "tile-group plan was not built"
    assert flag == 0 and head[_lib.TG_STATE_REDO] == 0 and head[_lib.TG_STATE_SAMPLED] == 1
    assert float(((Y - ref).abs() / scale.clamp_min(1e-30)).max()) <= 2.0 ** -19
    lib.sgnn_tune_set(b"tg_no_sample", 1)
    try:
        Ye, head_e, _ = run(X)
        assert head_e[2] == 0 and head_e[3] == 0
        assert float(((Ye - ref).abs() / scale.clamp_min(1e-30)).max()) <= 2.0 ** -19
        for what in ("outlier", "zero column", "just beyond"):
            Xo = X.clone()
            if what == "outlier":
                Xo[7, 3] = 5000.0                                   # row 7 is not sampled (rows 0, 32, 64, ...)
            elif what == "zero column":
                Xo[:, 5] = 0.0
                Xo[33, 5] = 1.0
            else:
                col = Xo[:, 9]
                col[1::32] = 0.0                                    # keep the sampled maximum where it is ...
                sampled_max = float((Xo[::32, 9] * adj.dinv[::32]).abs().max())
                Xo[1, 9] = 40.0 * sampled_max / float(adj.dinv[1])  # ... and put 40 x that (bound: 8 x, up to 16 x passes) in an unsampled row
            lib.sgnn_tune_set(b"tg_no_sample", 1)
            Y_exact, _, f1 = run(Xo)
            lib.sgnn_tune_set(b"tg_no_sample", 0)
            Y_redo, head_r, f2 = run(Xo)
            assert f1 == 0 and f2 == 0, what
            assert head_r[2] == 1 and head_r[3] == 0, (what, head_r)      # the exact passes ran and left exact maxima
            assert torch.equal(Y_exact, Y_redo), what
        Xi = X.clone()
        Xi[12345, 17] = float("inf")
        lib.sgnn_tune_set(b"tg_no_sample", 0)
        Yi, _, fi = run(Xi, order=1)
        assert fi != 0
        adj.use_tilegroup, adj.use_rowgroup = False, False
        Yp = adj.propagate(Xi, 1)
        assert torch.equal(torch.nan_to_num(Yi, nan=1.5, posinf=2.5, neginf=-2.5), torch.nan_to_num(Yp, nan=1.5, posinf=2.5, neginf=-2.5))
    finally:
        lib.sgnn_tune_set(b"tg_no_sample", 0)


@pytest.mark.parametrize("F,order,weight,dist_kind", [(16, 2, None, "normal"), (41, 3, 2.0, "lognormal"), (130, 2, None, "lognormal"),
                                                      (602, 1, 2.0, "normal"), (64, 2, None, "sparse")])
def test_provisional_scales_across_widths_and_distributions(gpu_device, F, order, weight, dist_kind):
    """Just above the sampling threshold (70 000 rows): Gaussian data keeps the provisional scales,
    heavy-tailed (log-normal, sigma 3) and mostly-zero data send some operands through the exact
    passes; either way the result meets the 1e-5 bound against the plain kernel, and where the
    exact passes ran it is bit for bit the result of a library that never samples."""
    from sgnn_amd import _lib
    lib = _lib.load()
    n = 70_000
    adj, r, c, v = _graph(n, 900_000, seed=40 + F, dev=gpu_device, weight=weight)
    g = torch.Generator(device=gpu_device); g.manual_seed(F)
    X = torch.randn(n, F, device=gpu_device, generator=g)
    if dist_kind == "lognormal":
        X = torch.exp(3.0 * X) * torch.sign(torch.randn(n, F, device=gpu_device, generator=g))
    elif dist_kind == "sparse":
        X = X * (torch.rand(n, F, device=gpu_device, generator=g) < 0.002)
    adj.use_tilegroup, adj.use_rowgroup = False, False
    ref = adj.propagate(X, order)
    scale = adj.propagate(X.abs(), order)
    adj.use_tilegroup, adj.use_rowgroup = True, True
    try:
        lib.sgnn_tune_set(b"tg_no_sample", 0)
        Y = adj.propagate(X, order)
        torch.cuda.synchronize()
        assert adj._tg, "tile-group plan was not built"
        st = adj._tg_bufs[F]["states"]
        flags = st[:order, _lib.TG_STATE_FLAG].tolist()
        redo = int(st[0, _lib.TG_STATE_REDO].item())
        if any(flags):      # a dynamic range the planes refuse (log-normal columns may): the plain kernel's result, bit for bit
            assert torch.equal(Y, ref)
        else:
            assert float(((Y - ref).abs() / scale.clamp_min(1e-30)).max()) <= TOL
            assert float((Y - ref).norm() / ref.norm()) <= TOL
        if dist_kind == "normal":
            assert redo == 0 and not any(flags)
        lib.sgnn_tune_set(b"tg_no_sample", 1)
        Ye = adj.propagate(X, order)
        if redo:
            assert torch.equal(Y, Ye)
    finally:
        lib.sgnn_tune_set(b"tg_no_sample", 0)
