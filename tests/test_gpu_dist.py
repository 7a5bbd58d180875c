"""Two ranks sharing cuda:0 (gloo rendezvous, host-staged exchange): the
row-partitioned path with the real kernels equals the single-GPU result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, r, c, X, order, slices, q, w=None, exchange="allgather", overlap="local-block",
            partition="degree_cyclic", plain=False, planes=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sgnn_amd.dist import PartitionedAdj
        dev = torch.device("cuda:0")
        val = None if w is None else torch.full((r.size,), w, dtype=torch.float32)
        padj = PartitionedAdj.from_coo(torch.from_numpy(r), torch.from_numpy(c), val, n, dev, rank, world,
                                       exchange=exchange, overlap=overlap, partition=partition)
        assert (padj.halo is not None) == (exchange == "referenced")
        padj.use_planes = planes
        if plain:
            padj.local.use_tilegroup, padj.local.use_rowgroup = False, False
        rows = padj.rows.numpy()
        Xl = torch.from_numpy(X[rows].copy()).to(dev)
        Y = padj.propagate(Xl, order, slices=slices)
        Y2 = padj.propagate(Xl, order, slices=slices)          # the kept buffers are reused
        torch.cuda.synchronize()
        assert torch.equal(Y, Y2)
        q.put((rank, rows, Y.cpu().numpy(), padj.last_schedule))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("F,order,slices,w,exchange", [
    (96, 1, 1, None, "allgather"), (602, 2, 1, None, "allgather"), (602, 2, 2, None, "allgather"),
    (7, 2, 1, None, "allgather"), (128, 2, 1, 2.0, "allgather"),
    (100, 3, 1, None, "referenced"), (128, 2, 1, 2.0, "referenced")])
def test_two_ranks_one_gpu_match_single(gpu_device, F, order, slices, w, exchange):
    from sgnn_amd import NormAdj
    from sgnn_amd.graphgen import symmetric_simple_numpy
    n = 20_000
    r, c = symmetric_simple_numpy(n, 300_000, seed=4)
    X = np.random.default_rng(1).standard_normal((n, F)).astype(np.float32)
    adj = NormAdj.from_coo(r, c, None if w is None else np.full(r.size, w, np.float32), n, gpu_device)
    exp = adj.propagate(torch.from_numpy(X).to(gpu_device), order).cpu().numpy()
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(k, world, port, n, r, c, X, order, slices, q, w, exchange))
             for k in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    Y = np.zeros_like(exp)
    for rank, rows, y, sched in got:
        Y[rows] = y
        assert sched == ("sliced" if slices > 1 else "local-block")
    assert np.linalg.norm(Y - exp) / np.linalg.norm(exp) <= 1e-5


def _run_ranks(world, args_of, target=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target or _worker, args=args_of(k, port, q)) for k in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


@pytest.mark.parametrize("world,exchange", [(2, "allgather"), (4, "allgather")])
def test_one_block_schedule_with_the_plain_kernel_is_bit_identical_to_one_gpu(gpu_device, world, exchange):
    """The plain CSR kernel reduces a row in an order that depends only on that row's entries, so a
    row comes out the same whichever row block it is computed in: contiguous row blocks under the
    all-gather exchange (the column order inside a row is then the single-GPU one; the
    referenced-rows exchange and the degree-cyclic deal renumber the columns, which reorders the
    sums), one-block schedule, plain kernel on both sides -- equal bit for bit, chain included."""
    from sgnn_amd import NormAdj
    from sgnn_amd.graphgen import symmetric_simple_numpy
    n, F, order = 12_000, 70, 2
    r, c = symmetric_simple_numpy(n, 150_000, seed=14)
    X = np.random.default_rng(3).standard_normal((n, F)).astype(np.float32)
    from sgnn_amd import _lib
    adj = NormAdj.from_coo(r, c, None, n, gpu_device)
    adj.use_tilegroup, adj.use_rowgroup = False, False
    # the single-GPU kernel on the same operands the ranks multiply: X scaled by dinv in a pass of
    # its own (utils.py:93, as the ranks do before the exchange; the one-GPU chain folds that
    # scaling into its first hop, which rounds once less), then the two hops
    Xd = torch.from_numpy(X).to(gpu_device)
    Xs = torch.empty_like(Xd)
    _lib.check(_lib.load().sgnn_row_scale_f32(n, F, _lib.ptr(adj.dinv), _lib.ptr(Xd), F, _lib.ptr(Xs), F,
                                              _lib.current_stream(gpu_device)))
    mid = adj.spmm(Xs, scale_rows=True, scale_cols=False, epilogue=_lib.EPI_SCALE_ROW2)
    exp = adj.spmm(mid, scale_rows=True, scale_cols=False, epilogue=_lib.EPI_SCALE_ROW).cpu().numpy()
    assert np.linalg.norm(exp - adj.propagate(Xd, order).cpu().numpy()) / np.linalg.norm(exp) <= 1e-6
    got = _run_ranks(world, lambda k, port, q: (k, world, port, n, r, c, X, order, 1, q, None, exchange, "none",
                                                "contiguous", True))
    Y = np.zeros_like(exp)
    for rank, rows, y, sched in got:
        Y[rows] = y
        assert sched == "none"
    assert np.array_equal(Y, exp)


@pytest.mark.parametrize("world,F,order,w,exchange", [
    (2, 128, 2, None, "allgather"), (2, 200, 3, 2.0, "allgather"), (4, 96, 2, None, "allgather"),
    (2, 130, 2, 2.0, "referenced"), (2, 41, 2, None, "referenced"),
    # three and five 128-feature slices: the operand travels in units of two slices and the
    # remote-columns block is multiplied unit by unit (last unit: one slice, of 44 / 90 features)
    (2, 300, 2, 2.0, "allgather"), (2, 602, 3, None, "allgather"), (4, 602, 2, 2.0, "referenced")])
def test_planes_exchange_on_one_gpu_matches_single(gpu_device, world, F, order, w, exchange):
    """The matrix-core schedule of the partition, forced on a small graph: every rank splits only
    its own rows (after the ranks agreed on the per-feature maxima), fp16 planes travel slice by
    slice, the own-columns block is multiplied while they do, the remote-columns block adds its
    raw sums, and between hops the epilogue writes the rank's rows of the next operand into the
    next gather buffer.  Weight 2 exercises the diagonal correction rebuilt from the planes."""
    from sgnn_amd import NormAdj
    from sgnn_amd.graphgen import symmetric_simple_numpy
    n = 16_000
    r, c = symmetric_simple_numpy(n, 400_000, seed=15)
    X = np.random.default_rng(4).standard_normal((n, F)).astype(np.float32)
    adj = NormAdj.from_coo(r, c, None if w is None else np.full(r.size, w, np.float32), n, gpu_device)
    adj.use_tilegroup, adj.use_rowgroup = False, False
    Xd = torch.from_numpy(X).to(gpu_device)
    exp = adj.propagate(Xd, order).cpu().numpy()
    scale = adj.propagate(Xd.abs(), order).cpu().numpy()
    got = _run_ranks(world, lambda k, port, q: (k, world, port, n, r, c, X, order, 1, q, w, exchange, "local-block",
                                                "degree_cyclic", False, True))
    Y = np.zeros_like(exp)
    for rank, rows, y, sched in got:
        Y[rows] = y
        assert sched == "local-block, fp16 planes exchanged"
    assert np.linalg.norm(Y - exp) / np.linalg.norm(exp) <= 1e-5
    assert float((np.abs(Y - exp) / np.maximum(scale, 1e-30)).max()) <= 1e-5


def test_planes_exchange_refused_operand_falls_back_to_fp32_rows(gpu_device):
    """One infinite element on one rank: every rank must take the fp32 schedule for that call."""
    from sgnn_amd import NormAdj
    from sgnn_amd.graphgen import symmetric_simple_numpy
    n, F, world = 8_000, 64, 2
    r, c = symmetric_simple_numpy(n, 100_000, seed=16)
    X = np.random.default_rng(5).standard_normal((n, F)).astype(np.float32)
    X[123, 7] = np.inf
    adj = NormAdj.from_coo(r, c, None, n, gpu_device)
    adj.use_tilegroup, adj.use_rowgroup = False, False
    exp = adj.propagate(torch.from_numpy(X).to(gpu_device), 2).cpu().numpy()
    got = _run_ranks(world, lambda k, port, q: (k, world, port, n, r, c, X, 2, 1, q, None, "allgather", "local-block",
                                                "degree_cyclic", False, True))
    Y = np.zeros_like(exp)
    for rank, rows, y, sched in got:
        Y[rows] = y
        assert sched == "local-block"
    fin = np.isfinite(exp)
    assert np.array_equal(np.isfinite(Y), fin)
    assert np.linalg.norm(Y[fin] - exp[fin]) / np.linalg.norm(exp[fin]) <= 1e-5


def _tg_worker(rank, world, port, n, E, F, order, slices, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sgnn_amd import NormAdj
        from sgnn_amd.dist import PartitionedAdj
        from sgnn_amd.graphgen import rmat_undirected_device
        dev = torch.device("cuda:0")
        row, col = rmat_undirected_device(n, E, seed=5, device=dev)
        g = torch.Generator(device=dev); g.manual_seed(9)
        X = torch.randn(n, F, device=dev, generator=g)
        padj = PartitionedAdj.from_coo(row, col, None, n, dev, rank, world)
        rows = padj.rows.to(dev)
        Y = padj.propagate(X[rows].contiguous(), order, slices=slices)
        used = padj.last_schedule if slices == 1 else ("sliced, tile-group" if padj.local._tg else "sliced")
        ref = NormAdj.from_coo(row, col, None, n, dev)
        ref.use_tilegroup = False
        exp = ref.propagate(X, order)[rows]
        scale = ref.propagate(X.abs(), order)[rows]
        torch.cuda.synchronize()
        q.put((rank, used, float((Y - exp).norm() / exp.norm()), float(((Y - exp).abs() / scale.clamp_min(1e-30)).max())))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("slices", [1, 2])
def test_two_ranks_take_the_tile_group_kernel_on_a_dense_enough_graph(gpu_device, slices):
    """60 000 nodes, about 24 M stored entries, 12 M per rank of which half lie in other ranks' columns:
    the remote-columns block passes the size and sharing tests, so the default schedule exchanges
    fp16 planes and multiplies on the matrix cores without being told to; the feature-sliced
    schedule (fp32 rows, one block) takes the matrix-core kernel on the gathered operand."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tg_worker, args=(k, world, port, 60_000, 24_000_000, 200, 2, slices, q)) for k in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, used, normwise, elementwise in got:
        assert used == ("local-block, fp16 planes exchanged" if slices == 1 else "sliced, tile-group"), used
        assert normwise <= 1e-5 and elementwise <= 1e-5


def _nccl_single(port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        from sgnn_amd.dist import PartitionedAdj
        from sgnn_amd.graphgen import symmetric_simple_numpy
        n = 5000
        r, c = symmetric_simple_numpy(n, 40_000, seed=8)
        padj = PartitionedAdj.from_coo(torch.from_numpy(r), torch.from_numpy(c), None, n, dev, 0, 1)
        X = torch.randn(n, 64, device=dev)
        full = torch.zeros(padj.max_rows, 64, device=dev)
        full.copy_(X)
        padj._all_gather(full, full[0:padj.max_rows], force=True)      # RCCL, in place, one rank
        torch.cuda.synchronize()
        same = bool(torch.equal(full, X))
        Y1 = padj.propagate(X, 2)
        Y2 = padj.propagate(X, 2, slices=2)
        torch.cuda.synchronize()
        q.put((same, float((Y1 - Y2).abs().max()), float(Y1.abs().max())))
    finally:
        dist.destroy_process_group()


def test_rccl_backend_single_rank(gpu_device):
    """The production backend ("nccl" = RCCL) accepts the in-place all-gather the
    partitioned path issues, and the sliced schedule equals the plain one."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_single, args=(_free_port(), q))
    p.start()
    same, diff, scale = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert same
    assert diff <= 1e-5 * scale


def _grid_worker(rank, world, fg, port, n, r, c, X, order, slices, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sgnn_amd.dist import PartitionedAdj, grid_layout, make_row_groups
        dev = torch.device("cuda:0")
        row_rank, row_world, fgroup, f0, f1 = grid_layout(rank, world, fg, X.shape[1])
        group = make_row_groups(world, fg)
        padj = PartitionedAdj.from_coo(torch.from_numpy(r).to(dev), torch.from_numpy(c).to(dev), None, n, dev, row_rank,
                                       row_world, group=group)
        rows = padj.rows.numpy()
        Xl = torch.from_numpy(X[rows][:, f0:f1].copy()).to(dev)
        Y = padj.propagate(Xl, order, slices=slices)
        Y2 = padj.propagate(Xl, order, slices=slices)      # second call: the kept gather buffers are reused
        torch.cuda.synchronize()
        assert torch.equal(Y, Y2)
        q.put((rank, rows, f0, f1, Y.cpu().numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("slices", [1, 2])
def test_four_ranks_two_by_two_grid_on_one_gpu(gpu_device, slices):
    """2 feature groups x 2 row blocks, all four ranks on cuda:0, the COO handed over on the
    device (the deal and the column remap then run there): every (row block, feature slice)
    tile is produced once and the assembled result equals the single-GPU one."""
    from sgnn_amd import NormAdj
    from sgnn_amd.graphgen import symmetric_simple_numpy
    n, F, order, world, fg = 16_000, 200, 2, 4, 2
    r, c = symmetric_simple_numpy(n, 200_000, seed=6)
    X = np.random.default_rng(2).standard_normal((n, F)).astype(np.float32)
    exp = NormAdj.from_coo(r, c, None, n, gpu_device).propagate(torch.from_numpy(X).to(gpu_device), order).cpu().numpy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grid_worker, args=(k, world, fg, port, n, r, c, X, order, slices, q)) for k in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    Y = np.zeros_like(exp)
    hits = np.zeros(exp.shape, np.int32)
    for rank, rows, f0, f1, y in got:
        Y[np.ix_(rows, np.arange(f0, f1))] = y
        hits[np.ix_(rows, np.arange(f0, f1))] += 1
    assert (hits == 1).all()
    assert np.linalg.norm(Y - exp) / np.linalg.norm(exp) <= 1e-5


def _feature_worker(rank, world, port, n, r, c, X, order, q, w):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sgnn_amd.dist import FeatureSplitAdj
        dev = torch.device("cuda:0")
        val = None if w is None else torch.full((r.size,), w, dtype=torch.float32)
        fadj = FeatureSplitAdj.from_coo(torch.from_numpy(r), torch.from_numpy(c), val, n, dev, rank, world)
        rows = fadj.rows.numpy()
        Xl = torch.from_numpy(X[rows].copy()).to(dev)
        Y = fadj.propagate(Xl, order)
        Y2 = fadj.propagate(Xl, order)
        torch.cuda.synchronize()
        assert torch.equal(Y, Y2)
        q.put((rank, rows, Y.cpu().numpy(), fadj.last_schedule))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,F,order,w", [(2, 602, 2, None), (4, 130, 1, 2.0), (4, 41, 3, None)])
def test_feature_split_on_one_gpu_matches_single(gpu_device, world, F, order, w):
    """The feature-split schedule with the real kernels, ranks sharing cuda:0: every rank holds the
    whole graph (dense enough here for the matrix-core kernel at the wider slices) and propagates
    F / world columns; two all-to-alls carry the caller's rows there and back."""
    from sgnn_amd import NormAdj
    from sgnn_amd.graphgen import symmetric_simple_numpy
    n = 16_001                                             # blocks of unequal size
    r, c = symmetric_simple_numpy(n, 400_000, seed=17)
    X = np.random.default_rng(6).standard_normal((n, F)).astype(np.float32)
    adj = NormAdj.from_coo(r, c, None if w is None else np.full(r.size, w, np.float32), n, gpu_device)
    adj.use_tilegroup, adj.use_rowgroup = False, False
    Xd = torch.from_numpy(X).to(gpu_device)
    exp = adj.propagate(Xd, order).cpu().numpy()
    scale = adj.propagate(Xd.abs(), order).cpu().numpy()
    got = _run_ranks(world, lambda k, port, q: (k, world, port, n, r, c, X, order, q, w), target=_feature_worker)
    Y = np.zeros_like(exp)
    seen = np.zeros(n, bool)
    for rank, rows, y, sched in got:
        assert sched == "feature-split" and not seen[rows].any()
        seen[rows] = True
        Y[rows] = y
    assert seen.all()
    assert np.linalg.norm(Y - exp) / np.linalg.norm(exp) <= 1e-5
    assert float((np.abs(Y - exp) / np.maximum(scale, 1e-30)).max()) <= 1e-5
