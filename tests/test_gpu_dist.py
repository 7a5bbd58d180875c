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


def _worker(rank, world, port, n, r, c, X, order, slices, q, w=None, exchange="allgather"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sgnn_amd.dist import PartitionedAdj
        dev = torch.device("cuda:0")
        val = None if w is None else torch.full((r.size,), w, dtype=torch.float32)
        padj = PartitionedAdj.from_coo(torch.from_numpy(r), torch.from_numpy(c), val, n, dev, rank, world,
                                       exchange=exchange)
        assert (padj.halo is not None) == (exchange == "referenced")
        rows = padj.rows.numpy()
        Xl = torch.from_numpy(X[rows].copy()).to(dev)
        Y = padj.propagate(Xl, order, slices=slices)
        torch.cuda.synchronize()
        q.put((rank, rows, Y.cpu().numpy()))
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
    for rank, rows, y in got:
        Y[rows] = y
    assert np.linalg.norm(Y - exp) / np.linalg.norm(exp) <= 1e-5


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
        used = bool(padj.local._tg)
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
    """40 000 nodes, average degree ~250, 5 M stored entries per rank: each rank's 20 000 rows pass the size and sharing tests, so the
    rank-local multiply is the matrix-core kernel on the gathered (pre-scaled) operand."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tg_worker, args=(k, world, port, 40_000, 10_000_000, 200, 2, slices, q)) for k in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, used, normwise, elementwise in got:
        assert used, "the tile-group plan was not built"
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
