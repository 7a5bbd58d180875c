"""Row-partitioned propagation with world_size > 1 over gloo on the CPU.

The partitioning, padding, column remap, pre-scaling, all-gather and the hop
chaining are the product's (sgnn_amd.dist); the rank-local SpMM is injected from
the CPU oracle because there is no GPU here — the GPU variant of this test runs
the real kernels on two ranks that share cuda:0.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import sgnn_oracle as orc


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class OracleBlock:
    """CPU stand-in for the device row block: same interface, oracle arithmetic.
    Rows and columns arrive in the padded index space, where the rank's rows are the
    range [r0, r0 + n_rows)."""

    def __init__(self, lrow, lcol, lval, n_rows, n_pad, r0):
        v = np.ones(lrow.numel(), np.float32) if lval is None else lval.numpy().astype(np.float32)
        rowptr, colidx, vals, deg, dinv = orc.norm_csr(lrow.numpy(), lcol.numpy(), v, n_pad)
        e0, e1 = rowptr[r0], rowptr[r0 + n_rows]
        self.rowptr = (rowptr[r0:r0 + n_rows + 1] - e0).astype(np.int32)
        self.colidx = colidx[e0:e1].astype(np.int32)
        self.vals = vals[e0:e1]
        self.dinv = dinv[r0:r0 + n_rows]
        self.n_rows, self.n_cols = n_rows, n_pad

    def scale_rows_into(self, src, dst):
        dst.copy_(src * torch.from_numpy(self.dinv)[:, None])

    def split_columns(self, c0, c1):
        """The block over columns [c0, c1) and the block over all others (same rows, same dinv)."""
        import copy
        rows = np.repeat(np.arange(self.n_rows), np.diff(self.rowptr))
        inside = (self.colidx >= c0) & (self.colidx < c1)
        parts = []
        for keep in (inside, ~inside):
            b = copy.copy(self)
            b.rowptr = np.concatenate(([0], np.cumsum(np.bincount(rows[keep], minlength=self.n_rows)))).astype(np.int32)
            b.colidx, b.vals = self.colidx[keep], self.vals[keep]
            parts.append(b)
        return parts[0], parts[1]

    def spmm_raw_into(self, full, out):
        import scipy.sparse as sp
        A = sp.csr_matrix((self.vals.astype(np.float64), self.colidx, self.rowptr), shape=(self.n_rows, self.n_cols))
        out.copy_(torch.from_numpy((A @ full.numpy().astype(np.float64)).astype(np.float32)))

    def finish_into(self, s1, s2, out, final):
        d = torch.from_numpy(self.dinv)[:, None]
        y = (s1 + s2) * d
        out.copy_(y if final else y * d)

    def spmm_into(self, full, out, final):
        import scipy.sparse as sp
        A = sp.csr_matrix((self.vals.astype(np.float64), self.colidx, self.rowptr), shape=(self.n_rows, self.n_cols))
        s = (A @ full.numpy().astype(np.float64)).astype(np.float32)
        y = s * self.dinv[:, None]
        if not final:
            y = y * self.dinv[:, None]
        out.copy_(torch.from_numpy(y.astype(np.float32)))


def _worker(rank, world, port, n, r, c, X, order, q, partition="degree_cyclic", exchange="allgather",
            overlap="local-block"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sgnn_amd.dist import PartitionedAdj
        padj = PartitionedAdj.from_coo(torch.from_numpy(r), torch.from_numpy(c), None, n, "cpu", rank, world,
                                       local_factory=OracleBlock, partition=partition, exchange=exchange, overlap=overlap)
        assert padj.overlap == (overlap if world > 1 else "none")
        rows = padj.rows.numpy()
        if exchange == "referenced":
            assert padj.halo is not None and sum(padj.halo["recv_counts"]) == padj.halo["n_recv"]
        if partition == "contiguous":
            assert np.array_equal(rows, np.arange(padj.row_begin, padj.row_end))
        Xl = torch.from_numpy(X[rows].copy())
        Y = padj.propagate(Xl, order)
        if overlap == "local-block" and world > 1:
            # the one-block schedule on the same handle: same rows, to rounding of the two-block sum
            Y1 = padj.propagate(Xl, order, overlap="none")
            assert float((Y - Y1).norm() / Y1.norm()) <= 1e-6
        q.put((rank, rows, Y.numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,order,partition,exchange,overlap", [
    (2, 1, "degree_cyclic", "allgather", "local-block"), (2, 2, "degree_cyclic", "allgather", "none"),
    (3, 3, "degree_cyclic", "allgather", "local-block"),
    (3, 2, "contiguous", "allgather", "local-block"), (8, 2, "degree_cyclic", "allgather", "local-block"),
    (2, 2, "degree_cyclic", "referenced", "local-block"), (4, 3, "degree_cyclic", "referenced", "none"),
    (3, 2, "contiguous", "referenced", "local-block"),
    (8, 2, "degree_cyclic", "auto", "local-block")])
def test_row_partitioned_matches_single(world, order, partition, exchange, overlap):
    from sgnn_amd.graphgen import symmetric_simple_numpy
    n, F = 700, 12
    r, c = symmetric_simple_numpy(n, 4000, seed=3)
    X = np.random.default_rng(0).standard_normal((n, F)).astype(np.float32)
    exp = orc.propagate_scipy(__import__("scipy.sparse").sparse.coo_matrix(
        (np.ones(r.size, np.float32), (r, c)), shape=(n, n)), X, order)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(k, world, port, n, r, c, X, order, q, partition, exchange, overlap))
             for k in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    Y = np.zeros_like(exp)
    covered = np.zeros(n, bool)
    sizes = []
    for rank, rows, y in got:
        Y[rows] = y
        assert not covered[rows].any()
        covered[rows] = True
        sizes.append(rows.size)
    assert covered.all()
    if partition == "degree_cyclic":
        assert max(sizes) - min(sizes) <= 1                 # no padding to speak of in the exchange
    assert np.linalg.norm(Y - exp) / np.linalg.norm(exp) <= 1e-5


def _grid_worker(rank, world, fg, port, n, r, c, X, order, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sgnn_amd.dist import PartitionedAdj, grid_layout, make_row_groups
        row_rank, row_world, fgroup, f0, f1 = grid_layout(rank, world, fg, X.shape[1])
        group = make_row_groups(world, fg)
        padj = PartitionedAdj.from_coo(torch.from_numpy(r), torch.from_numpy(c), None, n, "cpu", row_rank, row_world,
                                       group=group, local_factory=OracleBlock)
        rows = padj.rows.numpy()
        Xl = torch.from_numpy(X[rows][:, f0:f1].copy())
        Y = padj.propagate(Xl, order)
        q.put((rank, rows, f0, f1, Y.numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,fg,order", [(4, 2, 2), (2, 2, 1), (3, 1, 1), (8, 2, 1)])
def test_feature_by_row_grid_matches_single(world, fg, order):
    """Ranks on a feature_groups x row_ranks grid: every (row block, feature slice)
    tile is produced exactly once and the assembled result is the single-process one."""
    from sgnn_amd.graphgen import symmetric_simple_numpy
    import scipy.sparse as sp
    n, F = 500, 10
    r, c = symmetric_simple_numpy(n, 3000, seed=4)
    X = np.random.default_rng(1).standard_normal((n, F)).astype(np.float32)
    exp = orc.propagate_scipy(sp.coo_matrix((np.ones(r.size, np.float32), (r, c)), shape=(n, n)), X, order)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grid_worker, args=(k, world, fg, port, n, r, c, X, order, q)) for k in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    Y = np.zeros_like(exp)
    covered = np.zeros((n, F), np.int32)
    for rank, rows, f0, f1, y in got:
        Y[rows, f0:f1] = y
        covered[rows, f0:f1] += 1
    assert (covered == 1).all()
    assert np.linalg.norm(Y - exp) / np.linalg.norm(exp) <= 1e-5


def test_grid_layout_slices():
    from sgnn_amd.dist import grid_layout
    for world, fg, F in [(8, 2, 602), (8, 8, 602), (4, 4, 128), (8, 4, 100), (2, 2, 3), (6, 3, 7)]:
        seen = []
        for rank in range(world):
            row_rank, row_world, fgroup, f0, f1 = grid_layout(rank, world, fg, F)
            assert row_world == world // fg and 0 <= row_rank < row_world and rank == fgroup * row_world + row_rank
            assert 0 <= f0 < f1 <= F
            if row_rank == 0:
                seen.append((f0, f1))
        assert seen[0][0] == 0 and seen[-1][1] == F
        assert all(a[1] == b[0] for a, b in zip(seen, seen[1:]))
        widths = [b - a for a, b in seen]
        assert max(widths) - min(widths) <= 2
        if F % 2 == 0 and F >= 4 * fg:
            assert all(a % 2 == 0 for a, _ in seen)
    assert grid_layout(5, 8, 2, 602) == (1, 4, 1, 302, 602)
    with pytest.raises(ValueError):
        grid_layout(0, 8, 3, 602)


def test_balanced_splits_and_padding():
    from sgnn_amd.dist import balanced_row_splits, pad_index_map
    counts = np.array([100, 1, 1, 1, 1, 50, 50, 1, 1, 1])
    s = balanced_row_splits(counts, 3)
    assert s[0] == 0 and s[-1] == 10 and all(a <= b for a, b in zip(s, s[1:]))
    loads = [counts[s[p]:s[p + 1]].sum() for p in range(3)]
    assert max(loads) <= 110                        # the heavy first row is a block of its own
    m, max_rows = pad_index_map(s)
    assert max_rows == max(s[p + 1] - s[p] for p in range(3))
    assert len(set(m.tolist())) == 10 and m.max() < 3 * max_rows
    for p in range(3):
        assert (m[s[p]:s[p + 1]] == p * max_rows + np.arange(s[p + 1] - s[p])).all()
    assert balanced_row_splits(np.ones(5), 1) == [0, 5]
    assert balanced_row_splits(np.ones(2), 4)[-1] == 2     # more ranks than rows: empty blocks allowed


def test_degree_cyclic_deal_balances_rows_and_entries():
    from sgnn_amd.dist import degree_cyclic_deal
    from sgnn_amd.graphgen import symmetric_simple_numpy
    n = 5000
    r, c = symmetric_simple_numpy(n, 60_000, seed=8)
    counts = np.bincount(r, minlength=n) + 1
    for world in (2, 4, 8):
        order, max_rows = degree_cyclic_deal(counts, world)
        order = order.numpy()
        assert np.array_equal(np.sort(order), np.arange(n)) and max_rows == -(-n // world)
        assert (np.diff(counts[order]) <= 0).all()             # degree order
        loads = [int(counts[order[p::world]].sum()) for p in range(world)]
        sizes = [order[p::world].size for p in range(world)]
        assert max(sizes) - min(sizes) <= 1
        assert max(loads) - min(loads) <= counts.max()          # within one row's entries of each other


class _OracleWhole:
    """CPU stand-in for the whole-graph device adjacency of the feature-split schedule."""

    def __init__(self, row, col, val, n):
        import scipy.sparse as sp
        r, c = np.asarray(row), np.asarray(col)
        v = np.ones(r.size, np.float32) if val is None else np.asarray(val, np.float32)
        self.A = sp.coo_matrix((v, (r, c)), shape=(n, n))
        self.nnz = int(r.size) + n

    def propagate(self, X, order):
        return torch.from_numpy(orc.propagate_scipy(self.A, X.numpy().copy(), order))


def _feature_worker(rank, world, port, n, r, c, X, order, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sgnn_amd.dist import FeatureSplitAdj
        fadj = FeatureSplitAdj.from_coo(r, c, None, n, "cpu", rank, world, adj_factory=_OracleWhole)
        rows = fadj.rows.numpy()
        Y = fadj.propagate(torch.from_numpy(X[rows].copy()), order)
        Y2 = fadj.propagate(torch.from_numpy(X[rows].copy()), order)      # buffers are reused: same result
        assert torch.equal(Y, Y2) and fadj.last_schedule == "feature-split"
        q.put((rank, rows, Y.numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,order,n,F", [(2, 1, 700, 12), (3, 2, 701, 10), (4, 3, 699, 7), (8, 2, 700, 5)])
def test_feature_split_matches_single(world, order, n, F):
    """Every rank holds the whole graph and propagates its slice of the feature columns; a
    caller who owns rows gets there and back by two all-to-alls.  Uneven blocks (n not a
    multiple of the world size), slices that come out short, and ranks whose slice is empty
    (8 ranks, 5 features in slices of 4: ranks 2..7 multiply nothing)."""
    from sgnn_amd.graphgen import symmetric_simple_numpy
    r, c = symmetric_simple_numpy(n, 4000, seed=5)
    X = np.random.default_rng(1).standard_normal((n, F)).astype(np.float32)
    exp = orc.propagate_scipy(__import__("scipy.sparse").sparse.coo_matrix(
        (np.ones(r.size, np.float32), (r, c)), shape=(n, n)), X, order)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_feature_worker, args=(k, world, port, n, r, c, X, order, q)) for k in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    Y = np.zeros_like(exp)
    covered = np.zeros(n, bool)
    for rank, rows, y in got:
        assert not covered[rows].any()
        covered[rows] = True
        Y[rows] = y
    assert covered.all()
    # columns are independent and every column is propagated whole on one rank: the oracle's
    # arithmetic on a column slice is the oracle's arithmetic on the whole matrix
    assert np.array_equal(Y, exp)
