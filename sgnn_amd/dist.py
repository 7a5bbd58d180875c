"""Row-partitioned propagation over the GPUs of one node (one process per GPU).

The reference's own multi-GPU mode is replication: every rank loads the whole
graph and recomputes the whole ``A_hat . X`` (main.py:39-41, model/SGNN.py:518-520
— the DDP wrapper is bypassed through ``.module.run``), so there is nothing to be
compatible with beyond the result.  Here the rows of ``A_hat`` are dealt out to the
``world`` ranks; rank p owns its rows of the matrix, of ``dinv`` and of X / Y
(``PartitionedAdj.rows`` lists them).  The default deal is cyclic over the rows in
degree order (row of degree rank k goes to rank k mod world): every rank gets the
same number of rows and, to within one row's worth, the same number of stored
entries, so the exchange carries no padding and the SpMMs finish together.  (A
contiguous split balanced by stored entries is kept as ``partition="contiguous"``;
on the Reddit-shape graph its blocks hold 3 107 ... 76 195 rows at 8 ranks and the
equal-piece all-gather moves 2.6 times the rows there are.)  One hop is

  1. scale the local rows of X by the local ``dinv``          (utils.py:93)
  2. all-gather the scaled blocks into one padded buffer      (RCCL over xGMI)
  3. SpMM of the local row block against the gathered buffer  (utils.py:94)
  4. scale the local output rows by the local ``dinv``        (utils.py:95, fused)

The local rows' CSR is also kept cut in two by column ownership -- the block over the rank's
own columns and the block over everyone else's -- and the default schedule
(``overlap="local-block"``) multiplies the own-columns block, which needs nothing from the
other ranks, on the compute stream while the exchange runs on a second stream; the
remote-columns block follows when the rows have landed and the two raw sums are added in that
fixed order before the row scaling (deterministic; ``overlap="none"`` keeps the one-block
schedule, whose rows are reduced exactly as on one GPU).  Where the remote-columns block is
large and shared enough for the tile-group (matrix-core) kernel, what travels is not fp32
rows but the operand's fp16 planes (the same four bytes per element), one all-gather per
128-feature slice: a rank splits only its own rows, after the ranks have agreed on the
per-feature maxima with one small all-reduce, and on later hops the epilogue writes the
rank's rows of the next operand straight into its piece of the next gather buffer.  The
planes travel in units of ``slices_per_exchange`` slices and the remote-columns block is
multiplied unit by unit as they land, so that only the last unit's multiply follows the last
byte; launches that run beside an exchange leave ``reserve_cus`` CUs to the collective.

Blocks are padded to a common row count so that the exchange is ONE
``all_gather_into_tensor`` of equal pieces; node ids are remapped once into the
padded index space (owner * max_rows + position in the owner's block), in which
every rank's rows are contiguous.  Pre-scaling before the exchange means
``dinv`` itself never travels.  With ``slices > 1`` the feature dimension is cut
into column slices and the all-gather of slice s+1 runs on a side stream while
slice s is multiplied, which hides most of the exchange behind the SpMM.

``exchange="referenced"`` replaces the all-gather by an all-to-all of exactly the rows each
rank's block references (the index lists are fixed at build time; columns are renumbered into
own rows followed by the received rows in rank order).  It moves ``referenced / all`` of the
bytes: 82 % on the Reddit shape at 8 ranks, where it is not worth its extra gather pass, but
39 % on ogbn-products and 25 % on R-MAT 10 M; ``exchange="auto"`` picks it when at most 60 %
of the rows a rank would receive are used.

The columns of X are independent, so the feature dimension can be split over ranks
with no exchange at all.  ``grid_layout`` arranges the ranks as ``feature_groups`` x
``row_ranks``: a rank owns one block of rows and one slice of features, and the
all-gather above runs only among the ranks that share its feature slice
(``make_row_groups`` builds those sub-groups).  ``feature_groups == world`` needs no
communication; 1 is the pure row partition; values in between trade the exchange
volume (halved per doubling) against narrower, less efficient rows per rank.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .norm_adj import TILEGROUP_MIN_NNZ, NormAdj, _keep, tilegroup_build, tilegroup_buffers, tilegroup_hop

TILEGROUP_MIN_ENTRIES_PER_OPERAND_ROW = 40


def ctypes_ptr(t, byte_offset=0):
    """Device pointer ``byte_offset`` bytes into a tensor."""
    import ctypes
    return ctypes.c_void_p(t.data_ptr() + int(byte_offset))


def grid_layout(rank, world, feature_groups, F, align=2):
    """Place ``rank`` on a ``feature_groups`` x ``world // feature_groups`` grid:
    consecutive ranks share a feature slice (and exchange rows with each other).
    Returns ``(row_rank, row_world, fgroup, f0, f1)``; slice boundaries are multiples
    of ``align`` floats where F allows, so that vector loads stay usable."""
    if feature_groups < 1 or world % feature_groups:
        raise ValueError(f"feature_groups={feature_groups} does not divide world={world}")
    if feature_groups > F:
        raise ValueError("more feature groups than features")
    row_world = world // feature_groups
    fgroup, row_rank = divmod(rank, row_world)

    def cut(k):
        if k == 0 or k == feature_groups:
            return 0 if k == 0 else F
        c = F * k // feature_groups
        a = (c + align // 2) // align * align
        return a if F * (k - 1) // feature_groups < a < F else c
    cuts = [cut(k) for k in range(feature_groups + 1)]
    if any(b <= a for a, b in zip(cuts, cuts[1:])):           # tiny F: fall back to unaligned cuts
        cuts = [F * k // feature_groups for k in range(feature_groups + 1)]
    return row_rank, row_world, fgroup, cuts[fgroup], cuts[fgroup + 1]


def make_row_groups(world, feature_groups):
    """Process group of the ranks that share this rank's feature slice (collective:
    every rank must call it, with the same arguments).  ``None`` when the default
    group already is that set (``feature_groups == 1``) or when the rank exchanges
    with nobody (``feature_groups == world``)."""
    row_world = world // feature_groups
    if feature_groups == 1 or row_world == 1:
        return None
    mine, rank = None, dist.get_rank()
    for g in range(feature_groups):
        ranks = list(range(g * row_world, (g + 1) * row_world))
        grp = dist.new_group(ranks)                             # every rank creates every group, in order
        if rank in ranks:
            mine = grp
    return mine


def balanced_row_splits(counts, world):
    """Split points r_0=0 <= ... <= r_world=n such that every block holds about
    the same number of stored entries.  ``counts`` = entries per row (1-D tensor
    or array); deterministic, so every rank computes the same answer."""
    c = torch.as_tensor(counts).to(torch.int64).cpu()
    n = c.numel()
    csum = torch.cumsum(c, 0)
    total = int(csum[-1]) if n else 0
    splits = [0]
    for p in range(1, world):
        target = (total * p + world - 1) // world
        r = int(torch.searchsorted(csum, torch.tensor(target, dtype=torch.int64)).item()) + 1
        splits.append(min(max(r, splits[-1]), n))
    splits.append(n)
    return splits


def pad_index_map(splits):
    """Column id -> position in the padded gather buffer (block p starts at
    p * max_rows), for a contiguous split."""
    world = len(splits) - 1
    max_rows = max(splits[p + 1] - splits[p] for p in range(world))
    n = splits[-1]
    m = np.empty(n, dtype=np.int64)
    for p in range(world):
        m[splits[p]:splits[p + 1]] = p * max_rows + np.arange(splits[p + 1] - splits[p])
    return m, max_rows


def degree_cyclic_deal(counts, world):
    """Deal the rows out cyclically in order of decreasing ``counts`` (stable): the row of
    degree rank k goes to rank ``k % world``, position ``k // world``.  Returns
    ``(order, max_rows)`` with ``order`` the rows by degree rank (int64 tensor on the
    device ``counts`` lives on: the sort runs there); deterministic, so every rank computes
    the same deal."""
    c = torch.as_tensor(counts).to(torch.int64)
    order = torch.argsort(c, descending=True, stable=True)
    return order, (c.numel() + world - 1) // world


def _scatter_hash(ids):
    """Deterministic pseudo-random key per node id (the same on every rank and device)."""
    x = (ids * 2654435761) & 0xFFFFFFFF
    x = x ^ (x >> 15)
    x = (x * 1540483477) & 0xFFFFFFFF          # < 2^31: the product stays inside int64
    return x ^ (x >> 13)


class PartitionedAdj:
    """Rank-local rows of ``A + I`` plus what one hop needs.  ``rows`` (int64 tensor,
    CPU) lists the global rows this rank owns, in the order of its local block: the
    caller hands ``X[rows]`` to ``propagate`` and gets ``Y[rows]`` back."""

    def __init__(self, local, rows, max_rows, rank, world, n, global_nnz, global_max_degree, group=None,
                 splits=None, halo=None, own=None, remote=None):
        self.local = local                      # block over (rows_local x padded columns)
        # the same rows cut by column ownership (None: one-block schedule only)
        self.own, self.remote = own, remote
        self.overlap = "local-block" if own is not None else "none"
        # exchange fp16 planes and multiply on the matrix cores: None = where the remote-columns
        # block merits the tile-group kernel (all ranks agree), True / False force it
        self.use_planes = None
        self._planes_decision = None
        self._plane_bufs = {}
        self._raw = {}
        self.last_schedule = None               # what the last propagate call ran (diagnostics, bench line)
        self.rows, self.rank, self.world, self.n = rows, rank, world, n
        self.n_local = int(rows.numel())
        self.max_rows = int(max_rows)
        self.splits = splits                    # contiguous partition only
        self.row_begin = splits[rank] if splits is not None else None
        self.row_end = splits[rank + 1] if splits is not None else None
        self.global_nnz, self.global_max_degree = global_nnz, global_max_degree
        self.group = group
        # referenced-rows exchange: dict(send_idx, send_counts, recv_counts, n_recv) or None (all-gather)
        self.halo = halo
        self._gather = {}
        self._slice_bufs = {}
        self._comm_stream = None
        self.reserve_cus = 32                   # see _reserved_cus
        self.slices_per_exchange = 2            # 128-feature slices per exchange / multiply unit of the planes schedule

    # ------------------------------------------------------------------ build
    @classmethod
    def from_coo(cls, row, col, val, n, device, rank, world, group=None, local_factory=None,
                 partition="degree_cyclic", exchange="allgather", overlap="local-block"):
        """Every rank passes the same COO (entries of other ranks' rows are dropped
        here).  ``local_factory`` is a test hook that builds the local block from
        ``(row_padded, col_padded, val, n_rows, n_padded, first_padded_row)``; the
        default builds the device CSR through the C ABI."""
        row = torch.as_tensor(row).to(torch.int64)
        col = torch.as_tensor(col).to(torch.int64)
        counts = torch.bincount(row, minlength=n) + 1            # + the self-loop every row gets
        global_nnz = int(counts.sum().item())
        global_max_degree = int(counts.max().item())
        splits = None
        if partition == "contiguous":
            splits = balanced_row_splits(counts, world)
            padmap, max_rows = pad_index_map(splits)
            pad = torch.from_numpy(padmap)
            rows = torch.arange(splits[rank], splits[rank + 1], dtype=torch.int64)
        elif partition == "degree_cyclic":
            order, max_rows = degree_cyclic_deal(counts, world)
            # Positions inside a rank's block are scattered (by a hash of the node id, the same
            # everywhere): left in degree order, the hubs of every block would be numbered next
            # to each other and the SpMM, which walks the columns block by block, would meet
            # all hot rows of X at once instead of a few per block (measured: 12.6 instead of
            # 10.4 ms).  Everything here runs on the device the COO lives on.
            pad = torch.empty(n, dtype=torch.int64, device=order.device)
            rows = None
            for p in range(world):
                mine_p = order[p::world]
                mine_p = mine_p[torch.argsort(_scatter_hash(mine_p), stable=True)]
                pad[mine_p] = p * max_rows + torch.arange(mine_p.numel(), dtype=torch.int64, device=order.device)
                if p == rank:
                    rows = mine_p.cpu()
        else:
            raise ValueError(f"unknown partition {partition!r}")
        # in the padded index space a rank's rows are the contiguous range
        # [rank * max_rows, rank * max_rows + n_local)
        pad = pad.to(row.device)
        prow = pad[row]
        r0 = rank * max_rows
        mine = (prow >= r0) & (prow < r0 + rows.numel())
        lrow, lcol = prow[mine], pad[col[mine]]
        lval = None if val is None else torch.as_tensor(val)[mine]
        n_pad = max_rows * world
        if exchange not in ("allgather", "referenced", "auto"):
            raise ValueError(f"unknown exchange {exchange!r}")
        halo = None
        if exchange != "allgather" and world > 1:
            # rows of other ranks this rank's block references, in padded (owner-major) order
            foreign = (lcol < r0) | (lcol >= r0 + max_rows)
            need = torch.unique(lcol[foreign])
            use = exchange == "referenced" or need.numel() <= 0.6 * (world - 1) * max_rows
            if exchange == "auto" and dist.is_initialized():
                # every rank of the exchange must take the same decision
                flag = torch.tensor([1 if use else 0], dtype=torch.int32,
                                    device="cpu" if dist.get_backend(group) == "gloo" else device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
                use = bool(flag.item())
            if use:
                pcol = pad[col]
                # rows of this rank's block that each other rank references: key = (target, position)
                theirs = ((prow < r0) | (prow >= r0 + max_rows)) & (pcol >= r0) & (pcol < r0 + max_rows)
                key = torch.unique((prow[theirs] // max_rows) * max_rows + (pcol[theirs] - r0))
                send_counts = torch.bincount(key // max_rows, minlength=world).tolist()
                recv_counts = torch.bincount(need // max_rows, minlength=world).tolist()
                halo = dict(send_idx=(key % max_rows).contiguous(), send_counts=send_counts, recv_counts=recv_counts,
                            n_recv=int(need.numel()))
                # compact column space: own rows at [0, max_rows), received rows behind them
                own = ~foreign
                ccol = torch.empty_like(lcol)
                ccol[own] = lcol[own] - r0
                ccol[foreign] = max_rows + torch.searchsorted(need, lcol[foreign])
                lrow, lcol = lrow - r0, ccol
                n_pad, r0 = max_rows + int(need.numel()), 0
                del pcol, theirs, key
        if local_factory is not None:
            local = local_factory(lrow, lcol, lval, int(rows.numel()), n_pad, r0)
        else:
            local = _LocalBlock.build(lrow, lcol, lval, n_pad, r0, r0 + int(rows.numel()), device)
        if halo is not None:
            halo["send_idx"] = halo["send_idx"].to(device)
        if overlap not in ("local-block", "none"):
            raise ValueError(f"unknown overlap {overlap!r}")
        own = remote = None
        if overlap == "local-block" and world > 1 and hasattr(local, "split_columns"):
            own, remote = local.split_columns(r0, r0 + max_rows)
        return cls(local, rows, max_rows, rank, world, n, global_nnz, global_max_degree, group, splits, halo, own, remote)

    # ------------------------------------------------------------------ apply
    def _buffers(self, F, device, dtype):
        key = (F, device, dtype)
        b = self._gather.get(key)
        if b is None:
            rows = self.max_rows * self.world if self.halo is None else self.max_rows + self.halo["n_recv"]
            Fp = (F + 3) // 4 * 4      # rows on 16-byte boundaries for the row-group kernel's loads
            b = [torch.zeros((rows, Fp), dtype=dtype, device=device) for _ in range(2)]
            _keep(self._gather, key, b)      # a stack of layers alternates between a few widths
        return b

    def _all_gather(self, full, piece, force=False):
        if self.world == 1 and not force:
            return
        if piece.is_cuda and dist.get_backend(self.group) == "gloo":
            # gloo has no GPU all-gather: stage through the host (several ranks sharing
            # one GPU in tests; production runs use the nccl backend = RCCL)
            host = torch.empty(full.shape, dtype=full.dtype)
            dist.all_gather_into_tensor(host.view(-1), piece.cpu().reshape(-1), group=self.group)
            # (the rank's own piece is already in place and may be being read by the own-columns multiply)
            lo, hi = self.rank * self.max_rows, (self.rank + 1) * self.max_rows
            full[:lo].copy_(host[:lo])
            full[hi:].copy_(host[hi:])
            return
        # on the GPU the piece is sent in place (it already sits in its slot of `full`)
        inp = piece if piece.is_cuda else piece.clone()
        dist.all_gather_into_tensor(full.view(-1), inp.reshape(-1), group=self.group)

    def _exchange_referenced(self, full):
        """All-to-all of the referenced rows: the rows other ranks need are gathered out of this
        rank's block (first ``max_rows`` rows of ``full``) and what this rank needs lands behind it."""
        h = self.halo
        send = full[:self.max_rows].index_select(0, h["send_idx"])
        recv = full[self.max_rows:]
        if full.is_cuda and dist.get_backend(self.group) == "gloo":      # tests: several ranks on one GPU
            host = torch.empty(recv.shape, dtype=recv.dtype)
            dist.all_to_all_single(host, send.cpu(), h["recv_counts"], h["send_counts"], group=self.group)
            recv.copy_(host)
            return
        dist.all_to_all_single(recv, send, h["recv_counts"], h["send_counts"], group=self.group)

    def propagate(self, X_local, order=1, slices=1, overlap=None):
        """``A_hat^order . X`` for the local rows; X_local = this rank's rows of X.
        Returns this rank's rows of the result.  ``overlap``: None = the schedule the handle
        was built for ("local-block" unless built with ``overlap="none"``)."""
        order = int(order)
        if order <= 0:
            return X_local
        rows = self.n_local
        if X_local.shape[0] != rows:
            raise ValueError(f"rank {self.rank} owns {rows} rows, got {X_local.shape[0]}")
        F = X_local.shape[1]
        mode = self.overlap if overlap is None else overlap
        if mode not in ("local-block", "none"):
            raise ValueError(f"unknown overlap {mode!r}")
        if self.own is None or self.world == 1:
            mode = "none"
        if slices > 1 and X_local.is_cuda and self.halo is None:
            self.last_schedule = "sliced"
            return self._propagate_sliced(X_local, order, slices)
        if mode == "local-block":
            if self._planes_ok(X_local):
                out = self._propagate_planes(X_local, order)
                if out is not None:
                    self.last_schedule = "local-block, fp16 planes exchanged"
                    return out
            self.last_schedule = "local-block"
            return self._propagate_split(X_local, order)
        self.last_schedule = "none"
        bufs = self._buffers(F, X_local.device, X_local.dtype)
        lo = self.rank * self.max_rows if self.halo is None else 0
        src = X_local
        out = None
        for h in range(order):
            full = bufs[h % 2]                       # [padded rows, F rounded up to 4]; whole rows travel
            piece = full[lo:lo + self.max_rows]
            if h == 0:
                self.local.scale_rows_into(src, piece[:rows, :F])       # utils.py:93
            # later hops: the previous epilogue already wrote the doubly scaled rows into `piece`
            if self.halo is None:
                self._all_gather(full, piece)
            else:
                self._exchange_referenced(full)
            last = h == order - 1
            if last:
                out = torch.empty((rows, F), dtype=X_local.dtype, device=X_local.device)
                self.local.spmm_into(full[:, :F], out, final=True)
            else:
                nxt = bufs[(h + 1) % 2][lo:lo + rows, :F]
                self.local.spmm_into(full[:, :F], nxt, final=False)     # epilogue: (s*d)*d
        return out

    # ---------------------------------------------------------- local-block overlap
    def _streams(self, device):
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=device)
        return torch.cuda.current_stream(device), self._comm_stream

    def _reserved_cus(self):
        """CUs the matrix-core kernel leaves free while an exchange runs beside it: its
        workgroups hold their CU (all of its LDS) until the launch's work is done, so a
        collective kernel that arrives later would otherwise wait for the multiply to end.
        Nothing to reserve where the exchange is staged through the host (gloo)."""
        if self.world == 1 or not dist.is_initialized() or dist.get_backend(self.group) == "gloo":
            return 0
        return int(self.reserve_cus)

    def _propagate_split(self, X_local, order):
        """Exchange of fp32 rows overlapped with the own-columns multiply: the own-columns block
        reads only the rank's own (already scaled) rows, so it starts at once on the compute
        stream while the exchange runs on the other; the remote-columns block follows when the
        rows have landed; the raw sums of the two are added in that order and row-scaled."""
        rows, F = self.n_local, X_local.shape[1]
        dev = X_local.device
        bufs = self._buffers(F, dev, X_local.dtype)
        key = (F, dev, X_local.dtype)
        raw = self._raw.get(key)
        if raw is None:
            raw = _keep(self._raw, key, [torch.empty((rows, F), dtype=X_local.dtype, device=dev) for _ in range(2)])
        s1, s2 = raw
        lo = self.rank * self.max_rows if self.halo is None else 0
        cuda = X_local.is_cuda
        main, comm = self._streams(dev) if cuda else (None, None)
        out = None
        for h in range(order):
            full = bufs[h % 2]
            piece = full[lo:lo + self.max_rows]
            if h == 0:
                self.local.scale_rows_into(X_local, piece[:rows, :F])       # utils.py:93
            if cuda:
                ready = torch.cuda.Event()
                ready.record(main)
                with torch.cuda.stream(comm):
                    comm.wait_event(ready)
                    self._all_gather(full, piece) if self.halo is None else self._exchange_referenced(full)
                    done = torch.cuda.Event()
                    done.record(comm)
                self.own.spmm_raw_into(full[:, :F], s1)          # while the other ranks' rows travel
                main.wait_event(done)
            else:
                self._all_gather(full, piece) if self.halo is None else self._exchange_referenced(full)
                self.own.spmm_raw_into(full[:, :F], s1)
            self.remote.spmm_raw_into(full[:, :F], s2)
            last = h == order - 1
            if last:
                out = torch.empty((rows, F), dtype=X_local.dtype, device=dev)
                self.local.finish_into(s1, s2, out, final=True)
            else:
                self.local.finish_into(s1, s2, bufs[(h + 1) % 2][lo:lo + rows, :F], final=False)
        return out

    def _agree(self, value, device, op):
        """One integer agreed on by the ranks of the exchange (collective)."""
        if self.world == 1 or not dist.is_initialized():
            return int(value)
        host = dist.get_backend(self.group) == "gloo"
        t = torch.tensor([int(value)], dtype=torch.int32, device="cpu" if host else device)
        dist.all_reduce(t, op=op, group=self.group)
        return int(t.item())

    def _planes_ok(self, X_local):
        """Whether this propagation exchanges fp16 planes and multiplies on the matrix cores: the
        remote-columns block must merit the tile-group kernel (uniform weights, at least 4 M
        stored entries, rows that share enough columns: the tests of NormAdj._tilegroup_plan, on
        what is by far the larger of the rank's two blocks) on every rank."""
        if not X_local.is_cuda or X_local.dtype != torch.float32 or self.use_planes is False:
            return False
        if self.own is None or not hasattr(self.own, "tilegroup_plan") or X_local.shape[1] < 16:
            return False
        if self._planes_decision is None:
            force = self.use_planes is True
            ok = self.remote.adj.w_off is not None and self.remote.adj.nnz > 0
            if ok and not force:
                ok = self.remote.adj.nnz >= TILEGROUP_MIN_NNZ
            if ok:
                ok = self.remote.tilegroup_plan(auto=not force) is not None
            ok = bool(self._agree(1 if ok else 0, X_local.device, dist.ReduceOp.MIN))
            if ok:
                ok = self.own.tilegroup_plan(auto=False) is not None
            self._planes_decision = ok
        return self._planes_decision

    def _planes_buffers(self, F, order, device):
        lib = _lib.load()
        b = self._plane_bufs.get(F)
        n_cols = self.local.n_cols
        S = (F + 127) // 128
        if b is None:
            nbytes = lib.sgnn_tilegroup_planes_bytes(n_cols, F)
            sb = lib.sgnn_tilegroup_state_bytes(F)
            b = dict(flat=[torch.zeros(nbytes, dtype=torch.uint8, device=device) for _ in range(2)], nbytes=nbytes,
                     sb=sb, states=torch.zeros((1, sb // 4), dtype=torch.int32, device=device),
                     s1=torch.empty((self.n_local, F), dtype=torch.float32, device=device),
                     ws_own=torch.empty(max(lib.sgnn_spmm_tilegroup_workspace_bytes(self.own._tg["slots"], F), 256),
                                        dtype=torch.uint8, device=device),
                     ws_rem=torch.empty(max(lib.sgnn_spmm_tilegroup_workspace_bytes(self.remote._tg["slots"], F), 256),
                                        dtype=torch.uint8, device=device))
            b["G"] = [f[:S * n_cols * 512].view(S, n_cols, 512) for f in b["flat"]]
            _keep(self._plane_bufs, F, b)
        if b["states"].shape[0] < order:
            b["states"] = torch.zeros((order, b["sb"] // 4), dtype=torch.int32, device=device)
        return b

    def _exchange_planes(self, G, s0=0, s1=None):
        """One exchange per 128-feature slice (slices s0 .. s1 - 1): a slice of the operand is
        [padded nodes, 512 bytes] and the rank's piece of it is contiguous."""
        lo = self.rank * self.max_rows if self.halo is None else 0
        for s in range(s0, G.shape[0] if s1 is None else s1):
            full = G[s]
            if self.halo is None:
                self._all_gather(full, full[lo:lo + self.max_rows])
            else:
                self._exchange_referenced(full)

    def _propagate_planes(self, X_local, order):
        """Local-block overlap with the operand exchanged as fp16 planes (see the module text).
        The operand travels in units of ``slices_per_exchange`` 128-feature slices, and the
        remote-columns block is multiplied unit by unit as they land: the multiply of one unit
        runs while the next is on the wire.
        Returns None when some rank's operand could not go through the planes (non-finite
        values, a dynamic range fp16 pairs cannot hold): the caller then runs the fp32 schedule."""
        lib = _lib.load()
        rows, F = self.n_local, X_local.shape[1]
        dev = X_local.device
        X_local = X_local if X_local.stride(1) == 1 else X_local.contiguous()
        b = self._planes_buffers(F, order, dev)
        n_cols = self.local.n_cols
        lo = self.rank * self.max_rows if self.halo is None else 0
        Fpad = (F + 127) // 128 * 128
        S = Fpad // 128
        per = max(1, int(self.slices_per_exchange))
        units = [(s, min(per, S - s)) for s in range(0, S, per)]
        reserve = self._reserved_cus()
        st = _lib.current_stream(dev)
        main, comm = self._streams(dev)
        states = b["states"]
        own, rem = self.own, self.remote
        dinv = self.local.dinv
        # operand 0: per-feature maxima of the local rows, agreed on by all ranks, then the local split
        _lib.check(lib.sgnn_tilegroup_amax_f32(rows, F, _lib.ptr(dinv), _lib.ptr(X_local), X_local.stride(0),
                                               _lib.ptr(states[0]), st))
        amax = states[0, _lib.TG_STATE_HEAD_WORDS:_lib.TG_STATE_HEAD_WORDS + Fpad]
        if dist.get_backend(self.group) == "gloo":
            host = amax.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.MAX, group=self.group)
            amax.copy_(host)
        else:
            dist.all_reduce(amax, op=dist.ReduceOp.MAX, group=self.group)
        _lib.check(lib.sgnn_tilegroup_planes_f32(rows, F, _lib.ptr(dinv), _lib.ptr(X_local), X_local.stride(0),
                                                 ctypes_ptr(b["flat"][0], lo * 512), n_cols, rows, _lib.ptr(states[0]), st))
        out = torch.empty((rows, F), dtype=torch.float32, device=dev)
        ready = torch.cuda.Event()
        ready.record(main)
        written = [ready] * len(units)            # when the rank's own rows of a unit's slices are in the gather buffer
        for h in range(order):
            G, flat, state = b["G"][h % 2], b["flat"][h % 2], states[h]
            last = h == order - 1
            landed = []
            with torch.cuda.stream(comm):
                for u, (s0, cnt) in enumerate(units):
                    comm.wait_event(written[u])
                    self._exchange_planes(G, s0, s0 + cnt)
                    ev = torch.cuda.Event()
                    ev.record(comm)
                    landed.append(ev)
            # own-columns block: raw pattern sums into s1, from the rank's own rows of the planes
            po = own._tg
            _lib.check(lib.sgnn_spmm_tilegroup_slices_f32(
                rows, n_cols, F, _lib.ptr(po["buf"]), po["bytes"], po["stages"], po["steps"], po["items"], po["slots"],
                po["n_long"], None, 1.0, None, _lib.ptr(flat), b["nbytes"], _lib.ptr(state), _lib.ptr(b["s1"]), F,
                _lib.EPI_SCALE_ROW, _lib.ptr(b["ws_own"]), b["ws_own"].numel(), own.adj.nnz, None, None, None, None, None, 0,
                None, 0, 0, None, 0, None, 0, 0, None, 0, 0, 0, reserve, st))
            # remote-columns block, unit by unit as the slices land: + s1, weights, diagonal
            # correction, row scaling; the last hop writes fp32, the others the rank's rows of the
            # next operand's planes
            pr = rem._tg
            nxt = b["flat"][(h + 1) % 2]
            written_next = []
            for u, (s0, cnt) in enumerate(units):
                main.wait_event(landed[u])
                _lib.check(lib.sgnn_spmm_tilegroup_slices_f32(
                    rows, n_cols, F, _lib.ptr(pr["buf"]), pr["bytes"], pr["stages"], pr["steps"], pr["items"], pr["slots"],
                    pr["n_long"], _lib.ptr(dinv), float(own.adj.w_off), _lib.ptr(own.adj.corr), _lib.ptr(flat), b["nbytes"],
                    _lib.ptr(state), _lib.ptr(out) if last else None, F,
                    _lib.EPI_SCALE_ROW if last else _lib.EPI_SCALE_ROW2, _lib.ptr(b["ws_rem"]), b["ws_rem"].numel(),
                    rem.adj.nnz, None, None, None, None, None, 0, None, 0, lo, _lib.ptr(b["s1"]), F,
                    None if last else _lib.ptr(nxt), n_cols, lo, None if last else _lib.ptr(states[h + 1]), 0,
                    s0, cnt, reserve if (u + 1 < len(units) or not last) else 0, st))
                if not last:                      # the next hop's exchange of this unit may start
                    ev = torch.cuda.Event()
                    ev.record(main)
                    written_next.append(ev)
            written = written_next
        # one agreement per call: did every rank's every operand go through the planes?
        head = states[:order, :8].cpu()           # synchronises
        refused = bool((head[:, _lib.TG_STATE_FLAG] != 0).any()
                       or (head[:, _lib.TG_STATE_TINY] > head[:, _lib.TG_STATE_TINY_LIMIT]).any())
        if self._agree(1 if refused else 0, dev, dist.ReduceOp.MAX):
            return None
        return out

    def _propagate_sliced(self, X_local, order, slices):
        """Feature-sliced pipeline: exchange slice s+1 while slice s is multiplied."""
        rows = self.n_local
        F = X_local.shape[1]
        dev = X_local.device
        bounds = [(F * s) // slices for s in range(slices + 1)]
        bounds = [b - b % 4 for b in bounds[:-1]] + [F]
        bounds = sorted(set(bounds))
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=dev)
        comm = self._comm_stream
        main = torch.cuda.current_stream(dev)
        lo = self.rank * self.max_rows
        cur = X_local
        # one gather buffer per slice, allocated (and its padding rows zeroed) once and kept: hop
        # h + 1 writes a slice's piece only after every SpMM of hop h, in stream order, and the
        # exchange of a slice waits for that write, so one set serves every hop
        key = (F, tuple(bounds), X_local.dtype, dev)
        fulls = self._slice_bufs.get(key)
        if fulls is None:
            fulls = [torch.zeros((self.max_rows * self.world, (bounds[s + 1] - bounds[s] + 3) // 4 * 4),
                                 dtype=X_local.dtype, device=dev) for s in range(len(bounds) - 1)]
            _keep(self._slice_bufs, key, fulls)
        for h in range(order):
            last = h == order - 1
            out = torch.empty((rows, F), dtype=X_local.dtype, device=dev)
            events = []
            for s in range(len(bounds) - 1):
                f0, f1 = bounds[s], bounds[s + 1]
                w = f1 - f0
                full = fulls[s]
                piece = full[lo:lo + self.max_rows]
                if h == 0:
                    self.local.scale_rows_into(cur[:, f0:f1], piece[:rows, :w])
                else:
                    piece[:rows, :w].copy_(cur[:, f0:f1])               # already doubly scaled
                ready = torch.cuda.Event()
                ready.record(main)
                with torch.cuda.stream(comm):
                    comm.wait_event(ready)
                    self._all_gather(full, piece)
                    done = torch.cuda.Event()
                    done.record(comm)
                events.append(done)
            for s in range(len(bounds) - 1):
                f0, f1 = bounds[s], bounds[s + 1]
                main.wait_event(events[s])
                self.local.spmm_into(fulls[s][:, :f1 - f0], out[:, f0:f1], final=last)
            cur = out
        return cur


class _LocalBlock:
    """Device CSR of one row block over the padded column space."""

    def __init__(self, adj, dinv_local, n_rows, n_cols_padded, x_row_offset):
        self.adj = adj                  # NormAdj whose rows are the local rows (square trick, see build)
        self.dinv = dinv_local
        self.n_rows, self.n_cols = n_rows, n_cols_padded
        self.x_row_offset = int(x_row_offset)   # where the block's own rows sit in the gathered X
        self._plans = {}
        self._ws = {}
        # tile-group (matrix-core) kernel on the rank's rows: None = where it pays (the same test
        # as NormAdj applies), True / False force it on / off
        self.use_tilegroup = None
        self.use_rowgroup = True        # False: the plain CSR kernel (whose rows are reduced exactly as on one GPU)
        self._tg = None

    @classmethod
    def build(cls, lrow, lcol, lval, n_pad, r0, r1, device):
        # Rows and columns arrive in the padded index space.  Build the block as an
        # n_pad x n_pad matrix that holds only the local rows (the library's builder is
        # square: it inserts the self-loops and computes the row degree), then cut the
        # local rows [r0, r1) out.
        full = NormAdj.from_coo(lrow.to(device), lcol.to(device), None if lval is None else lval.to(device),
                                n_pad, device)
        rp = full.rowptr
        e0, e1 = int(rp[r0].item()), int(rp[r1].item())
        rowptr = (rp[r0:r1 + 1] - e0).contiguous()
        colidx = full.colidx[e0:e1].contiguous()
        vals = full.vals[e0:e1].contiguous()
        # weight structure of the local rows only (the other rows of `full` hold just their self-loop)
        from .norm_adj import weight_structure
        unit, w_off, corr = weight_structure(r1 - r0, rowptr, colidx - r0, vals)
        block = NormAdj(r1 - r0, rowptr, colidx, vals, full.deg[r0:r1].contiguous(), full.dinv[r0:r1].contiguous(),
                        unit_weights=unit, w_off=w_off, corr=corr)
        return cls(block, block.dinv, r1 - r0, n_pad, r0)

    def scale_rows_into(self, src, dst):
        lib = _lib.load()
        src = src if src.stride(1) == 1 else src.contiguous()
        _lib.check(lib.sgnn_row_scale_f32(self.n_rows, src.shape[1], _lib.ptr(self.dinv), _lib.ptr(src), src.stride(0),
                                          _lib.ptr(dst), dst.stride(0), _lib.current_stream(src.device)))

    def _plan(self, F):
        a = self.adj
        if a.nnz == 0:
            return None
        lib = _lib.load()
        R = lib.sgnn_rowgroup_config_for(self.n_rows, a.nnz, F)
        if R == 0:
            return None
        plan = self._plans.get(R)
        if plan is None:
            dev = a.device
            pbytes = lib.sgnn_rowgroup_plan_bytes(self.n_rows, a.nnz, R)
            buf = torch.empty(pbytes, dtype=torch.uint8, device=dev)
            wbytes = lib.sgnn_rowgroup_workspace_bytes(self.n_rows, a.nnz, R)
            ws = torch.empty(wbytes, dtype=torch.uint8, device=dev)
            _lib.check(lib.sgnn_rowgroup_build(self.n_rows, self.n_cols, a.nnz, _lib.ptr(a.rowptr), _lib.ptr(a.colidx),
                                               R, _lib.ptr(buf), pbytes, _lib.ptr(ws), wbytes,
                                               _lib.current_stream(dev)))
            merged, items, slots, n_long = (int(v) for v in buf[:32].view(torch.int64).cpu())
            gw = None
            if a.w_off is None:         # stored values differ: per-entry weights ride with the plan
                gbytes = lib.sgnn_rowgroup_weights_bytes(merged, R)
                gw = torch.empty(gbytes, dtype=torch.uint8, device=dev)
                _lib.check(lib.sgnn_rowgroup_weights(self.n_rows, a.nnz, _lib.ptr(a.rowptr), _lib.ptr(a.colidx),
                                                     _lib.ptr(a.vals), R, _lib.ptr(buf), pbytes, merged,
                                                     _lib.ptr(gw), gbytes, _lib.current_stream(dev)))
            plan = dict(R=R, buf=buf, bytes=pbytes, items=items, slots=slots, n_long=n_long, gw=gw)
            self._plans[R] = plan
        return plan

    def _tilegroup_plan(self, F):
        a = self.adj
        if self.use_tilegroup is False or a.nnz == 0 or a.w_off is None or a.corr is not None:
            return None        # (a diagonal of its own would need the block's rows of X at an offset)
        auto = self.use_tilegroup is None
        if auto and (F < 16 or a.nnz < TILEGROUP_MIN_NNZ):     # the rule of NormAdj._tilegroup_plan, on the rank's block
            return None
        # A rank splits the whole operand it multiplies with (its own rows and everyone else's) but
        # multiplies only its own rows' entries: per node of the operand the max and split passes cost
        # what about 40 stored entries save against the row-group kernel (one GPU, per entry and
        # node: Reddit shape 44 ps saved, 2.2 ns spent; products 12 ps, 0.45 ns; R-MAT 10 M 25 ps,
        # 0.41 ns), so a block thinner than that stays on the row-group kernel.
        if auto and a.nnz < TILEGROUP_MIN_ENTRIES_PER_OPERAND_ROW * self.n_cols:
            return None
        if self._tg is None:
            self._tg = tilegroup_build(self.n_rows, self.n_cols, a.nnz, a.rowptr, a.colidx, a.device, auto)
        return self._tg or None

    def split_columns(self, c0, c1):
        """The same rows cut in two by column: the block over columns [c0, c1) -- the rank's own
        rows of the operand -- and the block over all other columns.  Both keep the padded column
        space, the rows' ``dinv`` and the uniform off-diagonal weight; the diagonal (and with it
        the diagonal correction of a weight-2 graph) belongs to the own-columns block."""
        a = self.adj
        dev = a.device
        counts = (a.rowptr[1:] - a.rowptr[:-1]).long()
        rows = torch.repeat_interleave(torch.arange(self.n_rows, device=dev), counts)
        inside = (a.colidx >= c0) & (a.colidx < c1)
        parts = []
        for keep, corr in ((inside, a.corr), (~inside, None)):
            cnt = torch.bincount(rows[keep], minlength=self.n_rows)
            rowptr = torch.zeros(self.n_rows + 1, dtype=torch.int32, device=dev)
            rowptr[1:] = torch.cumsum(cnt, 0).to(torch.int32)
            colidx = a.colidx[keep].contiguous()
            vals = None if a.unit_weights else a.vals[keep].contiguous()
            block = NormAdj(self.n_rows, rowptr, colidx, vals, a.deg, a.dinv, unit_weights=a.unit_weights,
                            w_off=a.w_off, corr=corr)
            parts.append(_LocalBlock(block, self.dinv, self.n_rows, self.n_cols, self.x_row_offset))
        return parts[0], parts[1]

    def tilegroup_plan(self, auto):
        """Tile-group plan of this block's pattern (kept); None when ``auto`` and its rows share too
        few columns, or when its weights differ."""
        a = self.adj
        if a.nnz == 0 or a.w_off is None:
            return None
        if self._tg is None or (self._tg is False and not auto):
            self._tg = tilegroup_build(self.n_rows, self.n_cols, a.nnz, a.rowptr, a.colidx, a.device, auto)
        return self._tg or None

    def spmm_into(self, full, out, final):
        """out[rows_local, F] = dinv . (block . full), or (dinv . s) . dinv when not final."""
        self._multiply(full, out, self.dinv, _lib.EPI_SCALE_ROW if final else _lib.EPI_SCALE_ROW2, True)

    def spmm_raw_into(self, full, out):
        """out[rows_local, F] = block . full with the block's weights and no row scaling: the raw
        sums of one column block, for ``finish_into``."""
        self._multiply(full, out, None, _lib.EPI_SCALE_ROW, False)

    def finish_into(self, s1, s2, out, final):
        """out = dinv . (s1 + s2), or that times dinv again when not final (sgnn_add_scale_f32)."""
        _lib.check(_lib.load().sgnn_add_scale_f32(self.n_rows, s1.shape[1], _lib.ptr(s1), s1.stride(0), _lib.ptr(s2),
                                                  s2.stride(0), _lib.ptr(self.dinv), 0 if final else 1, _lib.ptr(out),
                                                  out.stride(0), _lib.current_stream(s1.device)))

    def _multiply(self, full, out, dinv_row, epi, allow_tilegroup):
        lib = _lib.load()
        a = self.adj
        F = full.shape[1]
        dev = full.device
        tg = self._tilegroup_plan(F) if allow_tilegroup else None
        if tg is not None:
            # the gathered rows arrive scaled (utils.py:93 was applied before the exchange): split as they are
            nbytes = lib.sgnn_spmm_workspace_bytes(self.n_rows, a.nnz, F)
            ws = self._ws.get(("csr", F))
            if ws is None or ws.numel() < nbytes:
                ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
                self._ws[("csr", F)] = ws
            tilegroup_hop(tg, self.n_rows, self.n_cols, a.nnz, a.rowptr, a.colidx, a._vals, F, full, full.stride(0),
                          out, out.stride(0), dinv_row, None, a.w_off, None, epi, ws, dev)
            return
        aligned = full.data_ptr() % 16 == 0 and full.stride(0) % 4 == 0      # 16-byte row loads
        plan = self._plan(F) if aligned and self.use_rowgroup else None
        if plan is not None:
            nbytes = lib.sgnn_spmm_rowgroup_workspace_bytes(plan["slots"], plan["R"], F)
            ws = self._ws.get(("rg", F))
            if ws is None or ws.numel() < nbytes:
                ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
                self._ws[("rg", F)] = ws
            _lib.check(lib.sgnn_spmm_rowgroup_f32(
                self.n_rows, self.n_cols, a.nnz, F, plan["R"], _lib.ptr(plan["buf"]), plan["bytes"],
                _lib.ptr(plan["gw"]), plan["items"], plan["slots"], plan["n_long"], _lib.ptr(dinv_row), None,
                float(a.w_off if a.w_off is not None else 1.0), _lib.ptr(a.corr), self.x_row_offset,
                _lib.ptr(full), full.stride(0), _lib.ptr(out), out.stride(0), epi, _lib.ptr(ws), ws.numel(),
                _lib.current_stream(dev)))
            return
        nbytes = lib.sgnn_spmm_workspace_bytes(self.n_rows, a.nnz, F)
        ws = self._ws.get(("csr", F))
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
            self._ws[("csr", F)] = ws
        _lib.check(lib.sgnn_spmm_csr_f32(
            self.n_rows, self.n_cols, a.nnz, F, _lib.ptr(a.rowptr), _lib.ptr(a.colidx),
            None if a.unit_weights else _lib.ptr(a.vals), _lib.ptr(dinv_row), None,
            _lib.ptr(full), full.stride(0), _lib.ptr(out), out.stride(0), epi, _lib.ptr(ws), ws.numel(),
            _lib.current_stream(dev)))


class FeatureSplitAdj:
    """The other way to use P GPUs on ``A_hat^K . X``: the feature columns are independent, so
    every rank holds the WHOLE normalised adjacency and propagates ``F / P`` of the columns, with
    no exchange between hops at all.  A caller who owns rows of X (as with ``PartitionedAdj``:
    ``rows`` lists them, contiguous blocks of ``ceil(n / P)``) gets there and back by two
    all-to-alls per call -- each rank sends ``(P - 1) / P`` of its own block once in each
    direction, ``2 . n . F . 4 / P`` bytes in all, where the row partition's all-gather receives
    ``(P - 1) / P . n . F . 4`` per HOP (Reddit shape, 8 ranks: 123 MB per call against 491 MB per
    hop), and an all-to-all is the pattern xGMI's point-to-point links serve best.  The price
    is memory (every rank stores the graph and its plan: 1.3 GB on the Reddit shape) and narrower
    multiplies (the kernels' per-stage costs do not shrink with the width: one rank's hop at 76
    of 602 features takes 1.0 ms, not 5.3 / 8 = 0.66).  It replaces the reference's replicas
    (main.py:39-41, model/SGNN.py:518-520), which compute everything everywhere.

    Which of the two schedules is faster depends on the graph: wide features on a graph whose
    rows share many columns (Reddit) favour this one, narrow features on a sparse graph (R-MAT
    10 M at F = 128: 16 features per rank) the row partition.  ``bench.py --gpus N`` times both."""

    def __init__(self, adj, n, rank, world, group=None):
        self.adj, self.n, self.rank, self.world, self.group = adj, int(n), int(rank), int(world), group
        self.max_rows = (self.n + self.world - 1) // self.world
        lo = min(self.rank * self.max_rows, self.n)
        hi = min(lo + self.max_rows, self.n)
        self.rows = torch.arange(lo, hi, dtype=torch.int64)
        self.n_local = hi - lo
        self.global_nnz = adj.nnz if hasattr(adj, "nnz") else None
        self.last_schedule = None
        self._bufs = {}

    @classmethod
    def from_coo(cls, row, col, val, n, device, rank, world, group=None, adj_factory=None):
        """Every rank passes the same COO.  ``adj_factory(row, col, val, n)`` is a test hook that
        builds the object whose ``propagate(X, order)`` is the one-GPU product (default: the
        device adjacency through the C ABI)."""
        if adj_factory is not None:
            adj = adj_factory(row, col, val, n)
        else:
            adj = NormAdj.from_coo(torch.as_tensor(row).to(device), torch.as_tensor(col).to(device),
                                   None if val is None else torch.as_tensor(val).to(device), n, device)
        return cls(adj, n, rank, world, group)

    def feature_slice(self, F):
        """``(width of a slice, first feature of this rank's slice, its width)``: equal slices of
        ``ceil(F / P)`` features rounded up to a multiple of 4 (16-byte rows); the last ranks'
        slices may be short or empty."""
        Fw = ((F + self.world - 1) // self.world + 3) // 4 * 4
        f0 = min(self.rank * Fw, F)
        return Fw, f0, min(Fw, F - f0)

    def _all_to_all(self, recv, send):
        if self.world == 1:
            recv.copy_(send)
            return
        if send.is_cuda and dist.get_backend(self.group) == "gloo":          # tests: several ranks on one GPU
            host = torch.empty(recv.shape, dtype=recv.dtype)
            dist.all_to_all_single(host.view(-1), send.cpu().view(-1), group=self.group)
            recv.copy_(host)
            return
        dist.all_to_all_single(recv.view(-1), send.view(-1), group=self.group)

    def propagate(self, X_local, order=1):
        """``A_hat^order . X`` for this rank's rows (``X_local`` = rows ``self.rows`` of X)."""
        order = int(order)
        if order <= 0:
            return X_local
        if X_local.shape[0] != self.n_local:
            raise ValueError(f"rank {self.rank} owns {self.n_local} rows, got {X_local.shape[0]}")
        P, m, F = self.world, self.max_rows, X_local.shape[1]
        Fw, f0, wq = self.feature_slice(F)
        dev, dt = X_local.device, X_local.dtype
        key = (F, dev, dt)
        b = self._bufs.get(key)
        if b is None:
            # [P, m, Fw] blocks: block q of `send` goes to rank q; padding rows / features stay zero
            b = _keep(self._bufs, key, [torch.zeros((P, m, Fw), dtype=dt, device=dev) for _ in range(4)])
        send, recv, back, out_blocks = b
        nl = self.n_local
        for q in range(P):                       # my rows, cut by feature slice
            w = min(Fw, max(F - q * Fw, 0))
            if w > 0 and nl > 0:
                send[q, :nl, :w].copy_(X_local[:, q * Fw:q * Fw + w])
        self._all_to_all(recv, send)             # recv[p] = rank p's rows x my features: all rows, in order
        XF = recv.view(P * m, Fw)
        if wq > 0:
            YF = self.adj.propagate(XF[:self.n, :wq], order)
            back.view(P * m, Fw)[:self.n, :wq].copy_(YF)
        self._all_to_all(out_blocks, back)       # out_blocks[q] = my rows x slice q
        out = torch.empty((nl, F), dtype=dt, device=dev)
        for q in range(P):
            w = min(Fw, max(F - q * Fw, 0))
            if w > 0 and nl > 0:
                out[:, q * Fw:q * Fw + w].copy_(out_blocks[q, :nl, :w])
        self.last_schedule = "feature-split"
        return out
