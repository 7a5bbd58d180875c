"""``NormAdj`` — the device-resident form of the adjacency the propagation uses.

The reference keeps the adjacency as a torch sparse COO tensor (int64 indices,
fp32 values; utils.py:54-62) and, on EVERY call of
``process_data_with_adjacency_high_order``, rebuilds ``A + I``, the row degree
and ``1/sqrt(degree)`` from it (utils.py:82-89).  Here all of that is done once,
on the GPU, into an int32 CSR plus ``deg``/``dinv`` vectors that stay in HBM;
each later call is SpMM launches only.

Layout in HBM (n nodes, nnz stored entries of A+I):
  rowptr int32[n+1] | colidx int32[nnz] | vals fp32[nnz] | deg fp32[n] | dinv fp32[n]
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib


def _keep(cache, key, value, limit=4):
    """Put ``value`` under ``key`` in a small insertion-ordered cache, dropping the oldest
    entries beyond ``limit``.  The per-width buffers (planes, partial rows, workspaces) are
    kept this way: a stack of layers alternates between a few widths on one adjacency
    (Reddit: 602 and 128), and a single-entry cache would reallocate and clear half a
    gigabyte on every call."""
    cache.pop(key, None)
    cache[key] = value
    while len(cache) > limit:
        cache.pop(next(iter(cache)))
    return value


def _as_index_tensor(a, device):
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=torch.int64).contiguous()
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.int64))).to(device)


def weight_structure(n, rowptr, colidx, vals):
    """Classify the stored values of ``A (+ I)``.

    Returns ``(unit, w_off, corr)``: ``unit`` — every value is 1; ``w_off`` — the
    common value of all off-diagonal entries, or None if they differ (Reddit's
    ``A + A^T`` is 2.0 everywhere, data_loader.py:75); ``corr`` — per row,
    ``diagonal value - w_off`` where the row stores a diagonal entry, else 0
    (None when that is zero everywhere).  With these the weighted row sum is
    ``w_off * (pattern sum) + corr_i * x_i``, which is what lets a pattern-only
    kernel serve such graphs.
    """
    total = int(colidx.numel())
    if total == 0 or bool((vals == 1.0).all().item()):
        return True, 1.0, None
    counts = (rowptr[1:] - rowptr[:-1]).long()
    rows = torch.repeat_interleave(torch.arange(n, device=vals.device), counts)
    diag = rows == colidx.long()
    off = vals[~diag]
    if off.numel() == 0:
        w = 1.0
    else:
        w = float(off[0].item())
        if not bool((off == w).all().item()):
            return False, None, None
    corr = torch.zeros(n, dtype=torch.float32, device=vals.device)
    corr[rows[diag]] = vals[diag] - w
    if not bool((corr != 0).any().item()):
        corr = None
    return False, w, corr


# When the tile-group kernel is chosen automatically (tools/policy_probe.py, one hop against
# the row-group / plain kernels on R-MAT graphs): it needs enough stages of 128 merged columns
# to keep 256 CUs busy -- with 1.7 M and 2.5 M stored entries it is 1.36 and 1.08 times slower,
# with 6.5 M and more 0.75 to 0.95 times as long -- and then pays from little sharing on (stored
# entries per fetched row, on average over the groups of 512 rows): R-MAT with 10^7 nodes, at
# 1.6, takes 15.5 ms per hop on it against 19.2 ms, the ogbn-products shape, at 2.35, 5.05
# against 6.05 ms, the Reddit shape, at 9.5, 5.2 against 10.1 ms.
TILEGROUP_MIN_SHARING = 1.5
TILEGROUP_MIN_NNZ = 4_000_000


def tilegroup_build(n_rows, n_cols, nnz, rowptr, colidx, device, auto=True):
    """Tile-group plan of an ``n_rows x n_cols`` CSR pattern (``sgnn_tilegroup_analyze`` +
    ``_fill``), or ``False`` when ``auto`` and its rows share too few columns to pay for it."""
    lib = _lib.load()
    with torch.cuda.device(device):
        st = _lib.current_stream(device)
        wbytes = lib.sgnn_tilegroup_build_workspace_bytes(n_rows, nnz)
        ws = torch.empty(wbytes, dtype=torch.uint8, device=device)
        _lib.check(lib.sgnn_tilegroup_analyze(n_rows, n_cols, nnz, _lib.ptr(rowptr), _lib.ptr(colidx), _lib.ptr(ws),
                                              wbytes, st))
        merged, stages, steps, items, slots, n_long = (int(v) for v in ws[:48].view(torch.int64).cpu())
        if auto and merged * TILEGROUP_MIN_SHARING > nnz:
            return False
        pbytes = lib.sgnn_tilegroup_plan_bytes(n_rows, stages, steps, items, n_long)
        buf = torch.empty(pbytes, dtype=torch.uint8, device=device)
        _lib.check(lib.sgnn_tilegroup_fill(n_rows, n_cols, nnz, stages, steps, items, n_long, _lib.ptr(buf), pbytes,
                                           _lib.ptr(ws), wbytes, st))
        torch.cuda.current_stream(device).synchronize()      # the workspace goes away below
        del ws
    return dict(buf=buf, bytes=pbytes, merged=merged, stages=stages, steps=steps, items=items, slots=slots,
                n_long=n_long, bufs={})


def tilegroup_buffers(plan, n_cols, F, device, hops=1):
    """Per-width buffers of the tile-group kernel, kept in ``plan["bufs"]``: two sets of operand
    planes (the second only once a chain needs it), the partial rows of split groups, and one
    block of state words per hop (flag, counters, per-feature scales)."""
    lib = _lib.load()
    bufs = plan["bufs"].get(F)
    if bufs is None:
        pl_bytes = lib.sgnn_tilegroup_planes_bytes(n_cols, F)
        ws_bytes = lib.sgnn_spmm_tilegroup_workspace_bytes(plan["slots"], F)
        sb = lib.sgnn_tilegroup_state_bytes(F)
        bufs = dict(planes=torch.empty(pl_bytes, dtype=torch.uint8, device=device), planes_b=None, pl_bytes=pl_bytes,
                    ws=torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=device),
                    state_bytes=sb, states=torch.zeros((1, sb // 4), dtype=torch.int32, device=device))
        bufs["state"] = bufs["states"][0]
        _keep(plan["bufs"], F, bufs)
    if hops > 1 and bufs["planes_b"] is None:
        bufs["planes_b"] = torch.empty(bufs["pl_bytes"], dtype=torch.uint8, device=device)
    if bufs["states"].shape[0] < hops:
        bufs["states"] = torch.zeros((hops, bufs["state_bytes"] // 4), dtype=torch.int32, device=device)
    # the flag and the counters the tests and tools look at: those of the operand of the last hop run
    bufs["state"] = bufs["states"][hops - 1]
    return bufs


def tilegroup_hop(plan, n_rows, n_cols, nnz, rowptr, colidx, vals, F, X, ldx, Y, ldy, dinv_row, split_scale, w_off, corr,
                  epilogue, csr_ws, device):
    """One hop through the tile-group kernel: split X (times ``split_scale`` per row if given) into
    fp16 planes, multiply; the plain CSR kernel stands behind it for operands the planes cannot
    represent.  The per-width buffers live in ``plan["bufs"]``."""
    lib = _lib.load()
    bufs = tilegroup_buffers(plan, n_cols, F, device)
    st = _lib.current_stream(device)
    state = bufs["states"][0]
    bufs["state"] = state
    _lib.check(lib.sgnn_tilegroup_split_f32(n_cols, F, _lib.ptr(split_scale), _lib.ptr(X), ldx,
                                            _lib.ptr(bufs["planes"]), bufs["pl_bytes"], _lib.ptr(state), st))
    _lib.check(lib.sgnn_spmm_tilegroup_f32(
        n_rows, n_cols, F, _lib.ptr(plan["buf"]), plan["bytes"], plan["stages"], plan["steps"], plan["items"],
        plan["slots"], plan["n_long"], _lib.ptr(dinv_row), float(w_off), _lib.ptr(corr),
        _lib.ptr(bufs["planes"]), bufs["pl_bytes"], _lib.ptr(state), _lib.ptr(Y), ldy, int(epilogue),
        _lib.ptr(bufs["ws"]), bufs["ws"].numel(), nnz, _lib.ptr(rowptr), _lib.ptr(colidx),
        _lib.ptr(vals), _lib.ptr(split_scale), _lib.ptr(X), ldx, _lib.ptr(csr_ws), csr_ws.numel(),
        0, None, 0, None, 0, 0, None, 0, st))
    return bufs


def tilegroup_chain(plan, n, nnz, rowptr, colidx, vals, F, X, ldx, Y, ldy, dinv, w_off, corr, order, csr_ws, device):
    """``A_hat^order . X`` through the tile-group kernel (``sgnn_tilegroup_chain_f32``): X is split
    once, every hop but the last writes the next hop's operand planes from its epilogue."""
    lib = _lib.load()
    bufs = tilegroup_buffers(plan, n, F, device, hops=order)
    _lib.check(lib.sgnn_tilegroup_chain_f32(
        order, n, F, _lib.ptr(plan["buf"]), plan["bytes"], plan["stages"], plan["steps"], plan["items"], plan["slots"],
        plan["n_long"], _lib.ptr(dinv), float(w_off), _lib.ptr(corr), _lib.ptr(X), ldx, _lib.ptr(Y), ldy,
        _lib.ptr(bufs["planes"]), _lib.ptr(bufs["planes_b"]), bufs["pl_bytes"], _lib.ptr(bufs["states"]),
        bufs["states"].numel() * 4, _lib.ptr(bufs["ws"]), bufs["ws"].numel(), nnz, _lib.ptr(rowptr), _lib.ptr(colidx),
        _lib.ptr(vals), _lib.ptr(csr_ws), csr_ws.numel(), _lib.current_stream(device)))
    return bufs


def _require_gpu(device):
    device = torch.device(device)
    if device.type != "cuda":
        raise _lib.SgnnHipError(
            f"sgnn_amd runs on MI355X only (device={device}); there is no CPU path in this package")
    if not torch.cuda.is_available():
        raise _lib.SgnnHipError("no GPU visible to PyTorch-ROCm")
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return device


class NormAdj:
    """CSR of ``A (+ I)`` with degree vectors, resident on one GPU."""

    def __init__(self, n, rowptr, colidx, vals, deg, dinv, unit_weights=False, symmetric=None,
                 w_off=None, corr=None):
        self.n = int(n)
        self.rowptr, self.colidx = rowptr, colidx
        # all-ones values are not kept: no kernel reads them (NULL = 1.0 throughout the C ABI)
        # and on the Reddit shape they would hold 0.46 GB; `vals` rebuilds them on request
        self._vals = None if unit_weights else vals
        self.deg, self.dinv = deg, dinv
        self.nnz = int(colidx.numel())
        self.unit_weights = bool(unit_weights)
        # uniform off-diagonal weight (None: weights vary) and per-row diagonal
        # correction for the pattern-only row-group kernel, see weight_structure()
        self.w_off = 1.0 if unit_weights else w_off
        self.corr = None if unit_weights else corr
        self.device = rowptr.device
        self._symmetric = symmetric
        self._transposed = None
        self._ws = {}
        self._plans = {}                  # R -> row-group plan (built on first use)
        self.use_rowgroup = True          # uniform-weight graphs, supported widths: see _rowgroup_plan
        # tile-group (matrix-core) kernel: None = used where the graph's rows share enough
        # columns to pay for it (see _tilegroup_plan), True / False force it on / off
        self.use_tilegroup = None
        self._tg = None                   # its plan; False once it has been found not worthwhile
        self._tg_bufs = {}

    @property
    def vals(self):
        """Stored values, fp32[nnz] (for a unit-weight graph a fresh tensor of ones)."""
        if self._vals is None:
            return torch.ones(self.nnz, dtype=torch.float32, device=self.device)
        return self._vals

    # the reference's callers only ever look at .shape[0] of the adjacency handle
    # (model/SGNN.py:33)
    @property
    def shape(self):
        return torch.Size((self.n, self.n))

    def __repr__(self):
        return f"NormAdj(n={self.n}, nnz={self.nnz}, device={self.device}, unit_weights={self.unit_weights})"

    # ------------------------------------------------------------------ build
    @classmethod
    def from_coo(cls, row, col, val, n, device, strip_diag=False, self_loops=_lib.LOOPS_ADD,
                 drop_zeros=False, zero_inf=False, symmetrize=False, clamp_one=False):
        """Build from COO triples (duplicates summed; utils.py:54-62 + 82-89).

        ``symmetrize`` and ``clamp_one`` do the loaders' ``adj = adj.T + adj`` and
        ``adj = adj.minimum(1)`` (data_loader.py:451-452, 476-477, 490-491, 507-508)
        on the GPU as part of the same sort-and-merge, so a raw directed edge list
        goes in and the symmetric 0/1 adjacency is what gets normalised."""
        device = _require_gpu(device)
        lib = _lib.load()
        n = int(n)
        if n <= 0:
            raise ValueError("adjacency must have at least one node")
        with torch.cuda.device(device):
            row = _as_index_tensor(row, device)
            col = _as_index_tensor(col, device)
            if row.shape != col.shape or row.dim() != 1:
                raise ValueError("row and col must be 1-D and of equal length")
            nnz_in = int(row.numel())
            if val is not None:
                if not isinstance(val, torch.Tensor):
                    val = torch.from_numpy(np.ascontiguousarray(np.asarray(val, dtype=np.float32)))
                val = val.to(device=device, dtype=torch.float32).contiguous()
                if val.numel() != nnz_in:
                    raise ValueError("val must have one entry per (row, col) pair")
            cap = nnz_in * (2 if symmetrize else 1) + n
            rowptr = torch.empty(n + 1, dtype=torch.int32, device=device)
            colidx = torch.empty(cap, dtype=torch.int32, device=device)
            vals = torch.empty(cap, dtype=torch.float32, device=device)
            deg = torch.empty(n, dtype=torch.float32, device=device)
            dinv = torch.empty(n, dtype=torch.float32, device=device)
            nnz_out = torch.zeros(1, dtype=torch.int64, device=device)
            ws_bytes = lib.sgnn_build_workspace_bytes(n, nnz_in * (2 if symmetrize else 1))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
            flags = ((_lib.STRIP_DIAG if strip_diag else 0) | (_lib.DROP_ZEROS if drop_zeros else 0)
                     | (_lib.SYMMETRIZE if symmetrize else 0) | (_lib.CLAMP_ONE if clamp_one else 0))
            _lib.check(lib.sgnn_build_norm_csr(
                n, nnz_in, _lib.ptr(row), _lib.ptr(col), _lib.ptr(val), flags, int(self_loops),
                1 if zero_inf else 0, _lib.ptr(rowptr), _lib.ptr(colidx), _lib.ptr(vals), _lib.ptr(deg),
                _lib.ptr(dinv), _lib.ptr(nnz_out), _lib.ptr(ws), ws_bytes, _lib.current_stream(device)))
            total = int(nnz_out.item())          # synchronises the stream
            del ws
            if total < 0:
                raise ValueError(f"adjacency indices out of range for a {n} x {n} matrix")
            # the output buffers were sized for the worst case (no duplicates); copy the used
            # part out only when that frees a worthwhile amount of memory
            compact = (cap - total) * 16 > cap
            colidx = colidx[:total].clone() if compact else colidx[:total]
            vals = vals[:total]
            unit, w_off, corr = weight_structure(n, rowptr, colidx, vals)
            vals = None if unit else (vals.clone() if compact else vals)
        return cls(n, rowptr, colidx, vals, deg, dinv, unit_weights=unit, w_off=w_off, corr=corr)

    @classmethod
    def from_scipy(cls, mat, device, **kw):
        """scipy sparse (any format) -> NormAdj; values cast to fp32 as utils.py:61 does."""
        coo = mat.tocoo()
        if coo.shape[0] != coo.shape[1]:
            raise ValueError("adjacency must be square")
        return cls.from_coo(coo.row, coo.col, np.asarray(coo.data, dtype=np.float32), coo.shape[0], device, **kw)

    @classmethod
    def from_torch_sparse(cls, t, device=None, **kw):
        """torch sparse COO (as utils.py:59-62 builds it; need not be coalesced)."""
        if t.layout != torch.sparse_coo:
            raise ValueError("expected a torch sparse COO tensor")
        if t.shape[0] != t.shape[1]:
            raise ValueError("adjacency must be square")
        idx = t._indices()
        return cls.from_coo(idx[0], idx[1], t._values().to(torch.float32), t.shape[0],
                            device if device is not None else t.device, **kw)

    @classmethod
    def from_edge_index(cls, edge_index, num_nodes, device=None, edge_weight=None, add_self_loops=True):
        """PyG convention (model/SGC.py:14, model/GCN.py:14): messages flow
        ``edge_index[0] -> edge_index[1]`` and the degree is the weighted in-degree
        of the target.  As a matrix that is ``M[target, source]``, so targets are
        the CSR rows here and the SGNN row-degree formula applies unchanged.
        """
        dev = device if device is not None else edge_index.device
        return cls.from_coo(edge_index[1], edge_index[0], edge_weight, num_nodes, dev,
                            self_loops=_lib.LOOPS_REMAINING if add_self_loops else _lib.LOOPS_NONE,
                            zero_inf=True)

    # --------------------------------------------------------------- plumbing
    def _workspace(self, F):
        ws = self._ws.get(F)
        if ws is None:
            nbytes = _lib.load().sgnn_spmm_workspace_bytes(self.n, self.nnz, F)
            ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=self.device)
            _keep(self._ws, F, ws)        # callers alternate between few widths
        return ws

    def _rowgroup_plan(self, F):
        """Row-group plan for feature width F, or None when the plain CSR kernel
        must be used (unsupported width, switched off)."""
        if not self.use_rowgroup or self.nnz == 0:
            return None
        lib = _lib.load()
        R = lib.sgnn_rowgroup_config_for(self.n, self.nnz, F)
        if R == 0:
            return None
        plan = self._plans.get(R)
        if plan is None:
            with torch.cuda.device(self.device):
                pbytes = lib.sgnn_rowgroup_plan_bytes(self.n, self.nnz, R)
                buf = torch.empty(pbytes, dtype=torch.uint8, device=self.device)
                wbytes = lib.sgnn_rowgroup_workspace_bytes(self.n, self.nnz, R)
                ws = torch.empty(wbytes, dtype=torch.uint8, device=self.device)
                _lib.check(lib.sgnn_rowgroup_build(
                    self.n, self.n, self.nnz, _lib.ptr(self.rowptr), _lib.ptr(self.colidx), R,
                    _lib.ptr(buf), pbytes, _lib.ptr(ws), wbytes, _lib.current_stream(self.device)))
                counts = buf[:32].view(torch.int64).cpu()       # synchronises
                del ws
            merged, items, slots, n_long = (int(v) for v in counts)
            gw = None
            if self.w_off is None:
                # stored values differ: the plan also carries the value of every
                # (merged column, row of the group) pair
                gbytes = lib.sgnn_rowgroup_weights_bytes(merged, R)
                gw = torch.empty(gbytes, dtype=torch.uint8, device=self.device)
                _lib.check(lib.sgnn_rowgroup_weights(
                    self.n, self.nnz, _lib.ptr(self.rowptr), _lib.ptr(self.colidx), _lib.ptr(self.vals), R,
                    _lib.ptr(buf), pbytes, merged, _lib.ptr(gw), gbytes, _lib.current_stream(self.device)))
            plan = dict(R=R, buf=buf, bytes=pbytes, merged=merged, items=items, slots=slots, n_long=n_long, ws={},
                        gw=gw)
            self._plans[R] = plan
        return plan

    TILEGROUP_MIN_SHARING = TILEGROUP_MIN_SHARING

    def _tilegroup_plan(self, F):
        """Tile-group plan, or None when that kernel does not apply (off-diagonal weights that
        differ, fewer than 16 features, rows that share too few columns, switched off)."""
        if self.use_tilegroup is False or not self.use_rowgroup or self.nnz == 0:
            return None
        if self.w_off is None:
            return None
        auto = self.use_tilegroup is None
        if auto and (F < 16 or self.nnz < TILEGROUP_MIN_NNZ):
            return None
        if self._tg is None:
            self._tg = tilegroup_build(self.n, self.n, self.nnz, self.rowptr, self.colidx, self.device, auto)
        return self._tg or None

    def _tilegroup_hop(self, plan, X, ldx, Y, ldy, F, scale_rows, scale_cols, epilogue):
        """One hop through the tile-group kernel (``tilegroup_hop``) on the whole matrix."""
        bufs = tilegroup_hop(plan, self.n, self.n, self.nnz, self.rowptr, self.colidx, self._vals, F, X, ldx, Y, ldy,
                             self.dinv if scale_rows else None, self.dinv if scale_cols else None, self.w_off,
                             self.corr, epilogue, self._workspace(F), self.device)
        _keep(self._tg_bufs, F, bufs)

    def _rowgroup_ws(self, plan, F):
        ws = plan["ws"].get(F)
        if ws is None:
            nbytes = _lib.load().sgnn_spmm_rowgroup_workspace_bytes(plan["slots"], plan["R"], F)
            ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=self.device)
            _keep(plan["ws"], F, ws)
        return ws

    def _folds_col_scale(self, plan, ldx):
        """True when the row-group kernel applies the column scaling itself for this plan and
        an operand of row stride ``ldx``: a small graph (too few work items to fill the chip,
        planned with groups of 8) on a pattern plan.  It saves a launch and a pass over X;
        on large graphs the gather and multiply per fetched row cost more than that pass
        (products shape: main kernel 5.96 ms against 5.33 ms)."""
        small = self.n // 16 + self.nnz // 64 < 8192      # the rule of sgnn_rowgroup_config_for
        return (plan is not None and small and plan["R"] == 8 and plan["gw"] is None
                and self.n * ldx * 4 < 2 ** 32 - 8192)

    def _hop(self, X, ldx, Y, ldy, F, scale_rows, scale_cols, epilogue):
        """Enqueue one SpMM launch sequence: the row-group kernel when a plan
        exists for this width, the plain CSR kernel otherwise."""
        lib = _lib.load()
        dinv = _lib.ptr(self.dinv)
        # the row-group kernel reads rows of X with 16-byte loads and takes X already scaled
        # (plans of small graphs excepted)
        aligned = X.data_ptr() % 16 == 0 and ldx % 4 == 0
        plan = self._rowgroup_plan(F) if aligned else None
        if scale_cols and not self._folds_col_scale(plan, ldx):
            plan = None
        if plan is not None:
            ws = self._rowgroup_ws(plan, F)
            _lib.check(lib.sgnn_spmm_rowgroup_f32(
                self.n, self.n, self.nnz, F, plan["R"], _lib.ptr(plan["buf"]), plan["bytes"], _lib.ptr(plan["gw"]),
                plan["items"], plan["slots"], plan["n_long"],
                dinv if scale_rows else None, dinv if scale_cols else None,
                float(self.w_off if self.w_off is not None else 1.0), _lib.ptr(self.corr), 0,
                _lib.ptr(X), ldx, _lib.ptr(Y), ldy, int(epilogue), _lib.ptr(ws), ws.numel(),
                _lib.current_stream(self.device)))
            return
        ws = self._workspace(F)
        _lib.check(lib.sgnn_spmm_csr_f32(
            self.n, self.n, self.nnz, F, _lib.ptr(self.rowptr), _lib.ptr(self.colidx),
            None if self.unit_weights else _lib.ptr(self.vals),
            dinv if scale_rows else None, dinv if scale_cols else None,
            _lib.ptr(X), ldx, _lib.ptr(Y), ldy, int(epilogue), _lib.ptr(ws), ws.numel(),
            _lib.current_stream(self.device)))

    def _check_dense(self, X, rows):
        if not isinstance(X, torch.Tensor) or X.dim() != 2:
            raise ValueError("X must be a 2-D tensor")
        if X.device != self.device:
            raise ValueError(f"X is on {X.device}, adjacency on {self.device}")
        if X.dtype != torch.float32:
            raise ValueError("X must be float32")
        if X.shape[0] != rows:
            raise ValueError(f"X has {X.shape[0]} rows, adjacency expects {rows}")
        if X.stride(1) != 1 or (X.shape[0] > 1 and X.stride(0) < X.shape[1]):
            X = X.contiguous()
        return X

    # ------------------------------------------------------------------ apply
    def spmm(self, X, scale_rows=True, scale_cols=True, epilogue=_lib.EPI_SCALE_ROW, out=None):
        """One hop: ``diag(dinv) . (A+I) . diag(dinv) . X`` (utils.py:93-95)."""
        X = self._check_dense(X, self.n)
        F = X.shape[1]
        with torch.cuda.device(self.device):
            Y = torch.empty((self.n, F), dtype=torch.float32, device=self.device) if out is None else out
            if F == 0:
                return Y
            ldx = X.stride(0) if X.shape[0] > 1 else F
            tg = self._tilegroup_plan(F)
            if tg is not None:
                self._tilegroup_hop(tg, X, ldx, Y, Y.stride(0) if self.n > 1 else F, F, scale_rows, scale_cols, epilogue)
                return Y
            plan = self._rowgroup_plan(F)
            aligned = X.data_ptr() % 16 == 0 and ldx % 4 == 0
            if plan is not None and not (aligned and (not scale_cols or self._folds_col_scale(plan, ldx))):
                # stage X for the row-group kernel: column scaling applied (utils.py:93),
                # rows on 16-byte boundaries
                X = self._staged(X, ldx, F, scale_cols)
                ldx, scale_cols = X.stride(0), False
            self._hop(X, ldx, Y, Y.stride(0) if self.n > 1 else F, F, scale_rows, scale_cols, epilogue)
        return Y

    def _staged(self, X, ldx, F, scale):
        """Copy of X (times dinv per row if ``scale``) whose rows start on 16-byte boundaries."""
        Fp = (F + 3) // 4 * 4
        Xs = torch.empty((self.n, Fp), dtype=torch.float32, device=self.device)
        if scale:
            _lib.check(_lib.load().sgnn_row_scale_f32(self.n, F, _lib.ptr(self.dinv), _lib.ptr(X), ldx,
                                                      _lib.ptr(Xs), Fp, _lib.current_stream(self.device)))
        else:
            Xs[:, :F].copy_(X)
        return Xs

    def propagate(self, X, order=1):
        """``A_hat^order . X`` — the whole loop of utils.py:91-96 in ``order`` SpMM
        launches.  Between hops the two row scalings of the reference (utils.py:95
        of one hop, utils.py:93 of the next) are folded into the epilogue, so no
        separate pass touches the intermediates."""
        order = int(order)
        if order <= 0:
            return X                       # range(order) is empty in the reference: X comes back as is
        X = self._check_dense(X, self.n)
        F = X.shape[1]
        with torch.cuda.device(self.device):
            Y = torch.empty((self.n, F), dtype=torch.float32, device=self.device)
            if F == 0:
                return Y
            tg = self._tilegroup_plan(F)
            if tg is not None:
                # tile-group kernel: X is split into fp16 planes once (the split applies
                # utils.py:93); every hop but the last writes the next hop's planes from its
                # epilogue, so no intermediate exists as fp32 or is measured and split again
                bufs = tilegroup_chain(tg, self.n, self.nnz, self.rowptr, self.colidx, self._vals, F, X,
                                       X.stride(0) if X.shape[0] > 1 else F, Y, F, self.dinv, self.w_off, self.corr,
                                       order, self._workspace(F), self.device)
                _keep(self._tg_bufs, F, bufs)
                return Y
            if self._rowgroup_plan(F) is None:
                # plain CSR kernel: the library's own K-hop chain
                lib = _lib.load()
                tmp = torch.empty((self.n, F), dtype=torch.float32, device=self.device) if order > 1 else None
                ws = self._workspace(F)
                _lib.check(lib.sgnn_spmm_chain_f32(
                    order, self.n, self.nnz, F, _lib.ptr(self.rowptr), _lib.ptr(self.colidx),
                    None if self.unit_weights else _lib.ptr(self.vals), _lib.ptr(self.dinv),
                    _lib.ptr(X), X.stride(0) if X.shape[0] > 1 else F, _lib.ptr(Y), F, _lib.ptr(tmp),
                    _lib.ptr(ws), ws.numel(), _lib.current_stream(self.device)))
                return Y
            # Row-group kernel: scale X by dinv once up front (utils.py:93) instead of
            # gathering dinv[col] per fetched row inside the kernel — measured 6 % faster
            # on the Reddit shape, the extra pass over X included.
            # The kernel reads rows with 16-byte loads, so the scaled copy and the
            # intermediates of a chain live in buffers whose row stride is a multiple of
            # 4 floats; only the last hop writes the caller-visible Y.
            # A small graph's plan (groups of 8) scales the fetched rows inside the kernel on the
            # first hop instead: one launch and one pass over X less, and X itself is read.
            ldx = X.stride(0) if X.shape[0] > 1 else F
            fold = (self._folds_col_scale(self._rowgroup_plan(F), ldx) and F % 4 == 0 and ldx % 4 == 0
                    and X.data_ptr() % 16 == 0)
            src = X if fold else self._staged(X, ldx, F, True)
            Fp = ldx if fold else src.stride(0)
            width = F if fold else Fp
            spare = [torch.empty((self.n, width), dtype=torch.float32, device=self.device)] if order > 1 else []
            if order > 2:    # second intermediate: the staged copy once hop 0 has read it
                spare.append(torch.empty((self.n, width), dtype=torch.float32, device=self.device) if fold else src)
            for h in range(order):
                last = h == order - 1
                dst = Y if last else spare[h % 2]         # ping-pong: no hop reads what it writes
                self._hop(src, Fp, dst, dst.stride(0), F, True, fold and h == 0,
                          _lib.EPI_SCALE_ROW if last else _lib.EPI_SCALE_ROW2)
                src, Fp = dst, dst.stride(0)
        return Y

    # --------------------------------------------------------------- sampling
    def sample_block(self, idx, skip_self=True, out=None):
        """Dense ``A[idx, :][:, idx]`` as fp32 on the GPU (model/SGNN.py:167-170).

        The stored matrix carries the self-loops the build added; ``skip_self``
        writes 0 where ``idx[i] == idx[j]``, which is the block of the diagonal-free
        adjacency the reference samples from (model/SGNN.py:497-501)."""
        idx = idx.to(device=self.device, dtype=torch.int64).contiguous()
        if idx.dim() != 1:
            raise ValueError("idx must be 1-D")
        b = int(idx.numel())
        if out is None:
            out = torch.empty((b, b), dtype=torch.float32, device=self.device)
        elif out.shape != (b, b) or out.dtype != torch.float32 or out.device != self.device or out.stride(1) != 1:
            raise ValueError("out must be a [b, b] fp32 tensor with unit column stride on the adjacency's device")
        _lib.check(_lib.load().sgnn_sample_block_f32(
            self.n, self.nnz, _lib.ptr(self.rowptr), _lib.ptr(self.colidx), _lib.ptr(self._vals), b, _lib.ptr(idx),
            _lib.SAMPLE_SKIP_SELF if skip_self else 0, _lib.ptr(out), out.stride(0) if b > 1 else b,
            _lib.current_stream(self.device)))
        return out

    # -------------------------------------------------------------- transpose
    def transpose(self):
        """NormAdj of ``A_hat^T`` (same dinv on both sides; needed by autograd)."""
        if self._transposed is not None:
            return self._transposed
        lib = _lib.load()
        with torch.cuda.device(self.device):
            t_rowptr = torch.empty_like(self.rowptr)
            t_colidx = torch.empty_like(self.colidx)
            vals = self.vals
            t_vals = torch.empty_like(vals)
            nbytes = lib.sgnn_transpose_workspace_bytes(self.n, self.nnz)
            ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=self.device)
            _lib.check(lib.sgnn_csr_transpose(
                self.n, self.nnz, _lib.ptr(self.rowptr), _lib.ptr(self.colidx), _lib.ptr(vals),
                _lib.ptr(t_rowptr), _lib.ptr(t_colidx), _lib.ptr(t_vals), _lib.ptr(ws), ws.numel(),
                _lib.current_stream(self.device)))
            same = bool(torch.equal(t_rowptr, self.rowptr) and torch.equal(t_colidx, self.colidx)
                        and torch.equal(t_vals, vals))
            del ws, vals
        self._symmetric = same
        if same:
            self._transposed = self
        else:
            t = NormAdj(self.n, t_rowptr, t_colidx, t_vals, self.deg, self.dinv,
                        unit_weights=self.unit_weights, symmetric=False, w_off=self.w_off, corr=self.corr)
            t._transposed = self
            t.use_rowgroup = self.use_rowgroup
            self._transposed = t
        return self._transposed

    @property
    def symmetric(self):
        if self._symmetric is None:
            self.transpose()
        return self._symmetric

    # ------------------------------------------------------------------ export
    def to_scipy(self):
        """Host copy as a scipy CSR (tests and diagnostics)."""
        import scipy.sparse as sp
        return sp.csr_matrix((self.vals.cpu().numpy(), self.colidx.cpu().numpy(), self.rowptr.cpu().numpy()),
                             shape=(self.n, self.n))
