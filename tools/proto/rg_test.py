#!/usr/bin/env python3
"""Prototype driver: row-group SpMM (R rows share each fetched row of X)."""
import ctypes
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from sgnn_amd import NormAdj  # noqa: E402
from sgnn_amd.graphgen import SHAPES, rmat_undirected_device  # noqa: E402


class RgArgs(ctypes.Structure):
    _fields_ = [(k, ctypes.c_void_p) for k in
                ("item_group", "item_m0", "item_m1", "item_out", "grows", "gcol", "gmask", "dinv", "X", "Y", "partial")] + \
               [("items", ctypes.c_int32), ("F", ctypes.c_int32), ("row_bytes", ctypes.c_int32), ("x_bytes", ctypes.c_uint32)]


def build_groups(adj, R, seg, sort_by_degree=True):
    dev = adj.device
    n = adj.n
    deg = (adj.rowptr[1:] - adj.rowptr[:-1]).long()
    rows = torch.repeat_interleave(torch.arange(n, device=dev), deg)
    cols = adj.colidx.long()
    if sort_by_degree:
        order = torch.argsort(deg, descending=True, stable=True)
    else:
        order = torch.arange(n, device=dev)
    rank = torch.empty_like(order)
    rank[order] = torch.arange(n, device=dev)
    G = (n + R - 1) // R
    grows = torch.full((G * R,), -1, dtype=torch.int32, device=dev)
    grows[:n] = order.int()
    rk = rank[rows]
    key = (rk // R) * n + cols
    bit = (1 << (rk % R))
    key, perm = torch.sort(key)
    bit = bit[perm]
    ukey, inv = torch.unique_consecutive(key, return_inverse=True)
    mask = torch.zeros(ukey.numel(), dtype=torch.int64, device=dev)
    mask.index_add_(0, inv, bit)
    gid = ukey // n
    gcol = (ukey % n).int()
    gptr = torch.zeros(G + 1, dtype=torch.int64, device=dev)
    gptr[1:] = torch.cumsum(torch.bincount(gid, minlength=G), 0)
    glen = gptr[1:] - gptr[:-1]
    nseg = (glen + seg - 1) // seg
    nseg = torch.clamp(nseg, min=1)
    item_group = torch.repeat_interleave(torch.arange(G, device=dev), nseg)
    first = torch.cumsum(nseg, 0) - nseg
    k = torch.arange(item_group.numel(), device=dev) - first[item_group]
    m0 = gptr[item_group] + k * seg
    m1 = torch.minimum(m0 + seg, gptr[item_group + 1])
    multi = nseg[item_group] > 1
    slot = torch.cumsum(multi.long(), 0) - 1
    item_out = torch.where(multi, slot, torch.full_like(slot, -1))
    return dict(grows=grows, gcol=gcol, gmask=mask.int(), item_group=item_group.int(), m0=m0.int(), m1=m1.int(),
                item_out=item_out.int(), nslots=int(multi.sum()), nseg=nseg, first=first, G=G)


def column_aligned_items(g, n, R, target=768, wmin=1024, cap=1536):
    """Re-cut every group's merged list at column boundaries that are aligned
    across groups: group g uses blocks of width wmin * 2^j columns, j chosen so a
    block holds about `target` merged columns; blocks longer than `cap` are split
    by count."""
    dev = g["gcol"].device
    gcol = g["gcol"].long()
    G = g["G"]
    gptr = torch.zeros(G + 1, dtype=torch.long, device=dev)
    # recover gptr from the original (unsplit) items: m0 of first item / m1 of last item per group
    ig = g["item_group"].long()
    gptr[1:] = torch.zeros(G, dtype=torch.long, device=dev).scatter_reduce(0, ig, g["m1"].long(), reduce="amax")
    glen = gptr[1:] - gptr[:-1]
    width = torch.full((G,), wmin, dtype=torch.long, device=dev)
    for _ in range(12):
        width = torch.where(glen.float() * width.float() / n < target, width * 2, width)
    gid = torch.repeat_interleave(torch.arange(G, device=dev), glen)
    blk = gcol // width[gid]
    key = gid * (n + 1) + blk
    ukey, counts = torch.unique_consecutive(key, return_counts=True)
    start = torch.cumsum(counts, 0) - counts
    # split long blocks by count
    nsub = (counts + cap - 1) // cap
    sub_item = torch.repeat_interleave(torch.arange(ukey.numel(), device=dev), nsub)
    firstsub = torch.cumsum(nsub, 0) - nsub
    ksub = torch.arange(sub_item.numel(), device=dev) - firstsub[sub_item]
    per = (counts[sub_item] + nsub[sub_item] - 1) // nsub[sub_item]
    m0 = start[sub_item] + ksub * per
    m1 = torch.minimum(m0 + per, start[sub_item] + counts[sub_item])
    item_group = (ukey // (n + 1))[sub_item]
    nper_group = torch.bincount(item_group, minlength=G)
    multi = nper_group[item_group] > 1
    slot = torch.cumsum(multi.long(), 0) - 1
    item_out = torch.where(multi, slot, torch.full_like(slot, -1))
    out = dict(g)
    out.update(item_group=item_group.int(), m0=m0.int(), m1=m1.int(), item_out=item_out.int(), nslots=int(multi.sum()))
    return out


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "reddit"
    maxdeg = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) > 0 else None
    Fo = int(sys.argv[3]) if len(sys.argv) > 3 else None
    dev = torch.device("cuda:0")
    lib = ctypes.CDLL(os.path.join(HERE, "librg.so"))
    n, E, F = SHAPES[wl]
    if Fo:
        F = Fo
    row, col = rmat_undirected_device(n, E, seed=1, device=dev, max_degree=maxdeg)
    adj = NormAdj.from_coo(row, col, None, n, dev)
    del row, col
    X = torch.randn(n, F, device=dev)
    ref = adj.propagate(X, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        adj.propagate(X, 1)
    e1.record(); torch.cuda.synchronize()
    deg = adj.rowptr[1:] - adj.rowptr[:-1]
    print(f"{wl} n={n} nnz={adj.nnz} F={F} max_deg={int(deg.max())}: plain CSR kernel {e0.elapsed_time(e1)/3:.3f} ms", flush=True)
    Xs = (X * adj.dinv[:, None]).contiguous()
    cfgs = ((16, 2, 1024, True), (16, 102, 1024, True), (16, 202, 1024, True)) if F > 128 else \
           ((8, 8, 1024, True), (8, 108, 1024, True), (8, 208, 1024, True))
    for R, U, seg, sortdeg in cfgs:
        g = build_groups(adj, R, seg, sortdeg)
        if os.environ.get("ALIGNED"):
            tgt, wmin, cap = (int(v) for v in os.environ["ALIGNED"].split(","))
            g = column_aligned_items(g, n, R, tgt, wmin, cap)
        c0 = g["gcol"][g["m0"].long()].long() // int(os.environ.get("ORDERCOLS", "4096"))
        perm = torch.argsort(c0 * (g["G"] + 1) + g["item_group"].long(), stable=True)
        for kk in ("item_group", "m0", "m1", "item_out"):
            g[kk] = g[kk][perm].contiguous()
        if os.environ.get("XCDCHUNK"):
            # emulate XCD-aware assignment: 8 chunks of equal merged-column work; block b -> chunk b%8
            ln = (g["m1"] - g["m0"]).long()
            cs = torch.cumsum(ln, 0)
            tot = int(cs[-1])
            bounds = [0] + [int(torch.searchsorted(cs, torch.tensor(tot * k // 8, device=dev))) for k in range(1, 8)] + [ln.numel()]
            bounds = [(b + 3) // 4 * 4 for b in bounds[:-1]] + [ln.numel()]
            mx = max(bounds[k + 1] - bounds[k] for k in range(8)); mx = (mx + 3) // 4 * 4
            nblk = mx // 4
            idxmap = torch.full((8 * mx,), -1, dtype=torch.long, device=dev)
            for k in range(8):
                cnt = bounds[k + 1] - bounds[k]
                src = torch.arange(cnt, device=dev)
                blk_in = src // 4; w = src % 4
                pos = (blk_in * 8 + k) * 4 + w
                idxmap[pos] = bounds[k] + src
            valid = idxmap >= 0
            for kk in ("item_group", "m0", "m1", "item_out"):
                old = g[kk]
                new = torch.zeros(8 * mx, dtype=old.dtype, device=dev)
                new[valid] = old[idxmap[valid]]
                if kk == "item_out":
                    new[~valid] = -2
                g[kk] = new.contiguous()
            print("xcd chunks (items):", [bounds[k + 1] - bounds[k] for k in range(8)])
        Y = torch.zeros(n, F, device=dev)
        partial = torch.zeros(max(g["nslots"], 1) * R * F, device=dev)
        a = RgArgs()
        for name, t in (("item_group", g["item_group"]), ("item_m0", g["m0"]), ("item_m1", g["m1"]),
                        ("item_out", g["item_out"]), ("grows", g["grows"]), ("gcol", g["gcol"]), ("gmask", g["gmask"]),
                        ("dinv", adj.dinv), ("X", Xs), ("Y", Y), ("partial", partial)):
            setattr(a, name, t.data_ptr())
        a.items = g["item_group"].numel(); a.F = F; a.row_bytes = F * 4; a.x_bytes = n * F * 4
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = lib.rg_spmm(R, U, ctypes.byref(a), st)
        assert rc == 0, rc
        torch.cuda.synchronize()
        # combine partials (torch, prototype only)
        io = g["item_out"].long(); sel = io >= 0
        if sel.any():
            P = partial.view(-1, R, F)
            grp = g["item_group"].long()[sel]
            rowsel = g["grows"].view(-1, R).long()[grp]          # [items, R]
            valid = rowsel >= 0
            Yp = torch.zeros(n, F, device=dev)
            Yp.index_add_(0, rowsel[valid], P[io[sel]][valid])
            long_rows = torch.unique(rowsel[valid])
            Y[long_rows] = Yp[long_rows] * adj.dinv[long_rows, None]
        err = float((Y - ref).norm() / ref.norm()) if U < 100 else float("nan")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            lib.rg_spmm(R, U, ctypes.byref(a), st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"R={R:2d} U={U} seg={seg} sortdeg={sortdeg}: merged={g['gcol'].numel()} ({100*g['gcol'].numel()/adj.nnz:.1f}%) "
              f"items={a.items} {ms:7.3f} ms  rel.err {err:.1e}", flush=True)


if __name__ == "__main__":
    main()
