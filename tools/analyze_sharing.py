#!/usr/bin/env python3
"""How much a fetched row of X can be shared on the benchmark graph (run on the GPU box).

Prints, for rows ranked by degree and cut into groups of R rows:
  * merged(R): distinct (group, column) pairs = rows of X one group-cooperative kernel
    fetches, for R from 16 to 4096;
  * the share of the stored entries that lies in the top-K columns by degree;
  * the density histogram of 32-row x 64-column tiles with both axes in degree order:
    share of the entries that lies in tiles at least 1/8, 1/4, 1/2 full;
  * pairs(32 | R): distinct (32-row sub-tile, column) pairs, the matrix-core work of a
    kernel that shares fetches between the R rows of a workgroup but multiplies per
    32-row sub-tile, with its columns packed 16 at a time per window of W workgroup
    columns (padding counted).

    python tools/analyze_sharing.py [workload] [max_degree] > gpurun_out/sharing.txt
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgnn_amd import NormAdj  # noqa: E402
from sgnn_amd.graphgen import SHAPES, rmat_undirected_device  # noqa: E402

dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "reddit"
maxdeg = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] != "0" else None
n, E, F = SHAPES[wl]
row, col = rmat_undirected_device(n, E, seed=1, device=dev, max_degree=maxdeg)
adj = NormAdj.from_coo(row, col, None, n, dev)
del row, col
deg = (adj.rowptr[1:] - adj.rowptr[:-1]).long()
rows = torch.repeat_interleave(torch.arange(n, device=dev), deg)
cols = adj.colidx.long()
nnz = adj.nnz
print(f"workload={wl} max_degree_clip={maxdeg} n={n} nnz={nnz} max_deg={int(deg.max())}", flush=True)
order = torch.argsort(deg, descending=True, stable=True)
rank = torch.empty_like(order)
rank[order] = torch.arange(n, device=dev)
rr, rc = rank[rows], rank[cols]
del rows

qs = torch.tensor([0.5, 0.9, 0.99, 0.999], device=dev)
print("degree quantiles 50/90/99/99.9 %:", [int(v) for v in torch.quantile(deg.float(), qs)], flush=True)

merged = {}
for R in (16, 32, 64, 128, 256, 512, 1024, 2048, 4096):
    u = torch.unique((rr // R) * n + cols).numel()
    merged[R] = u
    print(f"R={R:5d}: merged {u:12d} = {100.0 * u / nnz:5.1f}% of entries; rows of X fetched {u * F * 4 / 1e9:7.1f} GB; "
          f"dense work merged*R = {u * R / 1e9:8.2f} G", flush=True)

for k in (1024, 4096, 16384, 65536):
    colb = int((rc < k).sum())
    inb = int(((rr < k) & (rc < k)).sum())
    print(f"top-{k:6d} columns hold {100.0 * colb / nnz:4.1f}% of the entries; hub x hub block {100.0 * inb / nnz:4.1f}% "
          f"(density {inb / (k * k):.3f})", flush=True)

# tile density histogram, both axes in degree order
TR, TC = 32, 64
ntc = (n + TC - 1) // TC
tile = (rr // TR) * ntc + rc // TC
cnt = torch.bincount(tile, minlength=((n + TR - 1) // TR) * ntc)
del tile
per_entry = cnt.float() / (TR * TC)
for thr in (0.03125, 0.0625, 0.125, 0.25, 0.5):
    sel = per_entry >= thr
    print(f"tiles {TR}x{TC} at least {thr:7.5f} full: {int(sel.sum()):9d} tiles hold {100.0 * int(cnt[sel].sum()) / nnz:5.1f}% "
          f"of the entries", flush=True)
del cnt, per_entry

# matrix-core work with per-sub-tile column packing inside windows of W workgroup columns
for R in (128, 256, 512):
    gkey = (rr // R) * n + cols                       # (workgroup, column)
    uniq, inv = torch.unique(gkey, return_inverse=True)   # position of every entry's column in the merged order
    gfirst = torch.searchsorted(uniq, (torch.arange((n + R - 1) // R, device=dev)) * n)
    pos_in_group = inv - gfirst[rr // R]
    del gkey, uniq
    for W in (32, 64, 128):
        win = pos_in_group // W
        nwin = int(win.max()) + 1
        sub = rr // 32
        # distinct (sub-tile, window, column-position) triples, then count per (sub-tile, window)
        tkey = torch.unique((sub * nwin + win) * W + pos_in_group % W)
        sw = tkey // W
        c = torch.unique(sw, return_counts=True)[1]
        chunks16 = int(((c + 15) // 16).sum())
        chunks32 = int(((c + 31) // 32).sum())
        print(f"workgroup rows {R:4d}, window {W:4d} columns: (sub-tile, column) pairs {tkey.numel():11d}; "
              f"16-column MFMA steps {chunks16:10d} (fill {tkey.numel() / (16.0 * chunks16):.2f}); "
              f"dense steps {merged[R] * (R // 32) // 16:10d}", flush=True)
        del tkey, sw, c, win
    del inv, pos_in_group, gfirst
