#!/usr/bin/env python3
"""Structure statistics of the benchmark graph (run on the GPU box): how many
row fetches a kernel that shares fetched rows of X between R output rows would
save, and how dense the hub x hub block is."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgnn_amd import NormAdj  # noqa: E402
from sgnn_amd.graphgen import SHAPES, rmat_undirected_device  # noqa: E402

dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "reddit"
maxdeg = int(sys.argv[2]) if len(sys.argv) > 2 else None
n, E, F = SHAPES[wl]
row, col = rmat_undirected_device(n, E, seed=1, device=dev, max_degree=maxdeg)
adj = NormAdj.from_coo(row, col, None, n, dev)
del row, col
deg = (adj.rowptr[1:] - adj.rowptr[:-1]).long()
rows = torch.repeat_interleave(torch.arange(n, device=dev), deg)
cols = adj.colidx.long()
nnz = adj.nnz
print(f"n={n} nnz={nnz} max_deg={int(deg.max())}")
order = torch.argsort(deg, descending=True, stable=True)
rank = torch.empty_like(order)
rank[order] = torch.arange(n, device=dev)
for label, rid in (("degree-sorted", rank), ("natural", torch.arange(n, device=dev))):
    for R in (2, 4, 8, 16, 32):
        key = (rid[rows] // R) * n + cols
        u = torch.unique(key).numel()
        print(f"{label:14s} R={R:3d}: fetches {u:12d} = {100.0 * u / nnz:5.1f}% of entries")
rr, rc = rank[rows], rank[cols]
for k in (1024, 2048, 4096, 8192, 16384, 32768):
    inb = int(((rr < k) & (rc < k)).sum())
    colb = int((rc < k).sum())
    print(f"top-{k:6d}: hub x hub entries {inb:11d} ({100.0 * inb / nnz:4.1f}% of all, block density {inb / (k * k):.3f}); "
          f"all rows x hub cols {100.0 * colb / nnz:4.1f}%")
