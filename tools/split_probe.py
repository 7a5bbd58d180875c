#!/usr/bin/env python3
"""Row-block sizes of the nnz-balanced contiguous partition on a benchmark graph: how much
padding the equal-piece all-gather carries."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgnn_amd.graphgen import SHAPES, rmat_undirected_device
from sgnn_amd.dist import balanced_row_splits
dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "reddit"
n, E, F = SHAPES[wl]
row, col = rmat_undirected_device(n, E, seed=1, device=dev)
counts = torch.bincount(row, minlength=n) + 1
for world in (2, 4, 8):
    s = balanced_row_splits(counts, world)
    rows = [s[p + 1] - s[p] for p in range(world)]
    nnz = [int(counts[s[p]:s[p + 1]].sum()) for p in range(world)]
    print(f"{wl} world={world}: rows per rank {rows}; padded gather rows {world * max(rows)} vs n={n} "
          f"({world * max(rows) / n:.2f}x); nnz per rank min/max {min(nnz)}/{max(nnz)}")
