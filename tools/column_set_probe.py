#!/usr/bin/env python3
"""How much of X does a rank of the row partition actually read?  (run on the GPU box)

For P = 2, 4, 8 ranks and the degree-cyclic deal of sgnn_amd.dist: the number of distinct
columns the rows of one rank reference, as a share of n.  An exchange of referenced rows only
(all-to-all-v) moves that share of what the all-gather moves; it can only help where the share
is well below 1.
    python tools/column_set_probe.py [workload ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgnn_amd.dist import degree_cyclic_deal  # noqa: E402
from sgnn_amd.graphgen import SHAPES, rmat_undirected_device  # noqa: E402

dev = torch.device("cuda:0")
for wl in (sys.argv[1:] or ["reddit", "products", "rmat10m"]):
    n, E, F = SHAPES[wl]
    row, col = rmat_undirected_device(n, E, seed=1, device=dev)
    counts = torch.bincount(row, minlength=n) + 1
    order, _ = degree_cyclic_deal(counts, 1)
    rank_of = torch.empty(n, dtype=torch.int64, device=dev)
    rank_of[order] = torch.arange(n, device=dev)
    for P in (2, 4, 8):
        owner = rank_of % P
        shares, weighted = [], []
        for p in range(P):
            mine = owner[row] == p
            seen = torch.zeros(n, dtype=torch.bool, device=dev)
            seen[col[mine]] = True
            seen[owner == p] = True                      # its own rows (self-loops)
            foreign = seen & (owner != p)
            shares.append(float(seen.float().mean()))
            weighted.append(float(foreign.sum()) / float((owner != p).sum()))
        print(f"{wl}: n={n} entries={row.numel()} P={P}: a rank references {min(shares) * 100:.1f}-{max(shares) * 100:.1f}% of "
              f"all rows of X; of the rows it would receive, {min(weighted) * 100:.1f}-{max(weighted) * 100:.1f}% are used",
              flush=True)
    del row, col
    torch.cuda.empty_cache()
