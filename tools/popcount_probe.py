#!/usr/bin/env python3
"""Distribution of rows-per-merged-column in degree-ranked groups of R rows."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgnn_amd import NormAdj
from sgnn_amd.graphgen import SHAPES, rmat_undirected_device
dev = torch.device("cuda:0")
wl = sys.argv[1]
n, E, F = SHAPES[wl]
row, col = rmat_undirected_device(n, E, seed=1, device=dev)
adj = NormAdj.from_coo(row, col, None, n, dev); del row, col
deg = (adj.rowptr[1:] - adj.rowptr[:-1]).long()
rows = torch.repeat_interleave(torch.arange(n, device=dev), deg)
cols = adj.colidx.long()
order = torch.argsort(deg, descending=True, stable=True)
rank = torch.empty_like(order); rank[order] = torch.arange(n, device=dev)
for R in (8, 16):
    _, cnt = torch.unique((rank[rows] // R) * n + cols, return_counts=True)
    hist = torch.bincount(cnt, minlength=R + 1).cpu().tolist()
    tot = sum(hist)
    ent = sum(k * h for k, h in enumerate(hist))
    print(f"{wl} R={R}: merged {tot}, entries {ent}")
    print("   share of merged columns by popcount:", " ".join(f"{k}:{100.0*h/tot:.1f}%" for k, h in enumerate(hist) if h))
    print("   share of entries by popcount:       ", " ".join(f"{k}:{100.0*k*h/ent:.1f}%" for k, h in enumerate(hist) if h))
