#!/usr/bin/env python3
"""How many row fetches would other row orders save?  Rows are cut into groups of R
consecutive rows of an order; a group fetches every distinct column once.  Orders:
degree rank (what the plans use), min-hash of the neighbour set inside degree bands,
and plain min-hash."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgnn_amd import NormAdj  # noqa: E402
from sgnn_amd.graphgen import SHAPES, rmat_undirected_device  # noqa: E402

dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "reddit"
R = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n, E, F = SHAPES[wl]
row, col = rmat_undirected_device(n, E, seed=1, device=dev)
adj = NormAdj.from_coo(row, col, None, n, dev)
del row, col
deg = (adj.rowptr[1:] - adj.rowptr[:-1]).long()
rows = torch.repeat_interleave(torch.arange(n, device=dev), deg)
cols = adj.colidx.long()
nnz = adj.nnz


def fetches(order):
    rank = torch.empty_like(order)
    rank[order] = torch.arange(n, device=dev)
    return torch.unique((rank[rows] // R) * n + cols).numel()


def minhash(seed, weight_by_rarity=False):
    h = (cols * (2654435761 + 2 * seed) + 40503 * seed) % 2147483647
    if weight_by_rarity:   # hubs get large hashes, so the minimum is decided by rare columns
        cdeg = torch.bincount(cols, minlength=n)
        h = h // 1024 + (cdeg[cols] // 64) * (2147483647 // 1024)
    out = torch.full((n,), 2 ** 62, dtype=torch.long, device=dev)
    return out.scatter_reduce(0, rows, h, "amin")


base = torch.argsort(deg, descending=True, stable=True)
print(f"{wl} R={R}: degree rank          {100.0 * fetches(base) / nnz:5.1f}%", flush=True)
for bands in (64, 256, 1024, 4096):
    band = torch.empty(n, dtype=torch.long, device=dev)
    band[base] = torch.arange(n, device=dev) * bands // n
    for rarity in (False, True):
        mh = minhash(1, rarity)
        key = band * (2 ** 40) + mh % (2 ** 40)
        order = torch.argsort(key, stable=True)
        print(f"  {bands:5d} degree bands + minhash(rarity={rarity}): {100.0 * fetches(order) / nnz:5.1f}%", flush=True)
mh = minhash(1)
print(f"  plain minhash: {100.0 * fetches(torch.argsort(mh, stable=True)) / nnz:5.1f}%")
