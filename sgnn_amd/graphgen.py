"""Synthetic graphs of the shapes the propagation path is measured on.

No dataset can be downloaded where this code runs, so throughput is reported on
seeded R-MAT graphs with the node count, edge count and feature width of the
datasets the reference is configured for (Reddit: config.json:480-503 and
data_loader.py:56-94; ogbn-products: data_loader.py:97-137).  The generators are
this package's own; the reference has none.

Two back-ends produce the same *kind* of graph (not the same edges):
``rmat_edges_numpy`` for small CPU test graphs and ``rmat_undirected_device`` for
the 10^8-edge benchmark graphs, which are drawn, symmetrised and de-duplicated on
the GPU so that a benchmark run does not spend minutes in host code.
"""
from __future__ import annotations

import numpy as np

RMAT_ABCD = (0.57, 0.19, 0.19, 0.05)

# shapes quoted by the benchmark configurations: (nodes, directed edge count, F)
SHAPES = {
    "reddit": (232_965, 114_615_892, 602),
    "reddit128": (232_965, 114_615_892, 128),
    "products": (2_449_029, 123_718_280, 100),
    "rmat10m": (10_000_000, 200_000_000, 128),
    "pubmed": (19_717, 88_648, 500),
}


def rmat_edges_numpy(n, n_edges, seed, abcd=RMAT_ABCD):
    """Draw ``n_edges`` directed R-MAT pairs with ids < n (host, numpy).

    Pairs with an id >= n are redrawn; self-loops and duplicates are NOT removed
    here (tests want them).
    """
    rng = np.random.default_rng(seed)
    scale = max(1, int(np.ceil(np.log2(max(n, 2)))))
    a, b, c, _ = abcd
    out_r = np.empty(0, dtype=np.int64)
    out_c = np.empty(0, dtype=np.int64)
    while out_r.size < n_edges:
        m = int((n_edges - out_r.size) * 1.3) + 16
        r = np.zeros(m, dtype=np.int64)
        cc = np.zeros(m, dtype=np.int64)
        for _ in range(scale):
            u = rng.random(m)
            rbit = (u >= a + b).astype(np.int64)
            cbit = (((u >= a) & (u < a + b)) | (u >= a + b + c)).astype(np.int64)
            r = (r << 1) | rbit
            cc = (cc << 1) | cbit
        ok = (r < n) & (cc < n)
        out_r = np.concatenate((out_r, r[ok]))
        out_c = np.concatenate((out_c, cc[ok]))
    return out_r[:n_edges], out_c[:n_edges]


def symmetric_simple_numpy(n, n_undirected, seed, abcd=RMAT_ABCD):
    """Symmetric, loop-free, duplicate-free R-MAT graph with about
    ``2*n_undirected`` stored entries.  Returns (row, col) int64, both directions.
    """
    r, c = rmat_edges_numpy(n, int(n_undirected * 1.5) + 16, seed, abcd)
    keep = r != c
    lo = np.minimum(r[keep], c[keep])
    hi = np.maximum(r[keep], c[keep])
    key = np.unique(lo * np.int64(n) + hi)
    if key.size > n_undirected:
        rng = np.random.default_rng(seed + 7919)
        key = np.sort(rng.choice(key, size=n_undirected, replace=False))
    lo, hi = key // n, key % n
    return np.concatenate((lo, hi)), np.concatenate((hi, lo))


def rmat_undirected_device(n, n_directed_edges, seed, device, abcd=RMAT_ABCD, max_degree=None,
                           batch=1 << 25):
    """Symmetric, loop-free, duplicate-free R-MAT graph drawn on ``device``.

    Returns int64 ``row``, ``col`` tensors holding exactly
    ``2 * (n_directed_edges // 2)`` entries (each undirected edge in both
    directions) unless the id space saturates first.  ``max_degree`` (optional)
    drops, per node, the edges beyond that many neighbours to mimic graphs whose
    hubs are milder than raw R-MAT's.
    """
    import torch

    target = n_directed_edges // 2
    scale = max(1, int(np.ceil(np.log2(max(n, 2)))))
    a, b, c, _ = abcd
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    keys = torch.empty(0, dtype=torch.int64, device=device)
    rounds = 0
    while keys.numel() < target and rounds < 64:
        rounds += 1
        m = int(min(batch, max((target - keys.numel()) * 2, 1 << 16)))
        r = torch.zeros(m, dtype=torch.int64, device=device)
        cc = torch.zeros(m, dtype=torch.int64, device=device)
        for _ in range(scale):
            u = torch.rand(m, device=device, generator=gen)
            rbit = (u >= a + b).to(torch.int64)
            cbit = (((u >= a) & (u < a + b)) | (u >= a + b + c)).to(torch.int64)
            r = (r << 1) | rbit
            cc = (cc << 1) | cbit
        ok = (r < n) & (cc < n) & (r != cc)
        r, cc = r[ok], cc[ok]
        lo, hi = torch.minimum(r, cc), torch.maximum(r, cc)
        keys = torch.unique(torch.cat((keys, lo * n + hi)))
        del r, cc, lo, hi, ok
    if keys.numel() > target:
        perm = torch.randperm(keys.numel(), device=device, generator=gen)[:target]
        keys = keys[perm]
    lo, hi = keys // n, keys % n
    if max_degree is not None:
        # rank every entry inside its row and drop the tail; applied to the
        # undirected list through the larger of the two endpoint ranks
        row = torch.cat((lo, hi))
        order = torch.argsort(row, stable=True)
        srow = row[order]
        start = torch.searchsorted(srow, srow)
        rank = torch.arange(srow.numel(), device=device) - start
        rank_full = torch.empty_like(rank)
        rank_full[order] = rank
        m = lo.numel()
        keep = (rank_full[:m] < max_degree) & (rank_full[m:] < max_degree)
        lo, hi = lo[keep], hi[keep]
    return torch.cat((lo, hi)), torch.cat((hi, lo))
