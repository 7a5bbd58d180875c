#!/usr/bin/env python3
"""Where does the tile-group kernel pay?  (run on the GPU box)  One hop on synthetic R-MAT graphs
of several sizes, average degrees and widths: the tile-group kernel forced on against the
kernel the row-group / plain policy picks, with the sharing figure (stored entries per fetched
row, nnz / merged) the automatic choice is made from."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgnn_amd import NormAdj  # noqa: E402
from sgnn_amd.graphgen import rmat_undirected_device  # noqa: E402

dev = torch.device("cuda:0")
CASES = [  # n, directed entries, widths
    (19_717, 88_648, (500, 64)),            # PubMed shape
    (169_343, 2_315_598, (128, 40)),        # ogbn-arxiv shape (symmetrised)
    (100_000, 1_600_000, (128,)),           # average degree 16
    (100_000, 6_400_000, (128,)),           # 64
    (1_000_000, 10_000_000, (128, 32)),     # 10
    (1_000_000, 30_000_000, (128,)),        # 30
    (2_449_029, 123_718_280, (100,)),       # ogbn-products shape
]


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for n, E, widths in CASES:
    row, col = rmat_undirected_device(n, E, seed=1, device=dev)
    adj = NormAdj.from_coo(row, col, None, n, dev)
    del row, col
    for F in widths:
        X = torch.randn(n, F, device=dev)
        adj.use_tilegroup, adj._tg = True, None
        Y = adj.propagate(X, 1)
        merged = adj._tg["merged"]
        reps = 200 if adj.nnz < 20_000_000 else 20
        t_tg = timeit(lambda: adj.propagate(X, 1), reps)
        adj.use_tilegroup = False
        Y2 = adj.propagate(X, 1)
        t_other = timeit(lambda: adj.propagate(X, 1), reps)
        err = float((Y - Y2).norm() / Y2.norm())
        print(f"n={n} nnz={adj.nnz} (avg degree {adj.nnz / n:.1f}) F={F}: sharing {adj.nnz / merged:.2f}; "
              f"tile-group {t_tg * 1e3:.1f} us, row-group / plain {t_other * 1e3:.1f} us, ratio {t_tg / t_other:.2f}; diff {err:.1e}",
              flush=True)
        del X, Y, Y2
    del adj
    torch.cuda.empty_cache()
