#!/usr/bin/env python3
"""Time A_hat^K . X for K = 1, 2, 3 on a benchmark shape through the kernel the handle picks
(run on the GPU box):   python tools/chain_probe.py [workload] [K ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgnn_amd import NormAdj, _lib  # noqa: E402
from sgnn_amd.graphgen import SHAPES, rmat_undirected_device  # noqa: E402

dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "reddit"
orders = [int(v) for v in sys.argv[2:]] or [1, 2, 3]
n, E, F = SHAPES[wl]
row, col = rmat_undirected_device(n, E, seed=1, device=dev)
adj = NormAdj.from_coo(row, col, None, n, dev)
del row, col
g = torch.Generator(device=dev); g.manual_seed(2)
X = torch.randn(n, F, device=dev, generator=g)


def timeit(fn, reps=10):
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


out = []
for K in orders:
    out.append(f"K={K}: {timeit(lambda: adj.propagate(X, K)):.3f} ms")
print(f"{wl}: n={n} nnz={adj.nnz} F={F} tile-group={bool(adj._tg)}  " + "  ".join(out), flush=True)
