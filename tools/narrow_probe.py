#!/usr/bin/env python3
"""Narrow feature widths on a benchmark shape: one hop through the kernel the policy picks and
through the tile-group kernel forced on (run on the GPU box).   python tools/narrow_probe.py [workload]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgnn_amd import NormAdj  # noqa: E402
from sgnn_amd.graphgen import SHAPES, rmat_undirected_device  # noqa: E402

dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "reddit"
n, E, _ = SHAPES[wl]
row, col = rmat_undirected_device(n, E, seed=1, device=dev)
adj = NormAdj.from_coo(row, col, None, n, dev)
del row, col


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


for F in (8, 16, 32, 41, 64, 96):
    X = torch.randn(n, F, device=dev)
    out = []
    ref = None
    for mode in (None, True, False):
        adj.use_tilegroup = mode
        try:
            Y = adj.propagate(X, 1)
            t = timeit(lambda: adj.propagate(X, 1))
            if ref is None:
                ref = Y
            err = float((Y - ref).norm() / ref.norm())
            out.append(f"{'policy' if mode is None else ('tile-group' if mode else 'no tile-group')} {t:.3f} ms (diff {err:.1e})")
        except Exception as e:          # noqa: BLE001
            out.append(f"{mode}: {type(e).__name__} {str(e)[:60]}")
    print(f"{wl} F={F}: " + "; ".join(out), flush=True)
