#!/usr/bin/env python3
"""Workgroups per split group of the tile-group combine pass (run on the GPU box, under
rocprofv3 --kernel-trace --stats to see k_tg_combine alone, or plain for the hop time):
    python tools/combine_probe.py <gy> [workload]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgnn_amd import NormAdj, _lib  # noqa: E402
from sgnn_amd.graphgen import SHAPES, rmat_undirected_device  # noqa: E402

dev = torch.device("cuda:0")
wl = sys.argv[2] if len(sys.argv) > 2 else "reddit"
n, E, F = SHAPES[wl]
row, col = rmat_undirected_device(n, E, seed=1, device=dev)
adj = NormAdj.from_coo(row, col, None, n, dev)
del row, col
X = torch.randn(n, F, device=dev)
lib = _lib.load()
for gy in [int(v) for v in sys.argv[1].split(",")]:
    lib.sgnn_tune_set(b"tg_combine_gy", gy)
    for _ in range(3):
        adj.propagate(X, 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        adj.propagate(X, 1)
    e1.record()
    torch.cuda.synchronize()
    print(f"{wl} combine workgroups per split group {gy or 'built-in'}: hop {e0.elapsed_time(e1) / 30:.3f} ms", flush=True)
