#!/usr/bin/env python3
"""Time one hop on the Reddit-shape graph for several feature widths (the vector width
and the loads per row the kernel picks depend on F)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgnn_amd import NormAdj
from sgnn_amd.graphgen import SHAPES, rmat_undirected_device
dev = torch.device("cuda:0")
n, E, _ = SHAPES["reddit"]
row, col = rmat_undirected_device(n, E, seed=1, device=dev)
adj = NormAdj.from_coo(row, col, None, n, dev); del row, col
for F in (int(v) for v in sys.argv[1:]):
    X = torch.randn(n, F, device=dev)
    for _ in range(3):
        adj.propagate(X, 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        adj.propagate(X, 1)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"F={F}: {ms:.3f} ms/hop, {ms / F * 1000:.2f} us per feature column", flush=True)
