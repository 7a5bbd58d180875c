#!/usr/bin/env python3
"""Does the kernel's speed depend on how the nodes are numbered?  Times one hop on the
benchmark graph as generated (R-MAT puts hubs at low ids), with the node ids shuffled, and
with the nodes numbered by falling degree.  Optional second argument: the column-block size
the work items are ordered by (``rg_order``)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgnn_amd import NormAdj
from sgnn_amd.graphgen import SHAPES, rmat_undirected_device
dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "reddit"
n, E, F = SHAPES[wl]
row, col = rmat_undirected_device(n, E, seed=1, device=dev)
X = torch.randn(n, F, device=dev)
if len(sys.argv) > 2:
    from sgnn_amd import _lib
    _lib.load().sgnn_tune_set(b"rg_order", int(sys.argv[2]))
for label in ("as generated", "ids shuffled", "by falling degree"):
    if label == "ids shuffled":
        p = torch.randperm(n, device=dev)
        row, col = p[row], p[col]
    elif label == "by falling degree":
        deg = torch.bincount(row, minlength=n)
        p = torch.empty(n, dtype=torch.int64, device=dev)
        p[torch.argsort(deg, descending=True)] = torch.arange(n, device=dev)
        row, col = p[row], p[col]
    adj = NormAdj.from_coo(row, col, None, n, dev)
    for rg in (True, False):
        adj.use_rowgroup = rg
        for _ in range(3):
            adj.propagate(X, 1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            adj.propagate(X, 1)
        e1.record(); torch.cuda.synchronize()
        print(f"{wl} {label}, rowgroup={rg}: {e0.elapsed_time(e1) / 10:.3f} ms/hop", flush=True)
    del adj
