#!/usr/bin/env python3
"""One hop on the benchmark graph with random edge weights (no uniform-weight shortcut):
row-group kernel with per-entry weights against the plain CSR kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgnn_amd import NormAdj
from sgnn_amd.graphgen import SHAPES, rmat_undirected_device
dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "reddit"
n, E, F = SHAPES[wl]
row, col = rmat_undirected_device(n, E, seed=1, device=dev)
val = torch.rand(row.numel(), device=dev) + 0.5
adj = NormAdj.from_coo(row, col, val, n, dev)
assert adj.w_off is None
X = torch.randn(n, F, device=dev)
out = {}
for rg in (True, False):
    adj.use_rowgroup = rg
    for _ in range(3):
        Y = adj.propagate(X, 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        Y = adj.propagate(X, 1)
    e1.record(); torch.cuda.synchronize()
    out[rg] = Y
    print(f"{wl} random weights, rowgroup={rg}: {e0.elapsed_time(e1) / 10:.3f} ms/hop", flush=True)
print("relative difference:", float((out[True] - out[False]).norm() / out[False].norm()))
