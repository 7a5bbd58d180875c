#!/usr/bin/env python3
"""Same-box comparison of two builds of the library (run on the GPU box): one Reddit-shape hop
through the library file named on the command line, relative to the repository root.
Build the variants on the CPU (`hipcc` as `__graft_entry__.build` does), drop them next to the
sources as e.g. ab_old.so / ab_new.so, and alternate:
    for v in old new old new; do python tools/ab_lib.py ab_$v.so; done
A second argument gives every off-diagonal entry that weight (2: Reddit's A + A^T).
(delete the files afterwards: every GPU call uploads the tree)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_lib_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from sgnn_amd import _lib  # noqa: E402
_lib.LIB_PATH = os.path.join(_lib_root, sys.argv[1])
from sgnn_amd import NormAdj
from sgnn_amd.graphgen import SHAPES, rmat_undirected_device
dev = torch.device("cuda:0")
n, E, F = SHAPES["reddit"]
row, col = rmat_undirected_device(n, E, seed=1, device=dev)
val = torch.full((row.numel(),), float(sys.argv[2]), device=dev) if len(sys.argv) > 2 else None
adj = NormAdj.from_coo(row, col, val, n, dev)
del row, col, val
X = torch.randn(n, F, device=dev)
lib = _lib.load()
for _ in range(5):
    adj.propagate(X, 1)
torch.cuda.synchronize()
lib.sgnn_tune_set(b"time_kernel", 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(40):
    adj.propagate(X, 1)
e1.record()
torch.cuda.synchronize()
k_ms, k_n = _lib.kernel_time_ms()
for K in [int(v) for v in os.environ.get("AB_K", "").split(",") if v]:
    adj.propagate(X, K)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        adj.propagate(X, K)
    e1.record()
    torch.cuda.synchronize()
    print(sys.argv[1], "K = %d: %.3f ms, flags %s" % (K, e0.elapsed_time(e1) / 5, adj._tg_bufs[F]["states"][:, :8].tolist()), flush=True)
Y = adj.propagate(X, 1)
adj.use_tilegroup = False
Yr = adj.propagate(X, 1)
print(sys.argv[1], "hop %.3f ms, main kernel %.3f ms, normwise difference from the row-group kernel %.2e"
      % (e0.elapsed_time(e1) / 40, k_ms / k_n, float((Y - Yr).norm() / Yr.norm())), flush=True)
