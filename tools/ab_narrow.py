#!/usr/bin/env python3
"""Same-box comparison of two builds of the library at narrow feature widths (run on the GPU box):
one Reddit-shape hop per width through the library file named on the command line, main kernel
time from the library's own events.    python tools/ab_narrow.py <lib.so> [widths ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from sgnn_amd import _lib  # noqa: E402
_lib.LIB_PATH = os.path.join(_root, sys.argv[1])
from sgnn_amd import NormAdj  # noqa: E402
from sgnn_amd.graphgen import SHAPES, rmat_undirected_device  # noqa: E402

dev = torch.device("cuda:0")
n, E, _ = SHAPES["reddit"]
row, col = rmat_undirected_device(n, E, seed=1, device=dev)
adj = NormAdj.from_coo(row, col, None, n, dev)
del row, col
lib = _lib.load()
if os.environ.get("TG_SEG"):          # stages per work item of the plan built below (0 / unset: the library's choice)
    lib.sgnn_tune_set(b"tg_seg", int(os.environ["TG_SEG"]))
for F in [int(v) for v in sys.argv[2:]] or [16, 32, 41, 64, 76, 96, 128]:
    X = torch.randn(n, F, device=dev)
    adj.use_tilegroup = True
    for _ in range(3):
        Y = adj.propagate(X, 1)
    torch.cuda.synchronize()
    lib.sgnn_tune_set(b"time_kernel", 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        adj.propagate(X, 1)
    e1.record()
    torch.cuda.synchronize()
    k_ms, k_n = _lib.kernel_time_ms()
    lib.sgnn_tune_set(b"time_kernel", 0)
    adj.use_tilegroup = False
    Yr = adj.propagate(X, 1)
    print(f"{sys.argv[1]} seg={os.environ.get('TG_SEG', 'auto')} items={adj._tg['items']} F={F}: hop {e0.elapsed_time(e1) / 20:.3f} ms, main kernel {k_ms / k_n:.3f} ms, "
          f"normwise difference from the other kernels {float((Y - Yr).norm() / Yr.norm()):.2e}", flush=True)
