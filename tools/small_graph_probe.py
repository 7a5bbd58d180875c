#!/usr/bin/env python3
"""PubMed-size graphs: one hop under the row-group kernel's group size / segment overrides
and under the plain kernel (run on the GPU box).   python tools/small_graph_probe.py [workload]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgnn_amd import NormAdj, _lib  # noqa: E402
from sgnn_amd.graphgen import SHAPES, rmat_undirected_device  # noqa: E402

dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "pubmed"
n, E, F = SHAPES[wl]
row, col = rmat_undirected_device(n, E, seed=1, device=dev)
X = torch.randn(n, F, device=dev)
lib = _lib.load()


def run(tag, R, seg, rowgroup=True):
    lib.sgnn_tune_set(b"rg_R", R)
    lib.sgnn_tune_set(b"rg_seg", seg)
    adj = NormAdj.from_coo(row, col, None, n, dev)
    adj.use_rowgroup = rowgroup
    for _ in range(5):
        adj.propagate(X, 1)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):                      # one graph replay per hop: launch overhead off the host
        Y = adj.propagate(X, 1)
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    lib.sgnn_tune_set(b"time_kernel", 1)
    for _ in range(20):
        adj.propagate(X, 1)
    k_ms, k_n = _lib.kernel_time_ms()
    lib.sgnn_tune_set(b"time_kernel", 0)
    plan = next(iter(adj._plans.values()), None)
    print(f"{wl} {tag}: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us per hop (graph replay), main kernel {k_ms / max(k_n, 1) * 1e3:.1f} us"
          + (f"; items {plan['items']} split groups {plan['n_long']} merged {plan['merged']}" if plan else ""), flush=True)


run("policy", 0, 0)
for R in (8, 16):
    for seg in (16, 32, 64, 128):
        run(f"R={R} seg={seg}", R, seg)
run("plain CSR", 0, 0, rowgroup=False)
