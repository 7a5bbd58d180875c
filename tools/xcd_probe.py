#!/usr/bin/env python3
"""Work-item order and per-XCD work queues of the tile-group kernel (run on the GPU box):
    python tools/xcd_probe.py [workload]                 the library's choice and the three forced combinations, twice, timed
    python tools/xcd_probe.py <workload> <order> <mode>   ten hops of one combination (for a counter pass)
order 0 = chosen by the graph (default); 1 = longest first; -1 = neighbours together (length buckets,
then column block, then group).
mode 0 = work counters as the plan's order wants them (default); 256 = one per XCD, runs of 32
neighbours; 1024 = one for the launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgnn_amd import NormAdj, _lib  # noqa: E402
from sgnn_amd.graphgen import SHAPES, rmat_undirected_device  # noqa: E402

dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "reddit"
n, E, F = SHAPES[wl]
row, col = rmat_undirected_device(n, E, seed=1, device=dev)
adj = NormAdj.from_coo(row, col, None, n, dev)
del row, col
g = torch.Generator(device=dev); g.manual_seed(2)
X = torch.randn(n, F, device=dev, generator=g)
lib = _lib.load()
adj.use_tilegroup = True


def run(reps=20):
    for _ in range(3):
        adj.propagate(X, 1)
    torch.cuda.synchronize()
    lib.sgnn_tune_set(b"time_kernel", 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        adj.propagate(X, 1)
    e1.record()
    torch.cuda.synchronize()
    k_ms, k_n = _lib.kernel_time_ms()
    lib.sgnn_tune_set(b"time_kernel", 0)
    return e0.elapsed_time(e1) / reps, k_ms / max(k_n, 1)


if len(sys.argv) > 3:
    lib.sgnn_tune_set(b"tg_order", int(sys.argv[2]))
    lib.sgnn_tune_set(b"tg_mode", int(sys.argv[3]))
    hop, ker = run(10)
    print(f"{wl} order {sys.argv[2]} mode {sys.argv[3]}: hop {hop:.3f} ms, main kernel {ker:.3f} ms", flush=True)
    sys.exit(0)
ref = None
for rep in range(2):
    for order, modes in ((0, (0,)), (1, (1024,)), (-1, (1024, 256))):
        lib.sgnn_tune_set(b"tg_order", order)
        adj._tg = None
        for mode in modes:
            lib.sgnn_tune_set(b"tg_mode", mode)
            Y = adj.propagate(X, 1)
            if ref is None:
                ref = Y.clone()
            ok = bool(torch.equal(Y, ref))
            hop, ker = run()
            print(f"{wl} order {order:2d} mode {mode:4d}: hop {hop:.3f} ms, main kernel {ker:.3f} ms, same bits as the first run: {ok}", flush=True)
lib.sgnn_tune_set(b"tg_mode", 0)
lib.sgnn_tune_set(b"tg_order", 0)
