#!/usr/bin/env python3
"""Time one hop of every SpMM kernel on a benchmark shape (run on the GPU box).
    python tools/tg_bench.py [workload] [seg]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgnn_amd import NormAdj, _lib  # noqa: E402
from sgnn_amd.graphgen import SHAPES, rmat_undirected_device  # noqa: E402

dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "reddit"
if len(sys.argv) > 2:
    _lib.load().sgnn_tune_set(b"tg_seg", int(sys.argv[2]))
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


adj.use_tilegroup = True
t0 = time.perf_counter()
Y_tg = adj.propagate(X, 1)
torch.cuda.synchronize()
print(f"{wl}: n={n} nnz={adj.nnz} F={F}; tile-group plan {adj._tg and {k: v for k, v in adj._tg.items() if k not in ('buf', 'bufs')}} "
      f"plan MB {adj._tg['bytes'] / 1e6:.0f}; first call {time.perf_counter() - t0:.2f} s; "
      f"flag {int(adj._tg_bufs[F]['state'][1].item())} tiny {int(adj._tg_bufs[F]['state'][4].item())}", flush=True)
t_tg = timeit(lambda: adj.propagate(X, 1))
adj.use_tilegroup = False
Y_rg = adj.propagate(X, 1)
t_rg = timeit(lambda: adj.propagate(X, 1))
err = float((Y_tg - Y_rg).norm() / Y_rg.norm())
scale = adj.propagate(X.abs(), 1)
el = float(((Y_tg - Y_rg).abs() / scale.clamp_min(1e-30)).max())
if os.environ.get("TG_CYCLES"):
    import ctypes
    lib = _lib.load()
    adj.use_tilegroup = True
    out = (ctypes.c_ulonglong * 8)()
    fn = lib.sgnn_tilegroup_debug_counters
    for extra in [int(v) for v in os.environ["TG_CYCLES"].split(",")]:
        lib.sgnn_tune_set(b"tg_mode", 16 | extra)
        adj.propagate(X, 1); torch.cuda.synchronize(); fn(out)
        adj.propagate(X, 1); torch.cuda.synchronize(); fn(out)
        lib.sgnn_tune_set(b"tg_mode", 0)
        tot, wait, issue, steps, waves, nsteps, nstages = [out[i] for i in range(7)]
        print(f"mode {extra}: per wavefront: total {tot / waves:.0f} cycles; stage wait {100 * wait / tot:.1f}% issue {100 * issue / tot:.1f}% "
              f"steps {100 * steps / tot:.1f}%; {steps / max(nsteps, 1):.0f} cycles per step, {nsteps / nstages:.2f} steps and "
              f"{wait / nstages:.0f} + {issue / nstages:.0f} + {steps / nstages:.0f} cycles per stage", flush=True)
if os.environ.get("TG_MODES"):
    adj.use_tilegroup = True
    for m in (1, 2, 3, 0):
        _lib.load().sgnn_tune_set(b"tg_mode", m)
        print(f"mode {m} (1 = no steps, 2 = no row DMA): {timeit(lambda: adj.propagate(X, 1)):.3f} ms/hop", flush=True)
print(f"tile-group {t_tg:.3f} ms/hop   row-group {t_rg:.3f} ms/hop   normwise diff {err:.2e}  element-wise {el:.2e}", flush=True)
if os.environ.get("TG_SWEEP"):
    adj.use_tilegroup = True
    lib = _lib.load()
    wls = {}
    for order in (4096, 0):
        lib.sgnn_tune_set(b"tg_order", order)
        for seg in (128, 160, 192, 224, 256, 288, 320, 384):
            lib.sgnn_tune_set(b"tg_seg", seg)
            adj._tg = None
            adj._tg_bufs = {}
            adj.propagate(X, 1)
            print(f"order {'column block' if order else 'longest first'} tg_seg {seg}: items {adj._tg['items']} partial slots {adj._tg['slots']}: {timeit(lambda: adj.propagate(X, 1)):.3f} ms/hop", flush=True)
    lib.sgnn_tune_set(b"tg_order", 0)
    lib.sgnn_tune_set(b"tg_seg", 0)
