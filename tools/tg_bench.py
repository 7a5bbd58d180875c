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
        lib.sgnn_tune_set(b"tg_mode", 16 | extra)   # extra: 1 no steps, 2 no row DMA
        adj.propagate(X, 1); torch.cuda.synchronize(); fn(out)
        lib.sgnn_tune_set(b"time_kernel", 1)
        adj.propagate(X, 1); torch.cuda.synchronize(); fn(out)
        k_ms, k_n = _lib.kernel_time_ms()
        lib.sgnn_tune_set(b"time_kernel", 0)
        lib.sgnn_tune_set(b"tg_mode", 0)
        tot, wait, issue, steps, waves, nsteps, nstages = [out[i] for i in range(7)]
        # workgroup residence (100 MHz wall clock, summed) against 256 CUs busy for the whole kernel
        nwg = min(16384, adj._tg["items"] * ((F + 127) // 128))
        spans = (ctypes.c_ulonglong * (8 * nwg))()
        lib.sgnn_tilegroup_debug_spans(spans, nwg)
        import numpy as np
        sp = np.array(spans[:], dtype=np.int64).reshape(nwg, 8)
        t0, t1 = sp[:, 0].min(), sp[:, 1].max()
        ev = np.concatenate((np.stack((sp[:, 0], np.ones(nwg, np.int64)), 1), np.stack((sp[:, 1], -np.ones(nwg, np.int64)), 1)))
        ev = ev[np.argsort(ev[:, 0], kind="stable")]
        live = np.cumsum(ev[:, 1])
        dt = np.diff(ev[:, 0], append=ev[-1, 0])
        span = float(t1 - t0)
        print(f"workgroups {nwg}: first start to last end {span * 1e-5:.3f} ms; mean resident {float((live * dt).sum()) / span:.1f}; "
              f"time with all 256 resident {float(dt[live >= 256].sum()) / span * 100:.1f}%, with under 200 {float(dt[live < 200].sum()) / span * 100:.1f}%")
        deciles = [float((live * dt)[(ev[:, 0] >= t0 + span * k / 10) & (ev[:, 0] < t0 + span * (k + 1) / 10)].sum()) / (span / 10) for k in range(10)]
        print("mean resident workgroups per tenth of the kernel:", [round(v) for v in deciles])
        ok = sp[:, 6] > 0
        print(f"per unit, us (median / mean): entry to first stage issued {np.median(sp[:, 0] - sp[:, 5]) * 1e-2:.1f} / {(sp[:, 0] - sp[:, 5]).mean() * 1e-2:.1f}; "
              f"stage loop {np.median(sp[:, 1] - sp[:, 0]) * 1e-2:.1f} / {(sp[:, 1] - sp[:, 0]).mean() * 1e-2:.1f}; "
              f"wavefront 0 writing its rows {np.median(sp[:, 4] - sp[:, 1]) * 1e-2:.1f} / {(sp[:, 4] - sp[:, 1]).mean() * 1e-2:.1f}; "
              f"until the workgroup has its next unit {np.median((sp[:, 6] - sp[:, 4])[ok]) * 1e-2:.1f} / {(sp[:, 6] - sp[:, 4])[ok].mean() * 1e-2:.1f}")
        bar = (sp[:, 7] & 0xFFFFFFFF) - (sp[:, 1] - sp[:, 0])
        first = (sp[:, 7] >> 32) - (sp[:, 1] - sp[:, 0])
        print(f"  of the write-out, us after wavefront 0 left the stage loop (median / mean): barrier passed {np.median(bar) * 1e-2:.1f} / {bar.mean() * 1e-2:.1f}; "
              f"first eight row pairs stored and acknowledged {np.median(first) * 1e-2:.1f} / {first.mean() * 1e-2:.1f}")
        dur = (sp[:, 1] - sp[:, 0]) * 1e-2
        gaps, busy_by_xcc, end_by_xcc = [], {}, {}
        for cu in np.unique(sp[:, 2]):
            mine = sp[sp[:, 2] == cu]
            mine = mine[np.argsort(mine[:, 0])]
            gaps.extend(((mine[1:, 0] - mine[:-1, 1]) * 1e-2).tolist())
            x = int(cu) >> 16
            busy_by_xcc[x] = busy_by_xcc.get(x, 0.0) + float((mine[:, 1] - mine[:, 0]).sum()) * 1e-5
            end_by_xcc[x] = max(end_by_xcc.get(x, 0.0), float(mine[:, 1].max() - t0) * 1e-5)
        gaps = np.array(gaps)
        print(f"distinct CUs {len(np.unique(sp[:, 2]))}; gap between consecutive workgroups of a CU, us: median {np.median(gaps):.1f} mean {gaps.mean():.1f} "
              f"p90 {np.percentile(gaps, 90):.1f} max {gaps.max():.0f}; summed {gaps.sum() * 1e-3 / 256:.3f} ms per CU")
        print("per XCC: busy ms per CU", {k: round(v / 32, 3) for k, v in sorted(busy_by_xcc.items())}, "last end ms", {k: round(v, 3) for k, v in sorted(end_by_xcc.items())})
        print(f"us per stage by workgroup length: long (>=150 stages) {dur[sp[:, 3] >= 150].sum() / sp[sp[:, 3] >= 150, 3].sum():.2f}, short {dur[sp[:, 3] < 150].sum() / max(1, sp[sp[:, 3] < 150, 3].sum()):.2f}")
        print(f"workgroup duration us: mean {dur.mean():.0f} max {dur.max():.0f}; last 256 to start: mean {dur[np.argsort(sp[:, 0])[-256:]].mean():.0f}")
        print(f"mode {extra}: kernel {k_ms / max(k_n, 1):.3f} ms; CUs occupied {out[7] * 1e-5 / (256 * k_ms / max(k_n, 1)) * 100:.1f}% of the kernel's time", flush=True)
        print(f"mode {extra}: per wavefront: total {tot / waves:.0f} cycles; stage wait {100 * wait / tot:.1f}% issue {100 * issue / tot:.1f}% "
              f"steps {100 * steps / tot:.1f}%; {steps / max(nsteps, 1):.0f} cycles per step, {nsteps / nstages:.2f} steps and "
              f"{wait / nstages:.0f} + {issue / nstages:.0f} + {steps / nstages:.0f} cycles per stage", flush=True)
if os.environ.get("TG_MODES"):
    adj.use_tilegroup = True
    for m in [int(v) for v in os.environ["TG_MODES"].split(",")]:
        _lib.load().sgnn_tune_set(b"tg_mode", m)
        print(f"mode {m} (1 = no steps, 2 = no row DMA, 32 = no rows written, 64 = all fetches from the first 4096 rows, 576 = from the first 512): {timeit(lambda: adj.propagate(X, 1)):.3f} ms/hop", flush=True)
print(f"tile-group {t_tg:.3f} ms/hop   row-group {t_rg:.3f} ms/hop   normwise diff {err:.2e}  element-wise {el:.2e}", flush=True)
if os.environ.get("TG_SWEEP"):
    adj.use_tilegroup = True
    lib = _lib.load()
    for rep in range(3):
        for mode in (0, 128):
            lib.sgnn_tune_set(b"tg_mode", mode)
            print(f"{'one workgroup per pair' if mode else 'one workgroup per CU, work counter'}: {timeit(lambda: adj.propagate(X, 1)):.3f} ms/hop", flush=True)
    lib.sgnn_tune_set(b"tg_mode", 0)
