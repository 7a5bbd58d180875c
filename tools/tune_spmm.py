#!/usr/bin/env python3
"""Sweep the SpMM slice width / unroll on one graph (run on the GPU box).

    python tools/tune_spmm.py [workload] [--tiles 640,256,128,64] [--unrolls 2,4]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgnn_amd import NormAdj, _lib  # noqa: E402
from sgnn_amd.graphgen import SHAPES, rmat_undirected_device  # noqa: E402


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", nargs="?", default="reddit")
    ap.add_argument("--tiles", default="640,320,256,128,64,32")
    ap.add_argument("--unrolls", default="2,4")
    ap.add_argument("--F", type=int, default=None)
    ap.add_argument("--n", type=int, default=None)
    ap.add_argument("--E", type=int, default=None)
    ap.add_argument("--dist", action="store_true", help="print the degree distribution")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    n, E, F = SHAPES[args.workload]
    if args.F:
        F = args.F
    if args.n:
        n = args.n
    if args.E:
        E = args.E
    row, col = rmat_undirected_device(n, E, seed=1, device=dev)
    adj = NormAdj.from_coo(row, col, None, n, dev)
    del row, col
    if args.dist:
        deg = (adj.rowptr[1:] - adj.rowptr[:-1]).long()
        s, _ = torch.sort(deg, descending=True)
        cs = torch.cumsum(s, 0).double() / s.sum()
        for k in (256, 640, 1742, 3500, 8000, 13900, 16000, 32000, 64000, 111000):
            if k < n:
                print(f"top {k:7d} columns hold {100 * cs[k - 1].item():5.1f}% of the entries (deg >= {s[k-1].item()})")
        print("rows longer than 1024:", int((deg > 1024).sum()), "holding",
              f"{100 * deg[deg > 1024].sum().item() / deg.sum().item():.1f}% of the entries")
    X = torch.randn(n, F, device=dev)
    lib.sgnn_tune_set(b"tile", 0); lib.sgnn_tune_set(b"unroll", 0)
    ref = adj.propagate(X, 1)
    gather_gb = adj.nnz * F * 4 / 1e9
    for tile in [int(t) for t in args.tiles.split(",")]:
        for u in [int(t) for t in args.unrolls.split(",")]:
            lib.sgnn_tune_set(b"tile", tile); lib.sgnn_tune_set(b"unroll", u)
            y = adj.propagate(X, 1)
            err = float((y - ref).norm() / ref.norm())
            ms = timeit(lambda: adj.propagate(X, 1))
            print(f"tile={tile:4d} unroll={u}  {ms:8.3f} ms   {adj.nnz * F / ms / 1e9:8.2f} T edge.feat/s  "
                  f"gather {gather_gb / ms:6.2f} TB/s  rel.diff {err:.1e}", flush=True)


if __name__ == "__main__":
    main()
