#!/usr/bin/env python3
"""Compute side of the multi-GPU layouts on one GPU: time the local SpMM of rank 0 for
several (feature groups x row ranks) grids of an N-rank job, without any exchange."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgnn_amd.dist import PartitionedAdj, grid_layout
from sgnn_amd.graphgen import SHAPES, rmat_undirected_device
dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "reddit"
n, E, F = SHAPES[wl]
row, col = rmat_undirected_device(n, E, seed=1, device=dev)
for world in (1, 2, 4, 8):
    for fg in [g for g in (1, 2, 4, 8) if world % g == 0]:
        row_rank, row_world, fgroup, f0, f1 = grid_layout(0, world, fg, F)
        padj = PartitionedAdj.from_coo(row, col, None, n, dev, row_rank, row_world)
        w = f1 - f0
        full = torch.randn(padj.max_rows * row_world, (w + 3) // 4 * 4, device=dev)
        out = torch.empty(padj.n_local, w, device=dev)
        for _ in range(3):
            padj.local.spmm_into(full[:, :w], out, final=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            padj.local.spmm_into(full[:, :w], out, final=True)
        e1.record(); torch.cuda.synchronize()
        gathered_mb = (row_world - 1) * padj.max_rows * w * 4 / 1e6
        print(f"{wl} N={world}: {fg} feature slices x {row_world} row blocks: local SpMM {e0.elapsed_time(e1) / 10:.3f} ms "
              f"(rows {padj.n_local}, width {w}), gathers {gathered_mb:.0f} MB per hop", flush=True)
        del padj, full, out

# Second part: the default schedule of a row partition (own-columns block, then remote-columns
# block; fp16 planes where the remote block merits the tile-group kernel), rank 0 of N, with the
# exchange itself stubbed out (one process: what is timed is everything a rank computes per hop,
# the split of its own rows and the small agreement calls included).
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29517")
dist.init_process_group("gloo", rank=0, world_size=1)
for world in (2, 4, 8):
    padj = PartitionedAdj.from_coo(row, col, None, n, dev, 0, world)
    padj._all_gather = lambda full, piece: None
    X_local = torch.randn(padj.n_local, F, device=dev)
    for order in (1, 2):
        for _ in range(3):
            padj.propagate(X_local, order)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            padj.propagate(X_local, order)
        e1.record(); torch.cuda.synchronize()
        print(f"{wl} N={world}, rank 0, order {order}: schedule {padj.last_schedule}, {e0.elapsed_time(e1) / 10:.3f} ms per call "
              f"(own block {padj.own.adj.nnz} entries, remote block {padj.remote.adj.nnz}; exchange stubbed)", flush=True)
    del padj, X_local
dist.destroy_process_group()
