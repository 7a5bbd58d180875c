#!/usr/bin/env python3
"""Experiment: does the ORDER of the row-group work items matter?  Re-sorts the
item arrays inside a built plan (segment-index-major instead of group-major) and
times the kernel again."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgnn_amd import NormAdj, _lib
from sgnn_amd.graphgen import SHAPES, rmat_undirected_device

def au(x, a=256): return (x + a - 1) // a * a

def seg_cols_for(nnz):
    seg = 64
    while seg < 1024 and nnz // seg > 32768: seg *= 2
    return seg

def item_views(plan, n, nnz, R):
    G = (n + R - 1) // R
    cap_long = nnz // seg_cols_for(nnz) + 1
    cap_items = G + cap_long
    off = au(64)
    off += au(G * R * 4); off += au((G + 1) * 4); off += au(nnz * 4); off += au(nnz * 4)
    views = []
    for _ in range(4):
        views.append(plan["buf"][off:off + cap_items * 4].view(torch.int32)); off += au(cap_items * 4)
    return views

def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "reddit"
n, E, F = SHAPES[wl]
row, col = rmat_undirected_device(n, E, seed=1, device=dev)
adj = NormAdj.from_coo(row, col, None, n, dev); del row, col
X = torch.randn(n, F, device=dev)
ref = adj.propagate(X, 1)
plan = adj._rowgroup_plan(F)
R, items = plan["R"], plan["items"]
print(f"R={R} items={items} merged={plan['merged']} slots={plan['slots']} long={plan['n_long']}")
print(f"group-major (as built): {timeit(lambda: adj.propagate(X, 1)):.3f} ms")
ig, m0, m1, io = item_views(plan, n, adj.nnz, R)
g = ig[:items].clone().long(); a0 = m0[:items].clone(); a1 = m1[:items].clone(); o = io[:items].clone()
first = torch.zeros(int(g.max()) + 2, dtype=torch.long, device=dev)
first.scatter_reduce_(0, g, torch.arange(items, device=dev), reduce="amin", include_self=False) if False else None
# segment index inside the group = item index - first item index of the group
fi = torch.full((int(g.max()) + 1,), items, dtype=torch.long, device=dev)
fi.scatter_reduce_(0, g, torch.arange(items, device=dev), reduce="amin")
k = torch.arange(items, device=dev) - fi[g]
# start / end column of every item (gcol lives right after gptr in the plan)
G = (n + R - 1) // R
goff = au(64) + au(G * R * 4) + au((G + 1) * 4)
gcol = plan["buf"][goff:goff + adj.nnz * 4].view(torch.int32)
c0 = gcol[a0.long()].long(); c1 = gcol[(a1 - 1).long()].long()
width = (c1 - c0)
print("item column-span percentiles:", [int(torch.quantile(width.float(), q)) for q in (0.1, 0.5, 0.9, 0.99)])
for name, key in (("segment-major", k * (int(g.max()) + 1) + g), ("start-column", c0 * (int(g.max()) + 1) + g),
                  ("mid-column", ((c0 + c1) // 2) * (int(g.max()) + 1) + g),
                  ("start-column/4096 then group", (c0 // 4096) * (int(g.max()) + 1) + g),
                  ("end-column", c1 * (int(g.max()) + 1) + g)):
    perm = torch.argsort(key, stable=True)
    ig[:items] = g[perm].int(); m0[:items] = a0[perm]; m1[:items] = a1[perm]; io[:items] = o[perm]
    y = adj.propagate(X, 1)
    err = float((y - ref).norm() / ref.norm())
    print(f"{name}: {timeit(lambda: adj.propagate(X, 1)):.3f} ms  rel.diff {err:.1e}")
