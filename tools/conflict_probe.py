#!/usr/bin/env python3
"""How much of the tile-group kernel's time are LDS bank conflicts of the transposed reads?
(run on the GPU box).  Times one Reddit-shape hop as planned, then with every step's row
offsets overwritten by a conflict-free pattern (column k of a step reads slot k: the four rows
of every pass then sit in four different bank quarters).  The second run's result is wrong;
only its time is of interest."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgnn_amd import NormAdj, _lib  # noqa: E402
from sgnn_amd.graphgen import SHAPES, rmat_undirected_device  # noqa: E402

dev = torch.device("cuda:0")
n, E, F = SHAPES["reddit"]
row, col = rmat_undirected_device(n, E, seed=1, device=dev)
adj = NormAdj.from_coo(row, col, None, n, dev)
del row, col
adj.use_tilegroup = True
X = torch.randn(n, F, device=dev)
lib = _lib.load()


def timeit(reps=20):
    for _ in range(3):
        adj.propagate(X, 1)
    torch.cuda.synchronize()
    lib.sgnn_tune_set(b"time_kernel", 1)
    for _ in range(reps):
        adj.propagate(X, 1)
    torch.cuda.synchronize()
    ms, k = _lib.kernel_time_ms()
    lib.sgnn_tune_set(b"time_kernel", 0)
    return ms / k


print("as planned: kernel %.3f ms" % timeit(), flush=True)
p = adj._tg
al = lambda x: (x + 255) // 256 * 256
G = (n + 511) // 512
stages, steps = p["stages"], p["steps"]
off = al(64) + al(G * 512 * 4) + al((G + 1) * 4) + al((stages * 128 + 384) * 4)
step0 = p["buf"][off: off + (stages * 16 + 1) * 4].view(torch.int32)
assert int(step0[0]) == 0 and int(step0[-1]) == steps and bool((step0[1:] >= step0[:-1]).all()), "plan layout changed"
off += al((stages * 16 + 1 + 128) * 4)
slots = p["buf"][off: off + steps * 32].view(torch.int16).view(steps, 16)
# entry 2 * (4 * (k >> 3) + (k & 3)) + ((k >> 2) & 1) of a step holds the offset of its column k
k = torch.arange(16, device=dev)
spos = 2 * (4 * (k >> 3) + (k & 3)) + ((k >> 2) & 1)
pat = torch.empty(16, dtype=torch.int16, device=dev)
pat[spos] = ((k >> 2) * 2048 + (k & 3) * 64).to(torch.int16)
hist = torch.bincount(((slots.view(-1).int() & 0xFFFF) >> 6) & 3, minlength=4)
print("bank classes of the planned offsets:", hist.tolist(), flush=True)
slots[:] = pat
print("conflict-free offsets: kernel %.3f ms (result wrong, time only)" % timeit(), flush=True)
