#!/usr/bin/env python3
"""Time the SGC baseline's full-batch epoch (model/SGC.py:19-29: forward through
SGConv(K=2), cross-entropy on the training nodes, backward, Adam step) on a synthetic
graph of a benchmark shape, with the propagation recomputed per epoch (what
torch_geometric's default ``cached=False`` does) and with it cached.

    python tools/sgc_epoch.py pubmed reddit
"""
import json
import os
import sys
import time

import torch
import torch.nn.functional as Fn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgnn_amd.graphgen import SHAPES, rmat_undirected_device  # noqa: E402
from sgnn_amd.nn import SGConv  # noqa: E402

dev = torch.device("cuda:0")
CLASSES = {"pubmed": 3, "reddit": 41}
for wl in sys.argv[1:]:
    n, E, F = SHAPES[wl]
    row, col = rmat_undirected_device(n, E, seed=1, device=dev)
    edge_index = torch.stack((row, col))
    g = torch.Generator(device=dev); g.manual_seed(3)
    x = torch.randn(n, F, device=dev, generator=g)
    y = torch.randint(0, CLASSES.get(wl, 16), (n,), device=dev, generator=g)
    train = torch.rand(n, device=dev, generator=g) < 0.5
    for cached in (False, True):
        conv = SGConv(F, CLASSES.get(wl, 16), K=2, cached=cached).to(dev)
        opt = torch.optim.Adam(conv.parameters(), lr=0.01, weight_decay=5e-4)

        def epoch():
            opt.zero_grad()
            out = Fn.log_softmax(conv(x, edge_index), dim=1)
            loss = Fn.cross_entropy(out[train], y[train])
            loss.backward()
            opt.step()
        for _ in range(3):
            epoch()
        torch.cuda.synchronize()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            epoch()
        torch.cuda.synchronize()
        print(json.dumps({"workload": wl, "n": n, "F": F, "K": 2, "cached": cached,
                          "ms_per_epoch": (time.perf_counter() - t0) / reps * 1e3}), flush=True)
