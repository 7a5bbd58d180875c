#!/usr/bin/env python3
"""Time a full-batch epoch of the two-layer GCN baseline (model/GCN.py:5-17: GCNConv ->
relu -> GCNConv -> log_softmax; the reference defines the model but no runner builds it,
so the hidden width is the customary 16) through ``sgnn_amd.nn.GCNConv``: two forward
propagations at widths hidden / classes and their two transposed propagations in backward.

    python tools/gcn_epoch.py pubmed reddit
"""
import json
import os
import sys
import time

import torch
import torch.nn.functional as Fn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgnn_amd.graphgen import SHAPES, rmat_undirected_device  # noqa: E402
from sgnn_amd.nn import GCNConv  # noqa: E402

dev = torch.device("cuda:0")
CLASSES = {"pubmed": 3, "reddit": 41}
for wl in sys.argv[1:]:
    n, E, F = SHAPES[wl]
    k = CLASSES.get(wl, 16)
    row, col = rmat_undirected_device(n, E, seed=1, device=dev)
    edge_index = torch.stack((row, col))
    g = torch.Generator(device=dev); g.manual_seed(3)
    x = torch.randn(n, F, device=dev, generator=g)
    y = torch.randint(0, k, (n,), device=dev, generator=g)
    train = torch.rand(n, device=dev, generator=g) < 0.5
    for hidden in (16, 64):
        c1, c2 = GCNConv(F, hidden).to(dev), GCNConv(hidden, k).to(dev)
        opt = torch.optim.Adam(list(c1.parameters()) + list(c2.parameters()), lr=0.01, weight_decay=5e-4)

        def epoch():
            opt.zero_grad()
            h = Fn.relu(c1(x, edge_index))
            out = Fn.log_softmax(c2(h, edge_index), dim=1)
            Fn.nll_loss(out[train], y[train]).backward()
            opt.step()
        for _ in range(3):
            epoch()
        torch.cuda.synchronize()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            epoch()
        torch.cuda.synchronize()
        print(json.dumps({"workload": wl, "n": n, "F": F, "hidden": hidden, "classes": k,
                          "ms_per_epoch": (time.perf_counter() - t0) / reps * 1e3}), flush=True)
