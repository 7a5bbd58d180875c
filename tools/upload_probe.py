#!/usr/bin/env python3
"""Host-to-device costs at the drop-in boundary: the reference's call sites hand over a
scipy matrix once per graph (utils.py:54-62) and call X.to(device) per propagation
(model/SGNN.py:404).  Times both for the benchmark shape, next to the resident step."""
import json, os, sys, time
import numpy as np, scipy.sparse as sp, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgnn_amd import utils
from sgnn_amd.graphgen import SHAPES, rmat_undirected_device
dev = torch.device("cuda:0")
n, E, F = SHAPES[sys.argv[1] if len(sys.argv) > 1 else "reddit"]
row, col = rmat_undirected_device(n, E, seed=1, device=dev)
A = sp.csr_matrix((np.ones(row.numel(), np.float32), (row.cpu().numpy(), col.cpu().numpy())), shape=(n, n))
del row, col
Xh = torch.randn(n, F)
torch.cuda.synchronize()
t0 = time.perf_counter(); adj = utils.csr_to_sparse_Tensor(A, dev); torch.cuda.synchronize(); t_adj = time.perf_counter() - t0
utils.process_data_with_adjacency_high_order(adj, Xh.to(dev), dev, order=1); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    Y = utils.process_data_with_adjacency_high_order(adj, Xh.to(dev), dev, order=1)
torch.cuda.synchronize(); t_host = (time.perf_counter() - t0) / 5
Xd = Xh.to(dev)
t0 = time.perf_counter()
for _ in range(5):
    Y = utils.process_data_with_adjacency_high_order(adj, Xd, dev, order=1)
torch.cuda.synchronize(); t_res = (time.perf_counter() - t0) / 5
Xp = Xh.pin_memory()
t0 = time.perf_counter()
for _ in range(5):
    Y = utils.process_data_with_adjacency_high_order(adj, Xp.to(dev, non_blocking=True), dev, order=1)
torch.cuda.synchronize(); t_pin = (time.perf_counter() - t0) / 5
units = adj.nnz * F
print(json.dumps({"scipy_to_device_adjacency_s": t_adj, "step_ms_X_resident": t_res * 1e3,
                  "step_ms_X_from_pageable_host": t_host * 1e3, "step_ms_X_from_pinned_host": t_pin * 1e3,
                  "edge_feat_per_s_resident": units / t_res, "edge_feat_per_s_pcie_pageable": units / t_host,
                  "edge_feat_per_s_pcie_pinned": units / t_pin}))
