#!/usr/bin/env python3
"""Time one SGNN "epoch" on the Reddit-shape synthetic graph: one backward sweep plus
one forward sweep over both decoupled layers (model/SGNN.py:555-559), with the
reference's Reddit hyper-parameters (config.json:480-503: 128/64 neurons,
leaky_relu=0.2 inner activation, lr 1e-4, max_iter 10000, batch 512, order 1,
eta 1000).  Also reports the reference's own figure of merit, seconds per optimiser
iteration (GNN_tasks.py:133-140), and how much of the epoch is propagation.

    python tools/sgnn_epoch.py [--max-iter 10000] [--task classification|clustering]

``--task clustering`` times the same sweeps for the auto-encoder stack with the
reference's Reddit clustering hyper-parameters (config.json, SGNN/Clustering/Reddit:
128/64 neurons, mask_rate 0.2, lr 1e-4, batch 512, eta 10, lam 1e-6; the runner does not
forward ``order``, so layers propagate once — GNN_tasks.py:325-356): every optimiser
iteration samples a 512 x 512 sub-adjacency from the CSR on the GPU.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgnn_amd import NormAdj, utils as prop  # noqa: E402
from sgnn_amd.graphgen import SHAPES, rmat_undirected_device  # noqa: E402
from sgnn_amd.stacked import (EmbeddingLayer, GAELayer, LayerParam, StackedGNN, SupervisedStackedGNN,  # noqa: E402
                              get_activation)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--max-iter", type=int, default=10000)
    ap.add_argument("--workload", default="reddit")
    ap.add_argument("--task", default="classification", choices=["classification", "clustering"])
    ap.add_argument("--no-graph", action="store_true", help="issue every optimiser iteration from Python")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    n, E, F = SHAPES[args.workload]
    row, col = rmat_undirected_device(n, E, seed=1, device=dev)
    adj = NormAdj.from_coo(row, col, None, n, dev)
    del row, col
    g = torch.Generator(device=dev); g.manual_seed(2)
    X = torch.randn(n, F, device=dev, generator=g)
    labels = torch.randint(0, 41, (n,), device=dev, generator=g)
    train = torch.rand(n, device=dev, generator=g) < 0.66
    clustering = args.task == "clustering"
    if clustering:
        layers = [LayerParam(nn_, inner_act=get_activation("relu"), act=get_activation("linear"),
                             gnn_type=LayerParam.GAE, mask_rate=0.2, learning_rate=1e-4, max_iter=args.max_iter,
                             lam=1e-6, batch_size=512, use_graph=not args.no_graph) for nn_ in (128, 64)]
    else:
        layers = [LayerParam(nn_, inner_act=get_activation("leaky_relu=0.2"), act=get_activation("linear"),
                             gnn_type=LayerParam.EGCN, learning_rate=1e-4, max_iter=args.max_iter, lam=1e-6,
                             batch_size=512, use_graph=not args.no_graph) for nn_ in (128, 64)]
    # assemble the stacked model around the prebuilt handle (its constructor takes scipy; skip that path)
    cls = StackedGNN if clustering else SupervisedStackedGNN
    model = cls.__new__(cls)
    model.device, model.adjacency_tensor, model.content = dev, adj, X
    model.labels, model.training_mask, model.BP_count = labels, train, 1
    model.eta = 10 if clustering else 1000
    model.labels_host = model.metric_func = None
    model.gnns, dim = [], F
    for p in layers:
        model.gnns.append(GAELayer(adj, dim, p.neurons, p, dev) if clustering
                          else EmbeddingLayer(adj, labels, train, dim, p.neurons, 41, p, dev))
        dim = p.neurons

    prop_ms = []
    orig = prop.process_data_with_adjacency_high_order

    def timed(a, x, d, order=1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); y = orig(a, x, d, order=order); e1.record()
        prop_ms.append((e0, e1))
        return y
    import sgnn_amd.stacked as st
    st.prop.process_data_with_adjacency_high_order = timed

    torch.cuda.synchronize(); t0 = time.perf_counter()
    inputs, emb = model._forward_sweep(False)
    torch.cuda.synchronize(); t_fwd0 = time.perf_counter() - t0
    n0 = len(prop_ms)
    t0 = time.perf_counter()
    model._backward_sweep(inputs)
    inputs, emb = model._forward_sweep(True)
    torch.cuda.synchronize(); t_epoch = time.perf_counter() - t0
    ms_all = [a.elapsed_time(b) for a, b in prop_ms]
    iters_epoch = 4 * args.max_iter
    out = {"workload": args.workload, "task": args.task, "hip_graph": not args.no_graph, "max_iter": args.max_iter,
           "first_forward_sweep_s": round(t_fwd0, 3), "epoch_s": round(t_epoch, 3),
           "s_per_iteration": t_epoch / iters_epoch,
           "propagations_first_sweep": n0, "propagations_in_epoch": len(ms_all) - n0,
           "propagation_ms_first_sweep": [round(v, 2) for v in ms_all[:n0]],
           "propagation_ms_in_epoch": [round(v, 2) for v in ms_all[n0:]],
           "propagation_share_of_epoch": sum(ms_all[n0:]) / 1e3 / t_epoch}
    if clustering:
        idx = torch.randperm(n, device=dev)[:512]
        adj.sample_block(idx)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            adj.sample_block(idx)
        e1.record(); torch.cuda.synchronize()
        out["sample_block_512_ms"] = e0.elapsed_time(e1) / 200
    print(json.dumps(out))


if __name__ == "__main__":
    main()
