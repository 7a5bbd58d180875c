#!/usr/bin/env python3
"""End-to-end golden: the REFERENCE's SupervisedStackedGNN run on the CPU on a small
labelled synthetic graph (no real labelled dataset is readable here: the reference's
Planetoid files are pickles, which are not opened).

    python tests/golden/make_golden_e2e.py [/root/reference]

Writes tests/golden/e2e_sgnn_planted.npz: the graph, features, labels, masks, the
hyper-parameters, and the reference's predictions / test accuracy for seeds 0..2.
"""
import json
import logging
import os
import sys
import tempfile
import types

import numpy as np
import scipy.sparse as sp
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"


def planted_graph(n=1200, classes=4, F=48, p_in=0.02, p_out=0.002, seed=7):
    rng = np.random.default_rng(seed)
    y = rng.integers(0, classes, n)
    iu = np.triu_indices(n, 1)
    same = y[iu[0]] == y[iu[1]]
    keep = rng.random(iu[0].size) < np.where(same, p_in, p_out)
    r, c = iu[0][keep], iu[1][keep]
    row, col = np.concatenate((r, c)), np.concatenate((c, r))
    centers = rng.choice([-1.0, 1.0], size=(classes, F))
    X = (0.35 * centers[y] + rng.standard_normal((n, F))).astype(np.float32)
    train = np.zeros(n, bool)
    for k in range(classes):
        train[rng.choice(np.flatnonzero(y == k), 20, replace=False)] = True
    return row, col, X, y, train


CONFIG = {"eta": 100, "BP_count": 2, "lam": 0.001,
          "layers": [{"neurons": 32, "inner_act": "linear", "activation": "leaky_relu=0.2", "order": 1,
                      "batch_size": 1200, "learning_rate": 0.01, "max_iter": 60},
                     {"neurons": 16, "inner_act": "linear", "activation": "linear", "order": 2,
                      "batch_size": 80, "learning_rate": 0.01, "max_iter": 60}]}


def main():
    sys.dont_write_bytecode = True
    stub = types.ModuleType("munkres"); stub.Munkres = object
    sys.modules.setdefault("munkres", stub)
    sys.path.insert(0, REF)
    work = tempfile.mkdtemp()
    os.chdir(work)
    with open("global_settings.json", "w") as f:           # the reference re-reads this file (utils.py:349-351)
        json.dump({"ddp": False, "logger": {}}, f)
    import utils as ref_utils
    from model.SGNN import Func, LayerParam, SupervisedStackedGNN

    def act(spec):                                          # GNN_tasks.py:387-404, without importing torch_geometric
        if "linear" in spec:
            return Func(None)
        if "leaky" in spec:
            return Func(torch.nn.functional.leaky_relu, negative_slope=float(spec.split("=")[1]))
        if spec == "relu":
            return Func(torch.nn.functional.relu)
        raise ValueError(spec)

    row, col, X, y, train = planted_graph()
    n = X.shape[0]
    A = sp.coo_matrix((np.ones(row.size, np.float32), (row, col)), shape=(n, n)).tocsc()
    test = ~train
    log = logging.getLogger("golden"); log.addHandler(logging.NullHandler())
    out = dict(row=row.astype(np.int32), col=col.astype(np.int32), X=X, y=y.astype(np.int64), train=train,
               config=np.array(json.dumps(CONFIG)))
    accs = []
    for seed in range(3):
        ref_utils.set_seed(seed)
        layers = [LayerParam(l["neurons"], inner_act=act(l["inner_act"]), act=act(l["activation"]),
                             gnn_type=LayerParam.EGCN, learning_rate=l["learning_rate"], order=l["order"],
                             max_iter=l["max_iter"], lam=CONFIG["lam"], batch_size=l["batch_size"])
                  for l in CONFIG["layers"]]
        model = SupervisedStackedGNN(X, A, layers, training_mask=train, val_mask=test, BP_count=CONFIG["BP_count"],
                                     eta=CONFIG["eta"], device=torch.device("cpu"), labels=y,
                                     metric_func=ref_utils.classification, logger=log)
        pred = model.run()
        acc = float((pred[test] == y[test]).mean())
        accs.append(acc)
        out[f"pred_seed{seed}"] = pred.astype(np.int64)
        print(f"seed {seed}: reference test accuracy {acc:.4f}")
    out["ref_test_acc"] = np.array(accs)
    np.savez_compressed(os.path.join(HERE, "e2e_sgnn_planted.npz"), **out)


if __name__ == "__main__":
    main()
