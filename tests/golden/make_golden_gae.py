#!/usr/bin/env python3
"""End-to-end golden for the clustering branch: the REFERENCE's StackedGNN with graph
auto-encoder layers (model/SGNN.py:120-251, 470-564) run on the CPU on the planted
graph of make_golden_e2e.py.

    python tests/golden/make_golden_gae.py [/root/reference]

Writes tests/golden/e2e_gae_planted.npz: hyper-parameters, the reference's final
embeddings for seeds 0..2 and their k-means ACC / NMI (k-means with fixed seeds, the
assignment step of ACC solved with scipy because the munkres package the reference
imports is not installed).

The auto-encoder layer of the reference logs through ``self.logger``, which its
constructor leaves at None (model/SGNN.py:127-131 does not forward a logger); the
script sets that attribute on the constructed layers, nothing else is touched.
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
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"

from make_golden_e2e import planted_graph  # noqa: E402
from sgnn_amd.stacked import k_means  # noqa: E402  (host-side sklearn/scipy metric code only)

CONFIG = {"eta": 10, "BP_count": 2, "lam": 1e-6, "mask_rate": 0.2,
          "layers": [{"neurons": 32, "inner_act": "linear", "activation": "leaky_relu=0.2",
                      "batch_size": 256, "learning_rate": 0.005, "max_iter": 150},
                     {"neurons": 16, "inner_act": "linear", "activation": "linear",
                      "batch_size": 256, "learning_rate": 0.005, "max_iter": 150}]}


def main():
    sys.dont_write_bytecode = True
    stub = types.ModuleType("munkres"); stub.Munkres = object
    sys.modules.setdefault("munkres", stub)
    sys.path.insert(0, REF)
    os.chdir(tempfile.mkdtemp())
    with open("global_settings.json", "w") as f:           # the reference re-reads this file (utils.py:349-351)
        json.dump({"ddp": False, "logger": {}}, f)
    import utils as ref_utils
    from model.SGNN import Func, LayerParam, StackedGNN

    def act(spec):
        if "linear" in spec:
            return Func(None)
        if "leaky" in spec:
            return Func(torch.nn.functional.leaky_relu, negative_slope=float(spec.split("=")[1]))
        raise ValueError(spec)

    row, col, X, y, _ = planted_graph()
    n = X.shape[0]
    A = sp.coo_matrix((np.ones(row.size, np.float32), (row, col)), shape=(n, n)).tocsr()
    log = logging.getLogger("golden"); log.addHandler(logging.NullHandler()); log.setLevel(logging.CRITICAL)
    out = dict(config=np.array(json.dumps(CONFIG)))
    accs, nmis = [], []
    for seed in range(3):
        ref_utils.set_seed(seed)
        layers = [LayerParam(l["neurons"], inner_act=act(l["inner_act"]), act=act(l["activation"]),
                             gnn_type=LayerParam.GAE, mask_rate=CONFIG["mask_rate"], lam=CONFIG["lam"],
                             max_iter=l["max_iter"], learning_rate=l["learning_rate"], batch_size=l["batch_size"])
                  for l in CONFIG["layers"]]
        model = StackedGNN(X, A, layers, overlooked_rates=None, BP_count=CONFIG["BP_count"], eta=CONFIG["eta"],
                           device=torch.device("cpu"), labels=None, metric_func=None, logger=log)
        for g in model.gnns:
            g.logger = log
        emb = model.run().detach().cpu().numpy().astype(np.float32)
        acc, nmi = k_means(emb, 4, y, replicates=3, seed=0)
        accs.append(acc); nmis.append(nmi)
        out[f"emb_seed{seed}"] = emb
        print(f"seed {seed}: reference embedding k-means ACC {acc:.4f} NMI {nmi:.4f}")
    acc0, nmi0 = k_means(X, 4, y, replicates=3, seed=0)
    print(f"raw features: ACC {acc0:.4f} NMI {nmi0:.4f}")
    out["ref_acc"], out["ref_nmi"] = np.array(accs), np.array(nmis)
    out["raw_acc"], out["raw_nmi"] = np.float64(acc0), np.float64(nmi0)
    np.savez_compressed(os.path.join(HERE, "e2e_gae_planted.npz"), **out)


if __name__ == "__main__":
    main()
