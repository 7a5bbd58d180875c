import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgnn_amd import NormAdj, _lib
from sgnn_amd.graphgen import SHAPES, rmat_undirected_device
dev = torch.device("cuda:0"); lib = _lib.load()
wl = sys.argv[1]; n, E, F = SHAPES[wl]
row, col = rmat_undirected_device(n, E, seed=1, device=dev)
adj = NormAdj.from_coo(row, col, None, n, dev); del row, col
X = torch.randn(n, F, device=dev)
adj.use_rowgroup = False; ref = adj.propagate(X, 1); adj.use_rowgroup = True
Rs = [int(s) for s in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
Os = [int(s) for s in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0]
for seg, Rv, oc in [(int(s), r, o) for s in sys.argv[2].split(",") for r in Rs for o in Os]:
    lib.sgnn_tune_set(b"rg_seg", seg); lib.sgnn_tune_set(b"rg_R", Rv); lib.sgnn_tune_set(b"rg_order", oc); adj._plans = {}
    y = adj.propagate(X, 1); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): adj.propagate(X, 1)
    e1.record(); torch.cuda.synchronize()
    p = list(adj._plans.values())[0]
    print(f"{wl} seg={seg} order={oc} R={p['R']}: {e0.elapsed_time(e1)/5:.3f} ms items={p['items']} slots={p['slots']} rel.diff {float((y-ref).norm()/ref.norm()):.1e}", flush=True)
