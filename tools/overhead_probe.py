import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgnn_amd import NormAdj
from sgnn_amd.graphgen import SHAPES, rmat_undirected_device
dev = torch.device("cuda:0")
for wl in sys.argv[1:]:
    n, E, F = SHAPES[wl]
    row, col = rmat_undirected_device(n, E, seed=1, device=dev)
    adj = NormAdj.from_coo(row, col, None, n, dev)
    X = torch.randn(n, F, device=dev)
    for flag in (True, False):
        adj.use_rowgroup = flag
        for _ in range(3):
            adj.propagate(X, 1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record()
        for _ in range(20):
            adj.propagate(X, 1)
        e1.record(); t_issue = time.perf_counter() - t0
        torch.cuda.synchronize(); t_all = time.perf_counter() - t0
        plan = adj._plans.get(8) or adj._plans.get(16)
        print(f"{wl} rowgroup={flag}: events {e0.elapsed_time(e1)/20:.3f} ms/step, host issue {t_issue/20*1e3:.3f} ms/step, wall {t_all/20*1e3:.3f}"
              + (f" merged={plan['merged']} items={plan['items']} slots={plan['slots']} long={plan['n_long']} R={plan['R']}" if plan and flag else ""), flush=True)
