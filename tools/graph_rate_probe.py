#!/usr/bin/env python3
"""What a HIP graph replay costs per node on this card: graphs of 1..80 trivial kernels,
time per replay and per node (GPU side and host issue).  The per-node figure (~1.8 us) is
the floor under an optimiser iteration of many tiny kernels."""
import time, torch
dev = torch.device("cuda:0")
x = torch.zeros(4096, device=dev)
for n_nodes in (1, 5, 10, 20, 40, 80):
    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            x.add_(1.0)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n_nodes):
            x.add_(1.0)
    for _ in range(50):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000):
        g.replay()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print(f"{n_nodes:3d} nodes: {t / 2000 * 1e6:8.1f} us per replay ({t / 2000 / n_nodes * 1e6:6.2f} us per node), host issue {t_issue / 2000 * 1e6:7.1f} us", flush=True)
