#!/usr/bin/env python3
"""Benchmark of the sparse-propagation hot path: ``A_hat . X`` on a synthetic
graph of Reddit's shape (n = 232 965, 114 615 892 stored entries, F = 602,
order 1 — the shape the reference's Reddit configuration runs:
config.json:480-503, GNN_tasks.py:76-80).

    python bench.py --gpus N --steps K --warmup W

One step = one call of ``process_data_with_adjacency_high_order`` on a prebuilt
adjacency handle with X resident in HBM.  Rank 0 prints ONE JSON line.
N > 1 (launched by torch.distributed.run, one rank per GPU) splits the same job
over the ranks (strong scaling) on a feature_groups x row_ranks grid: a rank owns a
block of rows of the graph and a slice of the feature columns, and all-gathers the
scaled row blocks of its slice over RCCL before its SpMM.  Feature slices need no
exchange at all, row blocks keep the rows wide; which grid is fastest depends on
the xGMI all-gather rate, so with ``--feature-groups 0`` (default) every grid that
divides N (and, for grids that exchange, both the plain and the overlapped schedule)
is timed for a few untimed steps and the fastest is benchmarked.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E peak, GB/s


def algorithmic_bytes(n, nnz, F):
    """Bytes one SpMM has to move at least once: CSR (int32 col + fp32 val per
    stored entry, rowptr), both dinv vectors, X read once, Y written once."""
    return nnz * 8 + (n + 1) * 4 + 2 * n * 4 + 2 * n * F * 4


def cpu_baseline(n, F, seed, device, sample_edges, repeats=2):
    """Time the CPU oracle (torch sparse-COO restatement of utils.py:81-96) on a
    bounded sample: same node count, feature width and R-MAT parameters as the
    GPU workload, fewer edges so that it finishes in tens of seconds."""
    import torch
    from oracle import sgnn_oracle as orc
    from sgnn_amd.graphgen import rmat_undirected_device

    row, col = rmat_undirected_device(n, sample_edges, seed=seed + 100, device=device)
    idx = torch.stack((row, col)).cpu()
    nnz = int(idx.shape[1])
    adj = torch.sparse_coo_tensor(idx, torch.ones(nnz), (n, n))
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(n, F, generator=g)
    best = float("inf")
    for _ in range(repeats):
        t0 = time.perf_counter()
        orc.propagate_high_order(adj, X, 1)
        best = min(best, time.perf_counter() - t0)
    return {
        "value": (nnz + n) * F / best,
        "unit": "edge·feat/s",
        "cores": int(torch.get_num_threads()),
        "kind": "port",
        "sample": f"reddit-shape R-MAT n={n} nnz(A)={nnz} F={F} order=1, one call = {best:.2f} s "
                  f"(best of {repeats}); full op sequence incl. A+I and degree, as the reference redoes them per call",
        "seconds": best,
    }


def load_traffic(workload):
    """HBM bytes per SpMM launch from the committed PMC profile, if one matches."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
        return float(d[workload]["hbm_fabric_bytes"])
    except Exception:
        return None


def copy_ceiling(device, mib=2048, reps=10):
    """Read + write GB/s of a device-to-device copy of ``mib`` MiB: the streaming rate this
    card delivers, to set beside the vendor peak."""
    import torch
    src = torch.empty(mib << 18, dtype=torch.float32, device=device).normal_()
    dst = torch.empty_like(src)
    for _ in range(3):
        dst.copy_(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * src.numel() * 4 * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def sgnn_epoch(adj, X, device, max_iter, seed):
    """One SGNN "epoch" on the benchmark graph: a backward sweep plus a forward sweep over two
    decoupled layers with a label head (model/SGNN.py:555-559) at the reference's Reddit
    hyper-parameters (config.json:480-503: 128/64 neurons, leaky_relu=0.2, lr 1e-4, batch 512,
    order 1, eta 1000, 41 classes) except for ``max_iter`` optimiser iterations per layer run
    instead of 10 000, so that the default bench stays short; the per-iteration time is the
    reference's own figure of merit (GNN_tasks.py:133-140)."""
    import torch
    from sgnn_amd.stacked import EmbeddingLayer, LayerParam, SupervisedStackedGNN, get_activation
    n, F = X.shape
    g = torch.Generator(device=device)
    g.manual_seed(seed + 2)
    labels = torch.randint(0, 41, (n,), device=device, generator=g)
    train = torch.rand(n, device=device, generator=g) < 0.66
    model = SupervisedStackedGNN.__new__(SupervisedStackedGNN)     # around the resident handle; its constructor takes scipy
    model.device, model.adjacency_tensor, model.content = device, adj, X
    model.labels, model.training_mask, model.BP_count, model.eta = labels, train, 1, 1000
    model.labels_host = model.metric_func = None
    model.gnns, dim = [], F
    for neurons in (128, 64):
        p = LayerParam(neurons, inner_act=get_activation("leaky_relu=0.2"), act=get_activation("linear"),
                       gnn_type=LayerParam.EGCN, learning_rate=1e-4, max_iter=max_iter, lam=1e-6, batch_size=512)
        model.gnns.append(EmbeddingLayer(adj, labels, train, dim, neurons, 41, p, device))
        dim = neurons
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    inputs, _ = model._forward_sweep(False)
    torch.cuda.synchronize()
    t_first = time.perf_counter() - t0
    t0 = time.perf_counter()
    model._backward_sweep(inputs)
    model._forward_sweep(True)
    torch.cuda.synchronize()
    t_epoch = time.perf_counter() - t0
    return {"task": "classification, 2 layers (128, 64), batch 512", "max_iter": max_iter,
            "first_forward_sweep_s": t_first, "epoch_s": t_epoch, "optimiser_iterations": 4 * max_iter,
            "s_per_iteration": t_epoch / (4 * max_iter),
            "epoch_s_at_10000_iterations_per_layer_run": t_epoch / max_iter * 10000}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="reddit", help="reddit | reddit128 | products | rmat10m | pubmed")
    ap.add_argument("--order", type=int, default=1)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-degree", type=int, default=None, help="clip hub degree (real Reddit's is ~21.7k)")
    ap.add_argument("--slices", type=int, default=2,
                    help="N > 1: feature slices of the exchange/SpMM pipeline (1 = no overlap)")
    ap.add_argument("--feature-groups", type=int, default=0,
                    help="N > 1: ranks per row block that split the feature columns (0 = time every divisor of N)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--epoch-iters", type=int, default=5000,
                    help="N = 1, reddit: optimiser iterations per layer run of the SGNN epoch timed after the "
                         "SpMM steps (0 = skip; the reference's Reddit config uses 10000)")
    ap.add_argument("--cpu-sample-edges", type=int, default=10_000_000)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from sgnn_amd import NormAdj
    from sgnn_amd.graphgen import SHAPES, rmat_undirected_device

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    # Rehearsal on a one-GPU box: SGNN_BENCH_REHEARSAL=1 puts every rank on cuda:0 and
    # exchanges through gloo (host-staged), to exercise the N > 1 code path without RCCL.
    rehearsal = world > 1 and os.environ.get("SGNN_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    n, E, F = SHAPES[args.workload]
    if rank == 0:
        print(f"[bench] {args.workload}: drawing the graph on {world} rank(s)", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    row, col = rmat_undirected_device(n, E, seed=args.seed, device=device, max_degree=args.max_degree)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t0
    g = torch.Generator(device=device)
    g.manual_seed(args.seed + 1)
    X = torch.randn(n, F, device=device, generator=g)

    extra = {}
    if world == 1:
        t0 = time.perf_counter()
        adj = NormAdj.from_coo(row, col, None, n, device)
        torch.cuda.synchronize()
        extra["adjacency_build_ms"] = (time.perf_counter() - t0) * 1e3
        del row, col
        nnz = adj.nnz
        step = lambda: adj.propagate(X, args.order)          # noqa: E731
        max_deg = int((adj.rowptr[1:] - adj.rowptr[:-1]).max().item())
    else:
        from sgnn_amd.dist import PartitionedAdj, grid_layout, make_row_groups

        def build_grid(fg):
            """One candidate grid: a list of (step function, info), one entry per exchange
            schedule that reproduces the plain one on this machine."""
            row_rank, row_world, fgroup, f0, f1 = grid_layout(rank, world, fg, F)
            group = make_row_groups(world, fg)
            t0 = time.perf_counter()
            padj = PartitionedAdj.from_coo(row, col, None, n, device, row_rank, row_world, group=group)
            torch.cuda.synchronize()
            build_ms = (time.perf_counter() - t0) * 1e3
            Xl = X[padj.rows.to(device)][:, f0:f1].contiguous()
            schedules = [1]
            want = max(1, args.slices)
            if row_world > 1 and want > 1:
                # the overlapped (feature-sliced) exchange competes only if it reproduces
                # the plain one; every rank of the job takes the same decision
                ok = torch.ones(1, device=device)
                try:
                    a = padj.propagate(Xl, args.order, slices=1)
                    b = padj.propagate(Xl, args.order, slices=want)
                    if not torch.allclose(a, b, rtol=1e-4, atol=1e-5):
                        ok.zero_()
                    del a, b
                except Exception as exc:          # noqa: BLE001
                    print(f"[rank {rank}] sliced schedule failed: {exc}", file=sys.stderr, flush=True)
                    ok.zero_()
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if ok.item() != 0:
                    schedules.append(want)
            out = []
            for sl in schedules:
                info = {"feature_groups": fg, "row_ranks": row_world, "features_per_rank": f1 - f0,
                        "exchange_slices": sl, "adjacency_build_ms": build_ms, "nnz": padj.global_nnz,
                        "max_deg": padj.global_max_degree}
                out.append(((lambda sl=sl: padj.propagate(Xl, args.order, slices=sl)), info))
            return out

        def probe(step_fn, reps=3):
            for _ in range(2):
                step_fn()
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(reps):
                step_fn()
            torch.cuda.synchronize()
            t = torch.tensor([(time.perf_counter() - t0) / reps], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

        if args.feature_groups > 0:
            candidates = [args.feature_groups]
        else:
            candidates = [g for g in range(1, world + 1) if world % g == 0 and g <= F]
        best = None
        probed = {}
        n_options = 0
        for fg in candidates:
            for step_fn, info in build_grid(fg):
                n_options += 1
                t = probe(step_fn)
                tag = f"{fg}x{world // fg}" + (f"/s{info['exchange_slices']}" if info["exchange_slices"] > 1 else "")
                probed[tag] = round(t * 1e3, 3)
                if rank == 0:
                    print(f"[bench] {fg} feature slices x {world // fg} row blocks, exchange in "
                          f"{info['exchange_slices']} slice(s): built in {info['adjacency_build_ms']:.0f} ms, "
                          f"probe {t * 1e3:.2f} ms/step", file=sys.stderr, flush=True)
                if best is None or t < best[0]:
                    best = (t, step_fn, info)
                del step_fn
            torch.cuda.empty_cache()
        _, step, info = best
        del row, col, X
        nnz, max_deg = info.pop("nnz"), info.pop("max_deg")
        extra.update(info)
        if n_options > 1:
            extra["grid_probe_ms"] = probed

    lib = __import__("sgnn_amd._lib", fromlist=["load"]).load()
    R = lib.sgnn_rowgroup_config(extra.get("features_per_rank", F))
    kernel_path = f"row-group SpMM, R={R}" if R else "plain CSR SpMM"
    tilegroup = world == 1 and bool(getattr(adj, "_tilegroup_plan", lambda F: None)(F))
    if tilegroup:
        kernel_path = "tile-group SpMM (512-row groups, fp16 hi/lo planes on v_mfma_f32_32x32x16_f16, fp32 accumulation)"
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([wall], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    ev_ms = ev0.elapsed_time(ev1) / args.steps              # HIP events on the launch stream

    copy_gbps = None
    if rank == 0 and world == 1:
        try:
            copy_gbps = copy_ceiling(device)
        except Exception as exc:
            print(f"[bench] copy ceiling not measured: {exc}", file=sys.stderr, flush=True)
    epoch = None
    if world == 1 and args.workload == "reddit" and args.epoch_iters > 0:   # outside the timed SpMM region
        try:
            epoch = sgnn_epoch(adj, X, device, args.epoch_iters, args.seed)
        except Exception as exc:                  # the SpMM line must come out whatever happens here
            epoch = {"error": f"{type(exc).__name__}: {exc}"}
    if rank == 0:
        ms_per_step = wall * 1e3 / args.steps
        units = nnz * F * args.order                          # (E + n) . F per hop
        b_alg = algorithmic_bytes(n, nnz, F) * args.order
        achieved = b_alg / (ev_ms * 1e-3) / 1e9
        out = {
            "metric": "Â·X SpMM edge·feat/s (Reddit F=602)" if args.workload == "reddit"
                      else f"Â·X SpMM edge·feat/s ({args.workload})",
            "value": units / (wall / args.steps),
            "unit": "edge·feat/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}-shape symmetric R-MAT(0.57,0.19,0.19,0.05): n={n}, "
                            f"nnz(A+I)={nnz}, F={F}, order={args.order}, max_degree={max_deg}, unit weights",
                "rows": "one GPU" if world == 1 else
                        f"{extra['feature_groups']} feature slices x {extra['row_ranks']} row blocks over {world} GPUs, "
                        "all-gather of scaled X blocks inside a feature slice",
                "graph_gen_s": round(t_gen, 2),
                "kernel": kernel_path,
                **{k: (round(v, 2) if isinstance(v, float) else v) for k, v in extra.items()},
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": load_traffic(args.workload) if world == 1 else None,
                "algorithmic_bytes": b_alg,
                "kernel_ms": ev_ms,                           # HIP events around the K steps / K
                "kernels_in_step": "k_tg_amax + k_tg_split + k_tg_spmm + k_tg_combine" if tilegroup else
                                   ("k_row_scale + k_rg_spmm + k_rg_combine"
                                    + (" (rocprof, profiles/r01_reddit_f602_rowgroup_x4.json: 0.19 + 9.65 + 0.58 ms)"
                                       if args.workload == "reddit" and world == 1 else "")) if R
                                   else "k_spmm (+ k_row_scale)",
                "gather_bytes_no_reuse": nnz * F * 4 * args.order,
            },
        }
        roof = out["roofline"]
        if copy_gbps:                                         # what a streaming copy reaches on this card
            roof["copy_ceiling"] = copy_gbps
            roof["frac_of_copy"] = achieved / copy_gbps
        if roof["traffic"]:                                   # bytes the L2s fetched or wrote, per launch
            roof["fabric_GBps"] = roof["traffic"] / (ev_ms * 1e-3) / 1e9
        if epoch:
            out["sgnn_epoch"] = {k: (round(v, 6) if isinstance(v, float) else v) for k, v in epoch.items()}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(n, F, args.seed, device, args.cpu_sample_edges)
            except Exception as exc:
                out["cpu_baseline"] = {"error": f"{type(exc).__name__}: {exc}"}
        print(json.dumps(out, ensure_ascii=False), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
