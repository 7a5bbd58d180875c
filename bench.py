#!/usr/bin/env python3
"""Benchmark of the sparse-propagation hot path: ``A_hat . X`` on a synthetic
graph of Reddit's shape (n = 232 965, 114 615 892 stored entries, F = 602,
order 1 — the shape the reference's Reddit configuration runs:
config.json:480-503, GNN_tasks.py:76-80).

    python bench.py --gpus N --steps K --warmup W

One step = one call of ``process_data_with_adjacency_high_order`` on a prebuilt
adjacency handle with X resident in HBM.  Rank 0 prints ONE JSON line.

N = 1 also reports, outside the timed region and under keys of their own: the same
step on the other benchmark shapes (hub-clipped and weight-2 Reddit, Reddit at F = 128,
ogbn-products, R-MAT 10 M, PubMed), the CPU restatement of utils.py:81-96 on the SAME
graph (median of 3 calls, checked against the GPU result), and one SGNN epoch of the
classification stack and one of the clustering (auto-encoder) stack at the reference's
10 000 optimiser iterations per layer run.

N > 1 (launched by torch.distributed.run, one rank per GPU) splits the same job
over the ranks (strong scaling).  Two schedules that serve the same caller (each rank hands
in its rows of X and gets its rows of the result) are probed for a few steps and the faster
one is benchmarked:
* the row partition: every rank owns a block of rows of the graph, multiplies its
  own-columns block while the other ranks' rows of the operand travel over RCCL (as fp16
  planes, one all-gather per 128-feature slice, where the matrix-core kernel applies; as
  fp32 rows otherwise) and its remote-columns block as they land (``--slices k`` lets the
  older feature-sliced pipeline compete inside it);
* the feature split: every rank holds the whole graph and propagates F / N columns; two
  all-to-alls per call carry the caller's rows to column slices and back, no exchange
  between hops (``--no-feature-split`` keeps it out).
``config.rows`` / ``config.overlap`` / ``config.collective`` name what ran,
``config.row_partition_ms_per_step`` keeps the row partition's probe time when the other
schedule won.  Grids that split rows and features (feature_groups x row_ranks, no
all-to-all: X is assumed to live in feature slices already) are timed too and reported as
``grid_probe_ms`` extras only.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E peak, GB/s


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def algorithmic_bytes(n, nnz, F):
    """Bytes one SpMM has to move at least once: CSR (int32 col + fp32 val per
    stored entry, rowptr), both dinv vectors, X read once, Y written once."""
    return nnz * 8 + (n + 1) * 4 + 2 * n * 4 + 2 * n * F * 4


def kernel_of(adj, F):
    """(label, kernels of one hop) of the path ``adj.propagate`` takes at width F."""
    lib = __import__("sgnn_amd._lib", fromlist=["load"]).load()
    if adj._tilegroup_plan(F):
        return ("tile-group SpMM (512-row groups share fetched rows through LDS; fp16 hi/lo planes of X, one scale per "
                "feature column, on v_mfma_f32_32x32x16_f16, fp32 accumulation)",
                "k_tg_amax over every 32nd row + k_tg_state_init + k_tg_split + the exact max / split passes behind them "
                "(return at once unless the sample missed an outlier) (first hop only; later hops of a chain read the planes "
                "the epilogue before them wrote) + k_tg_spmm + k_tg_combine + guarded plain-CSR fall-back (returns at once)",
                "k_tg_spmm")
    R = lib.sgnn_rowgroup_config_for(adj.n, adj.nnz, F) if adj.use_rowgroup else 0
    if R:
        return f"row-group SpMM, R={R}", "k_row_scale + k_rg_spmm + k_rg_combine", "k_rg_spmm"
    return "plain CSR SpMM", "k_spmm + k_spmm_combine", "k_spmm"


def timed_steps(step, steps, warmup, device, world=1, dist=None):
    """W untimed steps, then K steps between barriers and device syncs.  Returns the wall time of
    the K steps (max over ranks), the HIP-event time per step on the launch stream, and the
    average duration of the main SpMM kernel alone (HIP events the library records around it)."""
    import torch
    from sgnn_amd import _lib
    lib = _lib.load()
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    lib.sgnn_tune_set(b"time_kernel", 1)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
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
    k_ms, k_n = _lib.kernel_time_ms()
    lib.sgnn_tune_set(b"time_kernel", 0)
    return wall, ev0.elapsed_time(ev1) / steps, (k_ms / k_n if k_n else None), k_n


def load_traffic(key, kernel=None):
    """HBM bytes per launch of the main SpMM kernel from the committed PMC profile, if there is
    one for this workload taken with the kernel that ran."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            rec = json.load(f)[key]
        if kernel is not None and rec.get("kernel") not in (None, kernel):
            return None
        return float(rec["hbm_fabric_bytes"])
    except Exception:
        return None


def draw(workload, seed, device, max_degree=None, weight=None):
    """Graph and operand of one benchmark shape, resident on the device."""
    import torch
    from sgnn_amd import NormAdj
    from sgnn_amd.graphgen import SHAPES, rmat_undirected_device
    n, E, F = SHAPES[workload]
    t0 = time.perf_counter()
    row, col = rmat_undirected_device(n, E, seed=seed, device=device, max_degree=max_degree)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t0
    val = None if weight is None else torch.full((row.numel(),), float(weight), device=device)
    t0 = time.perf_counter()
    adj = NormAdj.from_coo(row, col, val, n, device)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    g = torch.Generator(device=device)
    g.manual_seed(seed + 1)
    X = torch.randn(n, F, device=device, generator=g)
    return adj, X, row, col, t_gen, t_build


def sub_record(workload, seed, device, max_degree=None, weight=None, order=1, min_region_s=0.3):
    """One benchmark shape beside the headline: ms per step, main kernel, roofline fraction."""
    import torch
    adj, X, row, col, t_gen, t_build = draw(workload, seed, device, max_degree, weight)
    del row, col
    n, F = X.shape
    step = lambda: adj.propagate(X, order)      # noqa: E731
    for _ in range(2):
        step()                                   # plans, buffers
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    one = time.perf_counter() - t0
    steps = max(10, min(2000, int(min_region_s / max(one, 1e-5))))
    wall, ev_ms, k_ms, _ = timed_steps(step, steps, 2, device)
    label, kernels, main = kernel_of(adj, F)
    b_alg = algorithmic_bytes(n, adj.nnz, F)
    max_deg = int((adj.rowptr[1:] - adj.rowptr[:-1]).max().item())
    key = workload + ("" if max_degree is None else f"_clip{max_degree}") + ("" if weight is None else f"_w{weight:g}")
    if order > 1:
        # a chain: X is split into planes once, every later hop reads what the epilogue before it wrote
        key += f"_order{order}"
    rec = {"workload": key, "n": n, "nnz": adj.nnz, "F": F, "order": order, "max_degree": max_deg,
           "weights": "unit" if weight is None else f"{weight:g} off the diagonal, 1 on it",
           "steps": steps, "ms_per_step": round(wall * 1e3 / steps, 4), "kernel": label.split(" (")[0],
           "main_kernel": main, "main_kernel_ms": None if k_ms is None else round(k_ms, 4),
           "value": adj.nnz * F * order / (wall / steps), "algorithmic_bytes": b_alg,
           "frac": b_alg / ((k_ms or ev_ms / order) * 1e-3) / 1e9 / HBM_PEAK_GBPS,
           "frac_whole_step": b_alg * order / (ev_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
           "traffic": load_traffic(key, main), "adjacency_build_ms": round(t_build * 1e3, 1)}
    if wall / steps < 0.5e-3:
        # a hop this short is bound by the host's launch calls: the same launches replayed from a
        # HIP graph (how the SGNN stack runs its inner loops) give the device-side figure
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            Y = adj.propagate(X, order)
        w2, _, _, _ = timed_steps(graph.replay, steps, 3, device)
        rec["ms_per_step_graph_replay"] = round(w2 * 1e3 / steps, 4)
        del graph, Y
    del adj, X
    torch.cuda.empty_cache()
    return rec


def cpu_baseline(row, col, X_dev, Y_dev, sample_edges, seed, budget_s=150.0):
    """The CPU restatement of utils.py:81-96 (torch sparse COO ops, as the reference runs them)
    timed on this host: by default on the SAME graph and operand the GPU steps used, up to three
    calls (median), stopping early once ``budget_s`` is spent; its result checks the GPU's."""
    import torch
    from oracle import sgnn_oracle as orc
    n, F = X_dev.shape
    if sample_edges > 0:
        from sgnn_amd.graphgen import rmat_undirected_device
        row, col = rmat_undirected_device(n, sample_edges, seed=seed + 100, device=X_dev.device)
        Y_dev = None
    idx = torch.stack((row, col)).cpu()
    nnz_a = int(idx.shape[1])
    adj = torch.sparse_coo_tensor(idx, torch.ones(nnz_a), (n, n))
    X = X_dev.cpu()
    times, t_all, Y = [], time.perf_counter(), None
    for _ in range(3):
        t0 = time.perf_counter()
        Y = orc.propagate_high_order(adj, X, 1)
        times.append(time.perf_counter() - t0)
        log(f"cpu baseline call {len(times)}: {times[-1]:.1f} s")
        if time.perf_counter() - t_all + times[-1] > budget_s:
            break
    med = statistics.median(times)
    out = {
        "value": (nnz_a + n) * F / med,
        "unit": "edge·feat/s",
        "cores": int(torch.get_num_threads()),
        "kind": "port",
        "sample": (f"the benchmark graph itself: reddit-shape R-MAT n={n}, nnz(A)={nnz_a}, F={F}, order=1"
                   if sample_edges <= 0 else
                   f"reddit-shape R-MAT n={n} with nnz(A)={nnz_a} (a sample, not the benchmark graph), F={F}, order=1")
                  + f"; median of {len(times)} call(s); full op sequence incl. A+I and degree, as the reference "
                    "redoes them per call",
        "seconds": med,
        "seconds_each": [round(t, 2) for t in times],
    }
    if Y_dev is not None:
        Yg = Y_dev.cpu()
        out["gpu_vs_cpu_rel_err"] = float((Yg - Y).norm() / Y.norm())
        out["gpu_vs_cpu_max_abs_err"] = float((Yg - Y).abs().max())
    return out


def copy_ceiling(device, mib=512, reps=10):
    """Read + write GB/s of a device-to-device copy of ``mib`` MiB through the library's own
    streaming kernel (16 bytes per lane, sgnn_stream_copy_f32): the streaming rate this card
    delivers, to set beside the vendor peak.  512 MiB is the size of the benchmark's operand
    (561 MB in, 561 MB out); buffers of 2 GiB and more stream about 10 % slower."""
    import torch
    from sgnn_amd import _lib
    lib = _lib.load()
    src = torch.empty(mib << 18, dtype=torch.float32, device=device).normal_()
    dst = torch.empty_like(src)
    st = _lib.current_stream(device)
    for _ in range(3):
        _lib.check(lib.sgnn_stream_copy_f32(_lib.ptr(src), _lib.ptr(dst), src.numel(), st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _lib.check(lib.sgnn_stream_copy_f32(_lib.ptr(src), _lib.ptr(dst), src.numel(), st))
    e1.record()
    torch.cuda.synchronize()
    assert torch.equal(src[:4096], dst[:4096]) and torch.equal(src[-4096:], dst[-4096:])
    return 2.0 * src.numel() * 4 * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def exact_fp32_beside(adj, X, device):
    """The same hop through the row-group kernel (fp32 operands all the way, no fp16 planes): what
    a caller who cannot accept the planes' input rounding gets, and how far the two results are
    apart (normwise, and element by element against |A_hat| |X|)."""
    import torch
    if not adj._tg:
        return None
    Y_tg = adj.propagate(X, 1)
    adj.use_tilegroup = False
    try:
        Y_rg = adj.propagate(X, 1)
        wall, ev_ms, k_ms, _ = timed_steps(lambda: adj.propagate(X, 1), 10, 2, device)
        scale = adj.propagate(X.abs(), 1)
    finally:
        adj.use_tilegroup = None
    out = {"fp32_exact_ms": round(ev_ms, 4), "fp32_exact_kernel_ms": None if k_ms is None else round(k_ms, 4),
           "fp32_exact_kernel": "k_rg_spmm (row-group, fp32 vector adds)",
           "tilegroup_vs_fp32_kernel_rel_err": float((Y_tg - Y_rg).norm() / Y_rg.norm()),
           "tilegroup_vs_fp32_kernel_max_err_of_abs_sum": float(((Y_tg - Y_rg).abs() / scale.clamp_min(1e-30)).max())}
    del Y_tg, Y_rg, scale
    return out


def load_step_composition(key):
    """Average duration (ms) of every kernel of one step, from the committed rocprofv3 kernel
    trace of this workload (profiles/step_composition.json, written by tools/profile_gpu.sh)."""
    try:
        with open(os.path.join(ROOT, "profiles", "step_composition.json")) as f:
            return json.load(f)[key]
    except Exception:
        return None


def sgnn_epoch(adj, X, device, max_iter, seed):
    """One SGNN "epoch" on the benchmark graph: a backward sweep plus a forward sweep over two
    decoupled layers with a label head (model/SGNN.py:555-559) at the reference's Reddit
    hyper-parameters (config.json:480-503: 128/64 neurons, leaky_relu=0.2, lr 1e-4, batch 512,
    order 1, eta 1000, 41 classes, 10 000 optimiser iterations per layer run unless
    ``max_iter`` says otherwise).  Labels are random: there is no dataset offline.  Also times
    the full-graph projection after every layer run (model/SGNN.py:73-78, 439: ``self(PX)`` over
    all n rows, plus ``X . U`` once U is trained) to state its share of the epoch."""
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
    finish_events = []
    for layer in model.gnns:
        def timed(X_in, PX, orig=layer._finish):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = orig(X_in, PX)
            e1.record()
            finish_events.append((e0, e1))
            return out
        layer._finish = timed
    (X[:4096] @ model.gnns[0].W).sum().item()          # rocBLAS loads its kernels on first use: not the sweep's cost
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    inputs, _ = model._forward_sweep(False)
    torch.cuda.synchronize()
    t_first = time.perf_counter() - t0
    first_finish = sum(a.elapsed_time(b) for a, b in finish_events) * 1e-3
    finish_events.clear()
    t0 = time.perf_counter()
    model._backward_sweep(inputs)
    model._forward_sweep(True)
    torch.cuda.synchronize()
    t_epoch = time.perf_counter() - t0
    epoch_finish = sum(a.elapsed_time(b) for a, b in finish_events) * 1e-3
    return {"task": "classification, 2 layers (128, 64), batch 512, random labels", "max_iter": max_iter,
            "first_forward_sweep_s": t_first, "epoch_s": t_epoch, "optimiser_iterations": 4 * max_iter,
            "s_per_iteration": t_epoch / (4 * max_iter),
            "full_graph_projection": {
                "what": "act(inner(PX . U) . W) over all n rows after every layer run, X . U included once U is trained "
                        "(stock rocBLAS GEMMs)",
                "calls_in_epoch": len(finish_events), "seconds_in_epoch": round(epoch_finish, 5),
                "share_of_epoch": round(epoch_finish / t_epoch, 5),
                "seconds_in_first_forward_sweep": round(first_finish, 5),
                "share_of_first_forward_sweep": round(first_finish / t_first, 5)}}


def sgnn_epoch_clustering(adj, X, device, max_iter, seed):
    """The same two sweeps for the auto-encoder stack of the clustering task (GNN_tasks.py:325-356)
    at the reference's Reddit clustering hyper-parameters (config.json, SGNN / Clustering /
    Reddit: 128/64 neurons, relu, mask_rate 0.2, lr 1e-4, batch 512, eta 10, lam 1e-6): every
    optimiser iteration samples a 512 x 512 block of the adjacency from the CSR on the GPU
    (model/SGNN.py:167-175) and reconstructs it (model/SGNN.py:138-164)."""
    import torch
    from sgnn_amd.stacked import GAELayer, LayerParam, StackedGNN, get_activation
    n, F = X.shape
    model = StackedGNN.__new__(StackedGNN)
    model.device, model.adjacency_tensor, model.content = device, adj, X
    model.labels = model.training_mask = model.labels_host = model.metric_func = None
    model.BP_count, model.eta = 1, 10
    model.gnns, dim = [], F
    for neurons in (128, 64):
        p = LayerParam(neurons, inner_act=get_activation("relu"), act=get_activation("linear"),
                       gnn_type=LayerParam.GAE, mask_rate=0.2, learning_rate=1e-4, max_iter=max_iter, lam=1e-6,
                       batch_size=512)
        model.gnns.append(GAELayer(adj, dim, neurons, p, device))
        dim = neurons
    torch.manual_seed(seed + 3)
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
    return {"task": "clustering, 2 auto-encoder layers (128, 64), batch 512, mask rate 0.2", "max_iter": max_iter,
            "first_forward_sweep_s": t_first, "epoch_s": t_epoch, "optimiser_iterations": 4 * max_iter,
            "s_per_iteration": t_epoch / (4 * max_iter)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="reddit", help="reddit | reddit128 | products | rmat10m | pubmed")
    ap.add_argument("--order", type=int, default=1)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-degree", type=int, default=None, help="clip hub degree (real Reddit's is ~21.7k)")
    ap.add_argument("--weight", type=float, default=None,
                    help="off-diagonal weight (2.0 = the reference's A + A^T Reddit adjacency, data_loader.py:75)")
    ap.add_argument("--slices", type=int, default=1,
                    help="N > 1: also let the feature-sliced exchange/SpMM pipeline (fp32 rows, one block) of this many "
                         "slices compete with the default schedule (own-columns block multiplied while the exchange "
                         "runs); 1 = default schedule only")
    ap.add_argument("--feature-groups", type=int, default=1,
                    help="N > 1: extra grids to probe beside the row partition: 1 = none (every collective of the run "
                         "then uses the job's own process group), 0 = every divisor of N, "
                         "k = only k feature groups")
    ap.add_argument("--exchange", default="auto", choices=["allgather", "referenced", "auto"],
                    help="N > 1: rows of X a rank receives per hop: all of them (one all-gather), only the ones its "
                         "block references (all-to-all), or the latter where it saves at least 40 %% of the bytes")
    ap.add_argument("--no-feature-split", action="store_true",
                    help="N > 1: do not let the feature-split schedule (whole graph on every rank, F / N columns each, "
                         "two all-to-alls per call) compete with the row partition for the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="N = 1: skip the other shapes and the SGNN epoch")
    ap.add_argument("--epoch-iters", type=int, default=10000,
                    help="N = 1, reddit: optimiser iterations per layer run of the SGNN epoch timed after the "
                         "SpMM steps (0 = skip; 10000 is the reference's Reddit config)")
    ap.add_argument("--cpu-sample-edges", type=int, default=0,
                    help="0 = CPU baseline on the benchmark graph itself; > 0 = on an R-MAT sample of that many edges")
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
        log(f"{args.workload}: drawing the graph on {world} rank(s)")
    extra = {}
    cpu_inputs = None
    if world == 1:
        adj, X, row, col, t_gen, t_build = draw(args.workload, args.seed, device, args.max_degree, args.weight)
        extra["adjacency_build_ms"] = t_build * 1e3
        if not args.no_cpu_baseline and args.cpu_sample_edges <= 0:
            cpu_inputs = (row.cpu(), col.cpu())
        del row, col
        nnz = adj.nnz
        step = lambda: adj.propagate(X, args.order)          # noqa: E731
        max_deg = int((adj.rowptr[1:] - adj.rowptr[:-1]).max().item())
        kernel_path, kernels_in_step, main_kernel = kernel_of(adj, F)
    else:
        from sgnn_amd.dist import PartitionedAdj, grid_layout, make_row_groups
        t0 = time.perf_counter()
        row, col = rmat_undirected_device(n, E, seed=args.seed, device=device, max_degree=args.max_degree)
        torch.cuda.synchronize()
        t_gen = time.perf_counter() - t0
        g = torch.Generator(device=device)
        g.manual_seed(args.seed + 1)
        X = torch.randn(n, F, device=device, generator=g)

        def build_grid(fg):
            """One grid: a list of (step function, info), one entry per exchange schedule that
            reproduces the plain one on this machine."""
            row_rank, row_world, fgroup, f0, f1 = grid_layout(rank, world, fg, F)
            group = make_row_groups(world, fg)
            t0 = time.perf_counter()
            padj = PartitionedAdj.from_coo(row, col, None, n, device, row_rank, row_world, group=group,
                                           exchange=args.exchange)
            torch.cuda.synchronize()
            build_ms = (time.perf_counter() - t0) * 1e3
            Xl = X[padj.rows.to(device)][:, f0:f1].contiguous()
            schedules = [1]
            want = max(1, args.slices)
            if row_world > 1 and want > 1 and padj.halo is None:     # (the referenced-rows exchange is not sliced)
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
                        "max_deg": padj.global_max_degree,
                        "exchange_ranks": dist.get_world_size(group) if row_world > 1 else 1,
                        "exchange": "none" if row_world == 1 else
                                    ("all_gather_into_tensor" if padj.halo is None else "all_to_all_single of referenced rows"),
                        "exchange_bytes_per_rank_per_hop":
                            0 if row_world == 1 else
                            (padj.max_rows * row_world if padj.halo is None else padj.halo["n_recv"])
                            * ((f1 - f0 + 3) // 4 * 4) * 4}
                info["_padj"] = padj
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

        def probe_grid(fg, probed):
            best = None
            for step_fn, info in build_grid(fg):
                t = probe(step_fn)
                tag = f"{fg}x{world // fg}" + (f"/s{info['exchange_slices']}" if info["exchange_slices"] > 1 else "")
                probed[tag] = round(t * 1e3, 3)
                if rank == 0:
                    log(f"{fg} feature slices x {world // fg} row blocks, exchange in {info['exchange_slices']} "
                        f"slice(s): built in {info['adjacency_build_ms']:.0f} ms, probe {t * 1e3:.2f} ms/step")
                if best is None or t < best[0]:
                    best = (t, step_fn, info)
            return best

        probed = {}
        # Two ways to use the ranks compete for the headline, both serving a caller who owns rows
        # of X and both exchanging over the job's collective backend: the row partition (all
        # `world` ranks in one exchange per hop, overlapped with the multiply) and the feature
        # split (every rank holds the graph and propagates F / world columns; two all-to-alls per
        # call carry the rows there and back).
        t_row, step, info = probe_grid(1, probed)
        fs = None
        fs_ok = torch.ones(1, device=device)
        if not args.no_feature_split:
            from sgnn_amd.dist import FeatureSplitAdj
            t0 = time.perf_counter()
            try:                                  # (a rank that cannot hold the whole graph takes every rank out of this schedule)
                fadj = FeatureSplitAdj.from_coo(row, col, None, n, device, rank, world)
                torch.cuda.synchronize()
                Xf = X[fadj.rows.to(device)].contiguous()
            except Exception as exc:              # noqa: BLE001
                print(f"[rank {rank}] feature split not built: {exc}", file=sys.stderr, flush=True)
                fs_ok.zero_()
            dist.all_reduce(fs_ok, op=dist.ReduceOp.MIN)
        if not args.no_feature_split and fs_ok.item() != 0:
            fs_build_ms = (time.perf_counter() - t0) * 1e3
            fs_step = lambda: fadj.propagate(Xf, args.order)      # noqa: E731
            t_fs = probe(fs_step)
            probed["feature-split"] = round(t_fs * 1e3, 3)
            if rank == 0:
                log(f"feature split over {world} ranks (whole graph per rank, two all-to-alls per call): built in "
                    f"{fs_build_ms:.0f} ms, probe {t_fs * 1e3:.2f} ms/step; row partition {t_row * 1e3:.2f} ms/step")
            if t_fs < t_row:
                fs = (fadj, fs_step, fs_build_ms)
            else:
                del fadj, Xf
        if args.feature_groups == 0:
            others = [g for g in range(2, world + 1) if world % g == 0 and g <= F]
        else:
            others = [args.feature_groups] if args.feature_groups > 1 else []
        for fg in others:                            # extras: reported, never benchmarked
            probe_grid(fg, probed)
            torch.cuda.empty_cache()
        del row, col, X
        nnz, max_deg = info.pop("nnz"), info.pop("max_deg")
        padj = info.pop("_padj")
        if fs is not None:
            fadj, step, fs_build_ms = fs
            Fw, f0s, wq = fadj.feature_slice(F)
            step()
            extra.update({"feature_groups": world, "row_ranks": 1, "features_per_rank": wq if wq > 0 else Fw,
                          "adjacency_build_ms": fs_build_ms, "exchange_ranks": world,
                          "exchange": "2 x all_to_all_single per call (rows -> feature slices -> rows)",
                          "exchange_bytes_per_rank_per_call": 2 * world * fadj.max_rows * Fw * 4 * (world - 1) // world,
                          "overlap": "feature-split: no exchange between hops",
                          "collective": (f"all_to_all_single among {world} ranks, backend {dist.get_backend()} "
                                         f"({'gloo rehearsal on one GPU' if rehearsal else 'RCCL'}), world size "
                                         f"{dist.get_world_size()}"),
                          "row_partition_ms_per_step": round(t_row * 1e3, 3)})
            if len(probed) > 1:
                extra["grid_probe_ms"] = probed
            kernel_path, kernels_in_step, main_kernel = kernel_of(fadj.adj, extra["features_per_rank"])
            kernel_path = (f"feature split: every rank holds the whole graph and propagates {extra['features_per_rank']} of "
                           f"{F} columns; " + kernel_path)
            kernels_in_step = "pack + all_to_all + " + kernels_in_step + " + all_to_all + unpack"
            if rank == 0:
                log(f"headline: feature split over {world} ranks; {extra['collective']}")
            padj = None
        if padj is not None:
            local = padj.local            # which kernel the rank-local multiply took (decided on first use)
            extra.update(info)
            step()                                     # (sets padj.last_schedule for the schedule benchmarked)
            # which schedule the timed steps run: "local-block" = own-columns block multiplied while the
            # exchange runs, remote-columns block when it lands (with fp16 planes exchanged where the
            # matrix-core kernel applies); "sliced" = feature slices pipelined; "none" = exchange, then multiply
            extra["overlap"] = padj.last_schedule
            extra["collective"] = (f"{info['exchange']} among {info['exchange_ranks']} ranks, backend "
                                   f"{dist.get_backend()} ({'gloo rehearsal on one GPU' if rehearsal else 'RCCL'}), "
                                   f"world size {dist.get_world_size()}")
            if rank == 0:
                log(f"headline: row partition over {info['exchange_ranks']} ranks, overlap {extra['overlap']}; {extra['collective']}")
            if len(probed) > 1:
                extra["grid_probe_ms"] = probed
            lib = __import__("sgnn_amd._lib", fromlist=["load"]).load()
            planes = padj.last_schedule is not None and "planes" in padj.last_schedule
            if planes:
                Fr = extra["features_per_rank"]
                extra["exchange_bytes_per_rank_per_hop"] = (
                    (padj.max_rows * padj.world if padj.halo is None else padj.halo["n_recv"]) * ((Fr + 127) // 128 * 128) * 4)
                kernel_path = ("tile-group SpMM on the rank's own-columns and remote-columns blocks; every rank splits only "
                               "its own rows and fp16 planes travel, one exchange per 128-feature slice; the remote-columns "
                               f"block is multiplied in units of {padj.slices_per_exchange} slice(s) as they land")
                kernels_in_step = ("k_tg_amax + all_reduce(max) + k_tg_split (own rows) + exchange || [k_tg_spmm (own columns) "
                                   "+ per unit of slices: k_tg_spmm (remote columns, adds the own-columns sums) + k_tg_combine]")
                extra["exchange_unit_slices"] = padj.slices_per_exchange
                extra["reserved_cus_beside_exchange"] = padj._reserved_cus()
                main_kernel = "k_tg_spmm"
            else:
                blk = padj.remote if (padj.last_schedule == "local-block" and padj.remote is not None) else local
                R = lib.sgnn_rowgroup_config_for(blk.n_rows, blk.adj.nnz, extra["features_per_rank"])
                if getattr(local, "_tg", None) and padj.last_schedule in ("none", "sliced"):
                    kernel_path = "tile-group SpMM on the rank's rows (each rank splits the gathered operand)"
                    kernels_in_step = "k_row_scale + exchange + k_tg_amax + k_tg_split + k_tg_spmm + k_tg_combine"
                    main_kernel = "k_tg_spmm"
                else:
                    kernel_path = f"row-group SpMM, R={R}" if R else "plain CSR SpMM"
                    kernels_in_step = "k_row_scale + exchange + " + ("k_rg_spmm + k_rg_combine" if R else "k_spmm")
                    if padj.last_schedule == "local-block":
                        kernels_in_step = ("k_row_scale + exchange || own-columns " + ("k_rg_spmm" if R else "k_spmm")
                                           + " + remote-columns " + ("k_rg_spmm" if R else "k_spmm") + " + k_add_scale")
                    main_kernel = "k_rg_spmm" if R else "k_spmm"

    wall, ev_ms, k_ms, k_n = timed_steps(step, args.steps, args.warmup, device, world, dist if world > 1 else None)

    copy_gbps = copy_gbps_2g = None
    exact = None
    epoch = epoch_gae = None
    others = []
    cpu = None
    if rank == 0 and world == 1:
        try:
            copy_gbps = copy_ceiling(device)
            copy_gbps_2g = copy_ceiling(device, mib=2048)
        except Exception as exc:
            log(f"copy ceiling not measured: {exc}")
        try:
            exact = exact_fp32_beside(adj, X, device) if not args.no_extras else None
        except Exception as exc:
            exact = {"error": f"{type(exc).__name__}: {exc}"}
        if not args.no_cpu_baseline:
            try:
                Y = adj.propagate(X, 1)
                if cpu_inputs is not None:
                    cpu = cpu_baseline(cpu_inputs[0], cpu_inputs[1], X, Y, 0, args.seed)
                else:
                    cpu = cpu_baseline(None, None, X, None, args.cpu_sample_edges, args.seed)
                del Y
            except Exception as exc:
                cpu = {"error": f"{type(exc).__name__}: {exc}"}
            cpu_inputs = None
        if args.workload == "reddit" and args.epoch_iters > 0 and not args.no_extras:   # outside the timed SpMM region
            log(f"SGNN epoch at {args.epoch_iters} iterations per layer run")
            try:
                epoch = sgnn_epoch(adj, X, device, args.epoch_iters, args.seed)
            except Exception as exc:                  # the SpMM line must come out whatever happens here
                epoch = {"error": f"{type(exc).__name__}: {exc}"}
            log(f"SGNN clustering epoch at {args.epoch_iters} iterations per layer run")
            try:
                epoch_gae = sgnn_epoch_clustering(adj, X, device, args.epoch_iters, args.seed)
            except Exception as exc:
                epoch_gae = {"error": f"{type(exc).__name__}: {exc}"}
        if args.workload == "reddit" and args.max_degree is None and args.weight is None and not args.no_extras:
            del adj, X
            torch.cuda.empty_cache()
            for wl, md, wt, od in (("reddit", 22000, None, 1), ("reddit", None, 2.0, 1), ("reddit", 22000, 2.0, 1),
                                   ("reddit", None, None, 2), ("reddit128", None, None, 1), ("products", None, None, 1),
                                   ("products", None, None, 2), ("rmat10m", None, None, 1), ("rmat10m", None, None, 2),
                                   ("pubmed", None, None, 1)):
                log(f"other shapes: {wl} max_degree={md} weight={wt} order={od}")
                try:
                    others.append(sub_record(wl, args.seed, device, md, wt, od))
                except Exception as exc:
                    others.append({"workload": wl, "error": f"{type(exc).__name__}: {exc}"})
    if rank == 0:
        ms_per_step = wall * 1e3 / args.steps
        units = nnz * F * args.order                          # (E + n) . F per hop
        if world == 1:
            b_alg = algorithmic_bytes(n, nnz, F)               # per launch of the main kernel (one hop)
        else:
            # one rank's launch: its share of the CSR and of Y, its slice of the whole (gathered) X
            rr, fr = extra["row_ranks"], extra["features_per_rank"]
            b_alg = nnz // rr * 8 + (n // rr + 1) * 4 + 2 * (n // rr) * 4 + n * fr * 4 + (n // rr) * fr * 4
        launches_per_step = (k_n / args.steps) if k_n else args.order
        kernel_ms = k_ms if k_ms else ev_ms / launches_per_step
        if world > 1 and k_ms:
            # a rank's hop is several launches of the main kernel (own-columns block, remote-columns
            # block unit by unit): what is priced against the rank's bytes is their sum per hop
            kernel_ms = k_ms * launches_per_step / args.order
        achieved = b_alg / (kernel_ms * 1e-3) / 1e9
        key = args.workload + ("" if args.max_degree is None else f"_clip{args.max_degree}") \
            + ("" if args.weight is None else f"_w{args.weight:g}")
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
                            f"nnz(A+I)={nnz}, F={F}, order={args.order}, max_degree={max_deg}, "
                            + ("unit weights" if args.weight is None else f"weight {args.weight:g} off the diagonal"),
                "rows": "one GPU" if world == 1 else
                        (f"row partition: {extra['row_ranks']} row blocks over {world} GPUs, one exchange of the "
                         "operand's rows per hop" if extra.get("row_ranks", 1) > 1 else
                         f"feature split: whole graph on each of {world} GPUs, {extra['features_per_rank']} columns each; "
                         "the caller's row blocks travel by all-to-all before the first and after the last hop"),
                "arithmetic": "fp32 in, fp32 out, fp32 accumulation; the tile-group kernel feeds the matrix cores X as "
                              "two fp16 planes (hi + lo/2048 at one power-of-two scale per feature column, taken from a sample of the rows and checked element by element) that reproduce each fp32 "
                              "input to 2^-22 relative; operands they cannot represent take the plain fp32 kernel"
                              if main_kernel == "k_tg_spmm" else "fp32 throughout",
                "graph_gen_s": round(t_gen, 2),
                "kernel": kernel_path,
                **{k: (round(v, 2) if isinstance(v, float) else v) for k, v in extra.items()},
            },
            "roofline": {
                "bound": "hbm",
                "kernel": main_kernel,
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": load_traffic(key, main_kernel) if world == 1 else None,
                "algorithmic_bytes": b_alg,
                "kernel_ms": kernel_ms,                       # HIP events around every launch of the main kernel
                "kernel_launches_timed": k_n,
                "step_ms": ev_ms,                             # HIP events around the K steps / K: all kernels of a step
                "frac_whole_step": b_alg * args.order / (ev_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,   # N > 1: exchange included
                "kernels_in_step": kernels_in_step,
                "gather_bytes_no_reuse": nnz * F * 4,
                **({} if world == 1 else {"scope": "one rank: kernel_ms is the sum of its launches of the main kernel per hop, "
                                                    "algorithmic_bytes its share of the CSR and of Y plus the operand it reads",
                                          "kernel_launches_per_hop": launches_per_step / args.order}),
            },
        }
        roof = out["roofline"]
        if exact:
            roof.update(exact)
        comp = load_step_composition(key) if world == 1 else None
        if comp:
            roof["step_composition_ms"] = comp          # from the committed kernel trace of this workload
        if copy_gbps:                                         # what a streaming copy reaches on this card
            roof["copy_ceiling"] = copy_gbps                  # 512 MiB buffers, hand-written float4 copy
            roof["copy_ceiling_2GiB_buffers"] = copy_gbps_2g
            roof["frac_of_copy"] = achieved / copy_gbps
        if roof["traffic"]:                                   # bytes the L2s fetched or wrote, per launch
            roof["fabric_GBps"] = roof["traffic"] / (kernel_ms * 1e-3) / 1e9
            # bytes actually moved over the L2's fabric side (Infinity-Cache hits included) over the vendor
            # peak: the "measured bytes" fraction, stated beside `frac`, which prices algorithmic bytes
            roof["fabric_frac_of_peak"] = roof["fabric_GBps"] / HBM_PEAK_GBPS
        if epoch:
            out["sgnn_epoch"] = {k: (round(v, 6) if isinstance(v, float) else v) for k, v in epoch.items()}
        if epoch_gae:
            out["sgnn_epoch_clustering"] = {k: (round(v, 6) if isinstance(v, float) else v) for k, v in epoch_gae.items()}
        if others:
            out["other_workloads"] = others
        if cpu:
            out["cpu_baseline"] = cpu
        print(json.dumps(out, ensure_ascii=False), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
