#!/usr/bin/env python3
"""Work balance of a tile-group plan (run on the GPU box): steps per wavefront and stage."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgnn_amd import NormAdj  # noqa: E402
from sgnn_amd.graphgen import SHAPES, rmat_undirected_device  # noqa: E402

dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "reddit"
n, E, F = SHAPES[wl]
row, col = rmat_undirected_device(n, E, seed=1, device=dev)
adj = NormAdj.from_coo(row, col, None, n, dev)
del row, col
adj.use_tilegroup = True
p = adj._tilegroup_plan(F)
al = lambda x: (x + 255) // 256 * 256
G = (n + 511) // 512
stages, steps = p["stages"], p["steps"]
off = al(64) + al(G * 512 * 4) + al((G + 1) * 4) + al((stages * 128 + 128) * 4)
step0 = p["buf"][off: off + (stages * 16 + 1) * 4].view(torch.int32).long()
per = (step0[1:] - step0[:-1]).view(stages, 16)                     # steps of (stage, sub-tile)
print("steps per (stage, sub-tile): mean %.2f" % per.float().mean().item(), "hist", torch.bincount(per.flatten(), minlength=9).tolist())
wave = per[:, :8] + per[:, 8:].flip(1)                               # wavefront w: sub-tiles w and 15 - w
wave_old = per[:, :8] + per[:, 8:]
wave_now = per[:, 0::2] + per[:, 1::2]                               # what the kernel does: sub-tiles 2 w and 2 w + 1
srt = per.sort(1).values
wave_best = srt[:, :8] + srt[:, 8:].flip(1)                          # per stage: largest with smallest (needs free ownership)
for name, wv in (("2w,2w+1 (kernel)", wave_now), ("w,15-w", wave), ("w,w+8", wave_old), ("sorted per stage", wave_best)):
    mx, mean = wv.max(1).values.float(), wv.float().mean(1)
    print(f"{name}: steps per wavefront and stage mean {mean.mean():.2f}; sum of max over stages {mx.sum():.0f} "
          f"vs sum of mean {mean.sum():.0f}: imbalance {mx.sum() / mean.sum():.3f}")
print("sub-tile mean steps by position in group:", [round(v, 2) for v in per.float().mean(0).tolist()])
tot = per.sum(1).float()
print("steps per stage: mean %.1f min %d max %d" % (tot.mean().item(), tot.min().item(), tot.max().item()))
hist = torch.bincount((tot.long() // 8), minlength=17).tolist()
print("stages by steps per stage (buckets of 8):", hist)
gst = p["buf"][al(64) + al(G * 512 * 4): al(64) + al(G * 512 * 4) + (G + 1) * 4].view(torch.int32).long()
gid = torch.repeat_interleave(torch.arange(G, device=dev), gst[1:] - gst[:-1])
gmax = torch.zeros(G, device=dev).scatter_reduce(0, gid, tot, "amax")
gsum = torch.zeros(G, device=dev).scatter_add(0, gid, tot)
for cap in (64, 74, 89, 96, 112):
    over = tot > cap
    print(f"cap {cap}: {over.float().mean() * 100:.1f}% of stages hold {tot[over].sum() / tot.sum() * 100:.1f}% of steps; "
          f"groups with a stage over it: {(gmax > cap).sum().item()} of {G}, holding {gsum[gmax > cap].sum() / tot.sum() * 100:.1f}% of steps")

# ---- launch-order simulation: greedy list scheduling of (item, slice) workgroups on 256 CUs ----
import heapq
ns = (gst[1:] - gst[:-1]).cpu()
stage_cost = (1500 + 550 * wave_now.max(1).values.float()).cpu()          # ticks: issue + the slowest wavefront's steps
cum = torch.cat((torch.zeros(1), stage_cost.cumsum(0)))
gst_c = gst.cpu()
total = int(gst_c[-1])
slice_w = [1.0, 1.0, 1.0, 1.0, 0.85]                                       # the fifth slice of F = 602 has 3 of 4 tiles


def items_for(seg):
    out = []
    for g in range(G):
        n_g, s0 = int(ns[g]), int(gst_c[g])
        pieces = (n_g + seg - 1) // seg if n_g > seg + seg // 4 else 1
        for k in range(pieces):
            a, b = s0 + n_g * k // pieces, s0 + n_g * (k + 1) // pieces
            out.append((b - a, float(cum[b] - cum[a]) + 6000.0))          # stages, ticks incl. start-up and write-out
    return out


def makespan(order):
    free = [0.0] * 256
    heapq.heapify(free)
    end = 0.0
    for c in order:
        t = heapq.heappop(free) + c
        end = max(end, t)
        heapq.heappush(free, t)
    return end


for seg in (128, max(32, min(224, total // 448)), 320):
    its = sorted(items_for(seg), key=lambda x: -x[0])
    ideal = sum(c for _, c in its) * sum(slice_w) / 256
    plain = [c * w for w in slice_w for _, c in its]
    front = [c * w for w in slice_w for n_, c in its if n_ >= 96] + [c * w for w in slice_w for n_, c in its if n_ < 96]
    glob = sorted(plain, reverse=True)
    rev = [c * w for w in reversed(slice_w) for _, c in its]
    print(f"seg {seg}: {len(its)} items; simulated makespan / ideal: slice-major longest-first {makespan(plain) / ideal:.3f}, "
          f"long items of all slices first {makespan(front) / ideal:.3f}, all workgroups longest-first {makespan(glob) / ideal:.3f}, "
          f"cheap slice first {makespan(rev) / ideal:.3f}")
