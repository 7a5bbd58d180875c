#!/usr/bin/env python3
"""Summarise a tools/profile_gpu.sh output directory: per-kernel time from the
kernel-trace stats and per-launch PMC counters for the SpMM kernel.

    python tools/summarize_prof.py gpurun_out/<tag> [kernel-substring]
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def one(pattern):
    files = glob.glob(pattern, recursive=True)
    return files[0] if files else None


def main():
    root = sys.argv[1]
    needle = sys.argv[2] if len(sys.argv) > 2 else "k_tg_spmm"
    out = {"dir": root, "kernel_filter": needle}
    stats = one(os.path.join(root, "trace", "**", "*_kernel_stats.csv"))
    if stats:
        rows = list(csv.DictReader(open(stats)))
        out["kernel_stats"] = [
            {"name": r["Name"][:110], "calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
             "total_ms": float(r["TotalDurationNs"]) / 1e6, "pct": float(r["Percentage"])}
            for r in rows[:12]]
    counters = defaultdict(list)
    durations = defaultdict(list)
    for sub in ("pmc_fetch", "pmc_write", "pmc_extra"):
        f = one(os.path.join(root, sub, "**", "*_counter_collection.csv"))
        if not f:
            continue
        for r in csv.DictReader(open(f)):
            if needle in r["Kernel_Name"]:
                counters[r["Counter_Name"]].append(float(r["Counter_Value"]))
                durations[sub].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    pmc = {k: sum(v) / len(v) for k, v in counters.items()}
    out["pmc_per_launch"] = pmc
    out["pmc_pass_kernel_ms"] = {k: sum(v) / len(v) for k, v in durations.items()}
    if "FETCH_SIZE" in pmc:
        out["fetch_bytes_raw"] = pmc["FETCH_SIZE"] * 1024
    if "WRITE_SIZE" in pmc:
        out["write_bytes"] = pmc["WRITE_SIZE"] * 1024
    if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        # gfx950: FETCH_SIZE counts 64-byte requests as 32 bytes, hence the factor 2
        out["hbm_fabric_bytes"] = 2 * out["fetch_bytes_raw"] + out["write_bytes"]
    if "TCC_HIT_sum" in pmc and "TCC_MISS_sum" in pmc:
        out["l2_hit_rate"] = pmc["TCC_HIT_sum"] / (pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"])
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
