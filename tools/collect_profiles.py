#!/usr/bin/env python3
"""Copy the outputs of tools/profile_gpu.sh (gpurun_out/<tag>/) into profiles/ and refresh the two
records bench.py reads: profiles/pmc_traffic.json (bytes over the L2's fabric side per launch of
the main kernel) and profiles/step_composition.json (average duration of every kernel of a step,
from the kernel trace).     python tools/collect_profiles.py <tag>=<bench key> ..."""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEP_KERNELS = ("k_tg_amax(", "k_tg_amax_fold", "k_tg_state_init", "k_tg_split", "k_tg_spmm", "k_tg_combine",
                "k_rg_spmm", "k_rg_combine", "k_row_scale", "k_spmm<", "k_spmm_combine", "k_collect_long", "fillBufferAligned")


def main():
    prof = os.path.join(ROOT, "profiles")
    traffic = json.load(open(os.path.join(prof, "pmc_traffic.json")))
    comp_path = os.path.join(prof, "step_composition.json")
    comp = json.load(open(comp_path)) if os.path.exists(comp_path) else {}
    for arg in sys.argv[1:]:
        tag, key = arg.split("=")
        src = os.path.join(ROOT, "gpurun_out", tag)
        summ = json.load(open(os.path.join(src, "summary.json")))
        summ["dir"] = f"gpurun_out/{tag}"
        json.dump(summ, open(os.path.join(prof, tag + ".json"), "w"), indent=1)
        shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(prof, tag + "_kernel_stats.csv"))
        shutil.copy(os.path.join(src, "bench_line_under_trace.json"), os.path.join(prof, tag + "_bench_line_under_trace.json"))
        if "hbm_fabric_bytes" in summ:
            traffic[key] = {"hbm_fabric_bytes": summ["hbm_fabric_bytes"], "fetch_bytes_raw_counter": summ["fetch_bytes_raw"],
                            "fetch_correction": 2.0, "write_bytes": summ["write_bytes"], "l2_hit_rate": summ.get("l2_hit_rate"),
                            "kernel": summ["kernel_filter"], "source": f"profiles/{tag}.json"}
        line = json.loads(open(os.path.join(src, "bench_line_under_trace.json")).read().strip().splitlines()[-1])
        steps = line["steps"] + line["warmup"]
        rows = list(csv.DictReader(open(os.path.join(src, "kernel_stats.csv"))))
        c = {}
        for r in rows:
            name = r["Name"]
            hit = next((k for k in STEP_KERNELS if k in name), None)
            if hit is None:
                continue
            label = {"k_tg_amax(": "k_tg_amax", "k_spmm<": "k_spmm (guarded fall-back)",
                     "k_spmm_combine": "k_spmm_combine (guarded fall-back)", "k_collect_long": "k_collect_long (guarded fall-back)", "fillBufferAligned": "memset"}.get(hit, hit)
            calls, avg = int(r["Calls"]), float(r["AverageNs"]) / 1e6
            if calls < steps:        # not part of every step (plan construction shares some names)
                continue
            if label == "memset":    # the trace cannot tell a step's fills from those of graph and plan construction
                c[label] = {"avg_ms": round(avg, 5), "launches_in_traced_run": calls,
                            "note": "graph and plan construction included; a one-hop step issues one fill "
                                    "(the work counters of the guarded fall-back), a chain one more per hop"}
                continue
            c[label] = {"avg_ms": round(avg, 5), "launches_per_step": round(calls / steps, 2)}
        c["source"] = f"profiles/{tag}_kernel_stats.csv ({steps} traced steps, warm-up included)"
        comp[key] = c
    json.dump(traffic, open(os.path.join(prof, "pmc_traffic.json"), "w"), indent=1)
    json.dump(comp, open(comp_path, "w"), indent=1)
    for k, v in comp.items():
        print(k, {n: x["avg_ms"] for n, x in v.items() if isinstance(x, dict)})
    for k in (a.split("=")[1] for a in sys.argv[1:]):
        t = traffic[k]
        print(k, "fabric GB", round(t["hbm_fabric_bytes"] / 1e9, 2), "L2 hit", round(t["l2_hit_rate"] or 0, 3))


if __name__ == "__main__":
    main()
