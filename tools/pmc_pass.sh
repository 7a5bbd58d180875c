#!/bin/bash
# One rocprofv3 counter pass of a python tool (run on the GPU box through gpurun):
#   tools/pmc_pass.sh <tag> "<counters>" <script> [args...]
# prints, per kernel name, the per-dispatch average of every counter.
TAG=$1; PMC=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d "$OUT" -- python3 "$REPO/$1" "${@:2}" > "$OUT/run.log" 2>&1 || { echo "pass failed"; tail -5 "$OUT/run.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set()
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    k = next((t for t in ("k_tg_spmm", "k_rg_spmm", "k_spmm", "k_tg_split", "k_tg_combine") if t in n), n[:40])
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r["Dispatch_Id"])
    if key not in seen:
        seen.add(key); cnt[k] += 1
for k in acc:
    if "k_tg" in k or "k_rg" in k or "k_spmm" in k:
        print(k, "dispatches", cnt[k], {c: round(v / cnt[k]) for c, v in acc[k].items()})
PY
