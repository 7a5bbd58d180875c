#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats plus two PMC passes of a
# short bench.py run.  Results land under gpurun_out/<tag>/: summary.json (per-kernel
# times, per-launch counters of the main SpMM kernel) and the kernel-stats CSV; the raw
# traces are deleted (they run to hundreds of MB).
#   tools/profile_gpu.sh <tag> <kernel-substring> [bench args...]
set -o pipefail
TAG=${1:-prof}; NEEDLE=${2:-k_tg_spmm}; shift 2
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --no-cpu-baseline --no-extras $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" $ARGS > "$OUT/trace.log" 2>&1 || { echo "trace pass failed"; tail -20 "$OUT/trace.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 "$REPO/bench.py" $ARGS > "$OUT/pmc_fetch.log" 2>&1 || { echo "pmc fetch pass failed"; tail -20 "$OUT/pmc_fetch.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 "$REPO/bench.py" $ARGS > "$OUT/pmc_write.log" 2>&1 || { echo "pmc write pass failed"; tail -20 "$OUT/pmc_write.log"; exit 1; }
python3 "$REPO/tools/summarize_prof.py" "$OUT" "$NEEDLE" > "$OUT/summary.json" || exit 1
find "$OUT" -name "*_kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
grep -h '^{' "$OUT/trace.log" > "$OUT/bench_line_under_trace.json" || true
rm -rf "$OUT/trace" "$OUT/pmc_fetch" "$OUT/pmc_write"
cat "$OUT/summary.json"
