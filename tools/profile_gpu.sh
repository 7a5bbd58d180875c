#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats plus two PMC passes of a
# short bench.py run.  Results land under gpurun_out/<tag>/.
#   tools/profile_gpu.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-prof}; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" $ARGS > "$OUT/trace.log" 2>&1 || { echo "trace pass failed"; tail -20 "$OUT/trace.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 "$REPO/bench.py" $ARGS > "$OUT/pmc_fetch.log" 2>&1 || { echo "pmc fetch pass failed"; tail -20 "$OUT/pmc_fetch.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 "$REPO/bench.py" $ARGS > "$OUT/pmc_write.log" 2>&1 || { echo "pmc write pass failed"; tail -20 "$OUT/pmc_write.log"; exit 1; }
find "$OUT" -name "*.csv" | head -20
