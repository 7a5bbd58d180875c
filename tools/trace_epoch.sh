#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace a short SGNN epoch (tools/sgnn_epoch.py,
# 300 iterations per layer run) and print call counts / average durations of the kernels an
# optimiser iteration consists of.   tools/trace_epoch.sh [classification|clustering]
set -o pipefail
TASK=${1:-classification}
OUT=$GRAFT_REPO_ROOT/gpurun_out/epoch_trace_$TASK
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 $GRAFT_REPO_ROOT/tools/sgnn_epoch.py --max-iter 300 --task $TASK > "$OUT/log.txt" 2>&1 || { tail -20 "$OUT/log.txt"; exit 1; }
find "$OUT" -name "*kernel_trace.csv" -delete
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = 0; cnt = 0
for r in rows:
    c = int(r['Calls'])
    if c >= 250:
        tot += int(r['TotalDurationNs']); cnt += c
        print(f"{c:6d} {int(r['TotalDurationNs'])/1e6:8.2f}ms avg {float(r['AverageNs'])/1e3:6.2f}us  {r['Name'][:150]}")
print("per-iteration kernels: calls", cnt, "time ms", tot / 1e6)
PY
