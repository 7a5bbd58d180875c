set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_gpu_tilegroup.py tests/test_gpu_planes_bound.py -x -q --timeout 300 --timeout-method thread -m gpu 2>&1 | tail -3 || exit 1
timeout -k 10 300 python -u tools/chain_probe.py reddit 1 2 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_chain -o chain -- python3 $GRAFT_REPO_ROOT/tools/chain_probe.py reddit 1 > /dev/null 2>&1; echo rocprof rc=$?
f=$(find /tmp/prof_chain -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
sel = [r for r in rows if "k_tg_" in r["Kernel_Name"] or "k_spmm" in r["Kernel_Name"] or "fill" in r["Kernel_Name"].lower()]
for r in sel[-11:]:
    print(r["Kernel_Name"][:60], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us")
PY
