set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_gpu_tilegroup.py tests/test_gpu_planes_bound.py -x -q --timeout 180 --timeout-method thread -m gpu > gpurun_out/r3_t3.log 2>&1; rc=$?; tail -3 gpurun_out/r3_t3.log
[ $rc -ne 0 ] && exit $rc
for wl in reddit products rmat10m; do
timeout -k 10 300 python -u tools/chain_probe.py $wl 1 2 > gpurun_out/r3_chain_new_$wl.log 2>&1 && tail -1 gpurun_out/r3_chain_new_$wl.log || exit 1
done
cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_chain -o chain -- python3 $GRAFT_REPO_ROOT/tools/chain_probe.py rmat10m 2 > $GRAFT_REPO_ROOT/gpurun_out/r3_prof_chain.log 2>&1; echo rocprof rc=$?
f=$(find /tmp/prof_chain -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
sel = [r for r in rows if "k_tg_" in r["Kernel_Name"] or "k_spmm" in r["Kernel_Name"] or "fill" in r["Kernel_Name"].lower() or "memset" in r["Kernel_Name"].lower()]
for r in sel[-16:]:
    print(r["Kernel_Name"][:60], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us")
PY
