set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_gpu_tilegroup.py -x -q --timeout 100 --timeout-method thread -k "slice_ranges" > gpurun_out/t_sl.log 2>&1; echo rc=$?; grep -E "passed|failed|^E " gpurun_out/t_sl.log | head -n 10
