set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -u -m pytest tests/test_gpu_tilegroup.py -x -v --timeout 150 --timeout-method thread -k "provisional" > gpurun_out/t_pv.log 2>&1; echo rc=$?; grep -E "PASSED|FAILED|^E |passed|failed" gpurun_out/t_pv.log | head -n 24
