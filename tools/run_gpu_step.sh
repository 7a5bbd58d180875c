set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python -u tools/ab_split.py sgnn_amd/libsgnn_hip.so > gpurun_out/split_final.log 2>&1 || exit 1
timeout -k 10 600 python -u -m pytest tests/test_gpu_tilegroup.py tests/test_gpu_planes_bound.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r03_tg_tests.log 2>&1 || { tail -n 30 gpurun_out/r03_tg_tests.log; exit 1; }
tail -n 3 gpurun_out/r03_tg_tests.log
