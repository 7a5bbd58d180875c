set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests/test_gpu_dist.py -x -q --timeout 300 --timeout-method thread -m gpu > gpurun_out/r3_t4.log 2>&1; rc=$?; tail -40 gpurun_out/r3_t4.log
exit $rc
