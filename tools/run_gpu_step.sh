set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests/test_gpu_planes_bound.py tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_random_sweep.py tests/test_zz_no_alloc.py -x -q --timeout 180 --timeout-method thread -m gpu > gpurun_out/r3_t2.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a gpurun_out/r3_t2.log
tail -5 gpurun_out/r3_t2.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -u tools/tg_bench.py reddit > gpurun_out/r3_tgbench_new.log 2>&1 && tail -3 gpurun_out/r3_tgbench_new.log && \
timeout -k 10 300 python -u ab_old/tools/tg_bench.py reddit > gpurun_out/r3_tgbench_old.log 2>&1 && tail -3 gpurun_out/r3_tgbench_old.log
