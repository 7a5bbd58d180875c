set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_gpu_tilegroup.py tests/test_gpu_planes_bound.py -x -q --timeout 180 --timeout-method thread -m gpu 2>&1 | tail -3 || exit 1
bash tools/profile_gpu.sh r03_reddit_f602_tilegroup k_tg_spmm > gpurun_out/p1.log 2>&1 && echo reddit ok && \
bash tools/profile_gpu.sh r03_reddit_order2_tilegroup k_tg_spmm --order 2 > gpurun_out/p4.log 2>&1 && echo order2 ok
head -12 gpurun_out/r03_reddit_order2_tilegroup/kernel_stats.csv | cut -c1-120
