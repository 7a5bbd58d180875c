set -o pipefail
cd $GRAFT_REPO_ROOT
bash tools/profile_gpu.sh r03_products_tilegroup k_tg_spmm --workload products > gpurun_out/p_products.log 2>&1 || { tail -20 gpurun_out/p_products.log; exit 1; }
bash tools/profile_gpu.sh r03_rmat10m_tilegroup k_tg_spmm --workload rmat10m > gpurun_out/p_rmat10m.log 2>&1 || { tail -20 gpurun_out/p_rmat10m.log; exit 1; }
tail -5 gpurun_out/p_products.log gpurun_out/p_rmat10m.log
