set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u tools/sgc_epoch.py pubmed reddit > gpurun_out/r03_sgc_epoch.jsonl 2> gpurun_out/r03_sgc.err || { tail -n 5 gpurun_out/r03_sgc.err; exit 1; }
timeout -k 10 300 python -u tools/gcn_epoch.py pubmed reddit > gpurun_out/r03_gcn_epoch.jsonl 2> gpurun_out/r03_gcn.err || { tail -n 5 gpurun_out/r03_gcn.err; exit 1; }
cat gpurun_out/r03_sgc_epoch.jsonl gpurun_out/r03_gcn_epoch.jsonl | cut -c1-200
bash tools/profile_gpu.sh r03_reddit128_tilegroup k_tg_spmm --workload reddit128 > gpurun_out/p_r128.log 2>&1 || { tail -n 20 gpurun_out/p_r128.log; exit 1; }
echo profiled
