set -o pipefail
cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; bash tools/profile_gpu.sh $tag k_tg_spmm "$@" > gpurun_out/p_$tag.log 2>&1 || { tail -n 20 gpurun_out/p_$tag.log; exit 1; }; echo "$tag done"; }
run r03_reddit_f602_tilegroup
run r03_reddit_w2_tilegroup --weight 2
run r03_reddit_clip22000_tilegroup --max-degree 22000
run r03_reddit_order2_tilegroup --order 2
run r03_reddit128_tilegroup --workload reddit128
run r03_products_tilegroup --workload products
run r03_rmat10m_tilegroup --workload rmat10m
timeout -k 10 900 python -u bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err || { tail -n 20 gpurun_out/r03_bench.err; exit 1; }
echo bench done
