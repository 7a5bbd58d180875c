set -o pipefail
cd $GRAFT_REPO_ROOT
start=$(date +%s)
timeout -k 10 900 python -u bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.log || { tail -n 30 gpurun_out/bench_final.log; exit 1; }
echo "wall $(( $(date +%s) - start )) s"
tail -c 600 gpurun_out/bench_final.json
