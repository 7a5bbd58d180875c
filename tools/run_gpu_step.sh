set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err || { tail -n 20 gpurun_out/r03_bench.err; exit 1; }
echo bench done
