set -o pipefail
cd $GRAFT_REPO_ROOT
export SGNN_BENCH_REHEARSAL=1
timeout -k 10 700 python -u -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29617 bench.py --gpus 4 --steps 5 --warmup 2 > gpurun_out/rehearsal4.json 2> gpurun_out/rehearsal4.log || { tail -n 40 gpurun_out/rehearsal4.log; exit 1; }
tail -c 1500 gpurun_out/rehearsal4.json
