set -o pipefail
cd $GRAFT_REPO_ROOT
export SGNN_BENCH_REHEARSAL=1
timeout -k 10 800 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29535 bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r03_rehearsal2.json 2> gpurun_out/r03_rehearsal2.err || { tail -n 30 gpurun_out/r03_rehearsal2.err; exit 1; }
grep "\[bench\]" gpurun_out/r03_rehearsal2.err | tail -n 6
