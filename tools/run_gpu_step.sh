set -o pipefail
cd $GRAFT_REPO_ROOT
export SGNN_BENCH_REHEARSAL=1
timeout -k 10 800 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 4 --steps 5 --warmup 2 > gpurun_out/r03_rehearsal4.json 2> gpurun_out/r03_rehearsal4.err || { tail -30 gpurun_out/r03_rehearsal4.err; exit 1; }
grep "\[bench\]" gpurun_out/r03_rehearsal4.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_rehearsal4.json').read().strip().splitlines()[-1])
print(json.dumps(d['config'])[:1800]); print({k:v for k,v in d['roofline'].items() if k!='kernels_in_step'}); print(d['value'], d['ms_per_step'])
PY
