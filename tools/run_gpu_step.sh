set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u tools/policy_probe.py > gpurun_out/r03_policy_probe.txt 2>&1; echo policy rc=$?; grep -v amdgpu.ids gpurun_out/r03_policy_probe.txt
SGNN_BENCH_REHEARSAL=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 4 --steps 5 --warmup 2 > gpurun_out/r03_rehearsal4.json 2> gpurun_out/r03_rehearsal4.err; echo rehearsal rc=$?; tail -8 gpurun_out/r03_rehearsal4.err; python3 -c "
import json
d=json.loads(open('gpurun_out/r03_rehearsal4.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['n_gpus']); print(json.dumps(d['config'], indent=0)[:2500])
"
