set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u tools/shard_probe.py reddit 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_shard_probe_reddit.txt; tail -12 gpurun_out/r03_shard_probe_reddit.txt
