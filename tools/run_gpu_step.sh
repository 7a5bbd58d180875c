set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -u tools/grouping_probe.py reddit 512 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_grouping_probe.txt || exit 1
timeout -k 10 500 python -u tools/grouping_probe.py products 512 2>&1 | grep -v amdgpu.ids >> gpurun_out/r03_grouping_probe.txt || exit 1
cat gpurun_out/r03_grouping_probe.txt
