set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests/test_gpu_tilegroup.py tests/test_gpu_planes_bound.py tests/test_gpu_fullsize.py -x -q --timeout 300 --timeout-method thread 2>&1 | tail -3 || exit 1
timeout -k 10 300 python -u tools/chain_probe.py reddit 2>&1 | grep -v amdgpu.ids | tail -2
