set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -u -m pytest tests/test_gpu_tilegroup.py tests/test_gpu_planes_bound.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3 || exit 1
timeout -k 10 300 python -u tools/xcd_probe.py products 2>&1 | grep -v amdgpu.ids | tail -4 || exit 1
timeout -k 10 300 python -u tools/xcd_probe.py reddit 2>&1 | grep -v amdgpu.ids | tail -4 || exit 1
timeout -k 10 300 python -u tools/xcd_probe.py rmat10m 2>&1 | grep -v amdgpu.ids | tail -4
