set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_gpu_tilegroup.py -x -q --timeout 180 --timeout-method thread -m gpu 2>&1 | tail -3 || exit 1
for wl in reddit products; do timeout -k 10 400 python -u tools/xcd_probe.py $wl 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03_xcd_probe.txt || exit 1; done
