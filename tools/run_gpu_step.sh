set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests/test_gpu_tilegroup.py tests/test_gpu_planes_bound.py tests/test_gpu_dist.py tests/test_gpu_fullsize.py -x -q --timeout 200 --timeout-method thread > gpurun_out/t_tg.log 2>&1; rc=$?; tail -n 2 gpurun_out/t_tg.log; [ $rc -eq 0 ] || { grep -E "^E |FAILED" gpurun_out/t_tg.log | head; exit 1; }
for v in ab_old.so sgnn_amd/libsgnn_hip.so ab_old.so sgnn_amd/libsgnn_hip.so; do timeout -k 10 200 python -u tools/ab_lib.py $v 2>&1 | grep -v amdgpu.ids | cut -c1-70 || exit 1; done
