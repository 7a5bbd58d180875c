cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -u tools/prefetch_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_prefetch_probe.txt
