set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u tools/combine_probe.py 0,8,16,32,64,128,0,16,32 2>&1 | grep -v amdgpu.ids
