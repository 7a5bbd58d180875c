set -o pipefail
cd $GRAFT_REPO_ROOT
for v in ab_old.so sgnn_amd/libsgnn_hip.so ab_old.so sgnn_amd/libsgnn_hip.so; do timeout -k 10 200 python -u tools/ab_lib.py $v 2>&1 | grep -v amdgpu.ids | cut -c1-120 || exit 1; done
