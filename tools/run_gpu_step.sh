set -o pipefail
cd $GRAFT_REPO_ROOT
for seg in 0 288 384 512; do TG_SEG=$seg timeout -k 10 200 python -u tools/ab_narrow.py sgnn_amd/libsgnn_hip.so 16 41 76 2>&1 | grep -v amdgpu.ids || exit 1; done
