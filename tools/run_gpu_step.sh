set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -5
