set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r03_full_tests.log 2>&1; echo rc=$?; tail -n 3 gpurun_out/r03_full_tests.log
timeout -k 10 300 python -u -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -n 2
