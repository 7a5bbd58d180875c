#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp
for cfg in "153000 602 128 random" "153000 602 128 contiguous" "2000 602 128 random" "153000 128 64 random"; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pp; rm -rf $OUT
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/proto/project_bench.py $cfg > /dev/null 2>&1 || exit 1
  echo "== $cfg"
  grep k_batch_project $(find $OUT -name "*kernel_stats.csv") | awk -F'",' '{print substr($1,length($1)-70,40), $2,$3,$4,$5,$6,$7}'
done
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pp
