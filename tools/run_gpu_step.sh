cd $GRAFT_REPO_ROOT
timeout -k 10 120 ./tools/proto/copy_probe > gpurun_out/r3_copy_probe.txt 2>&1; cat gpurun_out/r3_copy_probe.txt
