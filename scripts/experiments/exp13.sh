cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/e13.log 2>&1
tail -5 gpurun_out/e13.log | cut -c1-250
