cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_constraint.py -m gpu -x -q > gpurun_out/e12.log 2>&1
tail -25 gpurun_out/e12.log | cut -c1-250
