set -x
cd $GRAFT_REPO_ROOT
timeout 900 python scripts/irls_reuse.py 100000 10000 0 0.001 0.01 0.05 0.2 2>&1 | grep -v "it/s" | grep "theta\|enq\]"
timeout 1500 python scripts/irls_reuse.py 500000 50000 0 0.01 0.05 0.2 2>&1 | grep -v "it/s" | grep "theta\|enq\]"
