cd $GRAFT_REPO_ROOT
ADELIE_HIP_TRACE_ENQ=1 python scripts/py_profile.py 2>&1 | grep -v "it/s" | grep "build\]\|run\]\|cumulative\|solver.py\|state.py\|matrix.py\|_abi.py\|function calls" | head -40
