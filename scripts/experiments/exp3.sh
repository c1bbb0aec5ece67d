set -x
cd $GRAFT_REPO_ROOT
./scripts/ubench/copy_cost
ADELIE_HIP_TRACE_ENQ=1 python scripts/host_phases.py 2>&1 | tail -8
