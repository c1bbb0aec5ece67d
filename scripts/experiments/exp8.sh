set -x
cd $GRAFT_REPO_ROOT
for cfg in "ADELIE_HIP_PANEL_BSZ=128" "ADELIE_HIP_PANEL_BSZ=128 ADELIE_HIP_SIDE_STREAMS=1" "ADELIE_HIP_SIDE_STREAMS=1" "ADELIE_HIP_SIDE_STREAMS=3" "ADELIE_HIP_PREBUILD=0"; do
  echo "== $cfg"
  env $cfg timeout 900 python scripts/irls_reuse.py 500000 50000 0.01 2>&1 | grep "theta"
done
