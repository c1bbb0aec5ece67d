# One gpurun call: the -m gpu suite, then one bench line per BASELINE.json config.
# Usage: bash scripts/gpu_round.sh <tag> "<pytest -k expression or empty>" [configs...]
TAG=${1:-r02}; KEXPR=${2:-}; shift; shift
CFGS=${@:-"2 3 5 4"}
mkdir -p gpurun_out
if [ -n "$KEXPR" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q -k "$KEXPR" > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/${TAG}_pytest.log
else
  timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/${TAG}_pytest.log
fi
tail -5 gpurun_out/${TAG}_pytest.log
for c in $CFGS; do
  [ "$c" = "none" ] && break
  timeout 900 python bench.py --config $c > gpurun_out/${TAG}_cfg$c.json 2> gpurun_out/${TAG}_cfg$c.err; echo "cfg $c rc=$?"
  tail -c 600 gpurun_out/${TAG}_cfg$c.err | tail -3
  cut -c1-900 gpurun_out/${TAG}_cfg$c.json
done
