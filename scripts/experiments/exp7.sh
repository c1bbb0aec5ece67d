set -x
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --config 4 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/e7_cfg4.json 2> gpurun_out/e7_cfg4.err
python - <<PY
import json
d=json.loads(open("gpurun_out/e7_cfg4.json").read().strip().splitlines()[-1])
print("cfg4", round(d["value"],4), round(d["ms_per_step"],1), d["breakdown_ms_last_path"], d["roofline"]["frac"], d["roofline_sweep"]["frac"])
PY
python scripts/bench_binom_dense.py 2>&1 | tail -3
