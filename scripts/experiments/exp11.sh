set -x
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --no-cv-leg --no-extra-legs"
$B --config 3 --steps 3 --warmup 1 > gpurun_out/e11_cfg3.json 2>gpurun_out/e11_cfg3.err
ADELIE_HIP_FUSE_REDUCE=0 $B --config 3 --steps 3 --warmup 1 > gpurun_out/e11_cfg3_nofr.json 2>gpurun_out/e11_cfg3_nofr.err
for f in cfg3 cfg3_nofr; do python - <<PY
import json
d=json.loads(open("gpurun_out/e11_$f.json").read().strip().splitlines()[-1])
print("$f", round(d["value"],3), round(d["ms_per_step"],1), d["breakdown_ms_last_path"], d["roofline_panel_step"]["avg_launch_ms"] if d.get("roofline_panel_step") else None)
PY
done
python scripts/bench_multi.py 100000 10000 4 2>&1 | tail -2
