set -x
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_solver.py tests/test_gpu_scale.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -2
B="python bench.py --no-cpu-baseline --no-cv-leg --no-extra-legs --steps 5 --warmup 2"
$B > gpurun_out/e10_base.json 2>gpurun_out/e10_base.err
ADELIE_HIP_STEP_REPS=2 $B > gpurun_out/e10_r2.json 2>gpurun_out/e10_r2.err
ADELIE_HIP_STEP_REPS=2 ADELIE_HIP_SIDE_GRAMS=0 $B > gpurun_out/e10_r2ns.json 2>gpurun_out/e10_r2ns.err
ADELIE_HIP_SIDE_GRAMS=0 $B > gpurun_out/e10_ns.json 2>gpurun_out/e10_ns.err
for f in base r2 r2ns ns; do python - <<PY
import json
d=json.loads(open("gpurun_out/e10_$f.json").read().strip().splitlines()[-1])
print("$f", round(d["value"],3), round(d["ms_per_step"],1), d["breakdown_ms_last_path"], d["roofline_panel_step"]["avg_launch_ms"] if d.get("roofline_panel_step") else None)
PY
done
