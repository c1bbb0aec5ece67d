set -x
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_solver.py tests/test_gpu_scale.py tests/test_golden.py tests/test_cv.py -m gpu -x -q 2>&1 | tail -3
ADELIE_HIP_FUSE_REDUCE=1 timeout 1500 python -m pytest tests/test_gpu_solver.py tests/test_gpu_scale.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --no-cv-leg --no-extra-legs --steps 5 --warmup 2"
$B > gpurun_out/e9_base.json 2>gpurun_out/e9_base.err
ADELIE_HIP_FUSE_REDUCE=1 $B > gpurun_out/e9_fr.json 2>gpurun_out/e9_fr.err
$B --dtype f32 > gpurun_out/e9_f32.json 2>gpurun_out/e9_f32.err
ADELIE_HIP_FUSE_REDUCE=1 $B --dtype f32 > gpurun_out/e9_f32fr.json 2>gpurun_out/e9_f32fr.err
for f in base fr f32 f32fr; do python - <<PY
import json
d=json.loads(open("gpurun_out/e9_$f.json").read().strip().splitlines()[-1])
print("$f", round(d["value"],3), round(d["ms_per_step"],1), d["breakdown_ms_last_path"], d["roofline_panel_step"]["avg_launch_ms"] if d.get("roofline_panel_step") else None)
PY
done
