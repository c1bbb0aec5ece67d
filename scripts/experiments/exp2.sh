# round-3 experiment 2: staging arena, opening reorder, incremental cross rows
set -x
R=$GRAFT_REPO_ROOT
cd $R
B="python bench.py --no-cpu-baseline --no-cv-leg --steps 5 --warmup 2"
timeout 1500 python -m pytest tests/test_gpu_solver.py tests/test_gpu_scale.py tests/test_golden.py tests/test_cv.py tests/test_constraint.py -m gpu -x -q 2>&1 | tail -3
$B > gpurun_out/e2_base.json 2>gpurun_out/e2_base.err
ADELIE_HIP_STAGING=0 $B > gpurun_out/e2_nostage.json 2>gpurun_out/e2_nostage.err
ADELIE_HIP_CROSS_INCR=0 $B > gpurun_out/e2_noincr.json 2>gpurun_out/e2_noincr.err
for f in base nostage noincr; do python - <<PY
import json
d=json.loads(open("gpurun_out/e2_$f.json").read().strip().splitlines()[-1])
print("$f", round(d["value"],3), round(d["ms_per_step"],1), d["breakdown_ms_last_path"], d["roofline_panel_step"]["avg_launch_ms"] if d.get("roofline_panel_step") else None)
PY
done
cd /tmp && export TMPDIR=/tmp
for tag in base; do
  rm -rf $R/gpurun_out/tl_$tag
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/tl_$tag -o k -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-cv-leg > $R/gpurun_out/tl_$tag.json 2> $R/gpurun_out/tl_$tag.err
  python $R/scripts/timeline.py $(find $R/gpurun_out/tl_$tag -name "*results.db" | head -1) > $R/gpurun_out/timeline2_$tag.txt 2>&1
  rm -rf $R/gpurun_out/tl_$tag
done
cat $R/gpurun_out/timeline2_base.txt | cut -c1-180
