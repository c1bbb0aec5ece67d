# round-3 experiment 1: fuse_reduce with slice-major partials, timeline of the headline path
set -x
R=$GRAFT_REPO_ROOT
cd $R
B="python bench.py --no-cpu-baseline --no-cv-leg --steps 5 --warmup 2"
timeout 900 python -m pytest tests/test_gpu_solver.py -m gpu -x -q 2>&1 | tail -3
ADELIE_HIP_FUSE_REDUCE=1 timeout 900 python -m pytest tests/test_gpu_solver.py tests/test_gpu_scale.py -m gpu -x -q 2>&1 | tail -3
$B > gpurun_out/e1_base.json 2>gpurun_out/e1_base.err
ADELIE_HIP_FUSE_REDUCE=1 $B > gpurun_out/e1_fr.json 2>gpurun_out/e1_fr.err
ADELIE_HIP_SPECULATE=0 $B > gpurun_out/e1_nospec.json 2>gpurun_out/e1_nospec.err
ADELIE_HIP_SIDE_GRAMS=0 $B > gpurun_out/e1_noside.json 2>gpurun_out/e1_noside.err
for f in base fr nospec noside; do python - <<PY
import json
d=json.loads(open("gpurun_out/e1_$f.json").read().strip().splitlines()[-1])
print("$f", round(d["value"],3), round(d["ms_per_step"],1), d["breakdown_ms_last_path"], d["roofline_panel_step"]["avg_launch_ms"] if d.get("roofline_panel_step") else None)
PY
done
cd /tmp && export TMPDIR=/tmp
for tag in base fr; do
  [ $tag = fr ] && export ADELIE_HIP_FUSE_REDUCE=1
  rm -rf $R/gpurun_out/tl_$tag
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/tl_$tag -o k -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-cv-leg > $R/gpurun_out/tl_$tag.json 2> $R/gpurun_out/tl_$tag.err
  python $R/scripts/timeline.py $(find $R/gpurun_out/tl_$tag -name "*results.db" | head -1) > $R/gpurun_out/timeline_$tag.txt 2>&1
  rm -rf $R/gpurun_out/tl_$tag
done
tail -60 $R/gpurun_out/timeline_base.txt
