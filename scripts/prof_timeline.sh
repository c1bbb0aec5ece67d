# idle-gap analysis of one config's kernel trace: gpurun_out/<tag>_cfg<c>_timeline.txt
TAG=${1:-r05}; CFG=${2:-2}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_tl
timeout 900 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -o k -- python $R/bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-cv-leg --no-extra-legs > /dev/null 2> $R/gpurun_out/prof_tl.err
cd $R
python scripts/timeline.py $(find gpurun_out/prof_tl -name "*results.db" | head -1) --top 22 > gpurun_out/${TAG}_cfg${CFG}_timeline.txt 2>&1
rm -rf gpurun_out/prof_tl
cat gpurun_out/${TAG}_cfg${CFG}_timeline.txt
