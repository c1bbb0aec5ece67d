# Collects the per-round profile artefacts on the GPU box (run through gpurun): kernel trace + the two PMC passes of the
# headline bench, condensed by scripts/prof_summary.py.  Usage: bash scripts/prof_cmd.sh <tag>
set -x
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_k -o k -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/prof_${TAG}.err
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_${TAG}_f -o f -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_${TAG}_w -o w -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
cd $R
python scripts/prof_summary.py $(find gpurun_out/prof_${TAG}_k -name "*results.db" | head -1) $(find gpurun_out/prof_${TAG}_f -name "*results.db" | head -1) $(find gpurun_out/prof_${TAG}_w -name "*results.db" | head -1) > gpurun_out/${TAG}_rocprof_summary.txt 2>&1
