set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r1 -o k -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_r1_bench.json 2> $R/gpurun_out/prof_r1.err
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_r1f -o f -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_r1w -o w -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
cd $R
python scripts/prof_summary.py $(find gpurun_out/prof_r1 -name "*results.db" | head -1) $(find gpurun_out/prof_r1f -name "*results.db" | head -1) $(find gpurun_out/prof_r1w -name "*results.db" | head -1) > gpurun_out/r01b_rocprof_summary.txt 2>&1
tail -1 gpurun_out/prof_r1_bench.json | head -c 600
