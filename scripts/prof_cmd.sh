# Collects the per-round profile artefacts on the GPU box (run through gpurun): kernel trace + the two PMC passes of one bench
# config, condensed by scripts/prof_summary.py.  Usage: bash scripts/prof_cmd.sh <tag> [config] [extra bench args]
set -x
TAG=${1:-r03}
CFG=${2:-2}
shift; shift
EXTRA="$@"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=${TAG}_cfg${CFG}
# The kernel-trace pass of config 5 (eight host threads submitting to streams that share hardware queues) has crashed inside
# librocprofiler-sdk's queue-intercept layer in about two runs out of three (SIGSEGV at the end of a 1 MB queue ring mapping,
# below libhsa-runtime64, from whichever HIP call a fold thread happened to be in; never without the profiler attached, never in
# the PMC passes, which serialise dispatches): retry a few times.
for attempt in 1 2 3 4 5; do
  rm -rf $R/gpurun_out/prof_${N}_k
  timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${N}_k -o k -- python $R/bench.py --config $CFG --steps 1 --warmup 1 --no-cpu-baseline --no-cv-leg --no-extra-legs $EXTRA > $R/gpurun_out/${N}_bench_under_rocprof.json 2> $R/gpurun_out/prof_${N}.err
  rc=$?
  echo "kernel-trace attempt $attempt rc=$rc"
  [ $rc -eq 0 ] && break
done
timeout 900 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_${N}_f -o f -- python $R/bench.py --config $CFG --steps 1 --warmup 0 --no-cpu-baseline --no-cv-leg --no-extra-legs $EXTRA > /dev/null 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_${N}_w -o w -- python $R/bench.py --config $CFG --steps 1 --warmup 0 --no-cpu-baseline --no-cv-leg --no-extra-legs $EXTRA > /dev/null 2>&1
cd $R
python scripts/prof_summary.py $(find gpurun_out/prof_${N}_k -name "*results.db" | head -1) $(find gpurun_out/prof_${N}_f -name "*results.db" | head -1) $(find gpurun_out/prof_${N}_w -name "*results.db" | head -1) > gpurun_out/${N}_rocprof_summary.txt 2>&1
# the databases are large; only the summaries travel back
rm -rf gpurun_out/prof_${N}_k gpurun_out/prof_${N}_f gpurun_out/prof_${N}_w
