# kernel trace of the constrained-groups path (device visits only); writes gpurun_out/<tag>_cons_rocprof_summary.txt
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_cons_k
BENCH_CONS_DEVICE_ONLY=1 timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_cons_k -o k -- python $R/scripts/bench_cons.py 100000 10000 10 200 50 > $R/gpurun_out/${TAG}_cons_bench_under_rocprof.json 2> $R/gpurun_out/prof_cons.err
cd $R
python scripts/prof_summary.py $(find gpurun_out/prof_cons_k -name "*results.db" | head -1) > gpurun_out/${TAG}_cons_rocprof_summary.txt 2>&1
rm -rf gpurun_out/prof_cons_k
head -24 gpurun_out/${TAG}_cons_rocprof_summary.txt
