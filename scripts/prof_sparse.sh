# rocprofv3 kernel trace of scripts/bench_sparse.py (a 1M x 100k sparse-resident design), condensed into gpurun_out/<tag>_sparse_*
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
export SPARSE_STANDARDIZE=1
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/scripts/bench_sparse.py > $R/gpurun_out/${TAG}_sparse_bench.json 2> $R/gpurun_out/${TAG}_sparse_bench.err
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_sparse_k -o k -- python $R/scripts/bench_sparse.py > $R/gpurun_out/${TAG}_sparse_bench_under_rocprof.json 2> $R/gpurun_out/prof_${TAG}_sparse.err
cd $R
python scripts/prof_summary.py $(find gpurun_out/prof_${TAG}_sparse_k -name "*results.db" | head -1) > gpurun_out/${TAG}_sparse_rocprof_summary.txt 2>&1
rm -rf gpurun_out/prof_${TAG}_sparse_k
tail -c 1500 gpurun_out/${TAG}_sparse_bench.json
head -30 gpurun_out/${TAG}_sparse_rocprof_summary.txt
