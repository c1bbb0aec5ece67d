# FETCH_SIZE / WRITE_SIZE passes of scripts/bench_sparse.py (separate runs, as guides/MI355X_MICROARCH.md prescribes), condensed by
# scripts/prof_summary.py into gpurun_out/<tag>_sparse_pmc_summary.txt
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_sp_k -o k -- python $R/scripts/bench_sparse.py > /dev/null 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_${TAG}_sp_f -o f -- python $R/scripts/bench_sparse.py > /dev/null 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_${TAG}_sp_w -o w -- python $R/scripts/bench_sparse.py > /dev/null 2>&1
cd $R
python scripts/prof_summary.py $(find gpurun_out/prof_${TAG}_sp_k -name "*results.db" | head -1) $(find gpurun_out/prof_${TAG}_sp_f -name "*results.db" | head -1) $(find gpurun_out/prof_${TAG}_sp_w -name "*results.db" | head -1) > gpurun_out/${TAG}_sparse_pmc_summary.txt 2>&1
rm -rf gpurun_out/prof_${TAG}_sp_k gpurun_out/prof_${TAG}_sp_f gpurun_out/prof_${TAG}_sp_w
grep -E "csc_sweep_kernel|csr_axpy_kernel|csc_gram" gpurun_out/${TAG}_sparse_pmc_summary.txt | head -12
