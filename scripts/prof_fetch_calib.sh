# FETCH_SIZE calibration: known-byte-count kernels under the PMC pass; writes gpurun_out/<tag>_fetch_calib.txt
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_fc
$R/scripts/ubench/fetch_calib > $R/gpurun_out/${TAG}_fetch_calib.txt 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_fc -o f -- $R/scripts/ubench/fetch_calib > /dev/null 2>&1
cd $R
python - >> gpurun_out/${TAG}_fetch_calib.txt <<PY
import glob, sqlite3
db = glob.glob("gpurun_out/prof_fc/**/*results.db", recursive=True)[0]
c = sqlite3.connect(db)
known = 125056 * 16384 / 1024.0
print("\n# rocprofv3 --pmc FETCH_SIZE, per dispatch (KB as reported); known = %.1f KB" % known)
for name, n, av in c.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name = 'FETCH_SIZE' group by kernel_name"):
    print("%-60s n=%d FETCH_SIZE=%.1f KB  reported/known=%.3f" % (name[:60], n, av, av / known))
PY
rm -rf gpurun_out/prof_fc
cat gpurun_out/${TAG}_fetch_calib.txt
