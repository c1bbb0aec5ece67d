cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/scr_k
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/scr_k -o k -- python $R/scripts/dbg_screen.py 2000 50000 > $R/gpurun_out/scr.log 2>&1
cd $R
python scripts/prof_summary.py $(find gpurun_out/scr_k -name "*results.db" | head -1) 2>&1 | grep -i "screen_\|abs_grad\|calls" | head
rm -rf gpurun_out/scr_k
tail -3 gpurun_out/scr.log
for v in 1 0; do ADELIE_HIP_DEVICE_SCREEN=$v python scripts/dbg_screen.py 2000 50000 | tail -1; done
