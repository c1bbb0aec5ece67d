R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tl -o k -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-cv-leg --no-extra-legs > $R/gpurun_out/tl21.json 2> $R/gpurun_out/tl21.err
python $R/scripts/timeline.py $(find /tmp/tl -name "*results.db" | head -1) --top 30 > $R/gpurun_out/timeline21.txt 2>&1
cut -c1-170 $R/gpurun_out/timeline21.txt
