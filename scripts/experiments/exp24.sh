R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p3
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p3 -- python $R/bench.py --config 3 --steps 2 --warmup 1 --no-cpu-baseline --no-cv-leg --no-extra-legs > $R/gpurun_out/p3.json 2> $R/gpurun_out/p3.err
python3 $R/scripts/kstats.py /tmp/p3 | head -24
