cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-cv-leg --no-extra-legs"
for v in 1 0; do
for c in 3 4 5; do
st=2; [ $c = 4 ] && st=1
ADELIE_HIP_DEVICE_SCREEN=$v $B --config $c --steps $st --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg$c devscreen=$v', round(d['value'],4), round(d['ms_per_step'],1), {k: round(v,1) for k,v in d['breakdown_ms_last_path'].items()}, d['counters'].get('n_device_screens'), d['counters'].get('n_host_screens'))"
done
done
