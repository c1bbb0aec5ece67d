cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-cv-leg --no-extra-legs"
for v in 160 144 128 112 160 128; do
ADELIE_HIP_STRIP_WGS=$v $B --steps 4 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 strip_wgs=$v', round(d['value'],4), round(d['ms_per_step'],1), round(d['breakdown_ms_last_path']['gram_mfma'],1), round(d['breakdown_ms_last_path']['cd'],1), round(d['roofline_panel_step']['avg_launch_ms']*1e3,2))"
done
for v in 160 144 128 112; do
ADELIE_HIP_STRIP_WGS=$v $B --config 3 --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg3 strip_wgs=$v', round(d['value'],4), round(d['ms_per_step'],1), round(d['breakdown_ms_last_path']['gram_mfma'],1), round(d['breakdown_ms_last_path']['cd'],1))"
done
for v in 512 128; do
ADELIE_HIP_STRIP_WGS=$v $B --config 5 --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 strip_wgs=$v', round(d['value'],4), round(d['ms_per_step'],1))"
ADELIE_HIP_STRIP_WGS=$v $B --dtype f32 --steps 4 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f32 strip_wgs=$v', round(d['value'],4), round(d['ms_per_step'],1))"
done
