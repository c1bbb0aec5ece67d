#!/bin/bash
# other BASELINE.json configs at scale (not the headline line): group size 10 / alpha 0.5, f32, 8-fold CV on one GPU
set -x
python bench.py --steps 1 --warmup 1 --no-cpu-baseline --group-size 10 --alpha 0.5 2>&1 | tail -1 | cut -c1-1500
python bench.py --steps 1 --warmup 1 --no-cpu-baseline --dtype f32 2>&1 | tail -1 | cut -c1-1200
python - <<'PY'
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import adelie_amd as ad
from bench import make_data
X, y = make_data(100000, 10000, 0, torch.device("cuda", 0), torch.float64)
Xd = ad.matrix.dense(X)
t0 = time.perf_counter()
res = ad.cv_grpnet(Xd, ad.glm.gaussian(y), n_folds=8, seed=0)
print("cv_grpnet 8 folds 100k x 10k f64: %.2f s, best_idx %d, loss range %.4f..%.4f" % (time.perf_counter() - t0, res.best_idx, res.avg_losses.min(), res.avg_losses.max()))
PY
