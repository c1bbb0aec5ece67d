"""8-fold cv_grpnet, binomial family, dense 100k x 10k: folds in flight with and without shared sweeps."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adelie_amd as ad
from bench import make_data
X, y = make_data(100000, 10000, 0, torch.device("cuda", 0), torch.float64)
yb = (y > np.median(y)).astype(np.float64)
Xd = ad.matrix.dense(X); glm = ad.glm.binomial(yb)
for nc, sb in [(1, "1"), (3, "0"), (3, "1"), (8, "1")]:
    os.environ["ADELIE_HIP_SWEEP_BATCH"] = sb
    t0 = time.perf_counter()
    res = ad.cv_grpnet(Xd, glm, n_folds=8, seed=0, n_concurrent=nc)
    print("n_concurrent", nc, "shared sweeps", sb, "%.2f s" % (time.perf_counter() - t0), "best", res.best_idx, flush=True)
