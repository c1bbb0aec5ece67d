import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adelie_amd as ad
from bench import make_data
X, y = make_data(100000, 10000, 0, torch.device("cuda", 0), torch.float64)
Xd = ad.matrix.dense(X); glm = ad.glm.gaussian(y)
ref = None
for nc in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 1]:
    t0 = time.perf_counter()
    res = ad.cv_grpnet(Xd, glm, n_folds=8, seed=0, n_concurrent=nc)
    el = time.perf_counter() - t0
    if ref is None: ref = res.losses
    print("n_concurrent", nc, "%.2f s" % el, "best", res.best_idx, "max|dloss| vs nc=1 %.2e" % np.abs(res.losses - ref).max(), flush=True)
