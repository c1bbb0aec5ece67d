"""Elastic net on a lazily standardized view (penalty_l2 route) vs the materialised copy vs the oracle: pairwise max|dbeta|."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adelie_amd as ad
from oracle import oracle as orc
rng = np.random.RandomState(23)
n, p = 420, 260
Z = np.asfortranarray(rng.normal(size=(n, p)) * rng.uniform(0.5, 3, p) + rng.normal(size=p))
M = ad.matrix.dense(Z)
S = ad.matrix.standardize(M, lazy=True)
Sm = ad.matrix.standardize(M, lazy=False)
c, s = Z.mean(0), Z.std(0)
Xs = np.asfortranarray((Z - c) / s)
c2, s2 = rng.normal(size=p), rng.uniform(0.5, 2.0, p)   # (the draws of tests/test_gpu_matrix.py::test_lazy_standardized_view)
cols = rng.choice(p, 17, replace=False)
beta = np.zeros(p); beta[rng.choice(p, 30, replace=False)] = rng.normal(size=30)
y = Xs @ beta + 0.5 * rng.normal(size=n)
w = rng.uniform(0.2, 1.0, n); pen = rng.uniform(0.5, 2.0, p)
for tol in (1e-11, 1e-13):
    for glm, extra in [(ad.glm.gaussian(y), dict(alpha=0.6)), (ad.glm.gaussian(y, weights=w / w.sum()), dict(alpha=0.3, penalty=pen)),
                       (ad.glm.gaussian(y), dict(alpha=1.0))]:
        kw = dict(tol=tol, early_exit=False, lmda_path_size=14, min_ratio=5e-2, progress_bar=False, **extra)
        a = ad.grpnet(S, glm, **kw); b = ad.grpnet(Sm, glm, **kw); o = ad.grpnet(orc.dense(Xs), glm, **kw)
        A, B, O = a.betas.toarray(), b.betas.toarray(), o.betas.toarray()
        print("screen", np.array_equal(a.screen_set, o.screen_set), "lmdas %.1e" % np.abs(a.lmdas - o.lmdas).max(), "worst row", np.abs(A - O).max(1).argmax(), "passes", a.counters["n_cd_passes_screen"], o.counters.get("n_cd_passes_screen"))
        print(tol, extra.get("alpha"), "a-b %.2e  a-o %.2e  b-o %.2e" % (np.abs(A - B).max(), np.abs(A - O).max(), np.abs(B - O).max()),
              "nnz", (A != 0).sum(), (O != 0).sum())
