"""cv_grpnet on a lazily standardized 2-bit view (grouped / no-intercept fits: the panel form of the view) and on a sparse-resident
design under a binomial response (the panel engine over compressed columns), folds in flight on alias handles, against the
oracle on the dense copy: loss tables and best indices."""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import adelie_amd as ad
from oracle import oracle as orc
import scipy.sparse as sp
rng = np.random.RandomState(4)
n, p = 600, 260
cd = rng.choice(np.array([0, 1, 2, -9], dtype=np.int8), size=(n, p), p=[0.5, 0.3, 0.1, 0.1])
imp = ad.matrix.compute_impute(cd)
Z = np.where(cd < 0, imp[None], cd).astype(float)
M = ad.matrix.snp_calldata(cd, imp)
S = ad.matrix.standardize(M, lazy=True)
Xs = np.asfortranarray((Z - Z.mean(0)) / Z.std(0))
beta = np.zeros(p); beta[rng.choice(p, 20, replace=False)] = rng.normal(size=20)
y = Xs @ beta + 0.5 * rng.normal(size=n)
os.environ["ADELIE_HIP_CD_BLOCK_MIN_NV"] = "1"
for kw in [dict(groups=np.arange(0, p, 4), alpha=0.6), dict(intercept=False)]:
    kwc = dict(n_folds=4, seed=2, lmda_path_size=10, min_ratio=0.05, progress_bar=False, tol=1e-12, **kw)
    a = ad.cv_grpnet(S, ad.glm.gaussian(y), **kwc)
    b = ad.cv_grpnet(orc.dense(Xs), ad.glm.gaussian(y), **kwc)
    print(kw.keys(), "cv losses max rel diff %.2e" % (np.abs(a.losses - b.losses).max() / np.abs(b.losses).max()), "best", a.best_idx, b.best_idx)
D = rng.normal(size=(n, p)) * (rng.uniform(size=(n, p)) < 0.1)
yb = (D[:, :5] @ rng.normal(size=5) + rng.normal(size=n) > 0).astype(float)
Xsp = ad.matrix.sparse(sp.csc_matrix(D), resident="csc")
kwc = dict(n_folds=4, seed=2, lmda_path_size=10, min_ratio=0.1, progress_bar=False, tol=1e-12, irls_tol=1e-12)
a = ad.cv_grpnet(Xsp, ad.glm.binomial(yb), **kwc)
b = ad.cv_grpnet(orc.dense(np.asfortranarray(D)), ad.glm.binomial(yb), **kwc)
print("sparse binomial cv losses max rel diff %.2e" % (np.abs(a.losses - b.losses).max() / np.abs(b.losses).max()), "best", a.best_idx, b.best_idx)
