import sys, os, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
os.environ["ADELIE_HIP_CD_BLOCK_MIN_NV"] = sys.argv[1] if len(sys.argv) > 1 else "1"
import adelie_amd as ad
from oracle import oracle
from util import make_gaussian
n, p, alpha = 1500, 700, 0.6
d = make_gaussian(n, p, seed=11, sparsity=0.5, weights=True)
kw = dict(alpha=alpha, tol=1e-10, early_exit=False, lmda_path_size=25, min_ratio=1e-2)
glm = lambda: ad.glm.gaussian(d["y"], weights=d["weights"])
a = ad.grpnet(ad.matrix.dense(d["X"]), glm(), **kw)
b = ad.grpnet(oracle.dense(d["X"]), glm(), **kw)
D = np.abs(a.betas.toarray()-b.betas.toarray()).max(1)
print("err", a.error, b.error, len(a.lmdas), len(b.lmdas))
print(D)
print(a.counters["n_updates"], b.counters["n_updates"], a.active_sizes, b.active_sizes, np.array_equal(a.screen_set,b.screen_set))
