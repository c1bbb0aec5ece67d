import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import adelie_amd as ad
from oracle import oracle
from util import make_gaussian
d = make_gaussian(200, 80, seed=3)
ref = ad.grpnet(oracle.dense(d["X"]), ad.glm.gaussian(d["y"]), lmda_path_size=0)
path = ref.lmda_max * np.array([2.0, 1.5, 1.0, 0.7, 0.4, 0.2, 0.1])
for tol in [1e-10, 1e-14]:
  for rule in ("strong","pivot"):
    kw=dict(lmda_path=path, screen_rule=rule, early_exit=False, tol=tol)
    a = ad.grpnet(ad.matrix.dense(d["X"]), ad.glm.gaussian(d["y"]), **kw)
    b = ad.grpnet(oracle.dense(d["X"]), ad.glm.gaussian(d["y"]), **kw)
    D = np.abs(a.betas.toarray()-b.betas.toarray()).max(1)
    print(tol, rule, D, a.counters["n_updates"], b.counters["n_updates"], a.counters["n_cd_passes_active"], b.counters["n_cd_passes_active"], a.y_var)
