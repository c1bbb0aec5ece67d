import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import adelie_amd as ad
from oracle import oracle
from util import make_gaussian
d = make_gaussian(2000, 400, seed=1, sparsity=0.95)
kw = dict(groups=np.arange(0, 400, 10), alpha=0.5, early_exit=False, progress_bar=False)
a = ad.grpnet(ad.matrix.dense(d["X"]), ad.glm.gaussian(d["y"]), **kw)
b = ad.grpnet(oracle.dense(d["X"]), ad.glm.gaussian(d["y"]), **kw)
print(a.error, b.error, len(a.lmdas), len(b.lmdas))
A, B = a.betas.toarray(), b.betas.toarray()
D = np.abs(A-B).max(axis=1)
print("max diff per lambda:", np.round(D[:100:5], 9))
print("devs", np.abs(np.asarray(a.devs)-np.asarray(b.devs)).max(), "icpt", np.abs(a.intercepts-b.intercepts).max())
print("active", a.active_set_size, b.active_set_size, sorted(a.screen_set.tolist())==sorted(b.screen_set.tolist()))
print(a.counters["n_panel_blocks"], a.counters["n_cd_passes_active"], b.counters["n_cd_passes_active"])
