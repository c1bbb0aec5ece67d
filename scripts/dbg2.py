import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import adelie_amd as ad
from oracle import oracle
from util import make_gaussian
d = make_gaussian(200, 80, seed=3)
ref = ad.grpnet(oracle.dense(d["X"]), ad.glm.gaussian(d["y"]), lmda_path_size=0)
for L in [4,5,6]:
    path = ref.lmda_max * np.array([2.0, 1.5, 1.0, 0.7, 0.4, 0.2, 0.1])[:L]
    kw=dict(lmda_path=path, screen_rule="pivot", early_exit=False, tol=1e-10)
    a = ad.grpnet(ad.matrix.dense(d["X"]), ad.glm.gaussian(d["y"]), **kw)
    b = ad.grpnet(oracle.dense(d["X"]), ad.glm.gaussian(d["y"]), **kw)
    print(L, "screen equal", np.array_equal(a.screen_set,b.screen_set), len(a.screen_set), len(b.screen_set), "active eq", np.array_equal(a.active_set[:a.active_set_size], b.active_set[:b.active_set_size]))
    print("  ", a.screen_set[:20], b.screen_set[:20])
    print("  absgrad diff", np.abs(a.abs_grad-b.abs_grad).max(), "nvalid", a.n_valid_solutions, b.n_valid_solutions, a.screen_sizes, b.screen_sizes)
    print("  upd", a.counters["n_updates"], b.counters["n_updates"], np.abs(a.betas.toarray()-b.betas.toarray()).max(1))
