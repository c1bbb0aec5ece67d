"""Concurrent fused GROUP launches (eight cv_grpnet folds in flight, each with its spread tail reduce spinning on its own
counter) and a few repeated full-size paths: a hang or a divergence between repeats would show here.  Run under `timeout`."""
import os, sys, time, hashlib
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adelie_amd as ad
from bench import make_data
X, y = make_data(100000, 4000, 0, torch.device("cuda", 0), torch.float64)
Xd = ad.matrix.dense(X)
groups = np.arange(0, 4000, 10)
tabs = []
for rep in range(3):
    t0 = time.perf_counter()
    res = ad.cv_grpnet(Xd, ad.glm.gaussian(y), groups=groups, alpha=0.5, n_folds=8, seed=0, min_ratio=0.05, progress_bar=False)
    tabs.append(np.array(res.losses))
    print("grouped cv rep %d: %.2f s, best_idx %d, max rel diff to rep 0 %.2e" % (
        rep, time.perf_counter() - t0, res.best_idx, np.abs(tabs[-1] - tabs[0]).max() / np.abs(tabs[0]).max()), flush=True)
# (folds in flight share their full-gradient sweeps in whatever groups the timing gives: the K-wide kernel and the single sweep
#  sum a column's rows in different orders, so repeats agree to rounding, not bit for bit; with ADELIE_HIP_SWEEP_BATCH=0 they do)
assert max(np.abs(t - tabs[0]).max() for t in tabs) <= 1e-9 * np.abs(tabs[0]).max()
hs = []
for rep in range(4):
    t0 = time.perf_counter()
    st = ad.grpnet(Xd, ad.glm.gaussian(y), groups=groups, alpha=0.5, early_exit=False, progress_bar=False)
    hs.append(hashlib.sha1(st.betas.toarray().tobytes()).hexdigest()[:12])
    print("grouped path rep %d: %.3f s, blocks %d, hash %s" % (rep, time.perf_counter() - t0, st.counters["n_panel_blocks"], hs[-1]), flush=True)
assert len(set(hs)) == 1, hs
print("soak ok")
