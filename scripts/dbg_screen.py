"""Timing aid for the device screening kernels: a wide, shallow lasso path (many groups, few rows)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adelie_amd as ad
n, p = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(0)
X = np.asfortranarray(rng.standard_normal((n, p)))
beta = np.zeros(p); beta[rng.choice(p, 300, replace=False)] = rng.standard_normal(300)
y = X @ beta + rng.standard_normal(n)
Xd = ad.matrix.dense(X)
for it in range(2):
    t0 = time.perf_counter()
    st = ad.grpnet(Xd, ad.glm.gaussian(y), lmda_path_size=50, min_ratio=0.05, early_exit=False, progress_bar=False)
    print(time.perf_counter() - t0, st.counters["n_device_screens"], st.counters["n_host_screens"], len(st.screen_set))
