"""IRLS on a design kept sparse (full-Gram engines): the reuse threshold of the screen set's Gram (ADELIE_HIP_IRLS_REUSE, read at
the start of every solve) against always-rebuild: wall, Gram time, passes, IRLS iterations, max|dbeta| over the path.

    python scripts/sparse_irls_reuse.py [n p density lambdas]      (default 1000000 100000 0.001 50)
"""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adelie_amd as ad

n, p, dens, L = (int(float(sys.argv[1])), int(float(sys.argv[2])), float(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 \
    else (1_000_000, 100_000, 1e-3, 50)
rng = np.random.default_rng(0)
nnz = int(n * p * dens)
M = sp.csc_matrix((rng.normal(size=nnz), (rng.integers(0, n, size=nnz, dtype=np.int64), rng.integers(0, p, size=nnz, dtype=np.int64))),
                  shape=(n, p))
M.sum_duplicates()
M.sort_indices()
beta = np.zeros(p)
beta[rng.choice(p, 50, replace=False)] = rng.normal(size=50) * 3
y = M @ beta + rng.normal(size=n)
yb = (rng.uniform(size=n) < 1 / (1 + np.exp(-(y - y.mean()) / y.std()))).astype(float)
X = ad.matrix.sparse(M, resident="csc")
kw = dict(lmda_path_size=L, min_ratio=5e-2, early_exit=False, progress_bar=False)
ref = None
for theta in [float(t) for t in os.environ.get("THETAS", "0,0.1,0.3,1,3").split(",")]:
    os.environ["ADELIE_HIP_IRLS_REUSE"] = repr(theta)
    t0 = time.time()
    st = ad.grpnet(X, ad.glm.binomial(yb), **kw)
    el = time.time() - t0
    B = st.betas.toarray()
    if ref is None:
        ref = B
    print(json.dumps({"theta": theta, "path_s": round(el, 3), "gram_ms": round(st.timers["t_gram_ms"], 1), "cd_ms": round(st.timers["t_cd_ms"], 1),
                      "axpy_ms": round(st.timers["t_axpy_ms"], 1), "irls": int(st.counters["n_irls_iters"]),
                      "passes": int(st.counters["n_cd_passes_screen"] + st.counters["n_cd_passes_active"]),
                      "final_active": int(st.active_set_size), "same_active_sets": bool(np.array_equal(B != 0, ref != 0)),
                      "max_abs_dbeta_vs_first": float(np.abs(B - ref).max()), "max_abs_dintercept": float(np.abs(st.intercepts - (st.intercepts if ref is B else ref_i)).max()),
                      "error": st.error}), flush=True)
    if ref is B:
        ref_i = st.intercepts.copy()
