"""A design that does not fit dense, kept sparse in HBM (matrix.sparse(resident="csc")): wall of a Gaussian lasso path,
the full-gradient sweep's rate over the stored entries, and (when n*p*8 fits) the same path on the expanded dense copy.

    python scripts/bench_sparse.py [n p density lambdas]      (default 1000000 100000 0.001 100)
"""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adelie_amd as ad

n, p, dens, L = (int(float(sys.argv[1])), int(float(sys.argv[2])), float(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 \
    else (1_000_000, 100_000, 1e-3, 100)
rng = np.random.default_rng(0)
nnz = int(n * p * dens)
t0 = time.time()
rows = rng.integers(0, n, size=nnz, dtype=np.int64)
cols = rng.integers(0, p, size=nnz, dtype=np.int64)
vals = rng.normal(size=nnz)
M = sp.csc_matrix((vals, (rows, cols)), shape=(n, p))
M.sum_duplicates()
M.sort_indices()
del rows, cols, vals
beta = np.zeros(p)
beta[rng.choice(p, 50, replace=False)] = rng.normal(size=50) * 3
y = M @ beta + rng.normal(size=n)
t_gen = time.time() - t0
t0 = time.time()
X = ad.matrix.sparse(M, resident="csc")
t_up = time.time() - t0
kw = dict(lmda_path_size=L, min_ratio=1e-2, early_exit=False, progress_bar=False)
ad.grpnet(X, ad.glm.gaussian(y), lmda_path_size=5, min_ratio=0.5, early_exit=False, progress_bar=False)  # warm-up
t0 = time.time()
st = ad.grpnet(X, ad.glm.gaussian(y), **kw)
t_path = time.time() - t0
assert st.error == "", st.error
# the sweep alone: X.mul through the C ABI includes two host copies of n / p values; time many and subtract nothing
v, w, out = rng.normal(size=n), np.full(n, 1.0 / n), np.empty(p)
X.mul(v, w, out)
t0 = time.time()
R = 10
for _ in range(R):
    X.mul(v, w, out)
t_mul = (time.time() - t0) / R
ref = (M.T @ (v * w))
res = {
    "workload": f"Gaussian lasso, sparse design {n}x{p}, {M.nnz} stored entries ({M.nnz / (n * p):.2e} of the cells), "
                f"{L} lambdas, early_exit=False; dense form would be {n * p * 8 / 2**30:.0f} GiB",
    "resident_bytes": int(M.nnz * 24 + (n + p + 2) * 8),
    "path_s": t_path, "lambdas": len(st.lmdas), "final_active": int(st.active_set_size), "final_screen": int(len(st.screen_set)),
    "X_mul_ms_incl_host_copies": t_mul * 1e3,
    "sweep_algorithmic_GBps_lower_bound": (M.nnz * 12 + n * 8) / t_mul / 1e9,
    "mul_max_abs_err_vs_scipy": float(np.abs(out - ref).max()),
    "dev_ratio_last": float(st.devs[-1]),
    "generate_s": t_gen, "upload_s": t_up,
}
if os.environ.get("SPARSE_STANDARDIZE"):  # the standardized view of the same design (entries shared, epilogue corrections)
    t0 = time.time()
    Z = ad.matrix.standardize(X)
    t_std = time.time() - t0
    ad.grpnet(Z, ad.glm.gaussian(y), lmda_path_size=5, min_ratio=0.5, early_exit=False, progress_bar=False)
    t0 = time.time()
    sz = ad.grpnet(Z, ad.glm.gaussian(y), **kw)
    res["standardized"] = {"path_s": time.time() - t0, "lambdas": len(sz.lmdas), "final_active": int(sz.active_set_size),
                           "centers_scales_s": t_std, "error": sz.error}
if os.environ.get("SPARSE_BINOMIAL"):  # IRLS on the same design: the screen set's Gram is rebuilt under every iteration's weights
    yb = (rng.uniform(size=n) < 1 / (1 + np.exp(-(y - y.mean()) / y.std()))).astype(float)
    kwb = dict(lmda_path_size=int(os.environ["SPARSE_BINOMIAL"]), min_ratio=5e-2, early_exit=False, progress_bar=False)
    t0 = time.time()
    sb = ad.grpnet(X, ad.glm.binomial(yb), **kwb)
    res["binomial"] = {"path_s": time.time() - t0, "lambdas": len(sb.lmdas), "final_active": int(sb.active_set_size),
                       "error": sb.error, "timers": {k: round(float(v), 1) for k, v in sb.timers.items()},
                       "counters": {k: int(sb.counters[k]) for k in ("n_irls_iters", "n_cd_passes_screen", "n_cd_passes_active",
                                                                       "n_cd_visits_screen", "n_cd_visits_active", "n_panel_blocks")}}
if n * p * 8 < 100 * 2**30:
    Xd = ad.matrix.sparse(M, resident="dense")
    ad.grpnet(Xd, ad.glm.gaussian(y), lmda_path_size=5, min_ratio=0.5, early_exit=False, progress_bar=False)
    t0 = time.time()
    sd = ad.grpnet(Xd, ad.glm.gaussian(y), **kw)
    res["dense_path_s"] = time.time() - t0
    res["max_abs_dbeta_vs_dense"] = float(np.abs(st.betas.toarray() - sd.betas.toarray()).max())
print(json.dumps(res))
