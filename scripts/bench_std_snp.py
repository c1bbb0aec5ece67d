"""matrix.standardize over a 2-bit SNP design, lazily (the view shares the 2-bit matrix: 2 bits per call) and materialised
(a dense f64 copy: 8 bytes per call): wall of a Gaussian lasso path on both, resident bytes, max|dbeta|.

    python scripts/bench_std_snp.py [n p lambdas]      (default 200000 20000 100)

ALPHA=0.5: an elastic net (round 5: runs on the base design's panel engines through penalty_l2); VIEW_ENGINES=1 forces the view's
own full-Gram engines (an exit_cond that never fires keeps grpnet in the view's coordinates) for a before / after on one build;
BINOMIAL=1 adds config 4's response; SKIP_COPY=1 leaves the materialised copy out.
"""
import json
import os
import sys
import time

import numpy as np
import torch

import adelie_amd as ad

n, p, L = (int(float(sys.argv[1])), int(float(sys.argv[2])), int(sys.argv[3])) if len(sys.argv) > 3 else (200_000, 20_000, 100)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (the calldata generator of config 4: rates of adelie.data.snp_unphased, built by column chunks)

cd, imp, y_bin = bench.make_snp_data(n, p, 0, torch.device("cuda", 0))
X = ad.matrix.snp_calldata(cd, imp)
del cd
torch.cuda.empty_cache()
rng = np.random.default_rng(0)
beta = np.zeros(p)
beta[rng.choice(p, 50, replace=False)] = rng.normal(size=50)
eta = np.zeros(n)
X.btmul(0, p, beta, eta)
y = eta + float(os.environ.get("NOISE", "1.0")) * np.std(eta) * rng.normal(size=n)
alpha = float(os.environ.get("ALPHA", "1"))
kw = dict(lmda_path_size=L, min_ratio=float(os.environ.get("MIN_RATIO", "2e-2")), early_exit=False, progress_bar=False, alpha=alpha)
if os.environ.get("GROUP_SIZE"):        # groups of GROUP_SIZE columns (not GROUPS: bash keeps that name for itself) / INTERCEPT=0: fits the base-coordinates route cannot take (round 5: panel form of the view)
    kw["groups"] = np.arange(0, p, int(os.environ["GROUP_SIZE"]))
if os.environ.get("INTERCEPT") == "0":
    kw["intercept"] = False
if os.environ.get("VIEW_ENGINES"):
    kw["exit_cond"] = lambda state: False
res = {"workload": f"Gaussian {'lasso' if alpha == 1 else 'elastic net alpha=%g' % alpha} on standardize(snp {n}x{p}), {L} lambdas", "view_engines": bool(os.environ.get("VIEW_ENGINES")), "bytes_2bit": int(n * p / 4), "bytes_dense_copy": int(n * p * 8)}
for name, lazy in (("lazy_view", True), ("materialised", False)):
    if not lazy and (n * p * 8 > 150 * 2**30 or os.environ.get("SKIP_COPY")):
        continue
    t0 = time.time()
    Z = ad.matrix.standardize(X, lazy=lazy)
    t_make = time.time() - t0
    ad.grpnet(Z, ad.glm.gaussian(y), lmda_path_size=5, min_ratio=0.5, early_exit=False, progress_bar=False)
    t0 = time.time()
    st = ad.grpnet(Z, ad.glm.gaussian(y), **kw)
    res[name] = {"path_s": time.time() - t0, "make_s": t_make, "final_active": int(st.active_set_size), "error": st.error,
                 "n_panel_blocks": int(st.counters["n_panel_blocks"])}
    if lazy:
        ref = st.betas.toarray()
    else:
        res["max_abs_dbeta"] = float(np.abs(ref - st.betas.toarray()).max())
    del Z
if os.environ.get("BINOMIAL"):  # config 4's own response on the standardized view: IRLS on the panel engines of the 2-bit matrix
    Z = ad.matrix.standardize(X, lazy=True)
    t0 = time.time()
    sb = ad.grpnet(Z, ad.glm.binomial(y_bin), **kw)
    res["binomial_lazy_view"] = {"path_s": time.time() - t0, "lambdas": len(sb.lmdas), "final_active": int(sb.active_set_size),
                                 "n_irls_iters": int(sb.counters["n_irls_iters"]), "n_panel_blocks": int(sb.counters["n_panel_blocks"]),
                                 "error": sb.error}
print(json.dumps(res))
