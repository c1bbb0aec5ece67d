"""The golden cases (SURVEY.md 8c "Golden fixtures to commit"): config-1-sized problems, seeds recorded here."""
import numpy as np

from util import make_gaussian

CASES = ["c1_lasso", "c1_group10_enet", "c1_binomial", "c1_snp_gaussian", "c1_lasso_default_exit"]


def make_case(name):
    n, p = 1000, 200
    if name == "c1_lasso":
        d = make_gaussian(n, p, seed=0, sparsity=0.95)
        d["kw"] = dict(early_exit=False, tol=1e-10)
        d["glm"] = "gaussian"
    elif name == "c1_lasso_default_exit":  # high SNR: the default early_exit (dev >= 0.9) stops the path early
        d = make_gaussian(n, p, seed=0, sparsity=0.95, snr=50.0)
        d["kw"] = dict(tol=1e-10)
        d["glm"] = "gaussian"
    elif name == "c1_group10_enet":
        d = make_gaussian(n, p, seed=1, sparsity=0.95)
        d["kw"] = dict(groups=np.arange(0, p, 10), alpha=0.5, early_exit=False, tol=1e-10)
        d["glm"] = "gaussian"
    elif name == "c1_binomial":
        d = make_gaussian(n, p, seed=2, sparsity=0.95)
        rng = np.random.RandomState(12)
        eta = d["X"][:, :5] @ np.array([1.0, -1.0, 0.5, 2.0, -0.5])
        d["y"] = rng.binomial(1, 1 / (1 + np.exp(-eta))).astype(np.float64)
        d["kw"] = dict(early_exit=False, tol=1e-10, irls_tol=1e-10, lmda_path_size=40, min_ratio=5e-2)
        d["glm"] = "binomial"
    elif name == "c1_snp_gaussian":
        rng = np.random.RandomState(3)
        calldata = rng.choice([0, 1, 2, -9], size=(n, p), p=[0.6, 0.25, 0.05, 0.1]).astype(np.int8)
        valid = calldata >= 0
        imp = np.where(valid, calldata, 0).sum(0) / np.maximum(valid.sum(0), 1)
        Xd = np.where(valid, calldata, imp[None]).astype(np.float64)
        beta = rng.normal(size=p) * (rng.uniform(size=p) < 0.05)
        y = Xd @ beta + np.sqrt(max(beta @ beta, 1e-3)) * rng.normal(size=n)
        d = dict(calldata=np.asfortranarray(calldata), y=y, kw=dict(early_exit=False, tol=1e-10), glm="gaussian",
                 Xdense=np.asfortranarray(Xd))
    else:
        raise KeyError(name)
    return d


def solve_case(ad, X, case):
    glm = ad.glm.gaussian(case["y"]) if case["glm"] == "gaussian" else ad.glm.binomial(case["y"])
    return ad.grpnet(X, glm, **case["kw"])
