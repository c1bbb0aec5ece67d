"""HIP path against checkers that share NOTHING with it below ``adelie_amd.grpnet``'s arguments: scikit-learn's own solvers
and first-principles KKT systems written out in numpy here.  No CPU oracle, no golden file.

Sizes are chosen so that the multi-CU engines run at their DEFAULT thresholds (screen sets of several hundred coefficients:
panel blocks, look-ahead, strips, speculative passes; IRLS with 64-visit blocks), which the oracle-based tests only reach by
forcing them with hooks at sizes the oracle finishes in seconds.

(VERDICT r3 weak point 1.iv: both sides of every HIP-vs-oracle test run the same ``adelie_amd/{solver,state}.py``.)
"""
import numpy as np
import pytest

import adelie_amd as ad
from util import kkt_gaussian, make_gaussian

pytestmark = pytest.mark.gpu


def _dense_problem(n, p, seed, nnz):
    rng = np.random.RandomState(seed)
    X = np.asfortranarray(rng.normal(size=(n, p)))
    X[:, 1::2] += 0.5 * X[:, 0::2][:, : X[:, 1::2].shape[1]]          # correlated neighbours: CD needs many passes
    b = np.zeros(p)
    b[rng.choice(p, nnz, replace=False)] = rng.normal(size=nnz)
    y = X @ b + rng.normal(size=n)
    return X, y


@pytest.mark.parametrize("alpha", [1.0, 0.6])
def test_hip_lasso_and_elastic_net_match_sklearn(hip, alpha):
    from sklearn.linear_model import enet_path

    n, p = 1500, 900
    X, y = _dense_problem(n, p, seed=11, nnz=300)
    # (tol bounds max_k A_k d_k^2 of a pass: 1e-16 leaves coefficient changes of 1e-8)
    st = ad.grpnet(ad.matrix.dense(X), ad.glm.gaussian(y), penalty=np.ones(p), alpha=alpha, tol=1e-16, early_exit=False,
                   lmda_path_size=30, min_ratio=1e-2, progress_bar=False)
    assert st.error == ""
    B = st.betas.toarray()
    assert (B[-1] != 0).sum() > 300                                   # the panel engine's regime, by a wide margin
    Xc, yc = X - X.mean(0), y - y.mean()
    _, coefs, _ = enet_path(Xc, yc, l1_ratio=alpha, alphas=st.lmdas, tol=1e-14, max_iter=500000)
    assert np.abs(B - coefs.T).max() < 1e-6                           # (1.4e-7 on the CPU: sklearn's duality-gap rule is the coarser)
    assert np.allclose(st.intercepts, y.mean() - B @ X.mean(0), atol=1e-8)
    # ... and the certificate that does not depend on either solver's stopping rule
    worst = kkt_gaussian(X, y, None, np.arange(p), np.ones(p, dtype=int), np.ones(p), alpha, True, st.betas, st.intercepts,
                         st.lmdas, 0)
    assert worst < 5e-7, worst                                        # (5e-8 .. 7e-8 at this tol)


def test_hip_group_lasso_kkt_at_default_thresholds(hip):
    d = make_gaussian(1200, 1000, G=230, seed=4, sparsity=0.7, weights=True)
    st = ad.grpnet(ad.matrix.dense(d["X"]), ad.glm.gaussian(d["y"], weights=d["weights"]), groups=d["groups"], penalty=d["penalty"],
                   alpha=0.8, tol=1e-16, early_exit=False, lmda_path_size=25, min_ratio=1e-1, progress_bar=False)
    assert st.error == ""
    assert (st.betas[-1].toarray() != 0).sum() > 300
    worst = kkt_gaussian(d["X"], d["y"], d["weights"], d["groups"], d["group_sizes"], d["penalty"], 0.8, True, st.betas,
                         st.intercepts, st.lmdas, 0)
    assert worst < 5e-7, worst


def test_hip_multigaussian_matches_sklearn_multitask_lasso(hip):
    sk = pytest.importorskip("sklearn.linear_model")
    rng = np.random.RandomState(2)
    n, p, K = 800, 300, 4
    X = np.asfortranarray(rng.normal(size=(n, p)))
    Bt = np.zeros((p, K))
    Bt[rng.choice(p, 40, replace=False)] = rng.normal(size=(40, K))
    Y = X @ Bt + rng.normal(size=(n, K)) + rng.normal(size=K)
    st = ad.grpnet(ad.matrix.dense(X), ad.glm.multigaussian(Y), tol=1e-13, early_exit=False, lmda_path_size=20, min_ratio=5e-2,
                   progress_bar=False)
    assert st.error == ""
    for l in (6, 13, 19):
        # same objective up to the scaling alpha_sklearn = lmda * sqrt(K) * K (tests/test_multi.py header)
        m = sk.MultiTaskLasso(alpha=st.lmdas[l] * np.sqrt(K) * K, fit_intercept=True, tol=1e-14, max_iter=500000).fit(X, Y)
        B = st.betas[l].toarray().reshape(p, K)
        assert np.abs(B - m.coef_.T).max() < 1e-5
        assert np.abs(st.intercepts[l] - m.intercept_).max() < 1e-5


@pytest.mark.parametrize("family", ["binomial", "poisson"])
def test_hip_glm_lasso_kkt_from_first_principles(hip, family):
    """Stationarity of  lmda * ||b||_1 + sum_i w_i (A(eta_i) - y_i eta_i)  (glm/glm_binomial.ipp, glm_poisson.ipp: the
    gradient in eta is w * (y - mu)), written out here."""
    rng = np.random.RandomState(9)
    n, p = 2000, 500
    X = np.asfortranarray(rng.normal(size=(n, p)))
    b = np.zeros(p)
    b[rng.choice(p, 60, replace=False)] = 0.3 * rng.normal(size=60)
    eta = X @ b
    if family == "binomial":
        y = rng.binomial(1, 1 / (1 + np.exp(-eta))).astype(float)
        glm, mean = ad.glm.binomial(y), (lambda e: 1 / (1 + np.exp(-e)))
    else:
        y = rng.poisson(np.exp(0.3 * eta)).astype(float)
        glm, mean = ad.glm.poisson(y), np.exp
    st = ad.grpnet(ad.matrix.dense(X), glm, tol=1e-12, irls_tol=1e-11, early_exit=False, lmda_path_size=25, min_ratio=3e-2,
                   progress_bar=False)
    assert st.error == ""
    B = st.betas.toarray()
    assert (B[-1] != 0).sum() > 200
    w = np.full(n, 1 / n)
    worst = 0.0
    for l, lm in enumerate(st.lmdas):
        g = X.T @ (w * (y - mean(X @ B[l] + st.intercepts[l])))
        worst = max(worst, abs(np.sum(w * (y - mean(X @ B[l] + st.intercepts[l])))))
        nzm = B[l] != 0
        worst = max(worst, np.abs(g[nzm] - lm * np.sign(B[l][nzm])).max() if nzm.any() else 0.0)
        worst = max(worst, (np.abs(g[~nzm]) - lm).max())
    assert worst < 5e-7, worst


def test_hip_constrained_groups_match_scipy_slsqp(hip):
    """Box / one-sided constraint objects on groups of several coefficients (solved on the device by kernels_cons.hip) against
    scipy's SLSQP on the same convex programme written out here — no oracle, none of adelie_amd.constraint's numpy classes in
    the solve (VERDICT r4 weak point 1.iv: constraints had no independent check)."""
    from scipy.optimize import minimize

    rng = np.random.RandomState(5)
    n, G, gs = 400, 12, 4
    p = G * gs
    X = np.asfortranarray(rng.normal(size=(n, p)))
    beta = np.zeros(p)
    beta[: 3 * gs] = rng.normal(size=3 * gs) * 2
    y = X @ beta + rng.normal(size=n)
    groups = np.arange(0, p, gs)
    pen = np.sqrt(np.full(G, gs, dtype=float))
    alpha = 0.7
    lo, hi = np.full(p, -np.inf), np.full(p, np.inf)
    cons = [None] * G
    cons[0] = ad.constraint.box(np.full(gs, -0.3), np.full(gs, 0.4))
    lo[0:gs], hi[0:gs] = -0.3, 0.4
    cons[1] = ad.constraint.lower(np.zeros(gs))               # beta >= 0 on group 1
    lo[gs:2 * gs] = 0.0
    cons[2] = ad.constraint.upper(np.full(gs, 0.25))          # beta <= 0.25 on group 2
    hi[2 * gs:3 * gs] = 0.25
    st = ad.grpnet(ad.matrix.dense(X), ad.glm.gaussian(y), groups=groups, alpha=alpha, constraints=cons, tol=1e-14,
                   early_exit=False, lmda_path_size=6, min_ratio=0.05, progress_bar=False)
    assert st.error == "" and st.counters["n_dev_cons_visits"] > 0 and st.counters["n_host_cons_visits"] == 0
    B = st.betas.toarray()
    w = np.full(n, 1.0 / n)
    for k in (2, 5):
        lm = float(st.lmdas[k])

        def obj(z):
            b, b0 = z[:p], z[p]
            r = y - X @ b - b0
            val = 0.5 * np.sum(w * r * r)
            grad_b = -(X.T @ (w * r))
            for g in range(G):
                bg = b[g * gs:(g + 1) * gs]
                nb = np.sqrt(np.sum(bg * bg) + 1e-24)            # (smoothed at 0 for the SQP; 1e-12 in the norm)
                val += lm * pen[g] * (alpha * nb + 0.5 * (1 - alpha) * np.sum(bg * bg))
                grad_b[g * gs:(g + 1) * gs] += lm * pen[g] * (alpha * bg / nb + (1 - alpha) * bg)
            return val, np.concatenate([grad_b, [-np.sum(w * r)]])

        z0 = np.concatenate([np.clip(B[k], lo, hi), [st.intercepts[k]]])
        res = minimize(obj, z0 * 0.9, jac=True, method="SLSQP", bounds=list(zip(lo, hi)) + [(None, None)],
                       options=dict(maxiter=2000, ftol=1e-15))
        f_hip = obj(np.concatenate([B[k], [st.intercepts[k]]]))[0]
        # feasible to the resolution of the reference's dual solver (its `tol` = 1e-9 bounds the multipliers' change, not the primal)
        assert np.all(B[k] >= lo - 1e-6) and np.all(B[k] <= hi + 1e-6)
        assert f_hip <= res.fun + 1e-7 * max(1.0, abs(res.fun))                      # as good as the SQP's point
        assert np.abs(B[k] - res.x[:p]).max() < 2e-4                                 # and the same point (SLSQP's resolution)


def test_hip_cv_loss_table_matches_sklearn_folds(hip):
    """cv_grpnet's (folds, lambdas) loss table against one scikit-learn Lasso per fold and lambda on the training rows and the
    validation loss written out here (reference cv.py:247-314: the fold's loss is the full-data loss minus the training loss,
    per unit of validation weight)."""
    from sklearn.linear_model import Lasso

    rng = np.random.RandomState(3)
    n, p, K = 600, 80, 3
    X = np.asfortranarray(rng.normal(size=(n, p)))
    beta = np.zeros(p)
    beta[:8] = rng.normal(size=8) * 2
    y = X @ beta + rng.normal(size=n)
    cv = ad.cv_grpnet(ad.matrix.dense(X), ad.glm.gaussian(y), n_folds=K, seed=4, lmda_path_size=12, min_ratio=0.1, tol=1e-14,
                      progress_bar=False)
    np.random.seed(4)
    order = np.random.choice(n, n, replace=False)
    worst = 0.0
    for f, (b, e) in enumerate(ad.cv.fold_ranges(n, K)):
        val = order[b:e]
        train = np.setdiff1d(np.arange(n), val)
        for l in (3, 7, 11):
            # the fold's problem: weights 1/n_train on the training rows; sklearn's alpha is adelie's lmda for that scaling
            m = Lasso(alpha=float(cv.lmdas[l]), fit_intercept=True, tol=1e-14, max_iter=200000).fit(X[train], y[train])
            eta = X[val] @ m.coef_ + m.intercept_
            # Gaussian loss of adelie: sum_i w_i (eta_i^2 / 2 - y_i eta_i), here per unit of validation weight
            loss = np.mean(0.5 * eta ** 2 - y[val] * eta)
            worst = max(worst, abs(loss - cv.losses[f, l]) / max(1.0, abs(loss)))
    assert worst < 1e-6, worst
