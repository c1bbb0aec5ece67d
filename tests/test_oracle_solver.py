"""Pins the CPU oracle (oracle/grpnet_oracle.cpp) — the reference cannot be built here (no Eigen), so the oracle is
pinned the way SURVEY.md 8c prescribes: the reference's published known answers, scikit-learn on identical problems,
first-principles KKT certificates, and the state invariants of adelie/state.py:1563-1674."""
import numpy as np
import pytest
from sklearn.linear_model import enet_path, lasso_path

import adelie_amd as ad
from util import kkt_gaussian, make_gaussian


def _quickstart(seed=0, n=100, p=1000):
    np.random.seed(seed)
    X = np.asfortranarray(np.random.normal(0, 1, (n, p)))
    return X


# ---- (4) known answers printed in the reference's documentation notebooks ------------------------------------------
def test_known_answer_lasso_46_solutions(oracle):
    """docs/sphinx/user_guide/notebooks/quickstart.ipynb:98 — `46/100 ... [dev:90.5%]`."""
    X = _quickstart()
    y = X[:, -1] * np.random.normal(0, 1) + np.random.normal(0, 1, 100)
    st = ad.grpnet(oracle.dense(X), ad.glm.gaussian(y))
    assert st.error == ""
    assert len(st.lmdas) == 46
    assert abs(st.devs[-1] - 0.905) < 2e-3 and st.devs[-2] < 0.9
    assert np.isclose(st.lmda_max, 0.483276, atol=1e-6)


def test_known_answer_group_lasso_38_solutions(oracle):
    """quickstart.ipynb:255 — groups of 10, `38/100 ... [dev:90.2%]`."""
    X = _quickstart()
    y = X[:, -1] * np.random.normal(0, 1) + np.random.normal(0, 1, 100)
    st = ad.grpnet(oracle.dense(X), ad.glm.gaussian(y), groups=np.arange(0, 1000, 10))
    assert len(st.lmdas) == 38
    assert abs(st.devs[-1] - 0.902) < 2e-3 and st.devs[-2] < 0.9


def test_known_answer_binomial_57_solutions(oracle):
    """quickstart.ipynb:493 (data cell 35) — `57/100 ... [dev:90.2%]`."""
    X = _quickstart()
    eta = X[:, -1] * np.random.normal(0, 1) + np.random.normal(0, 1, 100)
    y = np.random.binomial(1, 1 / (1 + np.exp(-eta)))
    st = ad.grpnet(oracle.dense(X), ad.glm.binomial(y=y, dtype=np.float64))
    assert st.error == ""
    assert len(st.lmdas) == 57
    assert abs(st.devs[-1] - 0.902) < 2e-3 and st.devs[-2] < 0.9


# ---- (1) scikit-learn on identical problems ----------------------------------------------------------------------------
@pytest.mark.parametrize("n,p", [(100, 50), (300, 200)])
def test_lasso_matches_sklearn(oracle, n, p):
    d = make_gaussian(n, p, seed=3)
    X, y = d["X"], d["y"]
    st = ad.grpnet(oracle.dense(X), ad.glm.gaussian(y), penalty=np.ones(p), tol=1e-14, early_exit=False,
                   lmda_path_size=25, min_ratio=5e-2)
    Xc, yc = X - X.mean(0), y - y.mean()
    _, coefs, _ = lasso_path(Xc, yc, alphas=st.lmdas, tol=1e-14, max_iter=200000)
    assert np.abs(st.betas.toarray() - coefs.T).max() < 1e-6
    assert np.allclose(st.intercepts, y.mean() - st.betas @ X.mean(0), atol=1e-8)


def test_elastic_net_matches_sklearn(oracle):
    d = make_gaussian(150, 80, seed=5)
    X, y = d["X"], d["y"]
    a = 0.5
    st = ad.grpnet(oracle.dense(X), ad.glm.gaussian(y), penalty=np.ones(80), alpha=a, tol=1e-14, early_exit=False,
                   lmda_path_size=20, min_ratio=5e-2)
    Xc, yc = X - X.mean(0), y - y.mean()
    _, coefs, _ = enet_path(Xc, yc, l1_ratio=a, alphas=st.lmdas, tol=1e-14, max_iter=200000)
    assert np.abs(st.betas.toarray() - coefs.T).max() < 1e-6


# ---- (2) KKT certificates ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("alpha", [1.0, 0.5, 0.0])
@pytest.mark.parametrize("intercept", [True, False])
@pytest.mark.parametrize("n,p,G", [(10, 4, 2), (10, 100, 10), (10, 100, 20), (100, 23, 4), (100, 100, 50)])
def test_group_path_satisfies_kkt(oracle, n, p, G, intercept, alpha):
    """Shapes of the reference's test_solve_gaussian (tests/test_solver.py:596-649): weighted, random penalties
    with 5% zeros, tol=1e-10, min_ratio=1e-1, 30 lambdas."""
    d = make_gaussian(n, p, G=G, seed=0, sparsity=0.95, weights=True, zero_pen=0.05)
    st = ad.grpnet(oracle.dense(d["X"]), ad.glm.gaussian(d["y"], weights=d["weights"]), groups=d["groups"],
                   penalty=d["penalty"], alpha=alpha, intercept=intercept, tol=1e-10, min_ratio=1e-1,
                   lmda_path_size=30, early_exit=False)
    assert st.error == ""
    assert len(st.lmdas) == 30
    v = kkt_gaussian(d["X"], d["y"], st.weights, d["groups"], d["group_sizes"], d["penalty"], alpha, intercept,
                     st.betas, st.intercepts, st.lmdas, 1e-5)
    scale = max(1.0, float(np.max(np.abs(st.grad))))
    assert v < 3e-5 * scale, v


# ---- (3) state invariants ------------------------------------------------------------------------------------------------
def test_invariants_and_warm_start(oracle):
    d = make_gaussian(100, 60, G=20, seed=7, weights=True)
    Xo = oracle.dense(d["X"])
    glm = ad.glm.gaussian(d["y"], weights=d["weights"])
    # the CD stopping rule bounds A*dbeta^2 by tol*y_var, i.e. beta is resolved to ~sqrt(tol): use a tiny tol so that
    # two different warm-start trajectories agree to 1e-6
    kw = dict(groups=d["groups"], tol=1e-18, early_exit=False)
    full = ad.grpnet(Xo, glm, lmda_path_size=20, min_ratio=1e-1, **kw)
    w, X, y = glm.weights, d["X"], d["y"]
    beta = full.betas[-1].toarray().ravel()
    r = (y - np.sum(w * y)) - X @ beta
    assert np.allclose(full.resid, r, atol=1e-9)
    assert np.isclose(full.resid_sum, np.sum(w * r), atol=1e-10)
    g = X.T @ (w * r) - np.sum(w * r) * (X.T @ w)
    assert np.allclose(full.grad, g, atol=1e-9)
    assert np.isclose(full.rsq, np.sum(w * (y - np.sum(w * y)) ** 2) - np.sum(w * (r - np.sum(w * r)) ** 2), atol=1e-8)
    # warm start from the 10th solution reproduces the tail of the path (solver.py:866-873,905-912)
    head = ad.grpnet(Xo, glm, lmda_path=full.lmdas[:10], **kw)
    tail = ad.grpnet(Xo, glm, lmda_path=full.lmdas[10:], warm_start=head, **kw)
    assert np.abs(tail.betas.toarray() - full.betas[10:].toarray()).max() < 1e-6
    assert np.allclose(tail.intercepts, full.intercepts[10:], atol=1e-6)


def test_gaussian_through_irls_equals_gaussian(oracle):
    """glm.gaussian(opt=False) forces the IRLS solver; the paths must agree (reference glm.ipynb cells 21-28)."""
    d = make_gaussian(80, 40, G=10, seed=2)
    Xo = oracle.dense(d["X"])
    kw = dict(groups=d["groups"], tol=1e-13, irls_tol=1e-13, early_exit=False, lmda_path_size=15, min_ratio=1e-1)
    a = ad.grpnet(Xo, ad.glm.gaussian(d["y"]), **kw)
    b = ad.grpnet(Xo, ad.glm.gaussian(d["y"], opt=False), **kw)
    assert len(a.lmdas) == len(b.lmdas) == 15
    assert np.abs(a.betas.toarray() - b.betas.toarray()).max() < 1e-6
    assert np.allclose(a.devs, b.devs, atol=1e-7)


def test_binomial_kkt(oracle):
    rng = np.random.RandomState(4)
    n, p = 200, 30
    X = np.asfortranarray(rng.normal(size=(n, p)))
    eta = X[:, :3] @ np.array([1.0, -2.0, 0.5])
    y = rng.binomial(1, 1 / (1 + np.exp(-eta))).astype(float)
    st = ad.grpnet(oracle.dense(X), ad.glm.binomial(y), tol=1e-12, irls_tol=1e-12, early_exit=False,
                   lmda_path_size=15, min_ratio=5e-2)
    assert st.error == "" and len(st.lmdas) == 15
    B = st.betas.toarray()
    for l, lm in enumerate(st.lmdas):
        mu = 1 / (1 + np.exp(-(X @ B[l] + st.intercepts[l])))
        g = X.T @ ((y - mu) / n)
        assert abs(np.sum(y - mu) / n) < 1e-6
        z = B[l] == 0
        assert np.all(np.abs(g[z]) <= lm + 1e-6)
        assert np.allclose(g[~z], lm * np.sign(B[l][~z]), atol=1e-6)


# ---- (5) building blocks -----------------------------------------------------------------------------------------------
def test_newton_group_prox_is_the_minimiser(oracle):
    """adelie.bcd.solve semantics (tests/test_bcd.py:91-152): minimise 1/2 sum q x^2 - v.x + l1|x| + l2/2 |x|^2."""
    from scipy.optimize import minimize

    rng = np.random.RandomState(0)
    for q in [2, 5, 17]:
        L = rng.uniform(0.1, 2, q)
        v = rng.normal(size=q) * np.sqrt(L)
        l1, l2 = 0.3, 0.1
        x, it = oracle.bcd_newton(L, v, l1, l2)
        obj = lambda z: 0.5 * np.sum(L * z * z) - v @ z + l1 * np.linalg.norm(z) + 0.5 * l2 * z @ z
        if np.linalg.norm(v) <= l1:
            assert np.all(x == 0)
            continue
        res = minimize(obj, x + 1e-3, method="BFGS", options=dict(gtol=1e-10))
        assert obj(x) <= res.fun + 1e-10
        # stationarity: (L + l2) x + l1 x/|x| = v
        assert np.allclose((L + l2) * x + l1 * x / np.linalg.norm(x), v, atol=1e-9)
    x, _ = oracle.bcd_newton(np.ones(3), np.array([0.1, 0.1, 0.1]), 1.0, 0.0)
    assert np.all(x == 0)


def test_search_pivot_matches_bruteforce(oracle):
    """optimization/search_pivot.hpp:7-62 against explicit piecewise-linear least squares (tests/test_optimization.py:143)."""
    rng = np.random.RandomState(1)
    for n in [1, 2, 7, 40]:
        x = np.arange(n, dtype=float)
        y = np.sort(rng.normal(size=n))
        idx, mses = oracle.search_pivot(x, y)
        if n == 1:
            assert idx == 0
            continue
        best, best_i = np.inf, 0
        for i in range(1, n):
            t = np.maximum(x[i] - x, 0)
            tc, yc = t - t.mean(), y - y.mean()
            var_t = tc @ tc
            b1 = (tc @ yc) / var_t
            m = -b1 * b1 * var_t
            assert np.isclose(m, mses[i], rtol=1e-8, atol=1e-10)
            if m < best:
                best, best_i = m, i
        assert idx == best_i


def test_jacobi_eigh(oracle):
    rng = np.random.RandomState(2)
    for q in [2, 3, 10, 33]:
        A = rng.normal(size=(q, q))
        A = A @ A.T
        D, V = oracle.eigh(A)
        assert np.allclose(D, np.linalg.eigvalsh(A), atol=1e-10 * np.abs(A).max())
        assert np.allclose(V @ np.diag(D) @ V.T, A, atol=1e-10 * np.abs(A).max())
        assert np.allclose(V.T @ V, np.eye(q), atol=1e-12)


def test_error_taxonomy(oracle):
    """max_cds / max_screen errors are returned in `error` with the partial path (state.py:157-176)."""
    d = make_gaussian(50, 30, seed=1)
    Xo = oracle.dense(d["X"])
    st = ad.grpnet(Xo, ad.glm.gaussian(d["y"]), max_iters=1, early_exit=False)
    assert st.error.startswith("adelie_core solver: max coordinate descents reached at lambda index: 0.")
    st = ad.grpnet(Xo, ad.glm.gaussian(d["y"]), max_screen_size=2, early_exit=False)
    assert st.error == "adelie_core solver: maximum screen set size reached."
    assert 0 < len(st.lmdas) < 100
    with pytest.raises(RuntimeError, match="alpha must be in"):
        ad.grpnet(Xo, ad.glm.gaussian(d["y"]), alpha=1.5)


@pytest.mark.timeout(120)
def test_basil_loop_terminates_when_kkt_and_fallback_round_differently(oracle):
    """f32, alpha < 1, lambda_0 == lmda_max: the KKT check (lmda * alpha * penalty, solver_base.hpp:428) can flag a group
    that the screening fallback (lmda * penalty * alpha, :369) does not pick up; the reference's BASIL loop then never
    terminates.  The restatement (and the product, csrc/solver.hip::screen) screen such a group with KKT's own expression.
    Found by scripts/fuzz_parity.py (seed 4, case 70); the solve must come back (here with the f32 Newton error)."""
    rng = np.random.RandomState(4 * 1000 + 70)
    n = int(rng.choice([37, 150, 513, 1200, 2049])); p = int(rng.choice([40, 130, 300, 777]))
    rng.choice(["gaussian", "gaussian", "binomial"]); rng.choice(["dense", "dense", "snp"])
    rng.uniform(); rng.uniform()
    sizes = []
    while sum(sizes) < p:
        sizes.append(int(rng.choice([1, 1, 2, 3, 7, 16, 33])))
    sizes[-1] -= sum(sizes) - p
    if sizes[-1] <= 0:
        sizes.pop(); sizes[-1] += p - sum(sizes)
    groups = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    alpha = float(rng.choice([1.0, 0.7, 0.3])); rng.uniform(); rng.uniform()
    rng.choice([1, 1, 64, 256]); rng.choice(["", "32", "64", "128"])
    assert (n, p, alpha) == (37, 300, 0.3)
    cd = rng.choice(np.array([0, 1, 2, -9], dtype=np.int8), size=(n, p), p=[0.6, 0.22, 0.1, 0.08])
    imp = ad.matrix.compute_impute(cd)
    Xd = np.where(cd < 0, imp[None, :], cd).astype(np.float64)
    beta = rng.normal(size=p) * (rng.uniform(size=p) < 0.15)
    eta = Xd @ beta; eta = eta / max(eta.std(), 1e-9)
    y = eta + rng.normal(size=n)
    st = ad.grpnet(oracle.snp_calldata(cd, imp, dtype=np.float32), ad.glm.gaussian(y, dtype=np.float32), groups=groups,
                   alpha=alpha, early_exit=False, lmda_path_size=15, min_ratio=0.3, tol=1e-7, max_iters=4000)
    assert st.error == "" or st.error.startswith("adelie_core solver: Newton-ABS max iterations reached")


@pytest.mark.parametrize("family", ["gaussian", "binomial"])
def test_lasso_on_standardized_columns_is_the_lasso_on_raw_columns_with_scaled_penalties(oracle, family):
    """The identity ``adelie_amd.solver._lasso_in_raw_coordinates`` rests on (a lasso with an intercept on a lazily
    standardized view is solved on the base design with ``penalty * |s|``), on the CPU checker alone: the path on
    ``(Z - 1 c') diag(s)^-1`` and the path on ``Z`` with the rescaled penalty factors are the same iterates in two
    coordinate systems — same lambdas, deviances, screen and active sets, ``beta~ = s beta``, intercepts shifted by
    ``sum_j beta_j c_j`` — and the product's re-expression of the state reproduces every invariant."""
    from adelie_amd.solver import _to_standardized_coordinates

    rng = np.random.RandomState(3)
    n, p = 150, 40
    Z = np.asfortranarray(rng.normal(size=(n, p)) * rng.uniform(0.3, 4.0, p) + rng.normal(size=p) * 2)
    c, s = rng.normal(size=p), rng.uniform(0.5, 3.0, p) * rng.choice([-1.0, 1.0], p)   # any centres, scales of either sign
    Xs = np.asfortranarray((Z - c) / s)
    eta = Xs[:, :4] @ np.array([1.0, -2.0, 0.7, 0.4])
    if family == "gaussian":
        glm = ad.glm.gaussian(eta + 0.3 * rng.normal(size=n))
    else:
        glm = ad.glm.binomial((rng.uniform(size=n) < 1 / (1 + np.exp(-eta))).astype(float))
    pen = rng.uniform(0.5, 2.0, p)
    kw = dict(tol=1e-12, irls_tol=1e-11, early_exit=False, lmda_path_size=20, min_ratio=1e-2, progress_bar=False)
    a = ad.grpnet(oracle.dense(Xs), glm, penalty=pen, **kw)
    b = ad.grpnet(oracle.dense(Z), glm, penalty=pen * np.abs(s), **kw)
    assert a.error == "" and b.error == ""
    assert np.allclose(a.lmdas, b.lmdas, rtol=1e-11) and np.allclose(a.devs, b.devs, atol=1e-9)
    assert np.array_equal(a.screen_set, b.screen_set) and np.array_equal(a.active_set[:a.active_set_size], b.active_set[:b.active_set_size])
    b = _to_standardized_coordinates(b, None, s, c, pen)
    assert np.abs(a.betas.toarray() - b.betas.toarray()).max() < 1e-8
    for name in ("intercepts", "screen_beta", "grad", "abs_grad", "resid", "screen_X_means", "screen_vars"):
        u, v = np.asarray(getattr(a, name)), np.asarray(getattr(b, name))
        assert u.shape == v.shape and (u.size == 0 or np.abs(u - v).max() < 1e-8), name
    if family == "gaussian":
        assert abs(a.resid_sum - b.resid_sum) < 1e-9 and np.abs(a.X_means - b.X_means).max() < 1e-9
    else:
        assert abs(a.beta0 - b.beta0) < 1e-8
