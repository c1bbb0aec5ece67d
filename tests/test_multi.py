"""Multi-response (multigaussian) path — SURVEY.md 8(f) rank 3.

CPU part: the oracle's restatement of ``concatenate([kronecker_eye(ones), kronecker_eye(X)])`` + the Gaussian naive solver
(reference ``solver.py:700-816``, ``state.py:1100-1125,2027-2391``, ``matrix_naive_kronecker_eye.ipp``,
``solver_multigaussian_naive.hpp:31-44``) is pinned against

* the reference's own printed output: quickstart.ipynb cell 41-42 (n=100, p=1000, K=4, seed 0) shows
  ``61/100 ... [dev:90.2%]``;
* scikit-learn's ``MultiTaskLasso`` (an independent solver of the same objective up to the scaling
  ``alpha_sklearn = lmda * sqrt(K) * K``);
* first-principles KKT conditions;
* the same solver run on the explicitly materialised Kronecker matrix.

GPU part (``-m gpu``): the HIP path through the C ABI against the oracle on the same inputs.
"""
import numpy as np
import pytest
from scipy.sparse import csr_matrix

import adelie_amd as ad
from adelie_amd import matrix


def make_multi(n, p, K, seed=0, nnz=4, dtype=np.float64, weights=False):
    rng = np.random.RandomState(seed)
    X = np.asfortranarray(rng.normal(size=(n, p)), dtype=dtype)
    B = np.zeros((p, K))
    B[rng.choice(p, nnz, replace=False)] = rng.normal(size=(nnz, K))
    Y = (X @ B + 0.5 * rng.normal(size=(n, K)) + rng.normal(size=K)).astype(dtype)
    w = None
    if weights:
        w = rng.uniform(1, 2, n)
        w /= w.sum()
    return X, Y, w


def explicit(X, K, icpt):
    n = X.shape[0]
    Xa = np.kron(X, np.eye(K))
    if icpt:
        Xa = np.hstack([np.kron(np.ones((n, 1)), np.eye(K)), Xa])
    return np.asfortranarray(Xa)


def kkt_multi(X, Y, w, K, groups, group_sizes, penalty, alpha, intercept, betas, intercepts, lmdas):
    """Worst violation of the stationarity conditions of
    (1/K) sum_i w_i (||eta_i||^2/2 - y_i.eta_i) + lmda sum_g pen_g (alpha ||B_g|| + (1-alpha)/2 ||B_g||^2)."""
    n, p = X.shape
    if w is None:
        w = np.full(n, 1 / n)
    worst = 0.0
    Bs = betas.toarray()
    for l, lm in enumerate(lmdas):
        B = Bs[l].reshape(p, K)
        R = Y - X @ B - intercepts[l]
        if intercept:
            worst = max(worst, np.abs(w @ R).max() / K)
        Gr = X.T @ (w[:, None] * R) / K
        for g, gs, pen in zip(groups, group_sizes, penalty):
            gg, bb = Gr[g:g + gs].ravel(), B[g:g + gs].ravel()
            nb = np.linalg.norm(bb)
            if nb == 0:
                worst = max(worst, np.linalg.norm(gg) - lm * alpha * pen)
            else:
                worst = max(worst, np.linalg.norm(gg - lm * pen * (alpha * bb / nb + (1 - alpha) * bb)))
    return worst


@pytest.mark.parametrize("icpt", [False, True])
def test_view_matches_explicit_kronecker(oracle, icpt):
    rng = np.random.RandomState(0)
    n, p, K = 7, 5, 3
    X = np.asfortranarray(rng.normal(size=(n, p)))
    V = matrix._multi_view(oracle.dense(X), K, icpt)
    Xa = explicit(X, K, icpt)
    assert V.shape == Xa.shape
    v, w = rng.normal(size=n * K), rng.uniform(size=n * K)
    out = np.empty(Xa.shape[1])
    V.mul(v, w, out)
    np.testing.assert_allclose(out, Xa.T @ (v * w), atol=1e-12)
    for j in range(Xa.shape[1]):
        np.testing.assert_allclose(V.cmul(j, v, w), Xa[:, j] @ (v * w), atol=1e-12)
        o = np.zeros(n * K)
        V.ctmul(j, 1.5, o)
        np.testing.assert_allclose(o, 1.5 * Xa[:, j], atol=1e-12)
    j, q = K, 2 * K
    o, b = np.zeros(n * K), rng.normal(size=q)
    V.btmul(j, q, b, o)
    np.testing.assert_allclose(o, Xa[:, j:j + q] @ b, atol=1e-12)
    c = np.empty((q, q))
    V.cov(j, q, np.sqrt(w), c)
    np.testing.assert_allclose(c, Xa[:, j:j + q].T @ (w[:, None] * Xa[:, j:j + q]), atol=1e-12)
    o = np.empty(Xa.shape[1])
    V.sq_mul(w, o)
    np.testing.assert_allclose(o, (Xa ** 2).T @ w, atol=1e-12)
    Bm = csr_matrix(rng.normal(size=(4, Xa.shape[1])) * (rng.uniform(size=(4, Xa.shape[1])) < 0.3))
    o = np.empty((4, n * K))
    V.sp_tmul(Bm, o)
    np.testing.assert_allclose(o, (Xa @ Bm.T.toarray()).T, atol=1e-12)
    if icpt:  # the reference's way of assembling it
        V2 = matrix.concatenate([matrix.kronecker_eye(np.ones((n, 1)), K), matrix.kronecker_eye(oracle.dense(X), K)], axis=1)
        assert V2.shape == V.shape
    with pytest.raises(NotImplementedError):
        matrix.concatenate([matrix.kronecker_eye(oracle.dense(X), K)] * 2, axis=0)


def test_reference_quickstart_known_answer(oracle):
    """quickstart.ipynb cells 41-42: ``61/100 ... [dev:90.2%]``."""
    n, p, K = 100, 1000, 4
    np.random.seed(0)
    X = np.random.normal(0, 1, (n, p))
    y = X[:, -1:] @ np.random.normal(0, 1, (1, K)) + np.random.normal(0, 1, (n, K))
    s = ad.grpnet(X=oracle.dense(np.asfortranarray(X)), glm=ad.glm.multigaussian(y=y))
    assert s.error == ""
    assert len(s.lmdas) == 61
    assert f"{100 * s.devs[-1]:.1f}" == "90.2" and s.devs[-2] < 0.9
    assert s.betas.shape == (61, p * K) and s.intercepts.shape == (61, K)


def test_oracle_vs_sklearn_multitask_lasso(oracle):
    sk = pytest.importorskip("sklearn.linear_model")
    X, Y, _ = make_multi(60, 30, 3, seed=1)
    n, p, K = 60, 30, 3
    s = ad.grpnet(X=oracle.dense(X), glm=ad.glm.multigaussian(Y), tol=1e-13, early_exit=False, lmda_path_size=20)
    for l in (5, 12, 19):
        m = sk.MultiTaskLasso(alpha=s.lmdas[l] * np.sqrt(K) * K, fit_intercept=True, tol=1e-14, max_iter=200000).fit(X, Y)
        B = s.betas[l].toarray().reshape(p, K)
        assert np.abs(B - m.coef_.T).max() < 1e-5  # resolution of sklearn's own stopping rule; KKT below is the certificate
        assert np.abs(s.intercepts[l] - m.intercept_).max() < 1e-5


@pytest.mark.parametrize("icpt", [False, True])
@pytest.mark.parametrize("alpha", [1.0, 0.5])
def test_oracle_kkt_and_explicit_matrix(oracle, icpt, alpha):
    n, p, K = 50, 24, 3
    X, Y, w = make_multi(n, p, K, seed=2, weights=True)
    groups = np.array([0, 1, 2, 5, 6, 10, 11, 12, 20])
    gs = np.diff(np.concatenate([groups, [p]]))
    glm = ad.glm.multigaussian(Y, weights=w)
    kw = dict(groups=groups, alpha=alpha, intercept=icpt, tol=1e-13, early_exit=False, lmda_path_size=15, min_ratio=5e-2)
    s = ad.grpnet(X=oracle.dense(X), glm=glm, **kw)
    assert s.error == "" and len(s.lmdas) == 15
    pen = np.sqrt(gs * K)
    v = kkt_multi(X, Y, glm.weights, K, groups, gs, pen, alpha, icpt, s.betas, s.intercepts, s.lmdas)
    assert v < 1e-6, v
    eta = ad.diagnostic.predict(oracle.dense(X), s.betas, s.intercepts)
    assert eta.shape == (15, n, K)
    np.testing.assert_allclose(eta[-1], X @ s.betas[-1].toarray().reshape(p, K) + s.intercepts[-1], atol=1e-12)
    # the same state on the explicitly materialised matrix: identical algorithm, so the path agrees to rounding
    st = ad.state.multigaussian_naive
    Xa = oracle.dense(explicit(X, K, icpt))
    s2 = _solve_on(Xa, X, glm, kw, oracle)
    assert np.abs(s2.betas.toarray() - s.betas.toarray()).max() < 1e-9
    np.testing.assert_allclose(s2.lmdas, s.lmdas, rtol=1e-12)


def _solve_on(Xa, X, glm, kw, oracle):
    """Builds the multigaussian state through grpnet's preamble, then swaps in `Xa` (a plain dense design holding the
    materialised Kronecker matrix) as the matrix the solver runs on."""
    captured = {}
    real = ad.solver._STATE_OF["multigaussian"]

    def spy(**a):
        s = real(**a)
        s._X = Xa
        captured["s"] = s
        return s

    ad.solver._STATE_OF["multigaussian"] = spy
    try:
        return ad.grpnet(X=oracle.dense(X), glm=glm, **kw)
    finally:
        ad.solver._STATE_OF["multigaussian"] = real


def test_cv_grpnet_multi_response(oracle):
    """cv_grpnet on a multi-response family (reference cv.py:130-325 is family-agnostic): fold losses are
    validation-weighted losses of the interpolated solutions, recomputed here from first principles for one fold."""
    n, p, K = 60, 10, 3
    X, Y, _ = make_multi(n, p, K, seed=7, nnz=3)
    glm = ad.glm.multigaussian(Y)
    Xd = oracle.dense(X)
    cv = ad.cv_grpnet(Xd, glm, n_folds=3, seed=5, lmda_path_size=12, min_ratio=0.2, tol=1e-12)
    assert cv.losses.shape == (3, 12) and 0 <= cv.best_idx < 12
    np.random.seed(5)
    order = np.random.choice(n, n, replace=False)
    b, e = ad.cv.fold_ranges(n, 3)[1]
    val = order[b:e]
    w = np.full(n, 1 / n)
    w[val] = 0
    w /= w.sum()
    st = ad.grpnet(X=Xd, glm=ad.glm.multigaussian(Y, weights=w), lmda_path=cv.lmdas, early_exit=False, tol=1e-12)
    eta = ad.diagnostic.predict(Xd, st.betas, st.intercepts)
    per_obs = np.sum(0.5 * eta ** 2 - Y[None] * eta, axis=2) / K          # (L, n)
    expect = per_obs[:, val].mean(axis=1)                                   # weights 1/n on the validation rows, renormalised
    np.testing.assert_allclose(cv.losses[1], expect, atol=1e-7)
    fit = cv.fit(Xd, glm, lmda_path_size=5)
    assert fit.betas.shape == (5, p * K) and fit.intercepts.shape == (5, K)
    Xm, ym, _ = make_multinomial(60, 8, 3, seed=2)
    cvm = ad.cv_grpnet(oracle.dense(Xm), ad.glm.multinomial(ym), n_folds=3, seed=1, lmda_path_size=8, min_ratio=0.3)
    assert cvm.losses.shape == (3, 8) and np.all(np.isfinite(cvm.losses))


def test_multi_errors(oracle):
    X, Y, _ = make_multi(20, 6, 2)
    with pytest.raises(RuntimeError):
        ad.glm.multigaussian(Y[:, 0])
    with pytest.raises(NotImplementedError):
        ad.glm.multigaussian(Y, opt=False)
    with pytest.raises(RuntimeError):
        matrix._multi_view(oracle.dense(X), 0, True)


# ---- multinomial (IRLS on the view) ---------------------------------------------------------------------------------------
def make_multinomial(n, p, K, seed=0, nnz=3, weights=False):
    rng = np.random.RandomState(seed)
    X = np.asfortranarray(rng.normal(size=(n, p)))
    B = np.zeros((p, K))
    B[rng.choice(p, nnz, replace=False)] = 1.5 * rng.normal(size=(nnz, K))
    eta = X @ B + 0.3 * rng.normal(size=K)
    mu = np.exp(eta) / np.sum(np.exp(eta), axis=1)[:, None]
    y = np.array([rng.multinomial(1, m) for m in mu]).astype(np.float64)
    w = None
    if weights:
        w = rng.uniform(1, 2, n)
        w /= w.sum()
    return X, y, w


def test_multinomial_glm_members():
    """gradient = -d loss / d eta; hessian = 2 K^-1 W P (1 - P) (glm_multinomial.ipp:21-66)."""
    rng = np.random.RandomState(0)
    n, K = 9, 4
    y = np.eye(K)[rng.randint(0, K, n)]
    w = rng.uniform(1, 2, n)
    glm = ad.glm.multinomial(y, weights=w)
    eta = rng.normal(size=(n, K))
    g = np.empty((n, K))
    glm.gradient(eta, g)
    num = np.empty((n, K))
    for i in range(n):
        for k in range(K):
            e = eta.copy(); e[i, k] += 1e-6
            e2 = eta.copy(); e2[i, k] -= 1e-6
            num[i, k] = -(glm.loss(e) - glm.loss(e2)) / 2e-6
    np.testing.assert_allclose(g, num, atol=1e-8)
    P = np.exp(eta) / np.exp(eta).sum(1)[:, None]
    h = np.empty((n, K))
    glm.hessian(eta, g, h)
    np.testing.assert_allclose(h, 2 * glm.weights[:, None] * P * (1 - P) / K, atol=1e-12)
    assert glm.loss_full() == 0  # one-hot responses
    with pytest.raises(RuntimeError):
        ad.glm.multinomial(y[:, :1])


def test_reference_quickstart_multinomial_known_answer(oracle):
    """quickstart.ipynb cells 46-47: ``52/100 ... [dev:90.2%]``."""
    n, p, K = 100, 1000, 4
    np.random.seed(0)
    X = np.random.normal(0, 1, (n, p))
    eta = X[:, -1:] @ np.random.normal(0, 1, (1, K)) + np.random.normal(0, 1, (n, K))
    mu = np.exp(eta) / np.sum(np.exp(eta), axis=-1)[:, None]
    y = np.array([np.random.multinomial(1, m) for m in mu])
    s = ad.grpnet(X=oracle.dense(np.asfortranarray(X)), glm=ad.glm.multinomial(y=y, dtype=np.float64))
    assert s.error == ""
    assert len(s.lmdas) == 52
    assert f"{100 * s.devs[-1]:.1f}" == "90.2" and s.devs[-2] < 0.9
    assert s.betas.shape == (52, p * K) and s.intercepts.shape == (52, K)


def kkt_multinomial(X, glm, K, groups, group_sizes, penalty, alpha, intercept, betas, intercepts, lmdas):
    n, p = X.shape
    worst = 0.0
    Bs = betas.toarray()
    for l, lm in enumerate(lmdas):
        B = Bs[l].reshape(p, K)
        R = np.empty((n, K))
        glm.gradient(X @ B + intercepts[l], R)  # -(d loss / d eta)
        if intercept:
            worst = max(worst, np.abs(R.sum(0)).max())
        Gr = X.T @ R
        for g, gs, pen in zip(groups, group_sizes, penalty):
            gg, bb = Gr[g:g + gs].ravel(), B[g:g + gs].ravel()
            nb = np.linalg.norm(bb)
            if nb == 0:
                worst = max(worst, np.linalg.norm(gg) - lm * alpha * pen)
            else:
                worst = max(worst, np.linalg.norm(gg - lm * pen * (alpha * bb / nb + (1 - alpha) * bb)))
    return worst


@pytest.mark.parametrize("icpt", [False, True])
def test_oracle_multinomial_kkt(oracle, icpt):
    n, p, K = 80, 12, 3
    X, y, w = make_multinomial(n, p, K, seed=3, weights=True)
    glm = ad.glm.multinomial(y, weights=w)
    groups = np.array([0, 1, 3, 4, 8])
    gs = np.diff(np.concatenate([groups, [p]]))
    s = ad.grpnet(X=oracle.dense(X), glm=glm, groups=groups, intercept=icpt, alpha=0.7, tol=1e-14, irls_tol=1e-16,
                  early_exit=False, lmda_path_size=12, min_ratio=0.1)
    assert s.error == "" and len(s.lmdas) == 12
    v = kkt_multinomial(X, glm, K, groups, gs, np.sqrt(gs * K), 0.7, icpt, s.betas, s.intercepts, s.lmdas)
    assert v < 1e-6, v
    assert np.all(np.diff(s.devs) > -1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("K", [2, 3, 5])
@pytest.mark.parametrize("icpt", [False, True])
def test_hip_multinomial_matches_oracle(hip, oracle, K, icpt):
    X, y, w = make_multinomial(300, 30, K, seed=10 + K, weights=True)
    glm = ad.glm.multinomial(y, weights=w)
    kw = dict(intercept=icpt, tol=1e-13, irls_tol=1e-16, early_exit=False, lmda_path_size=20, min_ratio=5e-2)
    a, b = _both(oracle, X, glm, **kw)
    assert a.error == "" and b.error == "", (a.error, b.error)
    assert len(a.lmdas) == len(b.lmdas) == 20
    np.testing.assert_allclose(a.lmdas, b.lmdas, rtol=1e-9)
    assert np.abs(a.betas.toarray() - b.betas.toarray()).max() < 1e-7
    assert np.abs(a.intercepts - b.intercepts).max() < 1e-7
    np.testing.assert_allclose(a.devs, b.devs, atol=1e-8)
    np.testing.assert_allclose(a.eta, b.eta, atol=1e-6)
    np.testing.assert_allclose(a.resid, b.resid, atol=1e-8)
    np.testing.assert_allclose(a.loss_null, b.loss_null, rtol=1e-10)
    gs = np.ones(30, dtype=int)
    v = kkt_multinomial(X, glm, K, np.arange(30), gs, np.sqrt(gs * K), 1.0, icpt, a.betas, a.intercepts, a.lmdas)
    assert v < 1e-5, v


@pytest.mark.gpu
def test_hip_multinomial_reference_quickstart(hip):
    n, p, K = 100, 1000, 4
    np.random.seed(0)
    X = np.random.normal(0, 1, (n, p))
    eta = X[:, -1:] @ np.random.normal(0, 1, (1, K)) + np.random.normal(0, 1, (n, K))
    mu = np.exp(eta) / np.sum(np.exp(eta), axis=-1)[:, None]
    y = np.array([np.random.multinomial(1, m) for m in mu])
    s = ad.grpnet(X=ad.matrix.dense(np.asfortranarray(X)), glm=ad.glm.multinomial(y=y, dtype=np.float64))
    assert s.error == "" and len(s.lmdas) == 52 and f"{100 * s.devs[-1]:.1f}" == "90.2"


# ---- HIP path ----------------------------------------------------------------------------------------------------------
def _both(oracle, X, glm, **kw):
    a = ad.grpnet(X=ad.matrix.dense(X), glm=glm, **kw)
    b = ad.grpnet(X=oracle.dense(X), glm=glm, **kw)
    return a, b


def _assert_same(a, b, atol):
    assert a.error == "" and b.error == "", (a.error, b.error)
    assert len(a.lmdas) == len(b.lmdas)
    f32 = np.asarray(a.lmdas).dtype == np.float32
    np.testing.assert_allclose(a.lmdas, b.lmdas, rtol=1e-4 if f32 else 1e-9)
    assert np.abs(a.betas.toarray() - b.betas.toarray()).max() <= atol
    assert np.abs(a.intercepts - b.intercepts).max() <= atol
    assert np.abs(np.asarray(a.devs) - np.asarray(b.devs)).max() <= 10 * atol
    assert sorted(a.screen_set.tolist()) == sorted(b.screen_set.tolist())
    np.testing.assert_allclose(a.resid, b.resid, atol=10 * atol)
    np.testing.assert_allclose(a.grad, b.grad, atol=10 * atol)


@pytest.mark.gpu
@pytest.mark.parametrize("K", [1, 2, 3, 4, 8, 11])
@pytest.mark.parametrize("icpt", [False, True])
def test_hip_multigaussian_matches_oracle(hip, oracle, K, icpt):
    X, Y, w = make_multi(203, 40, K, seed=K, weights=True)
    glm = ad.glm.multigaussian(Y, weights=w)
    a, b = _both(oracle, X, glm, intercept=icpt, tol=1e-12, early_exit=False, lmda_path_size=25)
    _assert_same(a, b, 1e-8)
    assert a.betas.shape == (25, 40 * K) and a.intercepts.shape == (25, K)
    assert a.counters["n_panel_blocks"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("alpha", [1.0, 0.3])
def test_hip_multigaussian_feature_groups(hip, oracle, alpha):
    n, p, K = 300, 60, 3
    X, Y, w = make_multi(n, p, K, seed=5, nnz=8)
    rng = np.random.RandomState(3)
    groups = np.sort(np.concatenate([[0], rng.choice(np.arange(1, p), 19, replace=False)]))
    gs = np.diff(np.concatenate([groups, [p]]))
    glm = ad.glm.multigaussian(Y)
    kw = dict(groups=groups, alpha=alpha, tol=1e-12, early_exit=False, lmda_path_size=30)
    a, b = _both(oracle, X, glm, **kw)
    _assert_same(a, b, 1e-8)
    v = kkt_multi(X, Y, glm.weights, K, groups, gs, np.sqrt(gs * K), alpha, True, a.betas, a.intercepts, a.lmdas)
    assert v < 1e-5, v


@pytest.mark.gpu
def test_hip_multigaussian_many_blocks_and_f32(hip, oracle):
    # enough screened columns for several 128-value blocks per pass, odd n (scalar loads), C-ordered input
    n, p, K = 1001, 400, 4
    X, Y, _ = make_multi(n, p, K, seed=9, nnz=40)
    glm = ad.glm.multigaussian(Y)
    a, b = _both(oracle, X, glm, tol=1e-11, early_exit=False, lmda_path_size=30, min_ratio=1e-2)
    _assert_same(a, b, 1e-7)
    assert a.screen_set.size * K > 256
    X32, Y32 = np.ascontiguousarray(X, dtype=np.float32), Y.astype(np.float32)
    # (the default newton_tol = 1e-12 is below float32 resolution: both the oracle and the HIP path then report the
    #  reference's "Newton-ABS max iterations reached" error)
    a32, b32 = _both(oracle, X32, ad.glm.multigaussian(Y32), tol=1e-6, newton_tol=1e-6, lmda_path_size=20, early_exit=False)
    assert a32.error == "" and len(a32.lmdas) == len(b32.lmdas)
    assert np.abs(a32.betas.toarray() - b32.betas.toarray()).max() < 2e-3
    assert np.abs(a32.intercepts - b32.intercepts).max() < 2e-3


@pytest.mark.gpu
def test_hip_multigaussian_reference_quickstart(hip):
    n, p, K = 100, 1000, 4
    np.random.seed(0)
    X = np.random.normal(0, 1, (n, p))
    y = X[:, -1:] @ np.random.normal(0, 1, (1, K)) + np.random.normal(0, 1, (n, K))
    s = ad.grpnet(X=ad.matrix.dense(np.asfortranarray(X)), glm=ad.glm.multigaussian(y=y))
    assert s.error == "" and len(s.lmdas) == 61 and f"{100 * s.devs[-1]:.1f}" == "90.2"


@pytest.mark.gpu
def test_hip_multigaussian_warm_start_and_predict(hip, oracle):
    X, Y, _ = make_multi(150, 30, 3, seed=4)
    glm = ad.glm.multigaussian(Y)
    Xd = ad.matrix.dense(X)
    full = ad.grpnet(X=Xd, glm=glm, tol=1e-12, early_exit=False, lmda_path_size=20)
    head = ad.grpnet(X=Xd, glm=glm, tol=1e-12, early_exit=False, lmda_path=full.lmdas[:10])
    # warm_start takes the raw (pre-tidy) invariants of the solved state
    tail = ad.grpnet(X=Xd, glm=glm, tol=1e-12, early_exit=False, lmda_path=full.lmdas[10:], warm_start=head)
    assert tail.error == ""
    assert np.abs(tail.betas.toarray() - full.betas[10:].toarray()).max() < 1e-7
    eta = ad.diagnostic.predict(Xd, full.betas, full.intercepts)
    np.testing.assert_allclose(eta[-1], X @ full.betas[-1].toarray().reshape(30, 3) + full.intercepts[-1], atol=1e-10)
    with pytest.raises(RuntimeError):  # the view only serves the solver
        out = np.empty(full._X.cols())
        _abi_mul(full._X, out)


def _abi_mul(V, out):
    from adelie_amd import _abi
    v = np.zeros(V.rows())
    V._backend.check(V._backend.fn("design_mul")(V._handle, v.ctypes.data, v.ctypes.data, out.ctypes.data))


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["multigaussian", "multinomial"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_hip_multi_path_losses_equal_host_predictions(family, dtype):
    """adelie_hip_design_multi_path_losses (cv.py:281-314 for multi-response fits): the per-lambda losses under two weight
    vectors, computed on the device from the CSR coefficients, against diagnostic.predict + glm.loss on the host."""
    rng = np.random.RandomState(11)
    n, p, K, L = 257, 31, 3, 7
    X = np.asfortranarray(rng.normal(size=(n, p)), dtype=dtype)
    if family == "multinomial":
        Y = np.eye(K)[rng.randint(0, K, n)].astype(dtype)
    else:
        Y = rng.normal(size=(n, K)).astype(dtype)
    wa = rng.uniform(1, 2, n)
    wa /= wa.sum()
    wb = wa.copy()
    wb[: n // 4] = 0
    wb /= wb.sum()
    B = (rng.normal(size=(L, p * K)) * (rng.uniform(size=(L, p * K)) < 0.2)).astype(dtype)
    B[0] = 0                                      # an empty row
    icpt = rng.normal(size=(L, K)).astype(dtype)
    off = (0.1 * rng.normal(size=(n, K))).astype(dtype)
    Xd = matrix.dense(X)
    ga = getattr(ad.glm, family)(y=Y, weights=wa, dtype=dtype)
    gb = getattr(ad.glm, family)(y=Y, weights=wb, dtype=dtype)
    kind = {"multigaussian": 0, "multinomial": 3}[family]
    la, lb = Xd.multi_path_losses(kind, K, csr_matrix(B), icpt, off, Y, ga.weights, gb.weights)
    eta = (X.astype(np.float64) @ B.astype(np.float64).reshape(L, p, K).transpose(1, 0, 2).reshape(p, L * K)).reshape(n, L, K)
    eta = eta.transpose(1, 0, 2) + icpt[:, None].astype(np.float64) + off.astype(np.float64)
    g64a = getattr(ad.glm, family)(y=Y.astype(np.float64), weights=wa)
    g64b = getattr(ad.glm, family)(y=Y.astype(np.float64), weights=wb)
    tol = 1e-12 if dtype == np.float64 else 2e-5
    assert np.allclose(la, [g64a.loss(e) for e in eta], rtol=tol, atol=tol)
    assert np.allclose(lb, [g64b.loss(e) for e in eta], rtol=tol, atol=tol)


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["multigaussian", "multinomial"])
def test_hip_multi_cv_device_losses_equal_host_path(family, monkeypatch):
    rng = np.random.RandomState(12)
    n, p, K = 240, 25, 3
    X = np.asfortranarray(rng.normal(size=(n, p)))
    S = X[:, :3] @ rng.normal(size=(3, K)) + 0.5 * rng.normal(size=(n, K))
    Y = np.eye(K)[np.argmax(S, axis=1)] if family == "multinomial" else S
    glm = getattr(ad.glm, family)(y=Y)
    Xd = matrix.dense(X)
    a = ad.cv_grpnet(Xd, glm, n_folds=3, seed=0, lmda_path_size=20, progress_bar=False)
    monkeypatch.setattr(ad.cv, "_MULTI_KINDS", {})            # host predictions + glm.loss, as before
    b = ad.cv_grpnet(Xd, glm, n_folds=3, seed=0, lmda_path_size=20, progress_bar=False)
    assert np.array_equal(a.lmdas, b.lmdas)
    assert np.allclose(a.losses, b.losses, rtol=1e-10, atol=1e-13)
    assert a.best_idx == b.best_idx


@pytest.mark.gpu
def test_hip_multi_response_on_snp_design_equals_densified():
    """A 2-bit SNP base under the multi-response view: same path as on the densified matrix."""
    rng = np.random.RandomState(13)
    n, p, K = 300, 40, 3
    calls = rng.choice([0, 1, 2, -9], size=(n, p), p=[0.55, 0.3, 0.1, 0.05]).astype(np.int8)
    imp = matrix.compute_impute(calls)
    D = np.asfortranarray(np.where(calls < 0, imp[None], calls).astype(np.float64))
    Y = D[:, :3] @ rng.normal(size=(3, K)) + 0.5 * rng.normal(size=(n, K))
    s1 = ad.grpnet(matrix.snp_calldata(calls), ad.glm.multigaussian(y=Y), progress_bar=False)
    s2 = ad.grpnet(matrix.dense(D), ad.glm.multigaussian(y=Y), progress_bar=False)
    assert np.array_equal(s1.lmdas, s2.lmdas)
    assert np.allclose(s1.betas.toarray(), s2.betas.toarray(), atol=1e-13)
    assert np.allclose(s1.intercepts, s2.intercepts, atol=1e-13)
    cv = ad.cv_grpnet(matrix.snp_calldata(calls), ad.glm.multigaussian(y=Y), n_folds=3, seed=0, lmda_path_size=10,
                      progress_bar=False)
    assert np.all(np.isfinite(cv.losses))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("family,n,p,K", [("multigaussian", 1031, 90, 4), ("multigaussian", 517, 60, 9),
                                          ("multinomial", 802, 70, 3)])
def test_hip_multi_response_kernels_on_2bit_base(monkeypatch, oracle, family, n, p, K, dtype):
    """The K-wide kernels on the 2-bit layout (SnpOnesAcc: sweep, panel step, fused look-ahead launch, Gram builds, residual
    update) with the panel engine forced: the view over the SNP design against the view over its densified copy (the same
    kernels through DenseOnesAcc: equal to rounding) and against the CPU oracle on the dense matrix.  Row counts that are not
    multiples of the 4 calls of a byte, missing calls in every column."""
    monkeypatch.setenv("ADELIE_HIP_CD_BLOCK_MIN_NV", "1")
    rng = np.random.RandomState(n + K)
    calls = rng.choice([0, 1, 2, -9], size=(n, p), p=[0.5, 0.3, 0.15, 0.05]).astype(np.int8)
    imp = matrix.compute_impute(calls)
    D = np.asfortranarray(np.where(calls < 0, imp[None], calls).astype(dtype))
    B = np.zeros((p, K))
    B[rng.choice(p, 8, replace=False)] = rng.normal(size=(8, K))
    eta = (D - D.mean(0)) @ B
    if family == "multigaussian":
        Y = (eta + 0.5 * rng.normal(size=(n, K))).astype(dtype)
    else:
        P = np.exp(eta - eta.max(1, keepdims=True))
        P /= P.sum(1, keepdims=True)
        Y = np.stack([rng.multinomial(1, P[i]) for i in range(n)]).astype(dtype)
    kw = dict(early_exit=False, lmda_path_size=12, min_ratio=0.1, progress_bar=False)
    if dtype == np.float32:
        kw.update(tol=1e-6, newton_tol=1e-5)
    glm = lambda: getattr(ad.glm, family)(y=Y, dtype=dtype)
    s1 = ad.grpnet(matrix.snp_calldata(calls, dtype=dtype), glm(), **kw)
    s2 = ad.grpnet(matrix.dense(D), glm(), **kw)
    so = ad.grpnet(oracle.dense(D), glm(), **kw)
    assert s1.error == "" and s2.error == "" and so.error == ""
    assert np.array_equal(s1.lmdas, s2.lmdas)
    f32 = dtype == np.float32
    assert np.abs(s1.betas.toarray() - s2.betas.toarray()).max() < (2e-4 if f32 else 1e-11)
    assert np.abs(s1.intercepts - s2.intercepts).max() < (2e-4 if f32 else 1e-11)
    assert len(so.lmdas) == len(s1.lmdas)
    assert np.abs(s1.betas.toarray() - so.betas.toarray()).max() < (5e-3 if f32 else 1e-6)
    assert (s1.betas[-1].toarray() != 0).sum() >= 8 * K


# ---- per-group constraints on multi-response fits (reference solver.py:640-659: the list goes through to the view's solver;
# ---- state.py:24-45 left-pads it with None for the K intercept columns) ------------------------------------------------------
def _multi_constraints(p, K, rng, with_linear=True):
    from adelie_amd import constraint

    spec = [None] * p
    for j in rng.choice(p, 6, replace=False):
        spec[j] = ("box", -rng.uniform(0.02, 0.2, K), rng.uniform(0.02, 0.2, K))
    if with_linear:
        free = [j for j in range(p) if spec[j] is None]
        for j in rng.choice(free, 2, replace=False):
            spec[j] = ("linear", rng.normal(size=(2, K)), -rng.uniform(0.02, 0.1, 2), rng.uniform(0.02, 0.1, 2))

    def make():
        out = []
        for s in spec:
            if s is None:
                out.append(None)
            elif s[0] == "box":
                out.append(constraint.box(s[1], s[2]))
            else:
                out.append(constraint.linear(s[1], s[2], s[3]))
        return out
    return spec, make


@pytest.mark.parametrize("icpt", [False, True])
def test_oracle_multigaussian_constrained_path_is_optimal(oracle, icpt):
    """KKT system of the constrained multigaussian problem from first principles, with the returned duals: stationarity
    grad_g - lmda pen (alpha b/|b| + (1 - alpha) b) - A_g' mu_g = 0 on non-zero groups, |grad_g - A_g' mu_g| <= lmda alpha pen on
    zero groups, feasibility, sign and complementarity of the multipliers."""
    K, p = 3, 16
    X, Y, w = make_multi(250, p, K, seed=5, nnz=6, weights=True)
    spec, make = _multi_constraints(p, K, np.random.RandomState(2))
    cons = make()
    st = ad.grpnet(X=oracle.dense(X), glm=ad.glm.multigaussian(Y, weights=w), constraints=cons, intercept=icpt, alpha=0.8,
                   tol=1e-13, early_exit=False, lmda_path_size=12, min_ratio=0.03, progress_bar=False)
    assert st.error == ""
    n_d = sum(0 if c is None else c.duals() for c in cons)
    assert st.duals.shape == (12, n_d) and st.duals.nnz > 0
    Bs, Ds = st.betas.toarray(), st.duals.toarray()
    pen = np.sqrt(K)
    worst, bound = 0.0, 0
    for l, lm in enumerate(st.lmdas):
        B = Bs[l].reshape(p, K)
        R = Y - X @ B - st.intercepts[l]
        if icpt:
            worst = max(worst, np.abs(w @ R).max() / K)
        Gr = X.T @ (w[:, None] * R) / K
        off = 0
        for j in range(p):
            g, b = Gr[j], B[j]
            atmu = np.zeros(K)
            if spec[j] is not None:
                m = cons[j].duals()
                mu = Ds[l, off:off + m]
                off += m
                if spec[j][0] == "box":
                    atmu, z, lo, up = mu, b, spec[j][1], spec[j][2]
                else:
                    atmu, z, lo, up = spec[j][1].T @ mu, spec[j][1] @ b, spec[j][2], spec[j][3]
                # (the constraint solvers stop on their own tolerances: tol 1e-9 on the dual iteration, 1e-7 on its sub-problem)
                assert np.all(z <= up + 5e-5) and np.all(z >= lo - 5e-5)
                assert np.all(np.maximum(mu, 0) * (up - z) < 1e-4) and np.all(np.maximum(-mu, 0) * (z - lo) < 1e-4)
                bound += int(np.any(mu != 0))
            nb = np.linalg.norm(b)
            if nb == 0:
                worst = max(worst, np.linalg.norm(g - atmu) - lm * 0.8 * pen)
            else:
                worst = max(worst, np.linalg.norm(g - atmu - lm * pen * (0.8 * b / nb + 0.2 * b)))
    assert worst < 5e-5, worst
    assert bound >= 5


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["multigaussian", "multinomial"])
def test_hip_multi_constraints_match_oracle(hip, oracle, family):
    """Constraint objects on the groups of a multi-response view: the HIP path (group visits through the callbacks between two
    K-wide panel steps) against the oracle (its own C++ box / linear solvers)."""
    K, p = 3, 24
    if family == "multigaussian":
        X, Y, w = make_multi(300, p, K, seed=9, nnz=6, weights=True)
        mk = lambda: ad.glm.multigaussian(Y, weights=w)
        kw = dict(tol=1e-13)
    else:
        X, Y, w = make_multinomial(400, p, K, seed=9, nnz=5)
        mk = lambda: ad.glm.multinomial(Y)
        kw = dict(tol=1e-13, irls_tol=1e-11)
    _, make = _multi_constraints(p, K, np.random.RandomState(4))
    kw.update(alpha=0.9, early_exit=False, lmda_path_size=10, min_ratio=0.05, progress_bar=False)
    a = ad.grpnet(X=ad.matrix.dense(X), glm=mk(), constraints=make(), **kw)
    b = ad.grpnet(X=oracle.dense(X), glm=mk(), constraints=make(), **kw)
    assert a.error == "" and b.error == "", (a.error, b.error)
    assert len(a.lmdas) == len(b.lmdas) == 10
    assert np.abs(a.betas.toarray() - b.betas.toarray()).max() < 2e-6
    assert np.abs(a.intercepts - b.intercepts).max() < 2e-6
    assert a.duals.shape == b.duals.shape and b.duals.nnz > 0
    scale = max(1.0, float(np.abs(b.duals).max()))
    assert np.abs((a.duals - b.duals)).max() < 1e-4 * scale
    # the box / one-sided objects of the view's groups are visited on the device (kernels_cons.hip); `linear` ones on the host
    assert a.counters["n_dev_cons_visits"] > 0, a.counters
