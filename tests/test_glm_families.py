"""Poisson and binomial-probit families (reference glm_poisson.ipp:14-58, glm_binomial.ipp:100-190) through the IRLS path.

CPU: the Python GLM members against numeric differentiation; the oracle's path against first-principles KKT conditions.
GPU: the HIP path against the oracle; the device path losses of cv_grpnet against the members."""
import numpy as np
import pytest

import adelie_amd as ad


def make(family, n, p, seed=0, weights=True):
    rng = np.random.RandomState(seed)
    X = np.asfortranarray(rng.normal(size=(n, p)))
    beta = np.zeros(p)
    beta[rng.choice(p, 4, replace=False)] = rng.normal(size=4) * 0.6
    eta = X @ beta + 0.2
    if family == "poisson":
        y = rng.poisson(np.exp(eta)).astype(float)
    else:
        from scipy.special import erf
        y = (rng.uniform(size=n) < 0.5 * (1 + erf(eta / np.sqrt(2)))).astype(float)
    w = None
    if weights:
        w = rng.uniform(1, 2, n)
        w /= w.sum()
    return X, y, w


def glm_of(family, y, w):
    return ad.glm.poisson(y, weights=w) if family == "poisson" else ad.glm.binomial(y, weights=w, link="probit")


@pytest.mark.parametrize("family", ["poisson", "probit"])
def test_members_match_derivatives(family):
    X, y, w = make(family, 30, 5, seed=1)
    glm = glm_of(family, y, w)
    rng = np.random.RandomState(2)
    eta = rng.normal(size=30) * 0.7
    g = np.empty(30)
    glm.gradient(eta, g)
    h = np.empty(30)
    glm.hessian(eta, g, h)
    for i in range(0, 30, 7):
        e1, e2 = eta.copy(), eta.copy()
        e1[i] += 1e-5
        e2[i] -= 1e-5
        assert abs(-(glm.loss(e1) - glm.loss(e2)) / 2e-5 - g[i]) < 1e-8
        g1, g2 = np.empty(30), np.empty(30)
        glm.gradient(e1, g1)
        glm.gradient(e2, g2)
        assert abs(-(g1[i] - g2[i]) / 2e-5 - h[i]) < 1e-7
    assert np.isfinite(glm.loss_full())
    assert glm.loss_full() <= glm.loss(eta) + 1e-12


def kkt(X, glm, groups, gs, pen, alpha, icpt, betas, intercepts, lmdas):
    worst = 0.0
    B = betas.toarray()
    for l, lm in enumerate(lmdas):
        b = B[l]
        r = np.empty(X.shape[0])
        glm.gradient(X @ b + intercepts[l], r)
        if icpt:
            worst = max(worst, abs(r.sum()))
        gr = X.T @ r
        for g, q, pk in zip(groups, gs, pen):
            gg, bb = gr[g:g + q], b[g:g + q]
            nb = np.linalg.norm(bb)
            if nb == 0:
                worst = max(worst, np.linalg.norm(gg) - lm * alpha * pk)
            else:
                worst = max(worst, np.linalg.norm(gg - lm * pk * (alpha * bb / nb + (1 - alpha) * bb)))
    return worst


@pytest.mark.parametrize("family", ["poisson", "probit"])
def test_oracle_path_kkt(oracle, family):
    n, p = 120, 16
    X, y, w = make(family, n, p, seed=3)
    glm = glm_of(family, y, w)
    groups = np.array([0, 1, 2, 5, 6, 10])
    gs = np.diff(np.concatenate([groups, [p]]))
    s = ad.grpnet(oracle.dense(X), glm, groups=groups, alpha=0.8, tol=1e-13, irls_tol=1e-16, early_exit=False,
                  lmda_path_size=12, min_ratio=0.05)
    assert s.error == "" and len(s.lmdas) == 12
    v = kkt(X, glm, groups, gs, np.sqrt(gs), 0.8, True, s.betas, s.intercepts, s.lmdas)
    assert v < 1e-6, v
    assert np.all(np.diff(s.devs) > -1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["poisson", "probit"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_hip_matches_oracle(hip, oracle, family, dtype):
    n, p = 400, 60
    X, y, w = make(family, n, p, seed=5)
    X = np.asfortranarray(X, dtype=dtype)
    mk = (lambda: ad.glm.poisson(y.astype(dtype), weights=w)) if family == "poisson" else \
         (lambda: ad.glm.binomial(y.astype(dtype), weights=w, link="probit"))
    f32 = dtype == np.float32
    kw = dict(tol=1e-7 if f32 else 1e-13, irls_tol=1e-6 if f32 else 1e-14, early_exit=False, lmda_path_size=15, min_ratio=0.05)
    a = ad.grpnet(ad.matrix.dense(X), mk(), **kw)
    b = ad.grpnet(oracle.dense(X), mk(), **kw)
    assert a.error == "" and b.error == "", (a.error, b.error)
    assert len(a.lmdas) == len(b.lmdas) == 15
    tol = 2e-3 if f32 else 1e-7
    assert np.abs(a.betas.toarray() - b.betas.toarray()).max() < tol
    assert np.abs(np.asarray(a.intercepts) - np.asarray(b.intercepts)).max() < tol
    np.testing.assert_allclose(a.devs, b.devs, atol=10 * tol)
    np.testing.assert_allclose(a.loss_null, b.loss_null, rtol=1e-4 if f32 else 1e-10)
    if not f32:
        glm = mk()
        v = kkt(X.astype(float), glm, np.arange(p), np.ones(p, dtype=int), np.ones(p), 1.0, True, a.betas, a.intercepts, a.lmdas)
        assert v < 1e-6, v


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["poisson", "probit"])
def test_cv_device_path_losses(hip, family):
    """cv_grpnet's per-fold losses come from adelie_hip_design_glm_path_losses; same numbers as predict + glm.loss."""
    X, y, w = make(family, 150, 12, seed=7, weights=False)
    glm = glm_of(family, y, None)
    Xd = ad.matrix.dense(X)
    st = ad.grpnet(Xd, glm, lmda_path_size=8, early_exit=False, min_ratio=0.2)
    wa = np.full(150, 1 / 150)
    wb = wa.copy()
    wb[:30] = 0
    wb /= wb.sum()
    la, lb = Xd.glm_path_losses(glm.core_kind, st.betas, st.intercepts, np.zeros(150), y, wa, wb)
    eta = ad.diagnostic.predict(Xd, st.betas, st.intercepts)
    np.testing.assert_allclose(la, [glm.loss(e) for e in eta], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(lb, [glm.reweight(wb).loss(e) for e in eta], rtol=1e-10, atol=1e-12)
    cv = ad.cv_grpnet(Xd, glm, n_folds=3, seed=0, lmda_path_size=6, min_ratio=0.3)
    assert cv.losses.shape == (3, 6) and np.all(np.isfinite(cv.losses))
