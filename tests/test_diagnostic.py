"""KKT diagnostics — the numeric part of ``adelie.diagnostic`` (reference ``diagnostic.py:279-574,1248-1415``; the
reference's own ``tests/test_diagnostic.py:100-112`` only checks that ``diagnostic(state)`` runs).

CPU part: ``residuals`` / ``gradient_norms`` / ``gradient_scores`` against first-principles loops.
GPU part (``-m gpu``): ``gradients`` (one ABI call, the K-wide sweep) against ``resids @ X``, and the certificate the
scores exist for — along a solved path every score is at most its ``lmda`` and the active groups sit on it.
"""
import numpy as np
import pytest
from scipy.sparse import csr_matrix

import adelie_amd as ad
from adelie_amd import diagnostic as dg


def test_residuals_are_negative_gradients():
    rng = np.random.RandomState(0)
    n, L = 30, 4
    y = rng.normal(size=n)
    w = rng.uniform(1, 2, n)
    glm = ad.glm.gaussian(y=y, weights=w)
    etas = rng.normal(size=(L, n))
    r = dg.residuals(glm, etas)
    wn = w / w.sum()
    assert np.allclose(r, wn * (y - etas), atol=1e-15)
    yb = rng.binomial(1, 0.5, n).astype(float)
    glm = ad.glm.binomial(y=yb)
    r = dg.residuals(glm, etas)
    assert np.allclose(r, (yb - 1 / (1 + np.exp(-etas))) / n, atol=1e-15)
    Y = rng.normal(size=(n, 3))
    glm = ad.glm.multigaussian(y=Y)
    etas = rng.normal(size=(L, n, 3))
    assert np.allclose(dg.residuals(glm, etas), (Y - etas) / n / 3, atol=1e-15)


@pytest.mark.parametrize("alpha", [1.0, 0.3])
def test_gradient_norms_and_scores(alpha):
    rng = np.random.RandomState(1)
    L, p = 5, 12
    groups = np.array([0, 1, 4, 5, 9])
    sizes = np.array([1, 3, 1, 4, 3])
    penalty = np.array([1.0, 0.0, 2.0, 1.5, 0.7])
    grads = rng.normal(size=(L, p))
    B = rng.normal(size=(L, p)) * (rng.uniform(size=(L, p)) < 0.4)
    lmdas = np.exp(-np.arange(L) / 2.0)
    got = dg.gradient_norms(grads, csr_matrix(B), None, lmdas, groups=groups, alpha=alpha, penalty=penalty)
    want = np.empty((L, len(groups)))
    for l in range(L):
        for k, (g, gs) in enumerate(zip(groups, sizes)):
            want[l, k] = np.linalg.norm(grads[l, g:g + gs] - lmdas[l] * (1 - alpha) * penalty[k] * B[l, g:g + gs])
    assert np.allclose(got, want, atol=1e-14)
    # defaults: singleton groups, penalty sqrt(size) = 1
    got1 = dg.gradient_norms(grads, csr_matrix(B), None, lmdas, alpha=alpha)
    assert np.allclose(got1, np.abs(grads - lmdas[:, None] * (1 - alpha) * B), atol=1e-14)
    sc = dg.gradient_scores(got, lmdas, alpha=alpha, penalty=penalty)
    for k in range(len(groups)):
        if alpha * penalty[k] > 0:
            assert np.allclose(sc[:, k], want[:, k] / (alpha * penalty[k]))
        else:
            assert np.array_equal(sc[:, k], lmdas)
    # multi-response layout: (L, p, K) gradients, groups counted in features
    K = 3
    gm = rng.normal(size=(L, 4, K))
    Bm = rng.normal(size=(L, 4 * K))
    gotm = dg.gradient_norms(gm, csr_matrix(Bm), None, lmdas, alpha=alpha)
    wantm = np.linalg.norm((gm.reshape(L, -1) - lmdas[:, None] * (1 - alpha) * np.sqrt(K) * Bm).reshape(L, 4, K), axis=2)
    assert np.allclose(gotm, wantm, atol=1e-14)
    with pytest.raises(NotImplementedError):
        dg.gradient_norms(grads, csr_matrix(B), None, lmdas, constraints=[object()] * p)


# ---------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("L", [1, 3, 8, 21])
def test_gradients_batch_equals_matmul(dtype, L):
    rng = np.random.RandomState(L)
    n, p = 1003, 257
    X = np.asfortranarray(rng.normal(size=(n, p)), dtype=dtype)
    R = rng.normal(size=(L, n)).astype(dtype)
    tol = 1e-11 if dtype == np.float64 else 2e-3
    Xd = ad.matrix.dense(X)
    g = dg.gradients(Xd, R)
    assert g.shape == (L, p) and g.dtype == dtype
    assert np.allclose(g, R.astype(np.float64) @ X.astype(np.float64), atol=tol)
    # one X.mul per vector (what the reference does) gives the same numbers
    one = np.empty(p, dtype=dtype)
    Xd.mul(R[0], np.ones(n, dtype=dtype), one)
    assert np.allclose(g[0], one, atol=tol)
    # multi-response residuals
    K = 3
    Rm = rng.normal(size=(L, n, K)).astype(dtype)
    gm = dg.gradients(Xd, Rm)
    assert gm.shape == (L, p, K)
    assert np.allclose(gm, np.einsum("lnk,np->lpk", Rm.astype(np.float64), X.astype(np.float64)), atol=tol)
    assert dg.gradients(Xd, np.zeros((0, n), dtype=dtype)).shape == (0, p)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,p,L", [(500, 70, 5), (1003, 33, 21), (4099, 7, 8), (7, 3, 2)])
def test_gradients_snp_design(dtype, n, p, L):
    """2-bit SNP designs go through the same K-wide sweep (the calls of a column are decoded once per eight vectors)."""
    rng = np.random.RandomState(3)
    calls = rng.choice([0, 1, 2, -9], size=(n, p), p=[0.6, 0.25, 0.1, 0.05]).astype(np.int8)
    Xs = ad.matrix.snp_calldata(calls, dtype=dtype)
    imp = ad.matrix.compute_impute(calls)
    dense = np.where(calls < 0, imp[None], calls).astype(np.float64)
    R = rng.normal(size=(L, n)).astype(dtype)
    g = dg.gradients(Xs, R)
    assert g.dtype == dtype
    assert np.allclose(g, R.astype(np.float64) @ dense, atol=1e-11 if dtype == np.float64 else 5e-3)
    one = np.empty(p, dtype=dtype)
    Xs.mul(R[1], np.ones(n, dtype=dtype), one)
    assert np.allclose(g[1], one, atol=1e-11 if dtype == np.float64 else 5e-3)


def _check_kkt(state, d, rtol):
    lm = np.asarray(state.lmdas)
    sc = d.gradient_scores
    assert sc.shape[0] == len(lm)
    assert np.all(sc <= lm[:, None] * (1 + rtol)), float(np.max(sc / lm[:, None]))
    return sc, lm


@pytest.mark.gpu
def test_diagnostic_gaussian_groups():
    rng = np.random.RandomState(0)
    n, p = 400, 120
    X = np.asfortranarray(rng.normal(size=(n, p)))
    beta = np.zeros(p)
    beta[:6] = rng.normal(size=6)
    y = X @ beta + rng.normal(size=n)
    groups = np.arange(0, p, 3)
    state = ad.grpnet(X, ad.glm.gaussian(y=y), groups=groups, alpha=0.8, tol=1e-14, progress_bar=False)
    d = dg.diagnostic(state)
    L = len(state.lmdas)
    assert d.linear_preds.shape == (L, n) and d.gradients.shape == (L, p) and d.gradient_norms.shape == (L, len(groups))
    sc, lm = _check_kkt(state, d, 1e-6)
    # active penalised groups sit on the boundary
    B = state.betas.toarray()
    for l in (L // 2, L - 1):
        act = np.array([np.any(B[l, g:g + 3] != 0) for g in groups])
        assert act.any()
        assert np.allclose(sc[l, act], lm[l], rtol=1e-6)
    # gradients are X^T of the residuals
    assert np.allclose(d.gradients, d.residuals @ X, atol=1e-12)


@pytest.mark.gpu
def test_diagnostic_binomial_and_multigaussian():
    rng = np.random.RandomState(1)
    n, p = 300, 40
    X = np.asfortranarray(rng.normal(size=(n, p)))
    eta = X[:, :3] @ np.array([1.0, -1.0, 0.5])
    yb = (rng.uniform(size=n) < 1 / (1 + np.exp(-eta))).astype(float)
    state = ad.grpnet(X, ad.glm.binomial(y=yb), irls_tol=1e-14, tol=1e-14, min_ratio=0.05, progress_bar=False)
    _check_kkt(state, dg.diagnostic(state), 1e-5)
    K = 3
    Y = X[:, :4] @ rng.normal(size=(4, K)) + rng.normal(size=(n, K))
    state = ad.grpnet(X, ad.glm.multigaussian(y=Y), tol=1e-14, progress_bar=False)
    d = dg.diagnostic(state)
    L = len(state.lmdas)
    assert d.gradients.shape == (L, p, K) and d.gradient_norms.shape == (L, p)
    sc, lm = _check_kkt(state, d, 1e-6)
    B = state.betas.toarray().reshape(L, p, K)
    act = np.any(B[-1] != 0, axis=1)
    assert act.any() and np.allclose(sc[-1, act], lm[-1], rtol=1e-6)


def test_coefficients_equals_one_coefficient_call_per_lambda():
    """``diagnostic.coefficients`` (what cv_grpnet evaluates every fold with) is ``diagnostic.coefficient`` (reference
    ``diagnostic.py:577-646``) for every query at once: interior points, saved lambdas themselves, both boundary solutions."""
    import logging
    import scipy.sparse as sp
    from adelie_amd.diagnostic import coefficient, coefficients

    rng = np.random.default_rng(0)
    L, p = 40, 200
    lm = np.sort(rng.uniform(0.1, 2, L))[::-1]
    B = sp.random(L, p, 0.2, format="csr", random_state=1)
    ic = rng.standard_normal(L)
    q = np.concatenate([lm[[0, 5, L - 1]], rng.uniform(0.05, 2.2, 60), [lm[3] * (1 - 1e-16)]])
    logging.getLogger("adelie_amd").setLevel(logging.ERROR)
    Bq, iq = coefficients(lmdas_new=q, betas=B, intercepts=ic, lmdas=lm)
    assert Bq.shape == (len(q), p)
    for k, l in enumerate(q):
        b, i0 = coefficient(lmda=l, betas=B, intercepts=ic, lmdas=lm)
        assert np.array_equal(b.toarray(), Bq[k].toarray()) and i0 == iq[k]
    with pytest.raises(RuntimeError):
        coefficients(lmdas_new=q, betas=B[:1], intercepts=ic[:1], lmdas=lm[:1])
