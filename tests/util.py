"""Shared helpers of the test-suite: seeded data, first-principles KKT certificates, path comparison."""
import numpy as np

import adelie_amd as ad


def make_gaussian(n, p, G=None, seed=0, sparsity=0.9, snr=1.0, weights=False, zero_pen=0.0, dtype=np.float64,
                  order="F"):
    """Mirrors the recipe of the reference's create_data_gaussian (tests/test_solver.py:219-333)."""
    rng = np.random.RandomState(seed)
    X = rng.normal(0, 1, (n, p))
    beta = rng.normal(0, 1, p)
    beta[rng.choice(p, int(sparsity * p), replace=False)] = 0
    noise = np.sqrt(max(np.sum(beta ** 2), 1e-8) / snr)
    y = X @ beta + noise * rng.normal(0, 1, n)
    if G is None or G == p:
        groups = np.arange(p)
    else:
        groups = np.sort(np.concatenate([[0], rng.choice(np.arange(1, p), size=G - 1, replace=False)])).astype(int)
    group_sizes = np.diff(np.concatenate([groups, [p]]))
    penalty = np.sqrt(group_sizes).astype(float)
    if zero_pen > 0:
        penalty[rng.choice(len(groups), max(1, int(zero_pen * len(groups))), replace=False)] = 0
    w = None
    if weights:
        w = rng.uniform(1, 2, n)
        w /= w.sum()
    X = np.asarray(X, dtype=dtype, order=order)
    return dict(X=X, y=y.astype(dtype), groups=groups, group_sizes=group_sizes, penalty=penalty.astype(dtype), weights=w)


def kkt_gaussian(X, y, w, groups, group_sizes, penalty, alpha, intercept, betas, intercepts, lmdas, tol):
    """max KKT violation over the path, from first principles (what solver_base.hpp:408-433 / state.py:1599-1628
    assert): zero groups ||g|| <= lmda*alpha*pen ; non-zero groups g = lmda*pen*(alpha*b/||b|| + (1-alpha)*b)."""
    n, p = X.shape
    if w is None:
        w = np.full(n, 1 / n)
    worst = 0.0
    B = betas.toarray() if hasattr(betas, "toarray") else np.asarray(betas)
    for l, lm in enumerate(lmdas):
        b = B[l]
        r = y - X @ b - (intercepts[l] if intercept else 0)
        if intercept:
            worst = max(worst, abs(np.sum(w * r)))
        grad = X.T @ (w * r)
        for g, gs, pen in zip(groups, group_sizes, penalty):
            gg, bb = grad[g:g + gs], b[g:g + gs]
            nb = np.linalg.norm(bb)
            if nb == 0:
                worst = max(worst, np.linalg.norm(gg) - lm * alpha * pen)
            else:
                worst = max(worst, np.linalg.norm(gg - lm * pen * (alpha * bb / nb + (1 - alpha) * bb)))
    return worst


def assert_same_path(a, b, atol, check_sets=True):
    """Two solved states describe the same path within `atol` (betas, intercepts, devs, lmdas)."""
    assert a.error == "" and b.error == "", (a.error, b.error)
    assert len(a.lmdas) == len(b.lmdas), (len(a.lmdas), len(b.lmdas))
    f32 = np.asarray(a.lmdas).dtype == np.float32 or np.asarray(b.lmdas).dtype == np.float32
    np.testing.assert_allclose(a.lmdas, b.lmdas, rtol=1e-4 if f32 else (1e-6 if atol > 1e-5 else 1e-9))
    A, B = a.betas.toarray(), b.betas.toarray()
    assert np.abs(A - B).max() <= atol, np.abs(A - B).max()
    assert np.abs(np.asarray(a.intercepts) - np.asarray(b.intercepts)).max() <= atol
    assert np.abs(np.asarray(a.devs) - np.asarray(b.devs)).max() <= max(atol, 1e-9) * 10
    if check_sets:
        # Same support and same screened groups.  The ORDER of insertion may differ: lambda_0 == lmda_max puts the first
        # group exactly on the soft-threshold boundary, so whether it activates at lambda_0 or lambda_1 is decided by the
        # last bit of a dot product (true of the reference across machines too); the sets agree, the paths agree to the
        # resolution of the stopping rule.
        assert a.active_set_size == b.active_set_size
        assert sorted(a.screen_set.tolist()) == sorted(b.screen_set.tolist())
        sa = sorted(a.screen_set[a.active_set[:a.active_set_size]].tolist())
        sb = sorted(b.screen_set[b.active_set[:b.active_set_size]].tolist())
        assert sa == sb
