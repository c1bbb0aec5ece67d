"""Per-group constraints on one-coefficient groups (reference adelie_core/constraint/*,
adelie/constraint.py): the descriptor objects against the KKT properties the reference's own tests/test_constraint.py:6-64
checks (d = 1 rows of its grid), the oracle's constrained path against an independent bound-constrained solve (the reference's
test_solver.py does this with cvxpy, which is not in this image: L-BFGS-B on the split problem instead), the error behaviour,
and — on the GPU — the HIP path against the oracle."""
import numpy as np
import pytest
from scipy.optimize import minimize

import adelie_amd as ad
from adelie_amd import constraint
from util import make_gaussian


# --------------------------------------------------------------------------------------------------------------------
# the objects (reference tests/test_constraint.py:6-64, d = 1)
# --------------------------------------------------------------------------------------------------------------------
def _run_object_test(cnstr, seed):
    atol = 5e-7
    assert cnstr.duals() == cnstr.dual_size == 1 and cnstr.primals() == cnstr.primal_size == 1
    np.random.seed(seed)
    quad = np.random.uniform(0, 1, 1)
    linear = np.sqrt(quad) * np.random.normal(0, 1, 1)
    l1 = 0.5 * np.linalg.norm(linear)
    l2 = 0.0
    for Q in (np.ones((1, 1)), -np.ones((1, 1))):  # the orthogonal 1x1 matrices (the pin solver passes +1)
        cnstr.clear()
        x = np.zeros(1)
        cnstr.solve(x, quad, linear, l1, l2, Q, None)
        actual = np.empty(1)
        cnstr.gradient(x, actual)
        mu = np.zeros(1)
        nnz = cnstr.duals_nnz()
        ind, val = np.empty(nnz, dtype=int), np.empty(nnz)
        cnstr.dual(ind, val)
        mu[ind] = val
        expected = np.empty(1)
        cnstr.gradient(x, mu, expected)
        assert np.allclose(actual, expected)
        Qx = Q @ x
        grad = np.empty(1)
        cnstr.gradient(Qx, mu, grad)
        lagr = (quad + l2) * x - linear + Q.T @ grad
        assert np.allclose(np.maximum(np.linalg.norm(lagr) - l1, 0), 0, atol=atol)     # first-order condition
        assert np.allclose(np.maximum(np.max(cnstr.evaluate(Qx)), 0), 0, atol=atol)     # primal feasibility
        z = Qx + np.random.normal(0, 3, 1)
        cnstr.project(z)
        assert np.allclose(np.maximum(np.max(cnstr.evaluate(z)), 0), 0, atol=atol)


@pytest.mark.parametrize("lower", [-1, -1e-14, 0])
@pytest.mark.parametrize("upper", [1, 1e-14, 0])
@pytest.mark.parametrize("seed", range(10))
def test_box_object(lower, upper, seed):
    np.random.seed(seed)
    lo = np.random.uniform(lower, 0, 1)
    up = np.random.uniform(0, upper, 1)
    _run_object_test(constraint.box(lo, up), seed)


@pytest.mark.parametrize("upper", [1, 1e-14, 0])
@pytest.mark.parametrize("seed", range(5))
def test_one_sided_objects(upper, seed):
    np.random.seed(seed)
    b = np.random.uniform(0, upper, 1)
    _run_object_test(constraint.one_sided(np.array([1.0]), b), seed)
    _run_object_test(constraint.one_sided(np.array([-1.0]), b), seed)
    _run_object_test(constraint.upper(b), seed)
    _run_object_test(constraint.lower(-b), seed)


def test_constructor_errors():
    with pytest.raises(RuntimeError, match="upper must be >= 0"):
        constraint.box(np.array([-1.0]), np.array([-0.5]))
    with pytest.raises(RuntimeError, match="lower must be <= 0"):
        constraint.box(np.array([0.5]), np.array([1.0]))
    with pytest.raises(RuntimeError, match=r"sgn must be a vector of \+/-1"):
        constraint.one_sided(np.array([2.0]), np.array([1.0]))
    with pytest.raises(RuntimeError, match="b must be >= 0"):
        constraint.upper(np.array([-1.0]))
    with pytest.raises(NotImplementedError):
        constraint.linear(np.eye(2), np.zeros(2), np.ones(2))


def test_solve_zero_any_size():
    c = constraint.box(np.array([0.0, -1.0, 0.0]), np.array([1.0, 0.0, 0.0]))
    v = np.array([-2.0, 3.0, -4.0])
    # free where the bound is exactly zero (lower of 0 and 2, upper of 1 and 2), zero elsewhere
    assert np.isclose(c.solve_zero(v, None), 0.0)
    assert np.allclose(c._mu, [-2.0, 3.0, -4.0])
    v = np.array([2.0, -3.0, 4.0])
    assert np.isclose(c.solve_zero(v, None), np.linalg.norm([2.0, 3.0]))
    assert np.allclose(c._mu, [0.0, 0.0, 4.0])


# --------------------------------------------------------------------------------------------------------------------
# the path, oracle vs an independent solver
# --------------------------------------------------------------------------------------------------------------------
def _bounded_lasso(X, y, w, lm, pen, lo, hi):
    """argmin 1/2 sum w (yc - Xc b)^2 + lm sum pen |b| s.t. lo <= b <= hi by L-BFGS-B on b = u - v, u, v >= 0."""
    p = X.shape[1]
    xm = w @ X
    Xc = X - xm
    yc = y - w @ y

    def f(z):
        b = z[:p] - z[p:]
        r = yc - Xc @ b
        g = -Xc.T @ (w * r)
        return 0.5 * np.sum(w * r * r) + lm * np.sum(pen * (z[:p] + z[p:])), np.concatenate([g + lm * pen, -g + lm * pen])

    bounds = [(0, h if np.isfinite(h) else None) for h in hi] + [(0, -l if np.isfinite(l) else None) for l in lo]
    res = minimize(f, np.zeros(2 * p), jac=True, bounds=bounds, method="L-BFGS-B",
                   options=dict(ftol=1e-16, gtol=1e-12, maxiter=50000, maxfun=200000))
    return res.x[:p] - res.x[p:]


def _mixed_constraints(p, rng, dtype=np.float64):
    """A mix of every supported kind and None; returns (objects, lo, hi)."""
    cons, lo, hi = [], np.full(p, -np.inf), np.full(p, np.inf)
    for j in range(p):
        k = j % 5
        if k == 0:
            cons.append(None)
        elif k == 1:
            cons.append(constraint.lower(np.zeros(1, dtype=dtype)))
            lo[j] = 0
        elif k == 2:
            b = rng.uniform(0.02, 0.3)
            cons.append(constraint.upper(np.array([b], dtype=dtype)))
            hi[j] = b
        elif k == 3:
            l, u = -rng.uniform(0.02, 0.2), rng.uniform(0.02, 0.2)
            cons.append(constraint.box(np.array([l], dtype=dtype), np.array([u], dtype=dtype)))
            lo[j], hi[j] = l, u
        else:
            cons.append(constraint.box(np.zeros(1, dtype=dtype), np.array([0.5], dtype=dtype)))
            lo[j], hi[j] = 0, 0.5
    return cons, lo, hi


def test_oracle_constrained_path_is_optimal(oracle):
    d = make_gaussian(150, 25, seed=3)
    X, y = d["X"], d["y"]
    rng = np.random.RandomState(0)
    cons, lo, hi = _mixed_constraints(25, rng)
    w = np.full(150, 1 / 150)
    st = ad.grpnet(oracle.dense(X), ad.glm.gaussian(y), constraints=cons, lmda_path_size=12, min_ratio=0.02,
                   early_exit=False, tol=1e-14, progress_bar=False)
    B = st.betas.toarray()
    assert st.error == "" and B.shape == (12, 25)
    assert (B >= lo - 1e-12).all() and (B <= hi + 1e-12).all()
    assert (np.abs(B[-1] - np.clip(B[-1], lo, hi)) == 0).all()
    hit = (B[-1] == lo) & (lo != 0) | (B[-1] == hi) & (hi != 0)
    assert hit.sum() >= 3  # the bounds bind somewhere, otherwise this test checks nothing
    for i in [3, 7, 11]:
        ref = _bounded_lasso(X, y, w, st.lmdas[i], np.ones(25), lo, hi)
        assert np.abs(B[i] - ref).max() < 2e-6
    # duals: one column per constrained group (render_dual_groups), stationarity of the Lagrangian at every solution
    assert st.duals.shape == (12, 20)
    dg = constraint.render_dual_groups(cons)
    xm = w @ X
    for i in range(12):
        r = (y - w @ y) - (X - xm) @ B[i]
        grad = (X - xm).T @ (w * r)
        mu = np.zeros(25)
        row = st.duals[i].toarray().ravel()
        for j, c in enumerate(cons):
            if c is not None:
                g = np.empty(1)
                c.gradient(B[i, j:j + 1], row[dg[j]:dg[j] + 1], g)
                mu[j] = g[0]
        act = B[i] != 0
        if act.any():
            assert np.abs(grad[act] - st.lmdas[i] * np.sign(B[i, act]) - mu[act]).max() < 1e-6
        assert (np.abs(grad[~act] - mu[~act]) <= st.lmdas[i] * (1 + 1e-6)).all()
        # complementary slackness: a multiplier only where a bound is active
        inside = (B[i] > lo) & (B[i] < hi)
        assert np.abs(mu[inside]).max(initial=0) == 0


def test_constraints_that_never_bind_leave_the_path_unchanged(oracle):
    d = make_gaussian(120, 20, seed=5)
    cons = [constraint.box(np.array([-1e3]), np.array([1e3])) for _ in range(20)]
    kw = dict(lmda_path_size=10, min_ratio=0.05, early_exit=False, tol=1e-12, progress_bar=False)
    a = ad.grpnet(oracle.dense(d["X"]), ad.glm.gaussian(d["y"]), constraints=cons, **kw)
    b = ad.grpnet(oracle.dense(d["X"]), ad.glm.gaussian(d["y"]), **kw)
    assert np.array_equal(a.betas.toarray(), b.betas.toarray())
    assert a.duals.nnz == 0 and a.duals.shape == (10, 20) and b.duals.shape == (10, 0)


def test_constraint_errors(oracle):
    d = make_gaussian(60, 8, seed=1)
    X, glm = oracle.dense(d["X"]), ad.glm.gaussian(d["y"])
    two = constraint.box(np.array([-1.0, -1.0]), np.array([1.0, 1.0]))
    with pytest.raises(NotImplementedError, match="one coefficient"):
        ad.grpnet(X, glm, groups=np.arange(0, 8, 2), constraints=[two, None, None, None], progress_bar=False)
    c = constraint.lower(np.zeros(1))
    with pytest.raises(RuntimeError, match="distinct objects"):
        ad.grpnet(X, glm, constraints=[c, c] + [None] * 6, progress_bar=False)
    with pytest.raises(RuntimeError, match="group's size"):
        ad.grpnet(X, glm, groups=np.arange(0, 8, 2), constraints=[c, None, None, None], progress_bar=False)

    class mine:
        primal_size = dual_size = 1

        def clear(self):
            pass

    with pytest.raises(NotImplementedError, match="user-defined"):
        ad.grpnet(X, glm, constraints=[mine()] + [None] * 7, progress_bar=False)
    y2 = np.stack([d["y"], -d["y"]], axis=1)
    with pytest.raises(NotImplementedError, match="multi-response"):
        ad.grpnet(X, ad.glm.multigaussian(y2), constraints=[two] + [None] * 7, progress_bar=False)


def test_cv_grpnet_passes_constraints_to_every_fold(oracle):
    d = make_gaussian(150, 20, seed=2)
    cons = [constraint.lower(np.zeros(1)) for _ in range(20)]
    kw = dict(n_folds=3, seed=1, lmda_path_size=8, progress_bar=False)
    a = ad.cv_grpnet(oracle.dense(d["X"]), ad.glm.gaussian(d["y"]), constraints=cons, **kw)
    b = ad.cv_grpnet(oracle.dense(d["X"]), ad.glm.gaussian(d["y"]), **kw)
    assert a.losses.shape == b.losses.shape == (3, 8)
    assert np.abs(a.losses - b.losses).max() > 1e-6  # the non-negativity constraint changes the fits


# --------------------------------------------------------------------------------------------------------------------
# HIP vs oracle
# --------------------------------------------------------------------------------------------------------------------
def _fit(X, glm, cons, **kw):
    return ad.grpnet(X, glm, constraints=cons, progress_bar=False, **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["gaussian", "binomial", "poisson"])
def test_hip_constrained_path_matches_oracle(hip, oracle, family):
    n, p = 500, 300   # the screen set spans three 128-visit blocks by the end of the path
    d = make_gaussian(n, p, seed=7, sparsity=0.8)
    X = d["X"]
    rng = np.random.RandomState(1)
    if family == "gaussian":
        mk = lambda: ad.glm.gaussian(d["y"])
        kw = dict(lmda_path_size=15, min_ratio=0.02, early_exit=False, tol=1e-13)
    elif family == "binomial":
        yb = (rng.uniform(size=n) < 1 / (1 + np.exp(-X[:, :6] @ np.ones(6)))).astype(float)
        mk = lambda: ad.glm.binomial(yb)
        kw = dict(lmda_path_size=10, min_ratio=0.1, early_exit=False, tol=1e-13, irls_tol=1e-11)
    else:
        yp = rng.poisson(np.exp(0.3 * X[:, :4] @ np.ones(4))).astype(float)
        mk = lambda: ad.glm.poisson(yp)
        kw = dict(lmda_path_size=10, min_ratio=0.1, early_exit=False, tol=1e-13, irls_tol=1e-11)
    cons_o, lo, hi = _mixed_constraints(p, np.random.RandomState(0))
    cons_h, _, _ = _mixed_constraints(p, np.random.RandomState(0))
    ref = _fit(oracle.dense(X), mk(), cons_o, **kw)
    st = _fit(ad.matrix.dense(X), mk(), cons_h, **kw)
    assert st.error == "" and ref.error == "" and len(st.lmdas) == len(ref.lmdas)
    assert st.counters["n_panel_blocks"] > 0
    B, R = st.betas.toarray(), ref.betas.toarray()
    assert (B >= lo).all() and (B <= hi).all()           # clipping is exact
    assert ((B[-1] == lo) & (lo != 0) | (B[-1] == hi) & (hi != 0)).sum() >= 2  # bounds away from zero bind
    assert np.abs(B - R).max() < 1e-7
    assert np.abs(st.intercepts - ref.intercepts).max() < 1e-7
    assert np.allclose(st.devs, ref.devs, rtol=0, atol=1e-9)
    assert st.duals.shape == ref.duals.shape == (len(ref.lmdas), sum(c is not None for c in cons_o))
    assert np.abs((st.duals - ref.duals)).max() < 1e-6
    assert np.array_equal(st.screen_set, ref.screen_set)
    assert np.abs(st.abs_grad - ref.abs_grad).max() < 1e-6
    # the objects are left holding the multipliers of the last fit
    mu_h = np.array([0.0 if c is None else c._mu[0] for c in cons_h])
    mu_o = np.array([0.0 if c is None else c._mu[0] for c in cons_o])
    assert np.abs(mu_h - mu_o).max() < 1e-6 and np.abs(mu_o).max() > 0


@pytest.mark.gpu
def test_hip_nonnegative_lasso_f32_and_warm_start(hip, oracle):
    d = make_gaussian(400, 150, seed=9, sparsity=0.7)
    X32 = d["X"].astype(np.float32)
    y32 = d["y"].astype(np.float32)
    mk = lambda: [constraint.lower(np.zeros(1, dtype=np.float32)) for _ in range(150)]
    kw = dict(lmda_path_size=12, min_ratio=0.05, early_exit=False, tol=1e-7)
    ref = _fit(oracle.dense(X32), ad.glm.gaussian(y32, dtype=np.float32), mk(), **kw)
    st = _fit(ad.matrix.dense(X32), ad.glm.gaussian(y32, dtype=np.float32), mk(), **kw)
    B = st.betas.toarray()
    assert B.dtype == np.float32 and (B >= 0).all() and (B[-1] > 0).sum() > 5
    assert np.abs(B - ref.betas.toarray()).max() < 2e-4
    # continue the path from the solved state (constraints' multipliers travel in through constraint_mu)
    d64 = make_gaussian(300, 80, seed=2, sparsity=0.7)
    cons = [constraint.box(np.array([-0.05]), np.array([0.1])) for _ in range(80)]
    Xh = ad.matrix.dense(d64["X"])
    first = _fit(Xh, ad.glm.gaussian(d64["y"]), cons, lmda_path_size=8, min_ratio=0.3, early_exit=False, tol=1e-13)
    lm2 = first.lmdas[-1] * np.array([0.8, 0.6, 0.4])
    first.lmda_path = lm2
    first.setup_lmda_max = first.setup_lmda_path = False
    cont = first.solve(progress_bar=False)
    cons2 = [constraint.box(np.array([-0.05]), np.array([0.1])) for _ in range(80)]
    cold = _fit(Xh, ad.glm.gaussian(d64["y"]), cons2, lmda_path=np.concatenate([first.lmdas, lm2]), early_exit=False,
                tol=1e-13)
    assert np.abs(cont.betas.toarray() - cold.betas.toarray()[-3:]).max() < 1e-7
    assert np.abs(cont.duals.toarray() - cold.duals.toarray()[-3:]).max() < 1e-6


def _mixed_group_problem(p, rng):
    """Groups of sizes 1 and 3 interleaved; every second group of size one carries a constraint."""
    groups, j, k = [], 0, 0
    while j < p:
        groups.append(j)
        j += 1 if (k % 2 == 0) else min(3, p - j)
        k += 1
    groups = np.array(groups)
    sizes = np.diff(np.append(groups, p))
    cons, lo, hi = [], np.full(p, -np.inf), np.full(p, np.inf)
    n1 = 0
    for g, q in zip(groups, sizes):
        if q == 1 and n1 % 2 == 0:
            if n1 % 4 == 0:
                cons.append(constraint.lower(np.zeros(1)))
                lo[g] = 0
            else:
                u = rng.uniform(0.02, 0.2)
                cons.append(constraint.box(np.array([-u]), np.array([u])))
                lo[g], hi[g] = -u, u
        else:
            cons.append(None)
        n1 += q == 1
    return groups, cons, lo, hi


def test_oracle_constrained_singletons_inside_a_grouped_problem(oracle):
    d = make_gaussian(200, 40, seed=4)
    groups, cons, lo, hi = _mixed_group_problem(40, np.random.RandomState(0))
    kw = dict(groups=groups, alpha=0.7, lmda_path_size=10, min_ratio=0.02, early_exit=False, tol=1e-13)
    st = _fit(oracle.dense(d["X"]), ad.glm.gaussian(d["y"]), cons, **kw)
    B = st.betas.toarray()
    assert st.error == "" and (B >= lo - 1e-12).all() and (B <= hi + 1e-12).all()
    assert ((B[-1] == lo) | (B[-1] == hi)).sum() >= 1
    free = ad.grpnet(oracle.dense(d["X"]), ad.glm.gaussian(d["y"]), progress_bar=False, **kw)
    assert np.abs(free.betas.toarray() - B).max() > 1e-3  # the constraints matter on this problem
    assert st.duals.shape == (10, sum(c is not None for c in cons)) and st.duals.nnz > 0


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["gaussian", "binomial"])
@pytest.mark.parametrize("rot", ["1", "0"])
def test_hip_constrained_singletons_inside_a_grouped_problem(hip, oracle, monkeypatch, family, rot):
    monkeypatch.setenv("ADELIE_HIP_GROUP_ROT", rot)
    n, p = 500, 260
    d = make_gaussian(n, p, seed=12, sparsity=0.8)
    X = d["X"]
    rng = np.random.RandomState(3)
    groups, cons_o, lo, hi = _mixed_group_problem(p, np.random.RandomState(0))
    _, cons_h, _, _ = _mixed_group_problem(p, np.random.RandomState(0))
    if family == "gaussian":
        mk = lambda: ad.glm.gaussian(d["y"])
        kw = dict(groups=groups, alpha=0.6, lmda_path_size=12, min_ratio=0.03, early_exit=False, tol=1e-13)
    else:
        yb = (rng.uniform(size=n) < 1 / (1 + np.exp(-X[:, :6] @ np.ones(6)))).astype(float)
        mk = lambda: ad.glm.binomial(yb)
        kw = dict(groups=groups, alpha=0.6, lmda_path_size=8, min_ratio=0.1, early_exit=False, tol=1e-13, irls_tol=1e-11)
    ref = _fit(oracle.dense(X), mk(), cons_o, **kw)
    st = _fit(ad.matrix.dense(X), mk(), cons_h, **kw)
    assert st.error == "" and ref.error == "" and len(st.lmdas) == len(ref.lmdas)
    B, R = st.betas.toarray(), ref.betas.toarray()
    assert (B >= lo).all() and (B <= hi).all()
    assert ((B[-1] == lo) | (B[-1] == hi)).sum() >= 3
    assert np.abs(B - R).max() < 1e-7
    assert np.abs((st.duals - ref.duals)).max() < 1e-6 and ref.duals.nnz > 0
    assert np.abs(st.abs_grad - ref.abs_grad).max() < 1e-6
    # same groups screened (the order inside one screening step follows the sort of scores that agree to rounding only)
    assert np.array_equal(np.sort(st.screen_set), np.sort(ref.screen_set))
