"""The covariance method: ``matrix.dense(method="cov")`` and ``gaussian_cov``.

Reference: ``adelie/solver.py:39-351``, ``solver_gaussian_cov.hpp``, ``solver_gaussian_pin_cov.hpp``, ``matrix_cov_dense.ipp``;
its tests ``tests/test_matrix.py:128-166`` (``run_cov``) and ``tests/test_solver.py:978-1028`` (``test_gaussian_cov``: the
covariance path equals the naive path on the same lambdas); known answer ``quickstart.ipynb`` cell 29: ``63/100 ... [rdev:0.1%]``."""
import numpy as np
import pytest

import adelie_amd as ad
from util import make_gaussian


def run_cov(A, cA, dtype, seed=0):
    """The reference's check list for a MatrixCovBase (tests/test_matrix.py:128-166)."""
    rng = np.random.RandomState(seed)
    p = A.shape[0]
    atol = 1e-5 if dtype == np.float32 else 1e-13
    v = rng.normal(0, 1, p)
    nnzs = rng.binomial(1, 0.7, p).astype(bool)
    v[~nnzs] = 0
    indices = np.arange(p)[nnzs]
    values = v[indices].astype(dtype)
    for i in range(1, p + 1):
        subset = np.sort(rng.choice(p, i, replace=False))
        out = np.empty(i, dtype=dtype)
        cA.bmul(subset, indices, values, out)
        assert np.allclose(v.T @ A[:, subset], out, atol=atol)
    out = np.empty(p, dtype=dtype)
    cA.mul(indices, values, out)
    assert np.allclose(v.T @ A, out, atol=atol)
    q = min(10, p)
    out = np.empty((q, q), dtype=dtype, order="F")
    for i in range(p - q + 1):
        cA.to_dense(i, q, out)
        assert np.allclose(A[i:i + q, i:i + q], out, atol=atol)
    assert cA.rows() == p and cA.cols() == p and cA.shape == (p, p) and cA.ndim == 2
    with pytest.raises(RuntimeError, match="bmul"):
        cA.bmul(np.arange(2), indices, values, np.empty(3, dtype=dtype))
    with pytest.raises(RuntimeError, match="mul"):
        cA.mul(indices, values, np.empty(p + 1, dtype=dtype))
    with pytest.raises(RuntimeError, match="to_dense"):
        cA.to_dense(p - q + 1, q, out)


def _cov_matrix(n, p, dtype, order, seed=0):
    rng = np.random.RandomState(seed)
    X = rng.normal(size=(n, p))
    return np.asarray(X.T @ X / n, dtype=dtype, order=order)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("order", ["C", "F"])
@pytest.mark.parametrize("n,p", [(10, 1), (30, 23)])
def test_oracle_cov_dense(oracle, n, p, dtype, order):
    A = _cov_matrix(n, p, dtype, order)
    run_cov(A.astype(np.float64), oracle.cov_dense(A), dtype)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("order", ["C", "F"])
@pytest.mark.parametrize("n,p", [(10, 1), (30, 23), (50, 301)])
def test_hip_cov_dense(hip, n, p, dtype, order):
    A = _cov_matrix(n, p, dtype, order)
    run_cov(A.astype(np.float64), ad.matrix.dense(A, method="cov"), dtype)
    # a non-symmetric matrix keeps its orientation (the cov_* operations are those of the reference on any square matrix)
    rng = np.random.RandomState(1)
    B = np.asarray(rng.normal(size=(9, 9)), dtype=dtype, order=order)
    cB = ad.matrix.dense(B, method="cov")
    out = np.empty(4, dtype=dtype)
    cB.bmul(np.array([1, 3, 4, 8]), np.array([0, 2]), np.array([1.0, -2.0], dtype=dtype), out)
    assert np.allclose(out, B[0, [1, 3, 4, 8]] - 2 * B[2, [1, 3, 4, 8]], atol=1e-5)
    blk = np.empty((3, 3), dtype=dtype, order="F")
    cB.to_dense(2, 3, blk)
    assert np.allclose(blk, B[2:5, 2:5])


def _quickstart(dense, cov_dense):
    """quickstart.ipynb cells 5, 19-20, 27-32."""
    n, p = 100, 1000
    np.random.seed(0)
    X = np.random.normal(0, 1, (n, p))
    y = X[:, -1] * np.random.normal(0, 1) + np.random.normal(0, 1, n)
    groups = 10 * np.arange(p // 10)
    state = ad.grpnet(dense(np.asfortranarray(X)), ad.glm.gaussian(y), groups=groups, progress_bar=False)
    Xc = X - np.mean(X, axis=0)[None]
    yc = y - np.mean(y)
    A = Xc.T @ Xc / n
    v = Xc.T @ yc / n
    return state, ad.gaussian_cov(A=cov_dense(A), v=v, groups=groups)


def _check_quickstart(state, state_cov, capsys):
    err = capsys.readouterr().err.strip().split("\r")[-1]
    assert state_cov.error == "" and len(state_cov.lmdas) == 63                   # cell 29: 63/100
    assert "| 63/100 [" in err and err.endswith("[rdev:0.1%]"), err
    k = min(state_cov.betas.shape[0], state.betas.shape[0])
    assert np.allclose(state_cov.betas[:k].toarray(), state.betas[:k].toarray(), atol=1e-3)   # cell 32
    assert np.all(state_cov.intercepts == 0)
    assert np.all(np.diff(state_cov.devs) >= 0)
    d = state_cov.devs
    assert (d[-1] - d[-2]) <= 1e-3 * d[-1] and (d[-2] - d[-3]) > 1e-3 * d[-2]    # the rdev_tol exit rule, exactly once


def test_oracle_reproduces_quickstart_cov(oracle, capsys):
    _check_quickstart(*_quickstart(oracle.dense, oracle.cov_dense), capsys)


@pytest.mark.gpu
def test_hip_reproduces_quickstart_cov(hip, capsys):
    _check_quickstart(*_quickstart(ad.matrix.dense, lambda A: ad.matrix.dense(A, method="cov")), capsys)


def _cov_equals_naive(dense, cov_dense, n, p, G, alpha, dtype, tol):
    """tests/test_solver.py:978-1028: the same problem through both methods, the covariance path on the naive path's lambdas."""
    d = make_gaussian(n, p, G=G, seed=n + p, dtype=dtype)
    X, y, groups = d["X"], d["y"], d["groups"]
    # (p > n: stop where the reference's test stops, adev_tol = 0.2 — further down the minimiser is not unique)
    # (the default newton_tol = 1e-12 is below float32 resolution: the reference's Newton root find then reports
    # "max iterations reached", and so do the oracle and the HIP path)
    nt = dict(newton_tol=1e-12 if dtype == np.float64 else 1e-5)
    sn = ad.grpnet(dense(X), ad.glm.gaussian(y, dtype=dtype), groups=groups, alpha=alpha, intercept=False,
                   adev_tol=0.2 if p > n else 0.6, tol=tol, progress_bar=False, **nt)
    A = np.asfortranarray(X.astype(np.float64).T @ X.astype(np.float64) / n).astype(dtype)
    v = (X.astype(np.float64).T @ y.astype(np.float64) / n).astype(dtype)
    sc = ad.gaussian_cov(A=cov_dense(A), v=v, groups=groups, alpha=alpha, lmda_path=sn.lmdas, tol=tol, early_exit=False,
                         progress_bar=False, **nt)
    assert sn.error == "" and sc.error == "" and len(sc.lmdas) == len(sn.lmdas) > 2
    return sn, sc


@pytest.mark.parametrize("n,p,G,alpha", [(10, 50, 10, 1.0), (40, 13, 7, 1.0), (60, 40, 40, 0.5), (80, 30, 9, 0.3)])
def test_oracle_cov_equals_naive(oracle, n, p, G, alpha):
    sn, sc = _cov_equals_naive(oracle.dense, oracle.cov_dense, n, p, G, alpha, np.float64, 1e-16)
    # Stated tolerance 1e-6 (the reference's f64 threshold, tests/test_solver.py:444-445).  Where n < group size the
    # within-group Gram is singular and the two methods stop a hair apart (tol * y_var vs tol): 2.3e-7 at tol = 1e-12 (hence 1e-16 here),
    # 2.3e-9 at 1e-16, 2.2e-11 at 1e-20 on the first shape — the stopping rule's resolution, not a different minimiser.
    assert np.abs(sn.betas.toarray() - sc.betas.toarray()).max() < 1e-6
    # devs of the covariance state are the unnormalised rsq of the naive state
    np.testing.assert_allclose(sc.devs, np.asarray(sn.devs) * sn.y_var, rtol=1e-6, atol=1e-10)


def _cov_kkt(A, v, st, groups, group_sizes, penalty, alpha):
    worst = 0.0
    B = st.betas.toarray()
    for l, lm in enumerate(st.lmdas):
        b = B[l]
        grad = v - A @ b
        for g, gs, pen in zip(groups, group_sizes, penalty):
            gg, bb = grad[g:g + gs], b[g:g + gs]
            nb = np.linalg.norm(bb)
            if nb == 0:
                worst = max(worst, np.linalg.norm(gg) - lm * alpha * pen)
            else:
                worst = max(worst, np.linalg.norm(gg - lm * pen * (alpha * bb / nb + (1 - alpha) * bb)))
    return worst


@pytest.mark.parametrize("alpha", [1.0, 0.4])
def test_oracle_cov_kkt_and_warm_start(oracle, alpha):
    d = make_gaussian(70, 45, G=14, seed=5, zero_pen=0.1)
    X, y = d["X"], d["y"]
    A, v = X.T @ X / 70, X.T @ y / 70
    cA = oracle.cov_dense(np.asfortranarray(A))
    kw = dict(groups=d["groups"], penalty=d["penalty"], alpha=alpha, tol=1e-13, early_exit=False, lmda_path_size=25,
              progress_bar=False)
    st = ad.gaussian_cov(cA, v, **kw)
    assert st.error == "" and len(st.lmdas) == 25
    assert _cov_kkt(A, v, st, d["groups"], d["group_sizes"], d["penalty"], alpha) < 1e-6
    assert np.abs(st.grad - (v - A @ st.betas[-1].toarray().ravel())).max() < 1e-10
    # warm start: continue the first half of the path from its final state
    half = ad.gaussian_cov(cA, v, **dict(kw, lmda_path=st.lmdas[:12]))
    rest = ad.gaussian_cov(cA, v, warm_start=half, **dict(kw, lmda_path=st.lmdas[12:]))
    assert np.abs(rest.betas.toarray() - st.betas[12:].toarray()).max() < 1e-9
    assert rest.check(method="assert")


def test_cov_argument_errors(oracle):
    A = np.eye(4)
    with pytest.raises(ValueError, match="MatrixCovBase"):
        ad.gaussian_cov(A=oracle.dense(np.asfortranarray(A)), v=np.zeros(4))
    with pytest.raises(RuntimeError, match=r"v must be \(p,\)"):
        ad.gaussian_cov(A=oracle.cov_dense(A), v=np.zeros(3))
    with pytest.raises(RuntimeError, match=r"mat must be \(p, p\)"):
        oracle.cov_dense(np.zeros((3, 4)))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,p,G,alpha", [(10, 50, 10, 1.0), (40, 13, 7, 1.0), (60, 40, 40, 0.5), (80, 30, 9, 0.3),
                                         (900, 700, 700, 1.0), (800, 640, 64, 0.7)])
def test_hip_cov_matches_oracle(hip, oracle, n, p, G, alpha, dtype):
    """HIP covariance path vs the oracle's restatement of the reference algorithm on identical inputs; the last two shapes run
    the multi-CU Gram engines (screen sets beyond 256 values), lasso and groups."""
    tol = 1e-16 if dtype == np.float64 else 1e-7
    hn, hc = _cov_equals_naive(ad.matrix.dense, lambda A: ad.matrix.dense(A, method="cov"), n, p, G, alpha, dtype, tol)
    on, oc = _cov_equals_naive(oracle.dense, oracle.cov_dense, n, p, G, alpha, dtype, tol)
    atol = 1e-7 if dtype == np.float64 else 2e-3
    assert len(hc.lmdas) == len(oc.lmdas)
    assert np.abs(hc.betas.toarray() - oc.betas.toarray()).max() < atol
    assert np.abs(hc.betas.toarray() - hn.betas.toarray()).max() < (1e-6 if dtype == np.float64 else 5e-3)
    np.testing.assert_allclose(hc.devs, oc.devs, rtol=1e-5 if dtype == np.float64 else 1e-2,
                               atol=1e-9 if dtype == np.float64 else 1e-4)
    assert np.abs(hc.grad - oc.grad).max() < (1e-9 if dtype == np.float64 else 1e-3)
    assert sorted(hc.screen_set.tolist()) == sorted(oc.screen_set.tolist()) or dtype == np.float32
    if p >= 640:
        assert len(hc.screen_set) * (p // G) >= 256


@pytest.mark.gpu
def test_hip_cov_warm_start_errors_and_exit_cond(hip):
    d = make_gaussian(70, 45, G=14, seed=5, zero_pen=0.1)
    X, y = d["X"], d["y"]
    A, v = X.T @ X / 70, X.T @ y / 70
    cA = ad.matrix.dense(np.asfortranarray(A), method="cov")
    kw = dict(groups=d["groups"], penalty=d["penalty"], alpha=0.6, tol=1e-13, early_exit=False, lmda_path_size=25,
              progress_bar=False)
    st = ad.gaussian_cov(cA, v, **kw)
    assert st.error == "" and len(st.lmdas) == 25
    assert _cov_kkt(A, v, st, d["groups"], d["group_sizes"], d["penalty"], 0.6) < 1e-6
    half = ad.gaussian_cov(cA, v, **dict(kw, lmda_path=st.lmdas[:12]))
    rest = ad.gaussian_cov(cA, v, warm_start=half, check_state=True, **dict(kw, lmda_path=st.lmdas[12:]))
    assert np.abs(rest.betas.toarray() - st.betas[12:].toarray()).max() < 1e-9
    cut = ad.gaussian_cov(cA, v, exit_cond=lambda s: len(s.lmdas) >= 7 and s.grad.shape == (45,), **kw)
    assert len(cut.lmdas) == 7
    with pytest.raises(RuntimeError, match="covariance matrix"):
        ad.grpnet(cA, ad.glm.gaussian(y))
    small = ad.gaussian_cov(cA, v, max_active_size=2, **kw)
    assert small.error.startswith("adelie_core solver: ") and 0 < len(small.lmdas) < 25


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["dense64", "dense32", "snp"])
def test_lazy_cov_is_xtx_and_solves_like_the_dense_cov_matrix(hip, kind):
    """matrix.lazy_cov (reference MatrixCovLazyCov): A = X^T X formed on the device; checked entry-wise through the
    MatrixCovBase members and through a gaussian_cov path against dense(A, method="cov")."""
    rng = np.random.RandomState(4)
    n, p = 700, 2300   # more than one 2048-column panel
    if kind == "snp":
        cd = rng.choice(np.array([0, 1, 2, -9], dtype=np.int8), size=(n, p), p=[0.6, 0.2, 0.1, 0.1])
        Xd = ad.matrix.snp_calldata(cd)
        imp = np.asarray(Xd.impute())
        X = np.where(cd < 0, imp[None, :], cd).astype(np.float64)
        A = ad.matrix.lazy_cov(Xd)
        tol = 1e-9
    else:
        dt = np.float64 if kind == "dense64" else np.float32
        X = np.asfortranarray(rng.normal(size=(n, p)).astype(dt))
        A = ad.matrix.lazy_cov(X)
        tol = 1e-9 if dt == np.float64 else 2e-3
    G = X.astype(np.float64).T @ X.astype(np.float64)
    assert A.shape == (p, p)
    for i, q in [(0, 5), (2040, 20), (p - 7, 7)]:
        out = np.empty((q, q), dtype=A.dtype, order="F")
        A.to_dense(i, q, out)
        assert np.abs(out - G[i:i + q, i:i + q]).max() < tol * n
    idx = np.array([3, 1000, 2047, 2048, p - 1])
    vals = rng.normal(size=5).astype(A.dtype)
    out = np.empty(p, dtype=A.dtype)
    A.mul(idx, vals, out)
    assert np.abs(out - vals.astype(np.float64) @ G[idx]).max() < tol * n * 5
    if kind == "dense64":
        y = X[:, :6] @ np.ones(6) + rng.normal(size=n)
        v = X.T @ y / n
        kw = dict(lmda_path_size=15, min_ratio=0.2, tol=1e-12, progress_bar=False)
        Al = ad.matrix.lazy_cov(X / np.sqrt(n))
        a = ad.gaussian_cov(Al, v, **kw)
        b = ad.gaussian_cov(ad.matrix.dense(np.asfortranarray(G / n), method="cov"), v, **kw)
        assert a.error == "" and len(a.lmdas) == len(b.lmdas)
        assert np.abs(a.betas.toarray() - b.betas.toarray()).max() < 1e-9


@pytest.mark.gpu
def test_block_diag_and_sparse_cov_matrices(hip):
    """matrix.block_diag(method="cov") and matrix.sparse(method="cov") (reference MatrixCovBlockDiag / MatrixCovSparse):
    materialised in HBM, the MatrixCovBase members agree with the dense matrix they stand for."""
    from scipy.sparse import csc_matrix
    rng = np.random.RandomState(8)
    blocks = []
    for q in (3, 17, 40):
        B = rng.normal(size=(q + 5, q))
        blocks.append(np.asfortranarray(B.T @ B / (q + 5)))
    A = ad.matrix.block_diag([blocks[0], ad.matrix.dense(blocks[1], method="cov"), blocks[2]], method="cov")
    p = 60
    ref = np.zeros((p, p))
    o = 0
    for b in blocks:
        ref[o:o + len(b), o:o + len(b)] = b
        o += len(b)
    out = np.empty((p, p), order="F")
    A.to_dense(0, p, out)
    assert np.array_equal(out, ref)
    S = csc_matrix(np.where(np.abs(ref) > 0.3, ref, 0.0))
    As = ad.matrix.sparse(S, method="cov")
    As.to_dense(0, p, out)
    assert np.array_equal(out, S.toarray())
    v = rng.normal(size=p)
    a = ad.gaussian_cov(A, v, lmda_path_size=10, min_ratio=0.1, tol=1e-12, progress_bar=False)
    b = ad.gaussian_cov(ad.matrix.dense(np.asfortranarray(ref), method="cov"), v, lmda_path_size=10, min_ratio=0.1, tol=1e-12,
                        progress_bar=False)
    assert np.array_equal(a.betas.toarray(), b.betas.toarray())
    with pytest.raises(NotImplementedError):
        ad.matrix.block_diag(blocks, method="naive")


# ---- per-group constraints with the covariance method (solver_gaussian_pin_cov.hpp:287-355,723-763) -------------------------
def _cov_constrained_problem(seed, n=400, p=48, gsz=3, tight=None):
    from adelie_amd import constraint

    rng = np.random.RandomState(seed)
    X = rng.randn(n, p)
    y = X[:, :8] @ rng.randn(8) + rng.randn(n)
    Xc, yc = X - X.mean(0), y - y.mean()
    A, v = Xc.T @ Xc / n, Xc.T @ yc / n
    groups = np.arange(0, p, gsz)
    G = len(groups)
    spec = [None] * G
    for g in rng.choice(G, 5, replace=False):
        spec[g] = ("box", -rng.uniform(0.01, 0.1, gsz), rng.uniform(0.01, 0.1, gsz))
    free = [g for g in range(G) if spec[g] is None]
    for g in rng.choice(free, 2, replace=False):
        spec[g] = ("linear", rng.randn(2, gsz), -rng.uniform(0.01, 0.05, 2), rng.uniform(0.01, 0.05, 2))

    def make():
        out = []
        for s in spec:
            if s is None:
                out.append(None)
            elif s[0] == "box":
                out.append(constraint.box(s[1], s[2], configs=tight))
            else:
                out.append(constraint.linear(s[1], s[2], s[3], configs=({**tight, "nnls_tol": 1e-12} if tight else None)))
        return out
    return np.asfortranarray(X), y, A, v, groups, make


def test_oracle_cov_with_constraints_equals_naive(oracle):
    """The covariance method with constraint objects against the naive method on the same (centred, 1/n-weighted) problem and
    lambdas: same coefficients and multipliers (the reference's test_gaussian_cov idea, tests/test_solver.py:978-1040, with its
    constraints list non-empty)."""
    tight = {"tol": 1e-13, "pinball_tol": 1e-12, "max_iters": 1000}
    X, y, A, v, groups, make = _cov_constrained_problem(3, tight=tight)
    kw = dict(groups=groups, alpha=0.8, tol=1e-13, early_exit=False, progress_bar=False)
    sc = ad.gaussian_cov(oracle.cov_dense(A), v, constraints=make(), lmda_path_size=12, min_ratio=0.05, **kw)
    sn = ad.grpnet(oracle.dense(X), ad.glm.gaussian(y), constraints=make(), lmda_path=sc.lmdas, **kw)
    assert sc.error == "" and sn.error == "" and sc.duals.nnz > 0
    # (the two methods stop on different scalings of the same rule: tol for the covariance method, tol * y_var for the naive one)
    assert np.abs(sc.betas.toarray() - sn.betas.toarray()).max() < 1e-6
    assert sc.duals.shape == sn.duals.shape and np.abs((sc.duals - sn.duals)).max() < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_hip_cov_with_constraints_matches_oracle(hip, oracle, dtype):
    """Constraint objects (host visits between Gram-engine blocks) and one-coefficient closed forms (clips in the group solve)
    under the covariance method: HIP vs the oracle."""
    from adelie_amd import constraint

    f32 = dtype == np.float32
    tight = None if f32 else {"tol": 1e-13, "pinball_tol": 1e-12, "max_iters": 1000}
    X, y, A, v, groups, make = _cov_constrained_problem(5, tight=tight)
    A, v = A.astype(dtype), v.astype(dtype)
    kw = dict(groups=groups, alpha=0.8, tol=1e-7 if f32 else 1e-13, early_exit=False, progress_bar=False, lmda_path_size=10,
              min_ratio=0.05)
    if f32:
        kw["newton_tol"] = 1e-5
    a = ad.gaussian_cov(ad.matrix.dense(A, method="cov"), v, constraints=make(), **kw)
    b = ad.gaussian_cov(oracle.cov_dense(A), v, constraints=make(), **kw)
    assert a.error == "" and b.error == "", (a.error, b.error)
    tol = 5e-3 if f32 else 1e-6
    assert np.abs(a.betas.toarray() - b.betas.toarray()).max() < tol
    assert a.duals.shape == b.duals.shape and b.duals.nnz > 0
    assert np.abs((a.duals - b.duals)).max() < (5e-2 if f32 else 1e-5)
    # singletons with closed-form bounds next to unconstrained singletons
    p = A.shape[0]
    mk1 = lambda: [constraint.box(np.array([-0.02]), np.array([0.03]), dtype=dtype) if j % 3 == 0 else
                   (constraint.lower(np.zeros(1), dtype=dtype) if j % 3 == 1 else None) for j in range(p)]
    kw1 = dict(alpha=1.0, tol=1e-7 if f32 else 1e-13, early_exit=False, progress_bar=False, lmda_path_size=10, min_ratio=0.05)
    a1 = ad.gaussian_cov(ad.matrix.dense(A, method="cov"), v, constraints=mk1(), **kw1)
    b1 = ad.gaussian_cov(oracle.cov_dense(A), v, constraints=mk1(), **kw1)
    assert a1.error == "" and b1.error == ""
    B = a1.betas.toarray()
    assert (B[:, 0::3] >= -0.02 - 1e-7).all() and (B[:, 0::3] <= 0.03 + 1e-7).all() and (B[:, 1::3] >= -1e-7).all()
    assert np.abs(B - b1.betas.toarray()).max() < tol
    assert np.abs((a1.duals - b1.duals)).max() < (5e-2 if f32 else 1e-6) and b1.duals.nnz > 0
