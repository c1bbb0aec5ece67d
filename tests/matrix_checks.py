"""`run_naive`: every MatrixNaiveBase method against numpy on the densified matrix — restatement of the check
list in the reference's tests/test_matrix.py:251-409."""
import numpy as np
from scipy.sparse import csr_matrix


def run_naive(cX, X, dtype):
    n, p = X.shape
    atol = 1e-4 if dtype == np.float32 else 1e-12
    rng = np.random.RandomState(42)
    scale = max(1.0, np.abs(X).max()) * n
    tol = atol * scale if dtype == np.float64 else atol * max(1.0, np.abs(X).max()) * np.sqrt(n) * 10
    w = rng.uniform(0, 1, n).astype(dtype)
    X64 = X.astype(np.float64)

    assert cX.rows() == n and cX.cols() == p and cX.shape == (n, p) and cX.ndim == 2

    # cmul / cmul_safe
    v = rng.normal(size=n).astype(dtype)
    for j in range(0, p, max(1, p // 7)):
        ref = X64[:, j] @ (v.astype(np.float64) * w)
        assert abs(cX.cmul(j, v, w) - ref) <= tol
        assert abs(cX.cmul_safe(j, v, w) - ref) <= tol
    # ctmul
    for j in range(0, p, max(1, p // 5)):
        out = rng.normal(size=n).astype(dtype)
        ref = out + 0.37 * X64[:, j]
        cX.ctmul(j, 0.37, out)
        assert np.abs(out - ref).max() <= tol
    # bmul / btmul over several blocks
    for j, q in [(0, p), (0, 1), (p // 2, p - p // 2), (max(p - 3, 0), min(3, p))]:
        out = np.empty(q, dtype=dtype)
        cX.bmul(j, q, v, w, out)
        assert np.abs(out - (v.astype(np.float64) * w) @ X64[:, j:j + q]).max() <= tol
        cX.bmul_safe(j, q, v, w, out)
        assert np.abs(out - (v.astype(np.float64) * w) @ X64[:, j:j + q]).max() <= tol
        c = rng.normal(size=q).astype(dtype)
        out = rng.normal(size=n).astype(dtype)
        ref = out + X64[:, j:j + q] @ c
        cX.btmul(j, q, c, out)
        assert np.abs(out - ref).max() <= tol
    # mul
    out = np.empty(p, dtype=dtype)
    cX.mul(v, w, out)
    assert np.abs(out - (v.astype(np.float64) * w) @ X64).max() <= tol
    # cov
    for j, q in [(0, min(p, 6)), (p // 3, min(4, p - p // 3)), (0, p if p <= 70 else 70)]:
        out = np.empty((q, q), dtype=dtype, order="F")
        cX.cov(j, q, np.sqrt(w), out)
        ref = (X64[:, j:j + q].T * w) @ X64[:, j:j + q]
        assert np.abs(out - ref).max() <= tol
        assert np.abs(out - out.T).max() <= (0 if dtype == np.float64 else 1e-5) + 1e-30
    # sq_mul
    out = np.empty(p, dtype=dtype)
    cX.sq_mul(w, out)
    assert np.abs(out - w @ (X64 ** 2)).max() <= tol * max(1.0, np.abs(X).max()) * (4 if dtype == np.float32 else 1)
    # sp_tmul
    L = 5
    dense = rng.normal(size=(L, p)) * (rng.uniform(size=(L, p)) < 0.3)
    dense[1] = 0
    V = csr_matrix(dense.astype(dtype))
    out = np.empty((L, n), dtype=dtype)
    cX.sp_tmul(V, out)
    assert np.abs(out - dense @ X64.T).max() <= tol
    # mean / var defaults and the python sugar
    out = np.empty(p, dtype=dtype)
    cX.mean(w, out)
    assert np.abs(out - w @ X64).max() <= tol
    centers = out.copy()
    var = np.empty(p, dtype=dtype)
    cX.var(centers, w, var)
    ref = w @ (X64 - centers[None]) ** 2
    # f32: cancellation; the centres here are weighted SUMS (w is not normalised), so for columns with a non-zero mean the
    # variances are ~ (sum w)^2 x mean^2 and the rounding error scales with them, not with |X|
    rel = (1e-5 if dtype == np.float32 else 1e-13) * np.abs(ref).max()
    assert np.abs(var - ref).max() <= tol * max(1.0, np.abs(X).max()) * (32 if dtype == np.float32 else 4) + rel
    b = rng.normal(size=p).astype(dtype)
    assert np.abs(cX @ b - X64 @ b).max() <= tol
    assert np.abs(cX.T @ v - X64.T @ v).max() <= tol
