"""Post-fit helpers used by ``cv_grpnet`` — mirrors the parts of ``adelie.diagnostic`` on the hot path:
``predict`` (reference ``adelie/diagnostic.py:30-121``), ``coefficient`` (``:577-646``) and ``objective``
(``:124-278``, the parity fall-back metric of the reference's own tests, ``tests/test_solver.py:447-466``).
Plotting and the other diagnostics are out of scope."""
import logging

import numpy as np
from scipy.sparse import csr_matrix

from . import matrix

logger = logging.getLogger("adelie_amd")


def predict(X, betas, intercepts, offsets=None, n_threads: int = 1):
    """Linear predictions ``eta_l = X beta_l + intercept_l + offsets`` for every row of ``betas``
    (reference ``diagnostic.py:30-121``, single-response branch).  ``betas`` CSR goes through
    ``X.sp_tmul`` — one device kernel over the resident design."""
    intercepts = np.atleast_1d(intercepts)
    is_multi = len(intercepts.shape) == 2
    if isinstance(X, np.ndarray):
        X = matrix.dense(X, method="naive", n_threads=n_threads)
    if is_multi:  # (L, n, K) predictions through X (x) I_K  (diagnostic.py:84-88)
        K = intercepts.shape[1]
        n = X.rows()
        X = matrix.kronecker_eye(X, K, n_threads=n_threads)
        dtype = X.dtype
        if offsets is None:
            offsets = np.zeros((n, K), dtype=dtype)
        if isinstance(betas, np.ndarray):
            betas = csr_matrix(np.atleast_2d(betas))
        L = betas.shape[0]
        etas = np.zeros((L, n * K), order="C", dtype=dtype)
        X.sp_tmul(betas, etas)
        return etas.reshape(L, n, K) + intercepts[:, None] + np.asarray(offsets, dtype=dtype).reshape(n, K)
    n = X.rows()
    dtype = X.dtype
    if offsets is None:
        offsets = np.zeros((n,), dtype=dtype)
    if isinstance(betas, np.ndarray):
        betas = np.atleast_2d(betas)
    L = betas.shape[0]
    etas = np.zeros((L, n), order="C", dtype=dtype)
    if isinstance(betas, np.ndarray):
        for i in range(L):
            X.btmul(0, X.cols(), betas[i], etas[i])
    elif isinstance(betas, csr_matrix):
        X.sp_tmul(betas, etas)
    else:
        raise RuntimeError("beta is not one of np.ndarray or scipy.sparse.csr_matrix.")
    etas += intercepts[:, None] + offsets
    return etas


def coefficient(*, lmda: float, betas: csr_matrix, intercepts: np.ndarray, lmdas: np.ndarray):
    """Linearly interpolated coefficient / intercept at ``lmda`` (reference ``diagnostic.py:577-646``)."""
    if len(lmdas) == 0:
        raise RuntimeError("lmdas must be non-empty!")
    if len(lmdas) == 1:
        return betas, lmdas  # (sic) reference diagnostic.py:623
    order = np.argsort(lmdas)
    idx = np.searchsorted(lmdas, lmda, sorter=order)
    idx = lmdas.shape[0] - idx
    if idx == 0 or idx == lmdas.shape[0]:
        logger.warning("lmda is not within the range of the saved lambdas. Returning boundary solution.")
        idx = np.clip(idx, 0, lmdas.shape[0] - 1)
        return betas[idx], intercepts[idx]
    left, right = betas[idx - 1], betas[idx]
    weight = (lmda - lmdas[idx]) / (lmdas[idx - 1] - lmdas[idx])
    beta = left.multiply(weight) + right.multiply(1 - weight)
    left, right = intercepts[idx - 1], intercepts[idx]
    intercept = weight * left + (1 - weight) * right
    return beta, intercept


def objective(X, glm, betas, intercepts, lmdas, *, groups=None, alpha: float = 1, penalty=None, offsets=None,
              relative: bool = True, add_penalty: bool = True, n_threads: int = 1):
    """Group elastic net objective ``loss(eta) - loss_full + lmda * sum_g w_g (alpha|b_g| + (1-alpha)/2 |b_g|^2)``
    (reference ``diagnostic.py:124-278``, penalty helper ``py_solver.cpp:10-80``)."""
    if isinstance(X, np.ndarray):
        X = matrix.dense(X, method="naive", n_threads=n_threads)
    p = X.cols()
    dtype = X.dtype
    if groups is None:
        groups = np.arange(p, dtype=int)
    group_sizes = np.concatenate([groups, [p]], dtype=int)
    group_sizes = group_sizes[1:] - group_sizes[:-1]
    if penalty is None:
        penalty = np.sqrt(group_sizes)
    etas = predict(X, betas, intercepts, offsets=offsets, n_threads=n_threads)
    objs = np.array([glm.loss(etas[i]) for i in range(etas.shape[0])], dtype=dtype)
    if relative:
        objs -= glm.loss_full()
    if add_penalty:
        B = betas.toarray() if isinstance(betas, csr_matrix) else np.atleast_2d(betas)
        pen = np.zeros(B.shape[0], dtype=dtype)
        for g, gs, w in zip(groups, group_sizes, penalty):
            nrm = np.linalg.norm(B[:, g:g + gs], axis=1)
            pen += w * (alpha * nrm + 0.5 * (1 - alpha) * nrm ** 2)
        objs += np.asarray(lmdas) * pen
    return objs
