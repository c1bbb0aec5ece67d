"""Post-fit helpers — mirrors the numeric parts of ``adelie.diagnostic``: ``predict`` (reference
``adelie/diagnostic.py:30-121``), ``coefficient`` (``:577-646``), ``objective`` (``:124-278``, the parity fall-back
metric of the reference's own tests, ``tests/test_solver.py:447-466``) and the KKT diagnostics ``residuals`` /
``gradients`` / ``gradient_norms`` / ``gradient_scores`` / ``diagnostic(state)`` (``:279-574,1248-1415``): the L
gradient sweeps ``X^T r_l`` go through one ABI call that streams the resident design once per eight vectors.
Plotting is out of scope."""
import logging

import numpy as np
from scipy.sparse import csr_matrix

from . import matrix

logger = logging.getLogger("adelie_amd")


def _linear_maps(X, B, out):
    """``out[l] = X @ B[l]`` for every row of ``B``: one ``sp_tmul`` kernel over the resident design for CSR rows, one
    ``btmul`` per row for a dense array."""
    if isinstance(B, csr_matrix):
        X.sp_tmul(B, out)
        return
    if not isinstance(B, np.ndarray):
        raise RuntimeError("beta is not one of np.ndarray or scipy.sparse.csr_matrix.")
    ncol = X.cols()
    for row, dst in zip(B, out):
        X.btmul(0, ncol, row, dst)


def predict(X, betas, intercepts, offsets=None, n_threads: int = 1):
    """Linear predictions ``eta_l = X beta_l + intercept_l + offsets`` for every solution ``l`` (semantics of the
    reference's ``diagnostic.py:30-121``): ``(L, n)`` for one response, ``(L, n, K)`` through the ``X (x) I_K`` view when
    ``intercepts`` is two-dimensional."""
    b0 = np.atleast_1d(intercepts)
    design = matrix.dense(X, method="naive", n_threads=n_threads) if isinstance(X, np.ndarray) else X
    n = design.rows()
    K = b0.shape[1] if b0.ndim == 2 else None
    if K is not None:
        design = matrix.kronecker_eye(design, K, n_threads=n_threads)
    dtype = design.dtype
    B = np.atleast_2d(betas) if isinstance(betas, np.ndarray) else betas
    if K is not None and isinstance(B, np.ndarray):
        B = csr_matrix(B)
    width = n if K is None else n * K
    etas = np.zeros((B.shape[0], width), dtype=dtype)
    _linear_maps(design, B, etas)
    if K is not None:
        etas = etas.reshape(-1, n, K)
    shift = 0 if offsets is None else np.asarray(offsets, dtype=dtype).reshape(etas.shape[1:])
    etas += b0[:, None] + shift
    return etas


def coefficient(*, lmda: float, betas: csr_matrix, intercepts: np.ndarray, lmdas: np.ndarray):
    """Solution at ``lmda`` by linear interpolation between the two saved solutions that bracket it (semantics of the
    reference's ``diagnostic.py:577-646``, including what it returns for a one-point path and its warning when ``lmda``
    falls outside the saved range)."""
    lmdas = np.asarray(lmdas)
    L = lmdas.shape[0]
    if L == 0:
        raise RuntimeError("lmdas must be non-empty!")
    if L == 1:
        return betas, lmdas  # the reference hands back (betas, lmdas) here, not (beta, intercept): diagnostic.py:623
    # On a decreasing path the number of saved lambdas >= lmda is the index of the first solution below lmda.
    below = int(np.count_nonzero(lmdas >= lmda))
    if below == 0 or below == L:
        logger.warning("lmda is not within the range of the saved lambdas. Returning boundary solution.")
        edge = min(below, L - 1)
        return betas[edge], intercepts[edge]
    above = below - 1
    t = (lmda - lmdas[below]) / (lmdas[above] - lmdas[below])
    beta = betas[above].multiply(t) + betas[below].multiply(1 - t)
    intercept = t * intercepts[above] + (1 - t) * intercepts[below]
    return beta, intercept


def coefficients(*, lmdas_new, betas: csr_matrix, intercepts: np.ndarray, lmdas: np.ndarray):
    """``coefficient`` for a whole vector of ``lmdas_new`` at once: ``(len(lmdas_new), p)`` CSR and the intercepts.  The same
    numbers as one call per value (``t * above + (1 - t) * below`` with the reference's ``t``, the boundary solution outside the
    saved range) through ONE sparse product with the (queries, L) interpolation matrix — ``cv_grpnet`` evaluates every fold at
    the full-data grid, and 100 calls of ``coefficient`` per fold, serialised by the interpreter lock across the folds in
    flight, were a fifth of an 8-fold CV's wall time."""
    lmdas = np.asarray(lmdas)
    q = np.atleast_1d(np.asarray(lmdas_new, dtype=lmdas.dtype))
    L = lmdas.shape[0]
    if L < 2:
        raise RuntimeError("coefficients() needs a path of at least two saved lambdas.")
    # lmdas is decreasing: the number of saved lambdas >= query is the index of the first solution below it
    below = L - np.searchsorted(lmdas[::-1], q, side="left")
    outside = (below == 0) | (below == L)
    if np.any(outside):
        logger.warning("lmda is not within the range of the saved lambdas. Returning boundary solution.")
    lo = np.clip(below, 1, L - 1)
    up = lo - 1
    t = (q - lmdas[lo]) / (lmdas[up] - lmdas[lo])
    edge = np.minimum(below, L - 1)
    t = np.where(outside, 1.0, t)
    up = np.where(outside, edge, up)
    lo = np.where(outside, edge, lo)
    one_minus = np.where(outside, 0.0, 1 - t)
    rows = np.arange(len(q))
    W = csr_matrix((np.concatenate([t, one_minus]), (np.concatenate([rows, rows]), np.concatenate([up, lo]))), shape=(len(q), L))
    ic = np.asarray(intercepts)
    return (W @ betas).tocsr(), t * ic[up] + one_minus * ic[lo]


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


def residuals(glm, etas):
    """``-grad loss(eta_l)`` for every row of ``etas`` (reference ``diagnostic.py:279-317``)."""
    etas = np.asarray(etas)
    resids = np.empty(etas.shape, dtype=glm.dtype)
    for eta, resid in zip(etas, resids):
        glm.gradient(eta, resid)
    return resids


def gradients(X, resids, *, n_threads: int = 1):
    """``X^T r_l`` for every residual ``r_l``: (L, n) -> (L, p); multi-response (L, n, K) -> (L, p, K), i.e.
    ``(X (x) I_K)^T vec(r_l^T)`` (reference ``diagnostic.py:320-387``, which calls ``X.mul`` once per residual and, for
    the multi-response case, once per view column group).  Here all the vectors of a call go to the device together
    (``adelie_hip_design_mul_batch``) and a dense design is read once per eight of them."""
    resids = np.asarray(resids)
    if isinstance(X, np.ndarray):
        X = matrix.dense(X, method="naive", n_threads=n_threads)
    if not hasattr(X, "mul_batch"):
        raise NotImplementedError("adelie_amd: gradients() needs a device-resident design (matrix.dense / snp_unphased).")
    dtype = X.dtype
    if resids.ndim == 3:
        L, n, K = resids.shape
        V = np.ascontiguousarray(np.transpose(resids, (0, 2, 1)), dtype=dtype).reshape(L * K, n)
        out = X.mul_batch(V).reshape(L, K, X.cols())
        return np.ascontiguousarray(np.transpose(out, (0, 2, 1)))
    return X.mul_batch(np.ascontiguousarray(resids, dtype=dtype))


def gradient_norms(grads, betas, duals, lmdas, *, constraints=None, groups=None, alpha: float = 1, penalty=None):
    """Group-wise norms ``|| gamma_g - lmda (1-alpha) w_g beta_g ||_2`` of the penalised gradient, (L, G)
    (reference ``diagnostic.py:389-520``).  Constraints are outside this package's scope: pass ``None``."""
    if constraints is not None and any(c is not None for c in constraints):
        raise NotImplementedError("adelie_amd: per-group constraints are outside the grpnet hot path (pass None).")
    grads = np.asarray(grads)
    lmdas = np.asarray(lmdas)
    if grads.ndim == 3:
        p, K = grads.shape[1:]
        groups = np.arange(p) if groups is None else np.asarray(groups)
        groups = groups * K
        total = p * K
    else:
        total = grads.shape[-1]
        groups = np.arange(total) if groups is None else np.asarray(groups)
    bounds = np.concatenate([groups, [total]]).astype(int)
    group_sizes = bounds[1:] - bounds[:-1]
    if penalty is None:
        penalty = np.sqrt(group_sizes)
    pen = np.repeat(np.asarray(penalty), group_sizes)
    L = grads.shape[0]
    B = betas if isinstance(betas, csr_matrix) else csr_matrix(np.atleast_2d(betas))
    G = grads.reshape(L, -1) - B.multiply(lmdas[:, None] * (1 - alpha) * pen[None])
    G = np.asarray(G)
    sq = np.add.reduceat(np.square(G), bounds[:-1], axis=1) if len(groups) else np.zeros((L, 0), dtype=G.dtype)
    return np.sqrt(sq)


def gradient_scores(grad_norms, lmdas, *, alpha: float = 1, penalty=None):
    """``h_g / (alpha w_g)`` where ``alpha w_g > 0`` and ``lmda`` elsewhere (reference ``diagnostic.py:523-574``): a
    solution satisfies the KKT conditions when every score is at most its ``lmda``."""
    grad_norms = np.asarray(grad_norms)
    lmdas = np.asarray(lmdas)
    denom = alpha * np.asarray(penalty, dtype=grad_norms.dtype)
    pos = denom > 0
    scores = np.empty_like(grad_norms)
    scores[:, pos] = grad_norms[:, pos] / denom[pos][None]
    scores[:, ~pos] = lmdas[:, None]
    return scores


class DiagnosticNaive:
    """The quantities ``adelie.diagnostic.DiagnosticNaive`` computes for a solved naive state (reference
    ``diagnostic.py:1248-1322``): linear predictions, residuals, gradients, gradient norms and scores along the path.
    The plotting methods of the reference class are not provided."""

    def __init__(self, state):
        self.state = state
        self.betas = state.betas
        self.duals = getattr(state, "duals", None)
        glm = state._glm
        self._is_multi = bool(glm.is_multi)
        self._n_classes = glm.y.shape[-1]
        groups, penalty = np.asarray(state.groups), np.asarray(state.penalty)
        if self._is_multi:  # drop the intercept columns of the view; groups are counted in features (diagnostic.py:1263-1270)
            p_begin = int(state.multi_intercept) * self._n_classes
            groups = groups[p_begin:] // self._n_classes - int(state.multi_intercept)
            penalty = penalty[p_begin:]
        self._args = {"groups": groups, "penalty": penalty, "constraints": None}
        X = getattr(state, "_X_raw", state._X)
        self.linear_preds = predict(X=X, betas=self.betas, intercepts=state.intercepts, offsets=state._offsets,
                                    n_threads=state.n_threads)
        self.residuals = residuals(glm=glm, etas=self.linear_preds)
        self.gradients = gradients(X=X, resids=self.residuals, n_threads=state.n_threads)
        self.gradient_norms = gradient_norms(grads=self.gradients, betas=self.betas, duals=self.duals, lmdas=state.lmdas,
                                             groups=groups, alpha=state.alpha, penalty=penalty)
        self.gradient_scores = gradient_scores(grad_norms=self.gradient_norms, lmdas=state.lmdas, alpha=state.alpha,
                                               penalty=penalty)


def diagnostic(state):
    """Diagnostic object of a solved state (reference ``diagnostic.py:1393-1415``; only naive states exist here)."""
    return DiagnosticNaive(state)
