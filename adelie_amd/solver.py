"""``grpnet`` — the entry point ``adelie.solver.grpnet`` of the reference, on an MI355X.

The reference's ``grpnet`` (``adelie/solver.py:354-958``) computes the invariants of the starting point in numpy
(``X_means``, ``y_mean``, ``y_var``, the residual and its gradient: two sweeps over ``X``), fills the keyword
arguments of a state constructor and calls ``state.solve()``.  The same happens here; the file is organised by *what
is computed* rather than by family:

=====================  =====================================================================================
``_Layout``            the grouping in solver coordinates (``groups``, ``group_sizes``, ``penalty``)
``_start_*``           the invariants of a cold start for each of the four state kinds
``_resume``            the same names read off a solved state (``warm_start``)
``_STATE_OF``          which constructor of ``adelie_amd.state`` takes them
=====================  =====================================================================================

Every ``X.mul`` below is one C-ABI call into ``libadelie_hip.so`` = one full-gradient sweep kernel over the resident
design.  Multi-response families (``glm.multigaussian``, ``glm.multinomial``) are solved in the
coordinates of the expanded design ``[1 (x) I_K, X (x) I_K]`` (reference ``solver.py:700-844``).
"""
import os
import warnings
from dataclasses import dataclass
from typing import Callable

import numpy as np

from . import matrix
from . import state as _state

# What a warm start hands over, per state kind (attribute names of a solved state == constructor keywords).
_SCREEN_FIELDS = ("lmda", "lmda_max", "screen_set", "screen_beta", "screen_is_active", "active_set_size", "active_set")
_INVARIANT_FIELDS = {
    "gaussian": ("X_means", "y_mean", "y_var", "rsq", "resid", "resid_sum", "grad"),
    "glm": ("beta0", "eta", "resid", "grad", "loss_null", "loss_full"),
    "multigaussian": ("X_means", "y_var", "rsq", "resid", "resid_sum", "grad"),
    "multiglm": ("eta", "resid", "grad", "loss_null", "loss_full"),
}
_STATE_OF = {
    "gaussian": _state.gaussian_naive,
    "glm": _state.glm_naive,
    "multigaussian": _state.multigaussian_naive,
    "multiglm": _state.multiglm_naive,
}


@dataclass
class _Layout:
    """Grouping of the solver's coordinates."""

    groups: np.ndarray       # (G,) first column of each group
    group_sizes: np.ndarray  # (G,)
    penalty: np.ndarray      # (G,)

    @classmethod
    def single(cls, groups, p, penalty, dtype):
        groups = np.arange(p, dtype=int) if groups is None else np.asarray(groups, dtype=int)
        sizes = np.diff(np.append(groups, p))
        pen = np.sqrt(sizes).astype(dtype) if penalty is None else np.asarray(penalty, dtype=dtype)
        return cls(groups, sizes, pen)

    @classmethod
    def multi(cls, groups, p, K, intercept, penalty, dtype):
        """Feature groups become groups of ``size*K`` view columns; with an intercept the first ``K`` view columns are ``K``
        unpenalised singleton groups (reference ``solver.py:705-727``)."""
        groups = np.arange(p, dtype=int) if groups is None else np.asarray(groups, dtype=int)
        groups = groups * K
        if intercept:
            groups = np.concatenate([np.arange(K), K + groups]).astype(int)
        sizes = np.diff(np.append(groups, (p + bool(intercept)) * K))
        if penalty is None:
            pen = np.sqrt(sizes).astype(dtype)
            if intercept:
                pen[:K] = 0
        else:
            pen = np.asarray(penalty, dtype=dtype)
            if intercept:
                pen = np.concatenate([np.zeros(K, dtype=dtype), pen]).astype(dtype)
        return cls(groups, sizes, pen)


def _start_screen(layout, alpha, dtype):
    """Cold start of the screen / active sets: every group the penalty cannot zero out (``penalty <= 0`` or no lasso part) is
    screened and active from the beginning, at ``beta = 0``; ``lmda = inf`` marks "nothing solved yet" (reference
    ``solver.py:855-866``)."""
    G = len(layout.groups)
    always_in = np.flatnonzero((layout.penalty <= 0) | (alpha <= 0))
    k = len(always_in)
    active_set = np.empty(G, dtype=int)
    active_set[:k] = np.arange(k)
    return {
        "lmda": np.inf,
        "lmda_max": None,
        "screen_set": always_in,
        "screen_beta": np.zeros(int(layout.group_sizes[always_in].sum()), dtype=dtype),
        "screen_is_active": np.ones(k, dtype=bool),
        "active_set_size": k,
        "active_set": active_set,
    }


def _lasso_in_raw_coordinates(X, glm, constraints, groups, alpha, intercept, warm_start):
    """``(base design, scales, centers)`` when a fit on the lazily standardized view ``X`` is a lasso that can run on the base
    design's own columns (see :func:`grpnet`), else ``None``."""
    sparse_view = getattr(X, "_kind", None) == "sparse" and getattr(X, "_std", None) is not None and getattr(X, "_keep", None) is not None
    if not (isinstance(X, matrix._StdView) or sparse_view) or warm_start is not None or not intercept or not (0 < alpha <= 1):
        return None
    if getattr(glm, "is_multi", False):
        return None
    if constraints is not None and any(c is not None for c in constraints):
        return None
    p = X.cols()
    if groups is not None and not (len(groups) == p and np.array_equal(np.asarray(groups), np.arange(p))):
        return None
    if sparse_view:  # the standardized view of a design kept sparse: its plain design, without the epilogue corrections
        return X._keep, np.asarray(X._std[1], dtype=X.dtype), np.asarray(X._std[0], dtype=X.dtype)
    return X._base, X._s, X._c


def _to_standardized_coordinates(state, view, s, c, penalty, alpha=1):
    """The state of a lasso solved on the raw columns, re-expressed for the standardized view: ``beta~ = s beta``, the
    intercepts take ``sum_j beta_j c_j``, gradients divide by ``s`` (``grad`` of the Gaussian state is the centred gradient,
    that of a GLM state ``X' resid``), means and variances of the screened columns follow."""
    dtype = state.dtype
    s = np.asarray(s, dtype=dtype)
    c = np.asarray(c, dtype=dtype)
    cols = np.asarray(state.groups)[np.asarray(state.screen_set, dtype=np.int64)]
    raw_screen = np.asarray(state.screen_beta)
    shift_now = float(np.dot(raw_screen, c[cols])) if len(cols) else 0.0
    state.intercepts = (np.asarray(state.intercepts) + np.asarray(state.betas @ c).reshape(-1)).astype(dtype)
    state.betas = state.betas.multiply(s[None]).tocsr().astype(dtype)
    state.screen_beta = (raw_screen * s[cols]).astype(dtype)
    if hasattr(state, "X_means"):   # Gaussian state: `grad` is centred (X_c' W r), so the centres drop out
        state.grad = (np.asarray(state.grad) / s).astype(dtype)
        state.X_means = ((np.asarray(state.X_means) - c) / s).astype(dtype)
        # its residual carries no intercept term (y_c - X beta; the intercept lives in resid_sum): X~ beta~ = X beta - sum_j beta_j c_j
        state.resid = (np.asarray(state.resid) + dtype(shift_now)).astype(dtype)
        state.resid_sum = dtype(state.resid_sum + shift_now * np.sum(state.weights))
    else:                           # GLM state: grad = X' resid with the weights inside resid
        state.grad = ((np.asarray(state.grad) - c * np.sum(state.resid)) / s).astype(dtype)
        if hasattr(state, "beta0"):
            state.beta0 = dtype(state.beta0 + shift_now)
    state.abs_grad = np.abs(state.grad)
    if alpha != 1 and len(cols) and np.any(state.screen_beta != 0):  # solver_base.hpp:20-110: minus the quadratic part's gradient
        shrink = np.zeros(len(s), dtype=dtype)
        shrink[cols] = dtype((1 - alpha) * state.lmda) * np.asarray(penalty, dtype=dtype)[cols] * state.screen_beta
        state.abs_grad = np.abs(state.grad - shrink)
    if len(np.asarray(state.screen_X_means)) == len(cols):  # (a GLM state keeps none: IRLS recomputes them per iteration)
        state.screen_X_means = ((np.asarray(state.screen_X_means) - c[cols]) / s[cols]).astype(dtype)
    if len(np.asarray(state.screen_vars)) == len(cols):
        state.screen_vars = (np.asarray(state.screen_vars) / s[cols] ** 2).astype(dtype)
    state.penalty = np.asarray(penalty, dtype=dtype)
    if getattr(state, "_penalty_l2", None) is not None:  # (the view's own coordinates have one factor per group again)
        state._penalty_l2 = None
    state._X = view
    if hasattr(state, "X"):
        state.X = view
    return state


def _resume(warm_start, kind):
    return {name: getattr(warm_start, name) for name in _SCREEN_FIELDS + _INVARIANT_FIELDS[kind]}


def _sweep(X, v, weights, dtype):
    """``X^T (v * weights)`` — one pass over the resident design."""
    out = np.empty(X.cols(), dtype=dtype)
    X.mul(np.ascontiguousarray(v, dtype=dtype), weights, out)
    return out


def _has_batched_sweep(X):
    """``X.mul_batch`` exists and the library behind ``X`` exports the batched sweep (user-defined matrix classes and
    libraries without the entry point take one ``X.mul`` per vector)."""
    backend = getattr(X, "_backend", None)
    if not hasattr(X, "mul_batch") or backend is None:
        return False
    # the entry point takes resident dense / 2-bit designs; covariance matrices and multi-response views go one vector at a time
    if isinstance(X, (matrix.MatrixCovBase64, matrix.MatrixCovBase32, matrix._MultiView, matrix._StdView)):
        return False
    try:
        backend.fn("design_mul_batch")
    except AttributeError:
        return False
    return True


def _start_gaussian(X, glm, offsets, intercept, dtype):
    """Invariants of ``beta = 0`` for the Gaussian loss ``sum_i w_i (eta_i^2 / 2 - y_i eta_i)`` (reference ``solver.py:886-906``).
    ``grad`` is handed over WITHOUT the ``-resid_sum * X_means`` centring term: at the start ``resid_sum`` is zero when there
    is an intercept, and without one the term does not exist."""
    w = glm.weights
    n = X.rows()
    y_off = glm.y - offsets
    y_mean = np.sum(y_off * w)
    resid = y_off - y_mean if intercept else y_off
    X_means = grad = None
    if _has_batched_sweep(X):
        # both sweeps in ONE pass over the resident design (two vectors side by side): the reference makes two X.mul calls.
        # (The two-vector kernel sums in another order than the one-vector sweep the solver uses later: the starting
        # invariants agree with it to rounding, not bit for bit.)
        try:
            X_means, grad = X.mul_batch(np.stack([np.asarray(w, dtype=dtype), np.asarray(resid * w, dtype=dtype)]))
        except RuntimeError as e:
            # a native design kind the batched entry point refuses (the class list above is a shortcut, the handle decides):
            # fall back to the two single-vector sweeps; anything else the library raised is a real error
            if "HIP error" in str(e):
                raise
            X_means = grad = None
    if grad is None:
        X_means = _sweep(X, np.ones(n, dtype=dtype), w, dtype)
        grad = _sweep(X, resid, w, dtype)
    return {
        "X_means": X_means,
        "y_mean": y_mean,
        "y_var": np.sum(w * resid ** 2),
        "rsq": 0,
        "resid": resid,
        "resid_sum": np.sum(w * resid),
        "grad": grad,
    }


def _start_glm(X, glm, offsets, dtype):
    """Invariants of ``beta = 0, beta0 = 0`` for a general GLM: ``resid`` is the negative gradient of the loss at
    ``eta = offsets`` and already carries the weights, hence the unit weights of the sweep (reference ``solver.py:925-937``)."""
    n = X.rows()
    resid = np.empty(n, dtype=dtype)
    glm.gradient(offsets, resid)
    return {
        "beta0": 0,
        "eta": offsets,
        "resid": resid,
        "grad": _sweep(X, resid, np.ones(n, dtype=dtype), dtype),
        "loss_null": None,
        "loss_full": glm.loss_full(),
    }


def _view_gradient(X, R, weights, intercept, dtype):
    """``[1 (x) I_K, X (x) I_K]^T vec(weights * R)`` for an ``(n, K)`` matrix ``R``, in view-column order (feature-major,
    response-minor): one sweep of the base design per response; the intercept block is the weighted column sums."""
    p, K = X.cols(), R.shape[1]
    out = np.empty((p + bool(intercept), K), dtype=dtype)
    for l in range(K):
        col = np.ascontiguousarray(R[:, l], dtype=dtype)
        out[bool(intercept):, l] = _sweep(X, col, weights, dtype)
        if intercept:
            out[0, l] = np.sum(col * weights)
    return out.ravel()


def _start_multigaussian(X, glm, offsets, intercept, dtype):
    """The multi-response Gaussian loss is the single-response one on the view with weights ``w_i / K`` (reference
    ``solver.py:742-800``).  With an intercept, R^2 is reported relative to the intercept-only model: it starts at
    ``-(gain of the intercepts)`` and the fit of the K unpenalised intercept columns brings it to zero."""
    K = glm.y.shape[-1]
    wK = glm.weights / K
    n = X.rows()
    means = np.repeat(_sweep(X, np.ones(n, dtype=dtype), wK, dtype), K)
    if intercept:
        means = np.concatenate([np.full(K, 1 / K), means]).astype(dtype)
    y_off = glm.y - offsets
    y_var = np.sum(wK[:, None] * y_off ** 2)
    rsq = 0
    if intercept:
        centred = y_off - (y_off.T @ glm.weights)[None]  # full weights here, as the reference has it (solver.py:769)
        centred_var = np.sum(wK[:, None] * centred ** 2)
        rsq, y_var = centred_var - y_var, centred_var
    return {
        "X_means": means,
        "y_var": y_var,
        "rsq": rsq,
        "resid": np.ascontiguousarray(y_off, dtype=dtype).ravel(),
        "resid_sum": np.sum(wK[:, None] * y_off),
        "grad": _view_gradient(X, y_off, wK, intercept, dtype),
    }


def _start_multiglm(X, glm, offsets, intercept, dtype):
    """Reference ``solver.py:818-831``; arrays cross into the state flattened ``(n, K)`` row-major."""
    resid = np.empty(offsets.shape, dtype=dtype)
    glm.gradient(offsets, resid)
    return {
        "eta": offsets.ravel(),
        "resid": resid.ravel(),
        "grad": _view_gradient(X, resid, np.ones(X.rows(), dtype=dtype), intercept, dtype),
        "loss_null": None,
        "loss_full": glm.loss_full(),
    }


def grpnet(
    X, glm, *,
    constraints: list = None, groups: np.ndarray = None, alpha: float = 1, penalty: np.ndarray = None,
    offsets: np.ndarray = None, lmda_path: np.ndarray = None,
    irls_max_iters: int = int(1e4), irls_tol: float = 1e-7,
    max_iters: int = int(1e5), tol: float = 1e-7, adev_tol: float = 0.9, ddev_tol: float = 0,
    newton_tol: float = 1e-12, newton_max_iters: int = 1000,
    n_threads: int = 1, early_exit: bool = True, intercept: bool = True,
    screen_rule: str = "pivot", min_ratio: float = 1e-2, lmda_path_size: int = 100,
    max_screen_size: int = None, max_active_size: int = None,
    pivot_subset_ratio: float = 0.1, pivot_subset_min: int = 1, pivot_slack_ratio: float = 1.25,
    check_state: bool = False, progress_bar: bool = True, warm_start=None, exit_cond: Callable = None,
    _penalty_l2: np.ndarray = None, _prepare_only: bool = False, _lmda_aug=None,
):
    """Group elastic net along a decreasing path of ``lmda`` on an MI355X (naive method).

    Minimises ``l(eta) + lmda * sum_g penalty_g (alpha ||beta_g|| + (1 - alpha) / 2 ||beta_g||^2)`` with
    ``eta = X beta + beta0 + offsets``.  Arguments, defaults and the returned state are those of the reference
    (``adelie/solver.py:388-620`` documents each of them); ``X`` may be an ndarray, a device tensor wrapped by
    ``adelie_amd.matrix.dense``, any other ``adelie_amd.matrix`` design, or a Python subclass of
    ``matrix.MatrixNaiveBase64/32`` (densified once into HBM through its own ``ctmul``); ``glm`` any ``adelie_amd.glm`` family
    or a Python subclass of ``glm.GlmBase64/32`` (evaluated by host callbacks once per IRLS iteration).

    Returns
    -------
    state
        The solved state (``betas`` CSR ``(L, p)``, ``intercepts``, ``devs``, ``lmdas`` and the invariants).
    """
    X = matrix.as_design(X, n_threads=n_threads)
    dtype = X.dtype
    p = X.cols()

    if constraints is not None and any(c is not None for c in constraints):
        # The constrained solves live in the panel engines.  A lazily standardized view of a dense / 2-bit design has them (the
        # sequential panel form on the base design's columns with the view's corrections), and so has a design kept sparse under
        # IRLS (the panel form over compressed columns): constrained fits run there, on the view.  What has no panel form -- a
        # Gaussian fit on a design kept sparse (its Gram is built once: the full-Gram engines), any fit with the view's panel form
        # switched off -- is materialised for such a fit (the reference composes any matrix with any constraint,
        # adelie/solver.py:257-313), with a warning because the copy costs n * p values of HBM; cold and warm-started fits
        # take the same route.
        is_view = isinstance(X, matrix._StdView)
        sparse_kept = matrix._is_kept_sparse(X) or (is_view and matrix._is_kept_sparse(getattr(X, "_base", None)))
        gaussian = getattr(glm, "name", "") == "gaussian" and not getattr(glm, "is_multi", False) and getattr(glm, "opt", True)
        panel_off = os.environ.get("ADELIE_HIP_STD_PANEL", "1") == "0" if is_view else False
        if sparse_kept and os.environ.get("ADELIE_HIP_SPARSE_PANEL", "1") == "0":
            panel_off = True
        if (sparse_kept and gaussian) or ((is_view or sparse_kept) and panel_off):
            warnings.warn(
                "adelie_amd: constraints on a lazily standardized / sparse-resident design run on its materialised dense copy "
                f"({X.rows()} x {X.cols()} values of device memory).", RuntimeWarning, stacklevel=2)
            X = X._materialize() if is_view else matrix._expanded(X)
    raw = None if (exit_cond is not None or _prepare_only) else _lasso_in_raw_coordinates(X, glm, constraints, groups, alpha, intercept, warm_start)
    if raw is not None:  # (an exit_cond callback reads the live state: it gets the view's own coordinates, i.e. the view's engines)
        # A lasso (alpha = 1, groups of one, intercept) on the standardized view (Z - 1 c') diag(s)^-1 of a resident design Z is the
        # lasso on Z itself with the penalty factors times |s|: beta~ = s beta, and the intercept absorbs the centres.  Every
        # coordinate update, the convergence measure A_jj d_j^2, the screening scores |g_j| / penalty_j and the KKT test are
        # the same numbers in both coordinate systems (eta is the same vector), so the solver runs its panel engines on the raw
        # columns -- the view itself only has the full-Gram engines (ROUNDS.md 9.9) -- and the state comes back in the
        # standardized coordinates.
        # An elastic net (0 < alpha < 1) is the same statement with the quadratic part's factors times s^2: the library takes
        # them as a separate vector (adelie_hip_grpnet_args::penalty_l2, ABI 8; the reference has one factor per group, hence
        # no public argument here).
        base, sc, ce = raw
        pen = np.ones(p, dtype=dtype) if penalty is None else np.asarray(penalty, dtype=dtype)
        state = grpnet(
            base, glm, groups=None, alpha=alpha, penalty=pen * np.abs(sc), offsets=offsets, lmda_path=lmda_path,
            irls_max_iters=irls_max_iters, irls_tol=irls_tol, max_iters=max_iters, tol=tol, adev_tol=adev_tol, ddev_tol=ddev_tol,
            newton_tol=newton_tol, newton_max_iters=newton_max_iters, n_threads=n_threads, early_exit=early_exit,
            intercept=True, screen_rule=screen_rule, min_ratio=min_ratio, lmda_path_size=lmda_path_size,
            max_screen_size=max_screen_size, max_active_size=max_active_size, pivot_subset_ratio=pivot_subset_ratio,
            pivot_subset_min=pivot_subset_min, pivot_slack_ratio=pivot_slack_ratio, check_state=check_state,
            progress_bar=progress_bar, _penalty_l2=None if alpha == 1 else (pen * sc * sc).astype(dtype))
        return _to_standardized_coordinates(state, X, sc, ce, pen, alpha)

    if isinstance(constraints, list):
        for c in constraints:  # cached dual state of a previous solve must not leak into this one (solver.py:638-642)
            if c is not None:
                c.clear()

    if offsets is None:
        offsets = np.zeros(glm.y.shape, dtype=dtype)
    else:
        if np.shape(offsets) != glm.y.shape:
            raise RuntimeError("offsets must be same shape as y if not None.")
        offsets = np.asarray(offsets, order="C", dtype=dtype)
    if lmda_path is not None:
        lmda_path = np.sort(np.asarray(lmda_path))[::-1].astype(dtype)  # decreasing, materialised (not a view)

    is_multi = bool(getattr(glm, "is_multi", False))
    closed_form = getattr(glm, "name", None) in ("gaussian", "multigaussian") and bool(getattr(glm, "opt", False))
    kind = ("multi" if is_multi else "") + ("gaussian" if closed_form else "glm")

    if is_multi:
        layout = _Layout.multi(groups, p, glm.y.shape[-1], intercept, penalty, dtype)
    else:
        layout = _Layout.single(groups, p, penalty, dtype)

    kwargs = dict(
        X=X, constraints=constraints, alpha=alpha, offsets=offsets, lmda_path=lmda_path,
        groups=layout.groups, group_sizes=layout.group_sizes, penalty=layout.penalty,
        max_iters=max_iters, tol=tol, adev_tol=adev_tol, ddev_tol=ddev_tol, newton_tol=newton_tol,
        newton_max_iters=newton_max_iters, n_threads=n_threads, early_exit=early_exit, intercept=intercept,
        screen_rule=screen_rule, min_ratio=min_ratio, lmda_path_size=lmda_path_size,
        max_screen_size=max_screen_size, max_active_size=max_active_size, pivot_subset_ratio=pivot_subset_ratio,
        pivot_subset_min=pivot_subset_min, pivot_slack_ratio=pivot_slack_ratio,
    )
    if closed_form:
        kwargs.update(y=glm.y, weights=glm.weights)
    else:
        kwargs.update(glm=glm, irls_max_iters=irls_max_iters, irls_tol=irls_tol)

    if warm_start is not None:
        kwargs.update(_resume(warm_start, kind))
    else:
        kwargs.update(_start_screen(layout, alpha, dtype))
        if kind == "gaussian":
            kwargs.update(_start_gaussian(X, glm, offsets, intercept, dtype))
        elif kind == "glm":
            kwargs.update(_start_glm(X, glm, offsets, dtype))
        elif kind == "multigaussian":
            kwargs.update(_start_multigaussian(X, glm, offsets, intercept, dtype))
        else:
            kwargs.update(_start_multiglm(X, glm, offsets, intercept, dtype))

    state = _STATE_OF[kind](**kwargs)
    if _penalty_l2 is not None:
        state._penalty_l2 = np.ascontiguousarray(_penalty_l2, dtype=dtype)
    if _lmda_aug is not None:  # (ratios, threshold): cv_grpnet's fold grid, joined to lmda_path inside the solve (ABI 10)
        state._lmda_aug = _lmda_aug
    if check_state:
        state.check(method="assert")
    if _prepare_only:  # (cv_grpnet: the folds' states are solved together by state.solve_many)
        return state
    return state.solve(progress_bar=progress_bar, exit_cond=exit_cond)


def gaussian_cov(
    A, v, *,
    constraints: list = None, groups: np.ndarray = None, alpha: float = 1, penalty: np.ndarray = None,
    lmda_path: np.ndarray = None, max_iters: int = int(1e5), tol: float = 1e-7, rdev_tol: float = 1e-3,
    newton_tol: float = 1e-12, newton_max_iters: int = 1000, n_threads: int = 1, early_exit: bool = True,
    screen_rule: str = "pivot", min_ratio: float = 1e-2, lmda_path_size: int = 100,
    max_screen_size: int = None, max_active_size: int = None,
    pivot_subset_ratio: float = 0.1, pivot_subset_min: int = 1, pivot_slack_ratio: float = 1.25,
    check_state: bool = False, progress_bar: bool = True, warm_start=None, exit_cond: Callable = None,
):
    """Gaussian group elastic net from summary statistics (covariance method) on an MI355X.

    Minimises ``1/2 beta' A beta - v' beta + lmda * sum_g penalty_g (alpha ||beta_g|| + (1 - alpha) / 2 ||beta_g||^2)``
    along a decreasing path of ``lmda`` (reference ``adelie.solver.gaussian_cov``, ``adelie/solver.py:39-351``; arguments,
    defaults and the returned state are the reference's).  ``A`` is a symmetric positive semi-definite ``(p, p)`` ndarray
    or an ``adelie_amd.matrix.dense(A, method="cov")`` handle (resident in HBM); ``v`` is ``(p,)``.  With ``A = X_c' W X_c``
    and ``v = X_c' W y_c`` this is the Gaussian ``grpnet`` problem (``devs`` then holds the unnormalised ``rsq``)."""
    if isinstance(A, np.ndarray):
        A = matrix.dense(A, method="cov", n_threads=n_threads)
    if not isinstance(A, (matrix.MatrixCovBase64, matrix.MatrixCovBase32)):
        raise ValueError("A must be an instance of MatrixCovBase32, MatrixCovBase64, or np.ndarray.")
    dtype = A.dtype
    p = A.cols()
    if isinstance(constraints, list):
        for c in constraints:
            if c is not None:
                c.clear()
    if lmda_path is not None:
        lmda_path = np.sort(np.asarray(lmda_path))[::-1].astype(dtype)
    layout = _Layout.single(groups, p, penalty, dtype)
    if warm_start is not None:
        start = {name: getattr(warm_start, name) for name in _SCREEN_FIELDS + ("rsq", "grad")}
    else:
        start = _start_screen(layout, alpha, dtype)
        # the gradient of beta = 0 is v (reference solver.py:279-290 subtracts A times the all-zero screen coefficients)
        start.update(rsq=0, grad=np.array(v, dtype=dtype, copy=True))
    state = _state.gaussian_cov(
        A=A, v=v, constraints=constraints, groups=layout.groups, group_sizes=layout.group_sizes, alpha=alpha,
        penalty=layout.penalty, lmda_path=lmda_path, max_iters=max_iters, tol=tol, rdev_tol=rdev_tol, newton_tol=newton_tol,
        newton_max_iters=newton_max_iters, n_threads=n_threads, early_exit=early_exit, screen_rule=screen_rule,
        min_ratio=min_ratio, lmda_path_size=lmda_path_size, max_screen_size=max_screen_size, max_active_size=max_active_size,
        pivot_subset_ratio=pivot_subset_ratio, pivot_subset_min=pivot_subset_min, pivot_slack_ratio=pivot_slack_ratio, **start)
    if check_state:
        state.check(method="assert")
    return state.solve(progress_bar=progress_bar, exit_cond=exit_cond)
