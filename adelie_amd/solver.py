"""``grpnet`` — mirrors ``adelie.solver.grpnet`` (reference ``adelie/solver.py:354-958``).

Same keyword arguments, same defaults, same preamble: the initial invariants are computed with two
``X.mul`` sweeps through the matrix plugin surface (reference ``solver.py:891-904``), the state object
is built by ``adelie_amd.state`` and ``state.solve()`` runs the whole lambda path on the MI355X.

``glm.multigaussian`` runs the same solver on the expanded design ``[1 (x) I_K, X (x) I_K]``
(reference ``solver.py:700-816``); the IRLS route for other multi-response families (multinomial) raises.
"""
from typing import Callable

import numpy as np

from . import matrix
from .state import gaussian_naive as state_gaussian_naive
from .state import glm_naive as state_glm_naive
from .state import multigaussian_naive as state_multigaussian_naive
from .state import multiglm_naive as state_multiglm_naive


def grpnet(
    X,
    glm,
    *,
    constraints: list = None,
    groups: np.ndarray = None,
    alpha: float = 1,
    penalty: np.ndarray = None,
    offsets: np.ndarray = None,
    lmda_path: np.ndarray = None,
    irls_max_iters: int = int(1e4),
    irls_tol: float = 1e-7,
    max_iters: int = int(1e5),
    tol: float = 1e-7,
    adev_tol: float = 0.9,
    ddev_tol: float = 0,
    newton_tol: float = 1e-12,
    newton_max_iters: int = 1000,
    n_threads: int = 1,
    early_exit: bool = True,
    intercept: bool = True,
    screen_rule: str = "pivot",
    min_ratio: float = 1e-2,
    lmda_path_size: int = 100,
    max_screen_size: int = None,
    max_active_size: int = None,
    pivot_subset_ratio: float = 0.1,
    pivot_subset_min: int = 1,
    pivot_slack_ratio: float = 1.25,
    check_state: bool = False,
    progress_bar: bool = False,
    warm_start=None,
    exit_cond: Callable = None,
):
    """Solves group elastic net via the naive method on an MI355X.

    Minimises ``l(eta) + lmda * sum_g penalty_g (alpha ||beta_g|| + (1-alpha)/2 ||beta_g||^2)`` with
    ``eta = X beta + beta0 + offsets`` along a decreasing path of ``lmda`` (see the reference docstring,
    ``adelie/solver.py:388-620``, for the meaning of every argument; they are identical).

    Returns
    -------
    state
        The solved state (``betas`` CSR ``(L, p)``, ``intercepts``, ``devs``, ``lmdas`` and the invariants).
    """
    X_raw = X
    if isinstance(X, np.ndarray):
        X = matrix.dense(X, method="naive", n_threads=n_threads)
    assert isinstance(X, (matrix.MatrixNaiveBase64, matrix.MatrixNaiveBase32))
    dtype = np.float64 if isinstance(X, matrix.MatrixNaiveBase64) else np.float32
    n, p = X.rows(), X.cols()

    is_multi = bool(getattr(glm, "is_multi", False))
    if is_multi and not ((glm.name == "multigaussian" and glm.opt) or glm.name == "multinomial"):
        raise NotImplementedError("adelie_amd.grpnet: of the multi-response GLMs, glm.multigaussian and glm.multinomial are on the device path.")
    if isinstance(constraints, list) and any(c is not None for c in constraints):
        raise NotImplementedError("adelie_amd.grpnet: constraints are outside the hot path (pass None).")

    if offsets is not None:
        offsets = np.asarray(offsets)
        if offsets.shape != glm.y.shape:
            raise RuntimeError("offsets must be same shape as y if not None.")
        offsets = np.asarray(offsets, order="C", dtype=dtype)
    else:
        offsets = np.zeros(glm.y.shape, dtype=dtype)

    if lmda_path is not None:
        lmda_path = np.array(np.flip(np.sort(lmda_path)), dtype=dtype)

    solver_args = {
        "X": X,
        "constraints": constraints,
        "alpha": alpha,
        "offsets": offsets,
        "lmda_path": lmda_path,
        "max_iters": max_iters,
        "tol": tol,
        "adev_tol": adev_tol,
        "ddev_tol": ddev_tol,
        "newton_tol": newton_tol,
        "newton_max_iters": newton_max_iters,
        "n_threads": n_threads,
        "early_exit": early_exit,
        "intercept": intercept,
        "screen_rule": screen_rule,
        "min_ratio": min_ratio,
        "lmda_path_size": lmda_path_size,
        "max_screen_size": max_screen_size,
        "max_active_size": max_active_size,
        "pivot_subset_ratio": pivot_subset_ratio,
        "pivot_subset_min": pivot_subset_min,
        "pivot_slack_ratio": pivot_slack_ratio,
    }

    is_gaussian_opt = (glm.name in ["gaussian", "multigaussian"]) and glm.opt  # solver.py:683-686
    if not is_gaussian_opt:
        solver_args["glm"] = glm
        solver_args["irls_max_iters"] = irls_max_iters
        solver_args["irls_tol"] = irls_tol
    else:
        solver_args["y"] = glm.y
        solver_args["weights"] = glm.weights

    if groups is None:
        groups = np.arange(p, dtype=int)
    groups = np.asarray(groups, dtype=int)

    if is_multi:
        return _grpnet_multi(X, glm, groups, penalty, offsets, intercept, alpha, warm_start, solver_args,
                                     check_state, progress_bar, exit_cond, n, p, dtype, n_threads)

    # single-response GLMs: solver.py:846-950
    group_sizes = np.concatenate([groups, [p]], dtype=int)
    group_sizes = group_sizes[1:] - group_sizes[:-1]
    G = len(groups)
    if penalty is None:
        penalty = np.sqrt(group_sizes).astype(dtype)
    penalty = np.asarray(penalty, dtype=dtype)

    if warm_start is None:
        lmda = np.inf
        lmda_max = None
        screen_set = np.arange(G)[(penalty <= 0) | (alpha <= 0)]
        screen_beta = np.zeros(np.sum(group_sizes[screen_set]), dtype=dtype)
        screen_is_active = np.ones(screen_set.shape[0], dtype=bool)
        active_set_size = screen_set.shape[0]
        active_set = np.empty(groups.shape[0], dtype=int)
        active_set[:active_set_size] = np.arange(active_set_size)
    else:
        lmda = warm_start.lmda
        lmda_max = warm_start.lmda_max
        screen_set = warm_start.screen_set
        screen_beta = warm_start.screen_beta
        screen_is_active = warm_start.screen_is_active
        active_set_size = warm_start.active_set_size
        active_set = warm_start.active_set

    solver_args["groups"] = groups
    solver_args["group_sizes"] = group_sizes
    solver_args["penalty"] = penalty
    solver_args["lmda"] = lmda
    solver_args["lmda_max"] = lmda_max
    solver_args["screen_set"] = screen_set
    solver_args["screen_beta"] = screen_beta
    solver_args["screen_is_active"] = screen_is_active
    solver_args["active_set_size"] = active_set_size
    solver_args["active_set"] = active_set

    if is_gaussian_opt:
        y = glm.y
        weights = glm.weights
        if warm_start is None:
            ones = np.ones(n, dtype=dtype)
            X_means = np.empty(p, dtype=dtype)
            X.mul(ones, weights, X_means)
            y_off = y - offsets
            y_mean = np.sum(y_off * weights)
            yc = y_off
            if intercept:
                yc = yc - y_mean
            y_var = np.sum(weights * yc ** 2)
            rsq = 0
            resid = yc
            resid_sum = np.sum(weights * resid)
            grad = np.empty(p, dtype=dtype)
            X.mul(resid, weights, grad)
        else:
            X_means = warm_start.X_means
            y_mean = warm_start.y_mean
            y_var = warm_start.y_var
            rsq = warm_start.rsq
            resid = warm_start.resid
            resid_sum = warm_start.resid_sum
            grad = warm_start.grad
        solver_args["X_means"] = X_means
        solver_args["y_mean"] = y_mean
        solver_args["y_var"] = y_var
        solver_args["rsq"] = rsq
        solver_args["resid"] = resid
        solver_args["resid_sum"] = resid_sum
        solver_args["grad"] = grad
        state = state_gaussian_naive(**solver_args)
    else:
        if warm_start is None:
            ones = np.ones(n, dtype=dtype)
            beta0 = 0
            eta = offsets
            resid = np.empty(n, dtype=dtype)
            glm.gradient(eta, resid)
            grad = np.empty(p, dtype=dtype)
            X.mul(resid, ones, grad)
            loss_null = None
            loss_full = glm.loss_full()
        else:
            beta0 = warm_start.beta0
            eta = warm_start.eta
            resid = warm_start.resid
            grad = warm_start.grad
            loss_null = warm_start.loss_null
            loss_full = warm_start.loss_full
        solver_args["beta0"] = beta0
        solver_args["grad"] = grad
        solver_args["eta"] = eta
        solver_args["resid"] = resid
        solver_args["loss_null"] = loss_null
        solver_args["loss_full"] = loss_full
        state = state_glm_naive(**solver_args)

    if check_state:
        state.check(method="assert")

    return state.solve(progress_bar=progress_bar, exit_cond=exit_cond)


def _grpnet_multi(X, glm, groups, penalty, offsets, intercept, alpha, warm_start, solver_args, check_state,
                          progress_bar, exit_cond, n, p, dtype, n_threads):
    """The multi-response branch of the reference's ``grpnet`` (``solver.py:700-844``): ``glm.multigaussian`` (Gaussian naive
    solver on the expanded design) and ``glm.multinomial`` (IRLS on the expanded design)."""
    K = glm.y.shape[-1]
    groups = groups * K  # flatten the grouping index across the classes
    if intercept:
        groups = np.concatenate([np.arange(K), K + groups], dtype=int)
    group_sizes = np.concatenate([groups, [(p + intercept) * K]], dtype=int)
    group_sizes = group_sizes[1:] - group_sizes[:-1]
    if penalty is None:
        penalty = np.sqrt(group_sizes).astype(dtype)
        if intercept:
            penalty[:K] = 0
    else:
        penalty = np.asarray(penalty, dtype=dtype)
        if intercept:
            penalty = np.concatenate([np.zeros(K), penalty], dtype=dtype)

    if warm_start is None:
        lmda = np.inf
        lmda_max = None
        screen_set = np.arange(groups.shape[0])[(penalty <= 0) | (alpha <= 0)]
        screen_beta = np.zeros(np.sum(group_sizes[screen_set]), dtype=dtype)
        screen_is_active = np.ones(screen_set.shape[0], dtype=bool)
        active_set_size = screen_set.shape[0]
        active_set = np.empty(groups.shape[0], dtype=int)
        active_set[:active_set_size] = np.arange(active_set_size)
    else:
        lmda = warm_start.lmda
        lmda_max = warm_start.lmda_max
        screen_set = warm_start.screen_set
        screen_beta = warm_start.screen_beta
        screen_is_active = warm_start.screen_is_active
        active_set_size = warm_start.active_set_size
        active_set = warm_start.active_set

    solver_args.update(groups=groups, group_sizes=group_sizes, penalty=penalty, lmda=lmda, lmda_max=lmda_max,
                       screen_set=screen_set, screen_beta=screen_beta, screen_is_active=screen_is_active,
                       active_set_size=active_set_size, active_set=active_set)

    if not (glm.name == "multigaussian" and glm.opt):  # IRLS route, solver.py:818-844
        if warm_start is None:
            eta = offsets
            resid = np.empty(eta.shape, dtype=dtype)
            glm.gradient(eta, resid)
            G = np.empty((p + (1 if intercept else 0), K), dtype=dtype)
            t = np.empty(p, dtype=dtype)
            ones = np.ones(n, dtype=dtype)
            for l in range(K):  # grad = X_aug^T resid, one sweep of the base design per class
                rl = np.ascontiguousarray(resid[:, l], dtype=dtype)
                X.mul(rl, ones, t)
                if intercept:
                    G[0, l] = np.sum(rl)
                    G[1:, l] = t
                else:
                    G[:, l] = t
            grad = G.ravel()
            resid = resid.ravel()
            loss_null = None
            loss_full = glm.loss_full()
            eta = eta.ravel()
        else:
            eta = warm_start.eta
            resid = warm_start.resid
            grad = warm_start.grad
            loss_null = warm_start.loss_null
            loss_full = warm_start.loss_full
        solver_args.update(grad=grad, eta=eta, resid=resid, loss_null=loss_null, loss_full=loss_full)
        state = state_multiglm_naive(**solver_args)
        return state.solve(progress_bar=progress_bar, exit_cond=exit_cond)

    y = glm.y
    weights = glm.weights
    weights_mscaled = weights / K
    if warm_start is None:
        ones = np.ones(n, dtype=dtype)
        X_means = np.empty(p, dtype=dtype)
        X.mul(ones, weights_mscaled, X_means)
        X_means = np.repeat(X_means, K)
        if intercept:
            X_means = np.concatenate([np.full(K, 1 / K), X_means], dtype=dtype)
        y_off = y - offsets
        y_var = np.sum(weights_mscaled[:, None] * y_off ** 2)
        # R^2 starts at (MSE of the intercept-only model) - y_var <= 0 and is brought to 0 by the fit of the unpenalised
        # intercept columns; normalising by the centred variance then makes it relative to the intercept model
        # (solver.py:760-772)
        if intercept:
            y_off_c = y_off - (y_off.T @ weights)[None]  # weights, not weights_mscaled (sic, solver.py:769)
            yc_var = np.sum(weights_mscaled[:, None] * y_off_c ** 2)
            rsq = yc_var - y_var
            y_var = yc_var
        else:
            rsq = 0
        resid = np.ascontiguousarray(y_off, dtype=dtype).ravel()
        resid_sum = np.sum(weights_mscaled[:, None] * y_off)
        # grad = X_aug^T (w' * resid): one sweep of the base design per response
        G = np.empty((p + (1 if intercept else 0), K), dtype=dtype)
        t = np.empty(p, dtype=dtype)
        for l in range(K):
            rl = np.ascontiguousarray(y_off[:, l], dtype=dtype)
            X.mul(rl, weights_mscaled, t)
            if intercept:
                G[0, l] = np.sum(rl * weights_mscaled)
                G[1:, l] = t
            else:
                G[:, l] = t
        grad = G.ravel()
    else:
        X_means = warm_start.X_means
        y_var = warm_start.y_var
        rsq = warm_start.rsq
        resid = warm_start.resid
        resid_sum = warm_start.resid_sum
        grad = warm_start.grad

    solver_args.update(X_means=X_means, y_var=y_var, rsq=rsq, resid=resid, resid_sum=resid_sum, grad=grad)
    state = state_multigaussian_naive(**solver_args)
    return state.solve(progress_bar=progress_bar, exit_cond=exit_cond)
