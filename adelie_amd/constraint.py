"""Per-group constraints — the host-side mirror of reference ``adelie/constraint.py`` (``box`` :18-135, ``lower`` :309-338,
``one_sided`` :341-480, ``upper`` :483-511) and of ``adelie_core/constraint/*``.

Two routes into the solver (``adelie_hip_grpnet_args::constraint_kind``):

* box / one-sided constraints on groups of ONE coefficient have closed forms (``constraint_box.ipp:51-96``,
  ``constraint_one_sided.ipp:12-49``): their bounds travel through the C ABI and the clipped coordinate update, the multipliers,
  ``abs_grad`` and the ``duals`` of the state are computed on the GPU;
* every other object — box / one-sided on groups of several coefficients (the reference's proximal-Newton dual solver,
  ``constraint/utils.hpp:24-243``, restated below in numpy), and any user subclass of :class:`ConstraintBase` — is visited on
  the host: the solver hands the group's quadratic model (``quad``, ``linear``, eigenbasis ``Q``) to ``solve`` between two
  panel steps and asks ``gradient`` / ``solve_zero`` / ``dual`` when it updates ``abs_grad`` and the duals, through the
  callbacks of ``adelie_hip_constraint_callbacks`` — the role the reference's ``PyConstraintBase`` trampoline plays.

After a solve every object holds the multipliers of the last fit (``dual`` / ``duals_nnz``), as the reference's objects do.
``linear`` (a general matrix ``A``) is such a host object too.
"""
from typing import Union

import numpy as np

MAX_SOLVER_VALUE = 1e100  # Configs.max_solver_value (configs.hpp:13)

KIND_BOX, KIND_ONE_SIDED, KIND_HOST = 1, 2, 3
NATIVE_BOX, NATIVE_ONE_SIDED, NATIVE_LINEAR = 4, 5, 6  # adelie_hip_grpnet_args::constraint_native (read by the CPU checker only)

_PROX_NEWTON_DEFAULTS = {"max_iters": 100, "tol": 1e-9, "pinball_max_iters": int(1e5), "pinball_tol": 1e-7, "slack": 1e-4}


def _group_prox(quad, v, l1, l2, tol=1e-12, max_iters=100000):
    """Minimiser of ``1/2 x' diag(quad) x - v' x + l1 |x|_2 + l2/2 |x|_2^2`` (the unconstrained group update,
    ``bcd/unconstrained/newton.hpp:35-142``): Newton on ``h = |x|`` from ``h = 0``.  Returns ``x`` and the two vectors the
    dual Hessian below is built from, ``quad + l2`` and ``1 / ((quad + l2) h + l1)``."""
    b1 = quad + l2
    if np.linalg.norm(v) <= l1:
        return np.zeros_like(v), b1, 1.0 / np.maximum(l1, np.finfo(float).tiny) * np.ones_like(v)
    if l1 <= 0:
        x = v / b1
        return x, b1, 1.0 / (b1 * np.linalg.norm(x))
    h = 0.0
    for _ in range(max_iters + 1):
        b2 = 1.0 / (b1 * h + l1)
        z = np.square(v * b2)
        t = z.sum()
        f = t - 1.0
        if abs(f) <= tol:
            break
        df = -(z * b1 * b2).sum() * (1 + np.sqrt(t)) / t
        h = max(h - f / df, 0.0)
    b2 = 1.0 / (b1 * h + l1)
    return h * v * b2, b1, b2


def _coerce(x, dtype):
    x = np.asarray(x)
    if dtype is None:
        if x.dtype not in (np.float32, np.float64):
            raise RuntimeError("dtype could not be inferred: pass float32 / float64 arrays or dtype=.")
        dtype = x.dtype.type
    return np.array(x, dtype=dtype, ndmin=1), dtype


class ConstraintBase:
    """Common surface of ``ConstraintBase{32,64}`` (``constraint_base.hpp:19-160``)."""

    kind = 0

    def __init__(self, d, dtype):
        self.dtype = dtype
        self.primal_size = int(d)
        self.dual_size = int(d)
        self._mu = np.zeros(d, dtype=dtype)

    def primals(self):
        return self.primal_size

    def duals(self):
        return self.dual_size

    def buffer_size(self):
        return 0

    def clear(self):
        self._mu[...] = 0

    def duals_nnz(self):
        return int(np.count_nonzero(self._mu))

    def dual(self, indices, values):
        nz = np.flatnonzero(self._mu)
        indices[: len(nz)] = nz
        values[: len(nz)] = self._mu[nz]

    # ABI descriptor (kind, a, b): the one-coefficient closed forms override this; everything else is a host object
    def _abi(self):
        return KIND_HOST, 0.0, 0.0

    # (native code, per-coefficient a, per-coefficient b, settings) for the CPU checker's own restatement, or None
    def _native(self):
        return None

    def solve(self, x, quad, linear, l1, l2, Q, buffer=None):
        raise NotImplementedError("a constraint class must provide solve(x, quad, linear, l1, l2, Q).")

    def gradient(self, x, *args):
        raise NotImplementedError("a constraint class must provide gradient(x, out).")

    def solve_zero(self, v, buffer=None):
        raise NotImplementedError("a constraint class must provide solve_zero(v).")


class _ProxNewton:
    """The dual proximal-Newton solver of the reference for linear inequality constraints ``A Q x <= b`` on a group of several
    coefficients (``constraint/utils.hpp:24-243``): ascent on the multipliers ``mu`` with the primal ``x*(mu)`` from the
    unconstrained group update, a quadratic model of the dual whose Hessian is ``|x| Q D Q' + l1 kappa |x| a a'`` and a
    coordinate-descent sub-solver that keeps ``mu`` feasible (pinball loss for a box, ``optimization/pinball_full.hpp:84-118``;
    sign-constrained QP for one-sided bounds, ``optimization/nnqp_full.hpp:150-178``); backtracking towards the ellipse
    ``|v - Q' A' mu| = l1`` when a step overshoots into the region where the primal is zero.  Subclasses give ``A`` (identity
    or ``diag(sgn)``), the bounds and the feasible set of ``mu``."""

    def _configure(self, configs):
        cfg = dict(_PROX_NEWTON_DEFAULTS)
        if configs:
            unknown = set(configs) - set(cfg)
            if unknown:
                raise TypeError(f"unknown configs: {sorted(unknown)}")
            cfg.update(configs)
        if cfg["tol"] < 0:
            raise RuntimeError("adelie_core: tol must be >= 0.")
        if cfg["pinball_tol"] < 0:
            raise RuntimeError("adelie_core: pinball_tol must be >= 0.")
        if not (0 < cfg["slack"] < 1):
            raise RuntimeError("adelie_core: slack must be in (0,1).")
        self._cfg = cfg

    # hooks ------------------------------------------------------------------------------------------------------------
    def _At(self, mu):            # A' mu, a vector of the group's size
        raise NotImplementedError

    def _nearest_at_zero(self, Qv, mu, l1=None):
        """Feasible multipliers (under complementary slackness at x = 0) whose ``A' mu`` is nearest to ``Qv``; ``l1``: any point
        with ``|Qv - A' mu| <= l1`` will do (the reference's iterative solver for ``linear`` stops there)."""
        raise NotImplementedError

    # bookkeeping hooks of classes that keep more than the dense vector of multipliers (``linear``: their insertion order)
    def _zero_fit_taken(self, taken):
        pass

    def _saved_prev(self):
        pass

    def _backtracked(self, mu, mu_prev):
        pass

    def _is_optimal(self, z, mu):  # z = Q x
        return False

    def _qp(self, hess, var, mu, z):
        """One proximal-Newton step: the minimiser over the feasible multipliers of the quadratic model with Hessian
        ``A hess A'`` around ``mu``, where ``z = Q x*(mu)``."""
        raise NotImplementedError

    # -----------------------------------------------------------------------------------------------------------------
    def _solve_multi(self, x, quad, linear, l1, l2, Q):
        cfg = self._cfg
        quad = np.asarray(quad, dtype=float)
        v = np.asarray(linear, dtype=float)
        Q = np.asarray(Q, dtype=float)
        mu = self._mu.astype(float)
        x0 = np.asarray(x, dtype=float).copy()

        def finish(xv, muv):
            x[...] = xv
            self._mu[...] = muv

        if np.linalg.norm(v) <= l1:
            return finish(0.0, 0.0)
        Qv = Q @ v

        def nearest_at_zero(mu_now, may_restore):
            """Multipliers that best explain v while the primal stays 0; keeps the old ones when even those do not."""
            cand = self._nearest_at_zero(Qv, mu_now, l1)
            gap = float(np.sum(np.square(Qv - self._At(cand))))
            if may_restore and gap > l1 * l1:
                self._zero_fit_taken(False)
                return mu_now, gap
            self._zero_fit_taken(True)
            return cand, gap

        x_zero_start = not np.any(x0)
        have_prev = False
        zero_checked = False
        Atmu_prev = z_prev = mu_prev = None
        rn_prev = -1.0
        if x_zero_start:
            zero_checked = True
            mu, gap = nearest_at_zero(mu, True)
            if gap <= l1 * l1:
                return finish(0.0, mu)
        xv = x0
        for it in range(1, int(cfg["max_iters"]) + 1):
            Atmu = self._At(mu)
            resid = v - Atmu @ Q
            rn = float(np.linalg.norm(resid))
            inside = rn <= l1
            xn = -1.0
            if not inside:
                xv, b1, b2 = _group_prox(quad, resid, l1, l2)
                xn = float(np.linalg.norm(xv))
                inside = xn <= 0
            if inside:
                if it == 1 and x_zero_start:
                    return finish(0.0, mu)
                if have_prev and abs(np.mean((Atmu - Atmu_prev) * z_prev)) <= cfg["tol"]:
                    return finish(0.0, mu)
                if not zero_checked:
                    zero_checked = True
                    had_prev = have_prev
                    if not had_prev:
                        rn_prev, have_prev = rn, True
                        mu_prev, Atmu_prev, z_prev = mu.copy(), Atmu.copy(), np.zeros(len(v))
                        self._saved_prev()
                    mu, gap = nearest_at_zero(mu, had_prev)
                    if gap <= l1 * l1:
                        return finish(0.0, mu)
                    if not had_prev:
                        continue
                    Atmu = self._At(mu)
                if (not have_prev) or rn_prev <= l1 * 0.9999 or rn > l1 * 1.0001:
                    raise RuntimeError("adelie_core: Possibly an unexpected error! Previous iterate should have been properly "
                                       "initialized. ")
                target = (1 - cfg["slack"]) * l1 + cfg["slack"] * rn_prev
                dAt = Atmu - Atmu_prev
                a = float(dAt @ dAt)
                b = float((Qv - Atmu) @ dAt)
                c = rn * rn - target * target
                t_star = (-b + np.sqrt(max(b * b - a * c, 0.0))) / a
                mu = mu_prev + min(max(1 - t_star, 0.0), 1.0) * (mu - mu_prev)
                self._backtracked(mu, mu_prev)
                continue
            z = Q @ xv
            if self._is_optimal(z, mu):
                return finish(xv, mu)
            if have_prev and abs(np.mean((Atmu - Atmu_prev) * (z_prev - z))) <= cfg["tol"]:
                return finish(xv, mu)
            rn_prev, have_prev = rn, True
            mu_prev, Atmu_prev, z_prev = mu.copy(), Atmu.copy(), z.copy()
            self._saved_prev()
            # dual Hessian and the variance scale of the sub-solver's stopping rule (Woodbury)
            a_t = xv * b2 / xn
            kappa = 1.0 / float(np.sum(xv * b1 * a_t))
            alpha = Q @ a_t
            hess = xn * (Q * b2) @ Q.T + (l1 * kappa * xn) * np.outer(alpha, alpha)
            xq = xv @ Q
            xy = float(xv @ xq)
            var = (float(np.sum(np.square(xq) / b2)) - xy * xy / (xn * xn / (l1 * kappa) + float(np.sum(np.square(xv) * b2)))) / xn
            mu = self._qp(hess, max(var, 0.0), mu, z)
        raise RuntimeError("adelie_core solver: ConstraintBase: proximal newton max iterations reached!")

    def _pinball(self, H, var, mu, g, lo_pen, up_pen):
        """Coordinate descent from ``mu`` on ``1/2 m'Hm - (g + H mu)'m + up_pen'(m)_+ + lo_pen'(m)_-`` (pinball loss,
        ``optimization/pinball_full.hpp:84-118``); ``g`` is the current gradient and is kept current."""
        mu, g = mu.copy(), g.copy()
        for _ in range(int(self._cfg["pinball_max_iters"])):
            worst = 0.0
            for i in range(len(mu)):
                h = H[i, i]
                if h <= 0:
                    continue
                g0 = g[i] + h * mu[i]
                new = np.copysign(max(-lo_pen[i] - g0, g0 - up_pen[i], 0.0), g0 + lo_pen[i]) / h
                dl = new - mu[i]
                if dl == 0:
                    continue
                mu[i] = new
                worst = max(worst, h * dl * dl)
                g -= dl * H[:, i]
            if worst < var * self._cfg["pinball_tol"]:
                return mu
        raise RuntimeError("adelie_core solver: StatePinballFull: max iterations reached!")


class _Box(_ProxNewton, ConstraintBase):
    """``lower <= x <= upper`` with ``lower <= 0 <= upper`` (``ConstraintBox``; the class stores ``l = -lower``)."""

    kind = KIND_BOX

    def __init__(self, lower, upper, dtype, configs=None):
        self._configure(configs)
        if lower.shape != upper.shape:
            raise RuntimeError("adelie_core: lower must be (d,) where upper is (d,).")
        if np.any(upper < 0):
            raise RuntimeError("adelie_core: upper must be >= 0.")
        if np.any(lower > 0):
            raise RuntimeError("adelie_core: lower must be <= 0.")
        super().__init__(len(upper), dtype)
        with np.errstate(over="ignore"):  # 1e100 is +inf in single precision (as in the reference's float32 classes)
            self._lower = np.maximum(lower, -MAX_SOLVER_VALUE).astype(dtype)
            self._upper = np.minimum(upper, MAX_SOLVER_VALUE).astype(dtype)

    def _abi(self):
        if self.primal_size != 1:
            return KIND_HOST, 0.0, 0.0
        return KIND_BOX, float(self._lower[0]), float(self._upper[0])

    def _native(self):
        c = self._cfg
        return NATIVE_BOX, self._lower, self._upper, (c["max_iters"], c["tol"], c["pinball_max_iters"], c["pinball_tol"], c["slack"])

    # hooks of _ProxNewton: A = I, multipliers mu = mu_+ - mu_-, free in sign only where the matching bound is 0
    def _At(self, mu):
        return mu

    def _nearest_at_zero(self, Qv, mu, l1=None):
        lo = np.where(self._lower >= 0, -MAX_SOLVER_VALUE, 0.0)
        hi = np.where(self._upper <= 0, MAX_SOLVER_VALUE, 0.0)
        return np.minimum(np.maximum(Qv, lo), hi)

    def _is_optimal(self, z, mu):
        u, l = self._upper.astype(float), self._lower.astype(float)
        return bool(np.all((z <= u) & (z >= l)) and np.all(np.maximum(mu, 0) * (z - u) == 0)
                    and np.all(np.minimum(mu, 0) * (z - l) == 0))

    def _qp(self, hess, var, mu, z):
        return self._pinball(hess, var, mu, z, -self._lower.astype(float), self._upper.astype(float))

    def evaluate(self, x):
        return np.concatenate([x - self._upper, self._lower - x])

    def project(self, x):
        x[...] = np.maximum(np.minimum(x, self._upper), self._lower)

    def gradient(self, x, *args):
        mu, out = (self._mu, args[0]) if len(args) == 1 else args
        out[...] = mu

    def solve_zero(self, v, buffer=None):
        lo = np.where(self._lower >= 0, -MAX_SOLVER_VALUE, 0.0)
        hi = np.where(self._upper <= 0, MAX_SOLVER_VALUE, 0.0)
        self._mu[...] = np.minimum(np.maximum(v, lo), hi)
        return float(np.linalg.norm(v - self._mu))

    def solve(self, x, quad, linear, l1, l2, Q, buffer=None):
        if self.primal_size != 1:
            return self._solve_multi(x, quad, linear, l1, l2, Q)
        A, q, v = float(np.asarray(Q).reshape(-1)[0]), float(quad[0]), float(linear[0])
        u, l = float(self._upper[0]), -float(self._lower[0])
        mp = 0.0 if u > 0 else max(A * v, 0.0)
        mn = 0.0 if l > 0 else max(-A * v, 0.0)
        if abs(v - A * (mp - mn)) <= l1:
            x[0] = 0
            self._mu[0] = mp - mn
            return
        x0 = A * max(min(A * np.copysign(abs(v) - l1, v) / (q + l2), u), -l)
        full = A * (v - ((q + l2) * x0 + np.copysign(l1, x0)))
        mp = 0.0 if A * x0 < u else max(full, 0.0)
        mn = 0.0 if A * x0 > -l else max(-full, 0.0)
        x[0] = x0
        self._mu[0] = mp - mn


class _OneSided(_ProxNewton, ConstraintBase):
    """``D x <= b`` with ``D = diag(+-1)``, ``b >= 0`` (``ConstraintOneSided``)."""

    kind = KIND_ONE_SIDED

    def __init__(self, D, b, dtype, configs=None):
        self._configure(configs)
        if D.shape != b.shape:
            raise RuntimeError("adelie_core: sgn be (d,) where b is (d,).")
        if np.any(np.abs(D) != 1):
            raise RuntimeError("adelie_core: sgn must be a vector of +/-1.")
        if np.any(b < 0):
            raise RuntimeError("adelie_core: b must be >= 0.")
        super().__init__(len(b), dtype)
        self._D = D.astype(dtype)
        with np.errstate(over="ignore"):
            self._b = np.minimum(b, MAX_SOLVER_VALUE).astype(dtype)

    def _abi(self):
        if self.primal_size != 1:
            return KIND_HOST, 0.0, 0.0
        return KIND_ONE_SIDED, float(self._D[0]), float(self._b[0])

    def _native(self):
        c = self._cfg
        return NATIVE_ONE_SIDED, self._D, self._b, (c["max_iters"], c["tol"], c["pinball_max_iters"], c["pinball_tol"], c["slack"])

    # hooks of _ProxNewton: A = diag(sgn), mu >= 0, positive only where the bound is 0 (at x = 0) or active
    def _At(self, mu):
        return self._D * mu

    def _nearest_at_zero(self, Qv, mu, l1=None):
        hi = np.where(self._b <= 0, MAX_SOLVER_VALUE, 0.0)
        return np.minimum(np.maximum(self._D * Qv, 0.0), hi)

    def _is_optimal(self, z, mu):
        g = self._D * z - self._b
        return bool(np.all(g <= 0) and np.all(mu * g == 0))

    def _qp(self, hess, var, mu, z):
        """Coordinate ascent on the sign-constrained quadratic model, in the coordinates ``sgn * mu``
        (``optimization/nnqp_full.hpp:150-178``)."""
        sgn = self._D.astype(float)
        m, g = mu * sgn, (sgn * z - self._b) * sgn
        d = len(m)
        for _ in range(int(self._cfg["pinball_max_iters"])):
            worst = 0.0
            for i in range(d):
                h = hess[i, i]
                step = 0.0 if h <= 0 else g[i] / h
                new = max(m[i] + step, 0.0) if sgn[i] > 0 else min(m[i] + step, 0.0)
                dl = new - m[i]
                if dl == 0:
                    continue
                m[i] = new
                worst = max(worst, h * dl * dl)
                g -= dl * hess[:, i]
            if worst < var * self._cfg["pinball_tol"]:
                return m * sgn
        raise RuntimeError("adelie_core solver: StateNNQPFull: max iterations reached!")

    def evaluate(self, x):
        return self._D * x - self._b

    def project(self, x):
        x[...] = self._D * np.minimum(self._D * x, self._b)

    def gradient(self, x, *args):
        mu, out = (self._mu, args[0]) if len(args) == 1 else args
        out[...] = self._D * mu

    def solve_zero(self, v, buffer=None):
        hi = np.where(self._b <= 0, MAX_SOLVER_VALUE, 0.0)
        self._mu[...] = np.minimum(np.maximum(self._D * v, 0.0), hi)
        return float(np.linalg.norm(v - self._D * self._mu))

    def solve(self, x, quad, linear, l1, l2, Q, buffer=None):
        if self.primal_size != 1:
            return self._solve_multi(x, quad, linear, l1, l2, Q)
        A = float(self._D[0]) * float(np.asarray(Q).reshape(-1)[0])
        q, v, b = float(quad[0]), float(linear[0]), float(self._b[0])
        mu0 = 0.0 if b > 0 else max(A * v, 0.0)
        if abs(v - A * mu0) <= l1:
            x[0] = 0
            self._mu[0] = mu0
            return
        x0 = A * min(A * np.copysign(abs(v) - l1, v) / (q + l2), b)
        x[0] = x0
        self._mu[0] = 0.0 if A * x0 < b else A * (v - ((q + l2) * x0 + np.copysign(l1, x0)))


def box(lower: np.ndarray, upper: np.ndarray, *, method: str = "proximal_newton", configs: dict = None,
        dtype: Union[np.float32, np.float64] = None):
    """Box constraint ``lower <= x <= upper`` (``lower <= 0 <= upper``); reference ``constraint.py:18-135``.  ``method`` /
    ``configs`` tune the reference's iterative solver for groups of several coefficients and have no effect on the
    one-coefficient closed form."""
    if method != "proximal_newton":
        raise KeyError(method)
    lower, ld = _coerce(lower, dtype)
    upper, ud = _coerce(upper, dtype)
    assert ld == ud
    return _Box(lower, upper, ld, configs)


def one_sided(D: np.ndarray, b: np.ndarray, *, method: str = "proximal_newton", configs: dict = None,
              dtype: Union[np.float32, np.float64] = None):
    """One-sided bound ``D x <= b`` (``D = diag(+-1)``, ``b >= 0``); reference ``constraint.py:341-480``."""
    if method not in ("proximal_newton", "admm"):
        raise KeyError(method)
    if method == "admm" and np.size(b) > 1:
        raise NotImplementedError("adelie_amd: one_sided(method='admm') is not implemented for several coefficients; "
                                  "use the default 'proximal_newton'.")
    b, dtype = _coerce(b, dtype)
    return _OneSided(np.array(D, dtype=dtype, ndmin=1), b, dtype, configs if method == "proximal_newton" else None)


def lower(b: np.ndarray, **kwargs):
    """``x >= b`` with ``b <= 0`` (reference ``constraint.py:309-338``: ``one_sided(D=-1, b=-b)``)."""
    b = np.asarray(b)
    return one_sided(D=np.full(np.atleast_1d(b).shape[0], -1.0), b=-np.atleast_1d(b), **kwargs)


def upper(b: np.ndarray, **kwargs):
    """``x <= b`` with ``b >= 0`` (reference ``constraint.py:483-511``)."""
    b = np.asarray(b)
    return one_sided(D=np.full(np.atleast_1d(b).shape[0], 1.0), b=np.atleast_1d(b), **kwargs)


class _Linear(_ProxNewton, ConstraintBase):
    """``lower <= A x <= upper`` with ``lower <= 0 <= upper`` and a general ``(m, d)`` matrix ``A`` (``ConstraintLinear``,
    ``constraint_linear.ipp``): ``m`` multipliers ``mu = mu_+ - mu_-``, the same dual proximal-Newton iteration with
    ``A' mu`` in place of ``mu`` (``:275-470``).  The Newton step minimises the pinball-penalised quadratic model with Hessian
    ``A hess A'`` (the reference's ``m < d`` branch, ``:408-418``; for ``m >= d`` it solves the same sub-problem through a
    low-rank active-set variant), the multipliers at ``x = 0`` come from a sign-bounded least-squares fit of ``A' mu`` to
    ``Q v`` (``:283-349``) by the reference's warm-started coordinate descent (``_bvls``).  Always a host object."""

    kind = KIND_HOST

    def __init__(self, A, lower, upper, dtype, configs=None, vars=None):
        cfg = dict(configs or {})
        # the bounded least squares here is scipy's; the reference's own settings for it are kept for the description of the
        # object that the CPU checker's restatement of the reference solver reads (_linear_descriptor)
        self._nnls_cfg = (int(cfg.pop("nnls_max_iters", int(1e5))), float(cfg.pop("nnls_tol", 1e-7)))
        cfg.pop("n_threads", None)
        if self._nnls_cfg[1] < 0:
            raise RuntimeError("adelie_core: nnls_tol must be >= 0.")
        self._configure(cfg)
        A = np.asarray(A.toarray() if hasattr(A, "toarray") else A, dtype=float)
        if A.ndim != 2:
            raise RuntimeError("adelie_core: A must be (m, d).")
        m, d = A.shape
        if lower.shape != (m,) or upper.shape != (m,):
            raise RuntimeError("adelie_core: lower and upper must be (m,) where A is (m, d).")
        if np.any(upper < 0):
            raise RuntimeError("adelie_core: upper must be >= 0.")
        if np.any(lower > 0):
            raise RuntimeError("adelie_core: lower must be <= 0.")
        ConstraintBase.__init__(self, d, dtype)
        self.dual_size = m
        self._mu = np.zeros(m, dtype=dtype)
        self._A = A
        self._lower = np.maximum(lower.astype(float), -MAX_SOLVER_VALUE)
        self._upper = np.minimum(upper.astype(float), MAX_SOLVER_VALUE)
        self._vars = np.sum(A ** 2, axis=1) if vars is None else np.asarray(vars, dtype=float).reshape(m)
        self._order = []   # the non-zero multipliers in the order they became non-zero (the reference keeps them sparse)

    def duals(self):
        return self.dual_size

    def _abi(self):
        return KIND_HOST, 0.0, 0.0

    def _native(self):
        c = self._cfg
        return NATIVE_LINEAR, None, None, (c["max_iters"], c["tol"], c["pinball_max_iters"], c["pinball_tol"], c["slack"])

    def _linear_descriptor(self):
        """``adelie_hip_linear_constraint`` of this object (ABI 5) and the arrays it points into."""
        from . import _abi

        c = self._cfg
        A = np.ascontiguousarray(self._A, dtype=np.float64)
        lo = np.ascontiguousarray(self._lower, dtype=np.float64)
        up = np.ascontiguousarray(self._upper, dtype=np.float64)
        va = np.ascontiguousarray(self._vars, dtype=np.float64)
        d = _abi.LinearConstraint()
        d.m, d.d = A.shape
        d.A, d.lower, d.upper, d.vars = A.ctypes.data, lo.ctypes.data, up.ctypes.data, va.ctypes.data
        for i, v in enumerate((c["max_iters"], c["tol"], self._nnls_cfg[0], self._nnls_cfg[1], c["pinball_max_iters"],
                               c["pinball_tol"], c["slack"])):
            d.cfg[i] = float(v)
        return d, (A, lo, up, va)

    def _At(self, mu):
        return self._A.T @ mu

    def _sign_bounds(self):
        lo = np.where(self._lower >= 0, -MAX_SOLVER_VALUE, 0.0)   # a negative multiplier needs an active lower bound at 0
        hi = np.where(self._upper <= 0, MAX_SOLVER_VALUE, 0.0)
        return lo, hi

    # ---- bounded least squares  min |target - A' mu|^2, lo <= mu <= hi, by coordinate descent with a screen / active set ----
    # The reference's solver for this sub-problem (solver_bvls.hpp through optimization/nnls.hpp) is iterative, warm-started
    # from the multipliers the object holds IN THE ORDER THEY BECAME NON-ZERO, and inside solve() it stops at the first
    # iterate whose residual is within l1 — so WHICH multipliers a group with x = 0 ends up with (state.duals) depends on
    # exactly this iteration; a direct least-squares solve returns a different, equally valid set.  Same loops here.
    def _bvls(self, target, mu, order, y_var, good_enough):
        A, va = self._A, self._vars
        m, d = A.shape
        lo, hi = self._sign_bounds()
        tol, max_iters = self._nnls_cfg[1], self._nnls_cfg[0]
        beta = np.array(mu, dtype=float)
        resid = target - A.T @ beta
        loss = 0.5 * float(resid @ resid)
        screen, active = list(order), list(order)
        is_screen, is_active = np.zeros(m, bool), np.zeros(m, bool)
        is_screen[screen] = True
        is_active[active] = True
        kappa = min(m, d)
        state = {"loss": loss, "iters": 0}

        def sweep(idx, add_active):
            worst = 0.0
            for k in idx:
                if good_enough(state["loss"]):
                    return worst
                vk = va[k]
                gk = float(A[k] @ resid)
                old = beta[k]
                new = min(max(old + (0.0 if vk <= 0 else gk / vk), lo[k]), hi[k])
                if new == old:
                    continue
                beta[k] = new
                dl = new - old
                sds = vk * dl * dl
                worst = max(worst, sds)
                state["loss"] -= dl * gk - 0.5 * sds
                resid[...] -= dl * A[k]
                if add_active and not is_active[k]:
                    active.append(k)
                    is_active[k] = True
            return worst

        def prune():
            keep = [k for k in active if lo[k] < beta[k] < hi[k]]
            for k in active:
                is_active[k] = False
            for k in keep:
                is_active[k] = True
            active[:] = keep

        def fit():
            while True:
                state["iters"] += 1
                worst = sweep(list(screen), True)
                if state["iters"] >= max_iters:
                    raise RuntimeError("adelie_core solver: bvls: max iterations reached!")
                if worst <= tol * y_var or good_enough(state["loss"]):
                    prune()
                    return
                while True:
                    state["iters"] += 1
                    worst = sweep(list(active), False)
                    if state["iters"] >= max_iters:
                        raise RuntimeError("adelie_core solver: bvls: max iterations reached!")
                    if worst <= tol * y_var:
                        break
                prune()

        n_kkt = 0
        by_violation = np.arange(m)
        while True:
            before = state["loss"]
            fit()
            if good_enough(state["loss"]):
                break
            if n_kkt > 0 and abs(state["loss"] - before) < 1e-6 * abs(y_var):
                break
            n_kkt += 1
            g = A @ resid
            viol = np.maximum(g, 0) * (beta < hi) - np.minimum(g, 0) * (beta > lo)
            by_violation = np.array(sorted(by_violation, key=lambda i: -viol[i]), dtype=int)
            n_old, passed = len(screen), True
            for k in by_violation:
                if is_screen[k] or viol[k] <= 0:
                    continue
                passed = False
                if len(screen) >= n_old + kappa:
                    break
                screen.append(int(k))
                is_screen[k] = True
            if passed:
                break
        kept = [k for k in active if abs(beta[k]) > 1e-16]
        out = np.zeros(m)
        out[kept] = beta[kept]
        return out, kept, state["loss"]

    def _active_order(self, mu):
        """The multipliers' insertion order, brought up to date with ``mu`` (``mu_to_sparse``, constraint_linear.ipp:94-108)."""
        order = [k for k in self._order if abs(mu[k]) > 1e-16]
        seen = set(self._order)
        order += [int(k) for k in np.flatnonzero(mu) if k not in seen and abs(mu[k]) > 1e-16]
        return order

    def _nearest_at_zero(self, Qv, mu, l1=None):
        self._order = self._active_order(mu)
        if l1 is not None and float(np.sum(np.square(Qv - self._A.T @ mu))) <= l1 * l1:   # constraint_linear.ipp:281-283
            self._order_cand = list(self._order)
            return np.array(mu, dtype=float)
        stop = (lambda loss: False) if l1 is None else (lambda loss: 2 * loss <= l1 * l1)
        out, self._order_cand, _ = self._bvls(Qv, mu, self._order, float(Qv @ Qv), stop)
        return out

    def _zero_fit_taken(self, taken):
        if taken:
            self._order = list(self._order_cand)

    def _saved_prev(self):
        self._order_prev = list(self._order)

    def _backtracked(self, mu, mu_prev):   # constraint_linear.ipp:364-382: the current multipliers first, then the previous ones
        order = list(self._order)
        order += [k for k in getattr(self, "_order_prev", []) if k not in set(order)]
        self._order = order

    def _qp(self, hess, var, mu, z):
        A = self._A
        self._order = self._active_order(mu)
        out = self._pinball(A @ hess @ A.T, var, mu, A @ z, -self._lower, self._upper)
        self._order = self._active_order(out)
        return out

    def evaluate(self, x):
        Ax = self._A @ x
        return np.concatenate([Ax - self._upper, self._lower - Ax])

    def gradient(self, x, *args):
        mu, out = (self._mu, args[0]) if len(args) == 1 else args
        out[...] = self._A.T @ mu

    def solve_zero(self, v, buffer=None):   # constraint_linear.ipp:520-603: warm-started from the multipliers held, run to the end
        v = np.asarray(v, dtype=float)
        mu = self._mu.astype(float)
        out, self._order, loss = self._bvls(v, mu, self._active_order(mu), float(v @ v), lambda loss: False)
        self._mu[...] = out
        return float(np.sqrt(max(2 * loss, 0.0)))

    def solve(self, x, quad, linear, l1, l2, Q, buffer=None):
        if np.linalg.norm(linear) <= l1:
            self._order = []
        out = self._solve_multi(x, quad, linear, l1, l2, Q)
        self._order = self._active_order(self._mu.astype(float))
        return out

    def clear(self):
        super().clear()
        self._order = []


def linear(A, lower: np.ndarray, upper: np.ndarray, *, vars: np.ndarray = None, copy: bool = False,
           method: str = "proximal_newton", configs: dict = None, dtype: Union[np.float32, np.float64] = None):
    """Linear constraint ``lower <= A x <= upper`` (``lower <= 0 <= upper``) for a dense / scipy-sparse ``(m, d)`` matrix ``A``;
    reference ``constraint.py:137-306``.  ``vars`` (``diag(A A')``, computed when absent) scales the coordinate steps of the
    reference's bounded least squares; ``copy`` is accepted and not needed here."""
    if method != "proximal_newton":
        raise KeyError(method)
    lower, ld = _coerce(lower, dtype)
    upper, _ = _coerce(upper, ld)
    return _Linear(A, lower, upper, ld, configs, vars)


def render_dual_groups(constraints):
    """Offsets of every group's multipliers in the rows of ``state.duals`` (reference ``state.py:48-55``)."""
    return np.cumsum(np.concatenate([[0], [0 if c is None else c.dual_size for c in constraints]]), dtype=int)[:-1]
