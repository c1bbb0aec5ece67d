"""Per-group constraints — the host-side mirror of reference ``adelie/constraint.py`` (``box`` :18-135, ``lower`` :309-338,
``one_sided`` :341-480, ``upper`` :483-511) for the part of ``adelie_core/constraint/*`` that ``grpnet`` runs on the device:
box and one-sided constraints on groups of ONE coefficient, where both classes have closed forms
(``constraint_box.ipp:51-96``, ``constraint_one_sided.ipp:12-49``).

The objects are descriptors: ``grpnet`` hands their bounds to the solver through the C ABI (``constraint_kind / _a / _b`` of
``adelie_hip_grpnet_args``) and the coordinate updates, the multipliers, ``abs_grad`` and the ``duals`` of the state are
computed on the GPU.  After a solve every object holds the multiplier of the last fit (``dual`` / ``duals_nnz``), as the
reference's objects do.  The elementwise members (``gradient``, ``project``, ``evaluate``, ``solve_zero``) work for any size;
``solve`` is the one-coefficient closed form.  Constraints over groups of several coefficients (the proximal-Newton / ADMM
solvers of the reference) and ``linear`` constraints are not implemented: ``grpnet`` raises for them.
"""
from typing import Union

import numpy as np

MAX_SOLVER_VALUE = 1e100  # Configs.max_solver_value (configs.hpp:13)

KIND_BOX, KIND_ONE_SIDED = 1, 2


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

    # ABI descriptor of a one-coefficient constraint: (kind, a, b)
    def _abi(self):
        raise NotImplementedError

    def _check_1d(self):
        if self.primal_size != 1:
            raise NotImplementedError(
                "adelie_amd: constraints are implemented for groups of one coefficient (box / lower / upper / one_sided); "
                "the proximal-Newton solvers for larger groups are not."
            )


class _Box(ConstraintBase):
    """``lower <= x <= upper`` with ``lower <= 0 <= upper`` (``ConstraintBox``; the class stores ``l = -lower``)."""

    kind = KIND_BOX

    def __init__(self, lower, upper, dtype):
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
        self._check_1d()
        return KIND_BOX, float(self._lower[0]), float(self._upper[0])

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
        self._check_1d()
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


class _OneSided(ConstraintBase):
    """``D x <= b`` with ``D = diag(+-1)``, ``b >= 0`` (``ConstraintOneSided``)."""

    kind = KIND_ONE_SIDED

    def __init__(self, D, b, dtype):
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
        self._check_1d()
        return KIND_ONE_SIDED, float(self._D[0]), float(self._b[0])

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
        self._check_1d()
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
    return _Box(lower, upper, ld)


def one_sided(D: np.ndarray, b: np.ndarray, *, method: str = "proximal_newton", configs: dict = None,
              dtype: Union[np.float32, np.float64] = None):
    """One-sided bound ``D x <= b`` (``D = diag(+-1)``, ``b >= 0``); reference ``constraint.py:341-480``."""
    if method not in ("proximal_newton", "admm"):
        raise KeyError(method)
    b, dtype = _coerce(b, dtype)
    return _OneSided(np.array(D, dtype=dtype, ndmin=1), b, dtype)


def lower(b: np.ndarray, **kwargs):
    """``x >= b`` with ``b <= 0`` (reference ``constraint.py:309-338``: ``one_sided(D=-1, b=-b)``)."""
    b = np.asarray(b)
    return one_sided(D=np.full(np.atleast_1d(b).shape[0], -1.0), b=-np.atleast_1d(b), **kwargs)


def upper(b: np.ndarray, **kwargs):
    """``x <= b`` with ``b >= 0`` (reference ``constraint.py:483-511``)."""
    b = np.asarray(b)
    return one_sided(D=np.full(np.atleast_1d(b).shape[0], 1.0), b=np.atleast_1d(b), **kwargs)


def linear(*args, **kwargs):
    raise NotImplementedError("adelie_amd: linear constraints (constraint_linear.ipp) are not implemented.")


def render_dual_groups(constraints):
    """Offsets of every group's multipliers in the rows of ``state.duals`` (reference ``state.py:48-55``)."""
    return np.cumsum(np.concatenate([[0], [0 if c is None else c.dual_size for c in constraints]]), dtype=int)[:-1]
