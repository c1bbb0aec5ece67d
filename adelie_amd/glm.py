"""GLM plugin surface — mirrors ``adelie.glm`` for the families on the grpnet hot path.

Reference: ``adelie/glm.py:40-55`` (weight normalisation), ``:83-196`` (``binomial``), ``:374-453``
(``gaussian`` with its ``opt`` flag), and the C++ classes ``glm/glm_base.ipp:23-37``,
``glm/glm_gaussian.ipp:15-63``, ``glm/glm_binomial.ipp:14-99``.

The objects below are host-side descriptors: they carry ``y``/``weights`` and the closed-form
member functions (written in numpy, used by the Python preamble of ``grpnet`` and by ``cv_grpnet``).
Inside the path solver the same functions run as fused HIP elementwise kernels, selected by
``core_kind`` (``adelie_hip_glm_kind`` in ``include/adelie_hip.h``).
"""
import numpy as np

from . import _abi
from . import configs as _configs


def _coerce_dtype(y, dtype):
    """Reference ``glm.py:12-33`` (with ``np.asarray`` instead of numpy-1 ``copy=False``)."""
    valid = (np.dtype("float32"), np.dtype("float64"))
    y = np.asarray(y, order="C")
    if dtype is None:
        if y.dtype not in valid:
            raise RuntimeError(
                "y must have an underlying type of np.float32 or np.float64, "
                "or dtype must be explicitly specified."
            )
        dtype = y.dtype.type
    else:
        if np.dtype(dtype) not in valid:
            raise RuntimeError("dtype must be either np.float32 or np.float64.")
        dtype = np.dtype(dtype).type
    return y.astype(dtype, copy=False), dtype


class GlmBase:
    """Base class of every family, and the plugin surface for user-defined ones (reference ``GlmBase{32,64}``,
    ``glm/glm_base.hpp:19-93``, Python-subclassable through ``py_glm.cpp:8-92``).

    A user-defined single-response family subclasses :class:`GlmBase64` (or :class:`GlmBase32`), calls
    ``GlmBase64.__init__(self, name, y, weights)`` and implements ``gradient(eta, grad)`` (the NEGATIVE gradient of the
    loss, written in place), ``hessian(eta, grad, hess)``, ``loss(eta)`` and ``loss_full()``; ``inv_hessian_gradient`` and
    ``inv_link`` are optional.  ``grpnet`` runs such a family through the IRLS solver on the device and calls these methods
    on host n-vectors once per IRLS iteration (``adelie_hip_glm_callbacks``)."""

    is_multi = False
    opt = False

    def __init__(self, name, y, weights):
        self.name = name
        self.y = y
        self.weights = weights

    def gradient(self, eta, grad):
        raise NotImplementedError("gradient() must be implemented by the GLM subclass.")

    def hessian(self, eta, grad, hess):
        raise NotImplementedError("hessian() must be implemented by the GLM subclass.")

    def loss(self, eta):
        raise NotImplementedError("loss() must be implemented by the GLM subclass.")

    def loss_full(self):
        raise NotImplementedError("loss_full() must be implemented by the GLM subclass.")

    def inv_link(self, eta, out):
        raise NotImplementedError("inv_link() must be implemented by the GLM subclass.")

    # glm_base.ipp:23-37
    def inv_hessian_gradient(self, eta, grad, hess, inv_hess_grad):
        hmin = _configs.Configs.hessian_min
        inv_hess_grad[...] = grad / (np.maximum(hess, 0) + hmin * (hess <= 0))


class GlmBase64(GlmBase):
    pass


class GlmBase32(GlmBase):
    pass


class glm_base:
    """Reference ``glm.py:36-55``."""

    def __init__(self, y, weights, dtype):
        self.y = np.array(y, copy=True, dtype=dtype)
        self.dtype = dtype
        if len(y.shape) != 1:
            raise RuntimeError("y must be 1-dimensional.")
        n = y.shape[0]
        if weights is not None:
            weights = np.asarray(weights)
            if weights.shape != (n,):
                raise RuntimeError("y and weights must have same length.")
            weights_sum = np.sum(weights)
            if not np.allclose(weights_sum, 1):
                weights = weights / weights_sum
        else:
            weights = np.full(n, 1 / n, dtype=dtype)
        self.weights = np.array(weights, copy=True, dtype=dtype)

    # glm_base.ipp:23-37
    def inv_hessian_gradient(self, eta, grad, hess, inv_hess_grad):
        hmin = self.dtype(_configs.Configs.hessian_min)
        inv_hess_grad[...] = grad / (np.maximum(hess, 0) + hmin * (hess <= 0))


def _mixin(dtype):
    return GlmBase64 if np.dtype(dtype) == np.float64 else GlmBase32


def gaussian(y, *, weights=None, dtype=None, opt: bool = True):
    """Gaussian family (reference ``adelie.glm.gaussian``, ``glm.py:374-453``)."""
    y, dtype = _coerce_dtype(y, dtype)

    class _gaussian(glm_base, _mixin(dtype)):
        name = "gaussian"

        def __init__(self):
            self.opt = opt
            glm_base.__init__(self, y, weights, dtype)
            self.core_kind = _abi.GLM_GAUSSIAN if opt else _abi.GLM_GAUSSIAN_IRLS

        # glm_gaussian.ipp:15-63
        def gradient(self, eta, grad):
            grad[...] = self.weights * (self.y - eta)

        def hessian(self, eta, grad, hess):
            hess[...] = self.weights

        def loss(self, eta):
            return np.sum(self.weights * (0.5 * np.square(eta) - self.y * eta))

        def loss_full(self):
            return -0.5 * np.sum(np.square(self.y) * self.weights)

        def inv_link(self, eta, out):
            out[...] = eta

        def reweight(self, weights=None):
            w = self.weights if weights is None else weights
            return gaussian(y=y, weights=w, dtype=dtype, opt=opt)

    return _gaussian()


class multiglm_base:
    """Reference ``glm.py:57-80``: ``y`` is ``(n, K)``, one weight per observation."""

    def __init__(self, y, weights, dtype):
        self.y = np.array(y, copy=True, dtype=dtype)
        self.dtype = dtype
        if len(y.shape) != 2:
            raise RuntimeError("y must be 2-dimensional.")
        n = y.shape[0]
        if weights is not None:
            weights = np.asarray(weights)
            if weights.shape != (n,):
                raise RuntimeError("y rows and weights must have same length.")
            weights_sum = np.sum(weights)
            if not np.allclose(weights_sum, 1):
                weights = weights / weights_sum
        else:
            weights = np.full(n, 1 / n, dtype=dtype)
        self.weights = np.array(weights, copy=True, dtype=dtype)


def multigaussian(y, *, weights=None, dtype=None, opt: bool = True):
    """MultiGaussian family (reference ``adelie.glm.multigaussian``, ``glm.py:456-535``; arithmetic
    ``glm_multigaussian.ipp:15-63``): ``loss = (1/K) sum_i w_i (||eta_i||^2 / 2 - y_i . eta_i)``.

    Only the optimised route (``opt=True``: the Gaussian naive solver on ``[1 (x) I_K, X (x) I_K]``) is on the device
    path; the IRLS route for multi-response GLMs is not."""
    y, dtype = _coerce_dtype(y, dtype)
    if not opt:
        raise NotImplementedError("adelie_amd.glm.multigaussian: only opt=True is on the device path.")

    class _multigaussian(multiglm_base, _mixin(dtype)):
        name = "multigaussian"
        is_multi = True

        def __init__(self):
            self.opt = opt
            multiglm_base.__init__(self, y, weights, dtype)

        def gradient(self, eta, grad):
            grad[...] = (self.weights[:, None] * (self.y - eta)) / self.y.shape[1]

        def hessian(self, eta, grad, hess):
            hess[...] = self.weights[:, None] / self.y.shape[1]

        def loss(self, eta):
            return np.sum(self.weights * np.sum(0.5 * np.square(eta) - self.y * eta, axis=1)) / self.y.shape[1]

        def loss_full(self):
            return -0.5 * np.sum(np.square(self.y) * self.weights[:, None]) / self.y.shape[1]

        def inv_link(self, eta, out):
            out[...] = eta

        def reweight(self, weights=None):
            w = self.weights if weights is None else weights
            return multigaussian(y=y, weights=w, dtype=dtype, opt=opt)

    return _multigaussian()


def multinomial(y, *, weights=None, dtype=None):
    """Multinomial family (reference ``adelie.glm.multinomial``, ``glm.py:538-618``; arithmetic ``glm_multinomial.ipp:21-115``):
    ``loss = (1/K) sum_i w_i (-y_i . eta_i + log sum_k exp eta_ik)``, diagonal majorant ``2 K^-1 W P (1 - P)`` as Hessian."""
    y, dtype = _coerce_dtype(y, dtype)
    if y.ndim == 2 and y.shape[1] <= 1:
        raise RuntimeError("adelie_core: y must have at least 2 columns (classes).")

    class _multinomial(multiglm_base, _mixin(dtype)):
        name = "multinomial"
        is_multi = True
        opt = False

        def __init__(self):
            multiglm_base.__init__(self, y, weights, dtype)
            self.core_kind = _abi.GLM_MULTINOMIAL

        def _prob(self, eta):
            e = np.exp(eta - np.max(eta, axis=1)[:, None])
            return e / np.sum(e, axis=1)[:, None]

        def gradient(self, eta, grad):
            grad[...] = (self.y - self._prob(eta)) * self.weights[:, None] / self.y.shape[1]

        def hessian(self, eta, grad, hess):
            K = self.y.shape[1]
            w = self.weights[:, None]
            h = self.y * w / K - grad
            hess[...] = h * (2 * (1 - K * (h / (w + (w <= 0)))))

        def inv_hessian_gradient(self, eta, grad, hess, inv_hess_grad):
            hmin = self.dtype(_configs.Configs.hessian_min)
            inv_hess_grad[...] = grad / (np.maximum(hess, 0) + hmin * (hess <= 0))

        def loss(self, eta):
            es = eta - np.max(eta, axis=1)[:, None]
            return np.sum(self.weights * (-np.sum(self.y * es, axis=1) + np.log(np.sum(np.exp(es), axis=1)))) / self.y.shape[1]

        def loss_full(self):
            with np.errstate(divide="ignore", invalid="ignore"):
                ly = np.log(self.y)
                t = np.where(np.isfinite(ly), self.y * ly, 0.0)
            return self.dtype(-np.sum(np.sum(t, axis=1) * self.weights) / self.y.shape[1])

        def inv_link(self, eta, out):
            out[...] = self._prob(eta)

        def reweight(self, weights=None):
            w = self.weights if weights is None else weights
            return multinomial(y=y, weights=w, dtype=dtype)

    return _multinomial()


def poisson(y, *, weights=None, dtype=None):
    """Poisson family, log link (reference ``adelie.glm.poisson``, ``glm.py:621-697``; arithmetic ``glm_poisson.ipp:14-58``)."""
    y, dtype = _coerce_dtype(y, dtype)

    class _poisson(glm_base, _mixin(dtype)):
        name = "poisson"

        def __init__(self):
            glm_base.__init__(self, y, weights, dtype)
            self.core_kind = _abi.GLM_POISSON

        def gradient(self, eta, grad):
            grad[...] = self.weights * (self.y - np.exp(eta))

        def hessian(self, eta, grad, hess):
            hess[...] = self.weights * self.y - grad

        def loss(self, eta):
            mx = np.finfo(self.dtype).max
            return np.sum(self.weights * (np.minimum(-eta, mx) * self.y + np.exp(eta)))

        def loss_full(self):
            mx = np.finfo(self.dtype).max
            with np.errstate(divide="ignore", invalid="ignore"):
                t = np.minimum(-np.log(self.y), mx) * self.y
            return self.dtype(np.sum(self.weights * (np.where(self.y > 0, t, 0.0) + self.y)))

        def inv_link(self, eta, out):
            out[...] = np.exp(eta)

        def reweight(self, weights=None):
            w = self.weights if weights is None else weights
            return poisson(y=y, weights=w, dtype=dtype)

    return _poisson()


def _binomial_probit(y, weights, dtype):
    """Binomial family, probit link (reference ``glm_binomial.ipp:100-190``)."""
    from scipy.special import erf

    class _probit(glm_base, _mixin(dtype)):
        name = "binomial_probit"

        def __init__(self):
            glm_base.__init__(self, y, weights, dtype)
            self.core_kind = _abi.GLM_BINOMIAL_PROBIT

        @staticmethod
        def _cdf(x):
            return 0.5 * (1 + erf(x / np.sqrt(2)))

        @staticmethod
        def _pdf(x):
            return np.exp(-0.5 * np.square(x)) / np.sqrt(2 * np.pi)

        def gradient(self, eta, grad):
            mx = np.finfo(self.dtype).max
            c = self._cdf(eta)
            with np.errstate(divide="ignore"):
                grad[...] = self.weights * self._pdf(eta) * (
                    self.y * np.minimum(1 / c, mx) - (1 - self.y) * np.minimum(1 / (1 - c), mx))

        def hessian(self, eta, grad, hess):
            mx = np.finfo(self.dtype).max
            c = self._cdf(eta)
            with np.errstate(divide="ignore"):
                hess[...] = self.weights * (
                    self.y * np.minimum(1 / np.square(c), mx) + (1 - self.y) * np.minimum(1 / np.square(1 - c), mx)
                ) * np.square(self._pdf(eta)) + eta * grad

        def loss(self, eta):
            mx = np.finfo(self.dtype).max
            c = self._cdf(eta)
            with np.errstate(divide="ignore"):
                return -np.sum(self.weights * (
                    self.y * np.maximum(np.log(c), -mx) + (1 - self.y) * np.maximum(np.log(1 - c), -mx)))

        def loss_full(self):
            with np.errstate(divide="ignore", invalid="ignore"):
                ly, l1y = np.log(self.y), np.log(1 - self.y)
                t1, t2 = self.weights * self.y * ly, self.weights * (1 - self.y) * l1y
            return self.dtype(-np.sum(t1[np.isfinite(ly)]) - np.sum(t2[np.isfinite(l1y)]))

        def inv_link(self, eta, out):
            out[...] = self._cdf(eta)

        def reweight(self, weights=None):
            w = self.weights if weights is None else weights
            return _binomial_probit(y, w, dtype)

    return _probit()


def binomial(y, *, weights=None, link: str = "logit", dtype=None):
    """Binomial family, logit or probit link (reference ``adelie.glm.binomial``, ``glm.py:83-196``)."""
    y, dtype = _coerce_dtype(y, dtype)
    if link == "probit":
        return _binomial_probit(y, weights, dtype)
    if link != "logit":
        raise RuntimeError("link must be one of 'logit' or 'probit'.")

    class _binomial(glm_base, _mixin(dtype)):
        name = "binomial_logit"

        def __init__(self):
            glm_base.__init__(self, y, weights, dtype)
            self.core_kind = _abi.GLM_BINOMIAL_LOGIT

        # glm_binomial.ipp:37-99
        def gradient(self, eta, grad):
            grad[...] = self.weights * (self.y - 1 / (1 + np.exp(-eta)))

        def hessian(self, eta, grad, hess):
            w = self.weights
            h = w * self.y - grad
            hess[...] = (h * (w - h)) / (w + (w <= 0))

        def loss(self, eta):
            mx = np.finfo(self.dtype).max
            return np.sum(self.weights * (
                ((eta > 0).astype(self.dtype) - self.y) * np.clip(eta, -mx, mx)
                + np.log1p(np.exp(-np.abs(eta)))
            ))

        def loss_full(self):
            # glm_binomial.ipp:14-33: non-finite logs are skipped
            with np.errstate(divide="ignore", invalid="ignore"):
                ly = np.log(self.y)
                l1y = np.log(1 - self.y)
                t1 = self.weights * self.y * ly
                t2 = self.weights * (1 - self.y) * l1y
            loss = 0.0
            loss -= np.sum(t1[np.isfinite(ly)])
            loss -= np.sum(t2[np.isfinite(l1y)])
            return self.dtype(loss)

        def inv_link(self, eta, out):
            out[...] = 1 / (1 + np.exp(-eta))

        def reweight(self, weights=None):
            w = self.weights if weights is None else weights
            return binomial(y=y, weights=w, dtype=dtype)

    return _binomial()
