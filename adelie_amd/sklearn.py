"""scikit-learn style estimator over the grpnet path — mirrors ``adelie.sklearn.GroupElasticNet`` (reference
``adelie/sklearn.py:43-250``): the same constructor parameters, fitted attributes (``glm_``, ``state_``, ``coef_``,
``intercept_``, ``lambda_``), error types and messages.  It is a thin caller of ``grpnet`` / ``cv_grpnet`` +
``CVGrpnetResult.fit`` + ``diagnostic.predict``; every numeric step runs on the device behind the C ABI.
``CSSModelSelection`` (``:253-``, column-subset selection on a covariance) is outside the grpnet path and not provided."""
import numpy as np
from sklearn.base import BaseEstimator, RegressorMixin

from . import glm as _glm
from .cv import CVGrpnetResult, cv_grpnet
from .diagnostic import predict as _predict
from .solver import grpnet

_FAMILIES = {
    "gaussian": _glm.gaussian,
    "binomial": _glm.binomial,
    "poisson": _glm.poisson,
    "multigaussian": _glm.multigaussian,
    "multinomial": _glm.multinomial,
}
_SOLVERS = {"grpnet": grpnet, "cv_grpnet": cv_grpnet}


class GroupElasticNet(BaseEstimator, RegressorMixin):
    """Group elastic net estimator.

    Parameters
    ----------
    solver : ``"grpnet"`` (whole path) or ``"cv_grpnet"`` (cross-validate, then refit down to the best lambda).
    family : ``"gaussian"``, ``"binomial"``, ``"poisson"``, ``"multigaussian"`` or ``"multinomial"``.
    """

    def __init__(self, solver: str = "grpnet", family: str = "gaussian"):
        self.solver = solver
        self.family = family

    def fit(self, X, y, **kwargs):
        """Fits the path (or the cross-validated model); ``kwargs`` go to the solver (reference ``sklearn.py:82-150``)."""
        self._validate_params()
        self.glm_ = _FAMILIES[self.family](y)
        self.state_ = _SOLVERS[self.solver](X=X, glm=self.glm_, **kwargs)
        if isinstance(self.state_, CVGrpnetResult):
            self.state_ = self.state_.fit(X=X, glm=self.glm_, **kwargs)
            self.coef_ = self.state_.betas[-1]
            self.intercept_ = np.array([self.state_.intercepts[-1]])
            self.lambda_ = np.array([self.state_.lmdas[-1]])
        else:
            self.coef_ = self.state_.betas
            self.intercept_ = self.state_.intercepts
            self.lambda_ = self.state_.lmdas
        return self

    def _linear(self, X):
        if not hasattr(self, "state_"):
            raise RuntimeError("The model has not been fitted yet. Call fit() first.")
        return _predict(X, self.coef_, self.intercept_)

    def predict_proba(self, X):
        """Class probabilities (binomial: two columns; multinomial: K), reference ``sklearn.py:152-186``."""
        if not hasattr(self, "state_"):
            raise RuntimeError("The model has not been fitted yet. Call fit() first.")
        if self.family not in ("binomial", "multinomial"):
            raise ValueError("predict_proba is only available for \"binomial\" and \"multinomial\" families.")
        eta = self._linear(X)
        if self.family == "binomial":
            pr = 0.5 * (1 + np.tanh(0.5 * eta))  # the logistic function, stable for either sign
            return np.stack((1 - pr, pr), axis=-1).squeeze()
        e = np.exp(eta - np.max(eta, axis=-1, keepdims=True))
        return (e / np.sum(e, axis=-1, keepdims=True)).squeeze()

    def predict(self, X):
        """Class labels for the binomial / multinomial families, linear predictions otherwise (``sklearn.py:188-214``)."""
        if self.family in ("binomial", "multinomial"):
            return np.argmax(self.predict_proba(X), axis=-1).squeeze()
        return self._linear(X).squeeze()

    def score(self, X, y):
        """R-squared of ``predict(X)`` clipped to [0, 1] (``sklearn.py:216-237``)."""
        yhat = self.predict(X)
        ss_res = np.sum((y - yhat) ** 2)
        ss_tot = np.sum((y - np.mean(y)) ** 2)
        return np.clip(1 - ss_res / ss_tot, 0, 1)

    def _validate_params(self):
        if self.solver not in _SOLVERS:
            raise ValueError(f"Unknown solver: {self.solver}")
        if self.family not in _FAMILIES:
            raise ValueError(f"Unknown family: {self.family}")
