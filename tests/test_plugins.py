"""The plugin surfaces of the reference that run user code: GLMs and matrices subclassed in Python, and ``exit_cond`` on
the live state (reference trampolines ``py_glm.cpp:8-92`` / ``py_matrix.cpp:626-825``, ``py_state.cpp:62-91``).

The user classes below are the ones the reference's documentation defines (``glm.ipynb`` cell 21, ``matrix.ipynb`` cell 15),
run on the notebooks' own random streams, so the printed outputs of those cells are the known answers."""
import numpy as np
import pytest

import adelie_amd as ad


class Gaussian(ad.glm.GlmBase64):      # glm.ipynb cell 21, verbatim API usage
    def __init__(self, y, w=None):
        self.y = y
        self.w = np.full(y.shape[0], 1 / y.shape[0]) if w is None else w / np.sum(w)
        ad.glm.GlmBase64.__init__(self, "my_gaussian", self.y, self.w)

    def gradient(self, eta, grad):
        grad[...] = self.w * (self.y - eta)

    def hessian(self, eta, grad, hess):
        hess[...] = self.w

    def loss(self, eta):
        return np.sum(self.w * (-self.y * eta + 0.5 * eta ** 2))

    def loss_full(self):
        return -0.5 * np.sum(self.w * self.y ** 2)


class Logit(ad.glm.GlmBase64):
    """A second user family with a non-constant Hessian and its own inv_hessian_gradient."""

    def __init__(self, y):
        self.y = y
        self.w = np.full(y.shape[0], 1 / y.shape[0])
        self.n_ihg = 0
        ad.glm.GlmBase64.__init__(self, "my_logit", self.y, self.w)

    def gradient(self, eta, grad):
        grad[...] = self.w * (self.y - 1 / (1 + np.exp(-eta)))

    def hessian(self, eta, grad, hess):
        mu = self.y - grad / self.w
        hess[...] = self.w * mu * (1 - mu)

    def inv_hessian_gradient(self, eta, grad, hess, out):
        self.n_ihg += 1
        out[...] = grad / (np.maximum(hess, 0) + 1e-24 * (hess <= 0))

    def loss(self, eta):
        return np.sum(self.w * (np.logaddexp(0, eta) - self.y * eta))

    def loss_full(self):
        return 0.0


class Dense(ad.matrix.MatrixNaiveBase64):  # matrix.ipynb cell 15
    def __init__(self, mat):
        self.mat = mat
        ad.matrix.MatrixNaiveBase64.__init__(self)

    def bmul(self, j, q, v, w, out):
        out[...] = self.mat[:, j:j + q].T @ (w * v)

    def btmul(self, j, q, v, out):
        out[...] += self.mat[:, j:j + q] @ v

    def cmul(self, j, v, w):
        return self.mat[:, j] @ (w * v)

    def ctmul(self, j, v, out):
        out[...] += self.mat[:, j] * v

    def rows(self):
        return self.mat.shape[0]

    def cols(self):
        return self.mat.shape[1]

    def mul(self, v, w, out):
        out[...] = self.mat.T @ (w * v)


def _glm_notebook_data():
    n, K, p = 100, 4, 1000
    np.random.seed(0)
    np.random.normal(0, 1, n)          # cell 5  y
    np.random.uniform(0, 1, n)         # cell 7  w
    np.random.normal(0, 1, n)          # cell 10 eta
    np.random.normal(0, 1, (n, K))     # cell 15
    np.random.normal(0, 1, (n, K))     # cell 17
    X = np.random.normal(0, 1, (n, p))                                      # cell 24
    y = X[:, -1] * np.random.normal(0, 1) + np.random.normal(0, 1, n)
    return np.asfortranarray(X), y


def _custom_glm_notebook(dense):
    """glm.ipynb cells 25-28: ``44/100 ... [dev:90.7%]`` for the Python-subclassed Gaussian, coefficients ``allclose`` to
    ``glm.gaussian(opt=False)``."""
    X, y = _glm_notebook_data()
    Xd = dense(X)
    st = ad.grpnet(Xd, Gaussian(y=y))
    ref = ad.grpnet(Xd, ad.glm.gaussian(y=y, opt=False))
    assert st.error == "" and ref.error == ""
    assert (len(st.lmdas), f"{100 * st.devs[-1]:.1f}") == (44, "90.7")
    assert np.allclose(st.betas.toarray(), ref.betas.toarray())             # cell 28
    assert np.abs(st.betas.toarray() - ref.betas.toarray()).max() < 1e-12   # same arithmetic, same order
    return st


def test_oracle_custom_glm_notebook(oracle):
    _custom_glm_notebook(oracle.dense)


@pytest.mark.gpu
def test_hip_custom_glm_notebook(hip):
    _custom_glm_notebook(ad.matrix.dense)


def _custom_logit(dense):
    rng = np.random.RandomState(3)
    n, p = 300, 40
    X = np.asfortranarray(rng.normal(size=(n, p)))
    y = (rng.uniform(size=n) < 1 / (1 + np.exp(-X[:, :3] @ [1.0, -2.0, 1.5]))).astype(float)
    glm = Logit(y)
    kw = dict(lmda_path_size=25, min_ratio=0.05, early_exit=False, tol=1e-10, irls_tol=1e-10, progress_bar=False)
    st = ad.grpnet(dense(X), glm, **kw)
    ref = ad.grpnet(dense(X), ad.glm.binomial(y), **kw)
    assert st.error == "" and len(st.lmdas) == 25 and glm.n_ihg > 25
    assert np.abs(st.betas.toarray() - ref.betas.toarray()).max() < 1e-8
    assert np.abs(st.intercepts - ref.intercepts).max() < 1e-8
    assert np.abs(st.devs - ref.devs).max() < 1e-8
    return st


def test_oracle_custom_logit_equals_builtin_binomial(oracle):
    _custom_logit(oracle.dense)


@pytest.mark.gpu
def test_hip_custom_logit_equals_builtin_binomial(hip, oracle):
    a = _custom_logit(ad.matrix.dense)
    b = _custom_logit(oracle.dense)
    assert np.abs(a.betas.toarray() - b.betas.toarray()).max() < 1e-8


def _glm_exception(dense):
    X, y = _glm_notebook_data()

    class Broken(Gaussian):
        calls = 0

        def hessian(self, eta, grad, hess):
            Broken.calls += 1
            if Broken.calls > 3:
                raise ValueError("user hessian failed")
            hess[...] = self.w

    with pytest.raises(ValueError, match="user hessian failed"):
        ad.grpnet(dense(X), Broken(y=y), progress_bar=False)


def test_oracle_custom_glm_exception_propagates(oracle):
    _glm_exception(oracle.dense)


@pytest.mark.gpu
def test_hip_custom_glm_exception_propagates(hip):
    _glm_exception(ad.matrix.dense)


def test_custom_glm_validation(oracle):
    X, y = _glm_notebook_data()

    class NotAGlm:
        name, weights, is_multi, opt = "x", np.full(100, 0.01), False, False

        def gradient(self, eta, grad):
            grad[...] = 0

        def loss_full(self):
            return 0.0

    NotAGlm.y = y
    with pytest.raises(RuntimeError, match="subclass of adelie_amd.glm.GlmBase64"):
        ad.grpnet(oracle.dense(X), NotAGlm())

    class G32(ad.glm.GlmBase32):
        def __init__(self, y):
            ad.glm.GlmBase32.__init__(self, "g32", y.astype(np.float32), np.full(len(y), 1 / len(y), dtype=np.float32))

        gradient, loss_full = Gaussian.gradient, lambda self: 0.0
        w = np.full(len(y), 1 / len(y))

    with pytest.raises(RuntimeError, match="same underlying value type"):
        ad.grpnet(oracle.dense(X), G32(y))


# ---- exit_cond on the live state -----------------------------------------------------------------------------------------

def _exit_cond_live(dense):
    X, y = _glm_notebook_data()
    Xd = dense(X)
    full = ad.grpnet(Xd, ad.glm.gaussian(y), early_exit=False, lmda_path_size=30, progress_bar=False)
    seen = []

    def cond(state):
        # the reference's users read the live state: lmdas / devs / betas so far, the screen set, the current lambda
        k = len(state.lmdas)
        assert k == state.n_solutions == len(state.devs) == len(state.intercepts) == state.betas.shape[0]
        assert state.betas.shape[1] == X.shape[1]
        assert state.lmda == state.lmdas[-1]
        assert state.alpha == 1 and len(state.penalty) == X.shape[1]          # constructor arguments are there too
        assert state.active_set_size <= len(state.screen_set)
        grad = state.grad                                                        # device-resident: copied on request
        r = state.resid
        assert np.abs(grad - (X.T @ (r / len(y)) - np.mean(r) * state.X_means)).max() < 1e-9
        seen.append((k, float(state.devs[-1])))
        return state.devs[-1] > 0.5

    st = ad.grpnet(Xd, ad.glm.gaussian(y), early_exit=False, lmda_path_size=30, progress_bar=False, exit_cond=cond)
    k = len(st.lmdas)
    assert 1 < k < 30 and st.devs[-1] > 0.5 and st.devs[-2] <= 0.5
    assert [s[0] for s in seen] == list(range(1, k + 1))
    np.testing.assert_allclose(st.betas.toarray(), full.betas[:k].toarray(), atol=1e-12)
    # the reference-style one-liner
    st2 = ad.grpnet(Xd, ad.glm.gaussian(y), early_exit=False, lmda_path_size=30, progress_bar=False,
                    exit_cond=lambda s: s.lmdas[-1] < 0.5 * s.lmda_max)
    assert st2.lmdas[-1] < 0.5 * st2.lmda_max <= st2.lmdas[-2]

    def bad(state):
        raise KeyError("oops")

    with pytest.raises(KeyError):
        ad.grpnet(Xd, ad.glm.gaussian(y), progress_bar=False, exit_cond=bad)
    with pytest.raises(AttributeError):
        ad.grpnet(Xd, ad.glm.gaussian(y), progress_bar=False, exit_cond=lambda s: s.no_such_attribute)


def test_oracle_exit_cond_live_state(oracle):
    _exit_cond_live(oracle.dense)


@pytest.mark.gpu
def test_hip_exit_cond_live_state(hip):
    _exit_cond_live(ad.matrix.dense)


def test_progress_bar_renders_reference_suffix(oracle, capsys):
    X, y = _glm_notebook_data()
    ad.grpnet(oracle.dense(X), ad.glm.gaussian(y))              # progress_bar defaults to True, as in the reference
    err = capsys.readouterr().err
    last = err.strip().split("\r")[-1]
    assert "| 44/100 [" in last and last.endswith("[dev:90.7%]"), last
    ad.grpnet(oracle.dense(X), ad.glm.gaussian(y), progress_bar=False)
    assert capsys.readouterr().err == ""


# ---- user-defined matrix ---------------------------------------------------------------------------------------------------

def _matrix_notebook_data():
    n, p = 100, 1000
    np.random.seed(0)
    X = np.random.normal(0, 1, (n, p))     # cell 5
    np.random.uniform(0, 1, n)             # cell 7  w
    np.random.normal(0, 1, n)              # cell 7  v
    np.random.normal(0, 1, 3)              # cell 11 values
    y = X[:, -1] * np.random.normal(0, 1) + np.random.normal(0, 1, n)   # cell 18
    return X, y


def test_densify_plugin_is_the_users_matrix():
    X, _ = _matrix_notebook_data()
    D = ad.matrix.densify_plugin(Dense(X))
    assert D.flags.f_contiguous and np.array_equal(D, X)
    with pytest.raises(RuntimeError, match="subclass of MatrixNaiveBase64"):
        ad.matrix.densify_plugin(object())


def test_oracle_custom_matrix_notebook(oracle):
    """matrix.ipynb cells 19-22 with the user's matrix evaluated once (what ``as_design`` does on the GPU)."""
    X, y = _matrix_notebook_data()
    st = ad.grpnet(oracle.dense(ad.matrix.densify_plugin(Dense(X))), ad.glm.gaussian(y=y), progress_bar=False)
    ref = ad.grpnet(oracle.dense(np.asfortranarray(X)), ad.glm.gaussian(y=y), progress_bar=False)
    assert (len(st.lmdas), f"{100 * st.devs[-1]:.1f}") == (52, "90.2")
    assert np.allclose(st.betas.toarray(), ref.betas.toarray())


@pytest.mark.gpu
def test_hip_custom_matrix_notebook(hip):
    X, y = _matrix_notebook_data()
    st = ad.grpnet(X=Dense(X), glm=ad.glm.gaussian(y=y))                   # cell 19
    ref = ad.grpnet(X=X, glm=ad.glm.gaussian(y=y))                         # cell 21
    assert (len(st.lmdas), f"{100 * st.devs[-1]:.1f}") == (52, "90.2")
    assert np.allclose(st.betas.toarray(), ref.betas.toarray())             # cell 22
    cv = ad.cv_grpnet(Dense(X), ad.glm.gaussian(y=y), n_folds=3, seed=0, lmda_path_size=10)
    assert cv.losses.shape == (3, 10)
