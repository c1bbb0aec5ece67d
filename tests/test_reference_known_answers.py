"""Known answers printed by the reference itself: docs/sphinx/user_guide/notebooks/sklearn_api.ipynb runs
``GroupElasticNet`` (= ``grpnet`` / ``cv_grpnet`` + ``CVGrpnetResult.fit`` + ``predict``, adelie/sklearn.py:43-262) on
scikit-learn's bundled datasets after ``np.random.seed(42)`` and shows

  cell 10   grpnet on the diabetes training split: 100/100 lambdas, progress suffix ``[dev:52.5%]``
  cell 17   cv_grpnet(min_ratio=1e-3): the five folds end at ``[dev:51.8%] [dev:50.8%] [dev:57.5%] [dev:52.6%] [dev:53.0%]``,
            the refit at the best lambda at ``[dev:52.6%]``
  cell 21   ``model_cv_gaussian.score(X_test, y_test)`` = 0.45673736460682524
  cell 27/29  binomial cv_grpnet on the breast-cancer split; test confusion matrix [[28, 16], [0, 70]]

Two more notebooks print the progress suffix of paths on seeded numpy data:

  faq.ipynb cells 4-12       n=100, p=200, seed 0: ``20/20 ... [dev:98.0%]`` for both ways of ending the path at a user
                             lambda, and the cell-10 assertion ``state.lmdas[-1] == lmda_star`` (exact equality)
  parallelism.ipynb cell 8   n=100, p=1000, seed 0: ``46/100 ... [dev:90.6%]`` (default early exit)

(quickstart.ipynb's 46 / 38 / 57 solutions at 90.5 / 90.2 / 90.2 % are checked in test_oracle_solver.py.)

The whole sequence is deterministic (the splits and the CV fold order draw from the seeded global numpy stream), so it can
be replayed here: this pins the oracle — and, on the GPU, the HIP path — against outputs of the reference binary, which
cannot be built in this environment."""
import numpy as np
import pytest

import adelie_amd as ad

sklearn_datasets = pytest.importorskip("sklearn.datasets")
from sklearn.model_selection import train_test_split  # noqa: E402

REF_SCORE = 0.45673736460682524            # sklearn_api.ipynb cell 21
REF_FOLD_DEVS = [51.8, 50.8, 57.5, 52.6, 53.0]  # cell 17, first five progress bars
REF_REFIT_DEV = 52.6                        # cell 17, last progress bar
REF_GRPNET_DEV = 52.5                       # cell 10
REF_CONFUSION = [[28, 16], [0, 70]]        # cell 29


def _replay(dense, monkeypatch):
    """Replays cells 2-29 of the notebook with `dense` as the matrix constructor; returns what the notebook prints."""
    fold_devs = []
    real_grpnet = ad.cv.grpnet

    def spy(*args, **kwargs):
        st = real_grpnet(*args, **kwargs)
        if kwargs.get("lmda_path") is not None and len(st.devs):
            fold_devs.append(100 * float(st.devs[-1]))
        return st

    monkeypatch.setattr(ad.cv, "grpnet", spy)
    out = {}
    np.random.seed(42)
    data = sklearn_datasets.load_diabetes()
    X_train, X_test, y_train, y_test = train_test_split(data.data, data.target, test_size=0.2)
    X_train, X_test = np.asfortranarray(X_train), np.asfortranarray(X_test)
    assert X_train.shape == (353, 10) and X_test.shape == (89, 10)   # cell 5 output
    Xd = dense(X_train)
    st = ad.grpnet(Xd, ad.glm.gaussian(y_train))                     # cell 10
    out["grpnet_nsol"], out["grpnet_dev"] = len(st.lmdas), 100 * st.devs[-1]
    glm = ad.glm.gaussian(y_train)
    cv = ad.cv_grpnet(Xd, glm, min_ratio=1e-3, n_concurrent=1)       # cell 17
    fit = cv.fit(Xd, glm, min_ratio=1e-3)
    out["fold_devs"] = list(fold_devs[:5])
    out["refit_dev"] = 100 * fit.devs[-1]
    yhat = ad.diagnostic.predict(dense(X_test), fit.betas[-1], np.array([fit.intercepts[-1]])).squeeze()
    out["score"] = 1 - np.sum((y_test - yhat) ** 2) / np.sum((y_test - np.mean(y_test)) ** 2)   # sklearn.py:239-245
    data = sklearn_datasets.load_breast_cancer()                     # cell 25
    X_train, X_test, y_train, y_test = train_test_split(data.data, data.target, test_size=0.2)
    X_train, X_test = np.asfortranarray(X_train), np.asfortranarray(X_test)
    assert X_train.shape == (455, 30) and X_test.shape == (114, 30)
    Xd = dense(X_train)
    glm = ad.glm.binomial(y_train.astype(np.float64))
    cv = ad.cv_grpnet(Xd, glm, n_concurrent=1)                       # cell 27
    fit = cv.fit(Xd, glm)
    eta = ad.diagnostic.predict(dense(X_test), fit.betas[-1], np.array([fit.intercepts[-1]])).squeeze()
    proba = 1 / (1 + np.exp(-eta))                                   # sklearn.py:183-186 (expit, argmax)
    yhat = np.argmax(np.stack((1 - proba, proba), axis=-1), axis=-1)
    out["confusion"] = [[int(np.sum((y_test == a) & (yhat == b))) for b in (0, 1)] for a in (0, 1)]
    return out


def _check(out):
    assert out["grpnet_nsol"] == 100
    assert round(out["grpnet_dev"], 1) == REF_GRPNET_DEV
    assert [round(d, 1) for d in out["fold_devs"]] == REF_FOLD_DEVS
    assert round(out["refit_dev"], 1) == REF_REFIT_DEV
    assert abs(out["score"] - REF_SCORE) < 1e-9, out["score"]
    assert out["confusion"] == REF_CONFUSION


def test_oracle_reproduces_reference_notebook(oracle, monkeypatch):
    _check(_replay(oracle.dense, monkeypatch))


@pytest.mark.gpu
def test_hip_reproduces_reference_notebook(hip, monkeypatch):
    _check(_replay(ad.matrix.dense, monkeypatch))


def _replay_small(dense):
    out = {}
    # faq.ipynb cells 4-12
    n, p, lmda_star = 100, 200, 1e-2
    np.random.seed(0)
    X = np.asfortranarray(np.random.normal(0, 1, (n, p)))
    y = X[:, 0] * np.random.normal(0, 1) + np.random.normal(0, 1, n)
    Xd = dense(X)
    lmda_max = ad.grpnet(Xd, ad.glm.gaussian(y), lmda_path_size=0, progress_bar=False).lmda_max
    st = ad.grpnet(Xd, ad.glm.gaussian(y), min_ratio=min(1, lmda_star / lmda_max), lmda_path_size=20, early_exit=False)
    out["faq1"] = (len(st.lmdas), round(100 * st.devs[-1], 1), abs(st.lmdas[-1] / lmda_star - 1))
    st = ad.grpnet(Xd, ad.glm.gaussian(y), lmda_path=np.linspace(lmda_max, lmda_star, num=20, endpoint=True),
                   early_exit=False)
    out["faq2"] = (len(st.lmdas), round(100 * st.devs[-1], 1))
    # parallelism.ipynb cells 5-8
    np.random.seed(0)
    X = np.random.normal(0, 1, (100, 1000))
    y = X[:, -1] * np.random.normal(0, 1) + np.random.normal(0, 1, 100)
    st = ad.grpnet(dense(np.asfortranarray(X)), ad.glm.gaussian(y))
    out["parallelism"] = (len(st.lmdas), 100 * st.devs[-1])
    return out


def _check_small(out):
    # cell 10 asserts exact equality; lmda_max * exp(log(lmda_star / lmda_max)) hits lmda_star exactly for the reference's
    # (and the oracle's) lmda_max, and within an ulp or two for a lmda_max whose dot products were summed in another order
    assert out["faq1"][:2] == (20, 98.0) and out["faq1"][2] < 1e-15
    assert out["faq2"] == (20, 98.0)
    # printed as 90.6 by a 4-thread run; this path ends at 90.54998 %, i.e. 2e-7 (in deviance fraction) below the rounding
    # boundary - inside the resolution of the stopping rule (tol = 1e-7) on a p = 10 n problem
    assert out["parallelism"][0] == 46 and abs(out["parallelism"][1] - 90.6) < 0.0501


def test_oracle_reproduces_faq_and_parallelism_notebooks(oracle):
    _check_small(_replay_small(oracle.dense))


@pytest.mark.gpu
def test_hip_reproduces_faq_and_parallelism_notebooks(hip):
    _check_small(_replay_small(ad.matrix.dense))


def _replay_glm_notebook(dense):
    """glm.ipynb cells 5-27: the notebook draws everything from one numpy stream seeded in cell 5, then runs the same data
    through ``glm.gaussian(y, opt=False)`` — the Gaussian family forced through the IRLS (StateGlmNaive) route — and prints
    ``44/100 ... [dev:90.7%]`` (cell 27; cell 25 prints the same for a Python-subclassed Gaussian)."""
    n, K, p = 100, 4, 1000
    np.random.seed(0)
    np.random.normal(0, 1, n)          # cell 5  y
    np.random.uniform(0, 1, n)         # cell 7  w
    np.random.normal(0, 1, n)          # cell 10 eta
    np.random.normal(0, 1, (n, K))     # cell 15 multi-response y
    np.random.normal(0, 1, (n, K))     # cell 17 multi-response eta
    X = np.random.normal(0, 1, (n, p))                                      # cell 24
    y = X[:, -1] * np.random.normal(0, 1) + np.random.normal(0, 1, n)
    Xd = dense(np.asfortranarray(X))
    out = {}
    for opt in (False, True):
        st = ad.grpnet(Xd, ad.glm.gaussian(y=y, opt=opt))
        assert st.error == ""
        out[opt] = (len(st.lmdas), f"{100 * st.devs[-1]:.1f}", st.devs[-2] < 0.9)
    return out


def test_oracle_reproduces_glm_notebook(oracle):
    out = _replay_glm_notebook(oracle.dense)
    assert out[False] == (44, "90.7", True) and out[True] == (44, "90.7", True)


@pytest.mark.gpu
def test_hip_reproduces_glm_notebook(hip):
    out = _replay_glm_notebook(ad.matrix.dense)
    assert out[False] == (44, "90.7", True) and out[True] == (44, "90.7", True)


def _replay_matrix_notebook(dense):
    """matrix.ipynb cells 5-21 (one numpy stream seeded in cell 5): ``52/100 ... [dev:90.2%]`` for both the Python-subclassed
    dense matrix (cell 19) and ``ad.matrix.dense`` (cell 21)."""
    n, p = 100, 1000
    np.random.seed(0)
    X = np.random.normal(0, 1, (n, p))     # cell 5
    np.random.uniform(0, 1, n)             # cell 7  w
    np.random.normal(0, 1, n)              # cell 7  v
    np.random.normal(0, 1, 3)              # cell 11 values
    y = X[:, -1] * np.random.normal(0, 1) + np.random.normal(0, 1, n)   # cell 18
    st = ad.grpnet(dense(np.asfortranarray(X)), ad.glm.gaussian(y=y))
    assert st.error == ""
    return len(st.lmdas), f"{100 * st.devs[-1]:.1f}", st.devs[-2] < 0.9


def test_oracle_reproduces_matrix_notebook(oracle):
    assert _replay_matrix_notebook(oracle.dense) == (52, "90.2", True)


@pytest.mark.gpu
def test_hip_reproduces_matrix_notebook(hip):
    assert _replay_matrix_notebook(ad.matrix.dense) == (52, "90.2", True)


REF_DIGITS_CONFUSION = [                    # sklearn_api.ipynb cell 39
    [29, 0, 0, 0, 0, 0, 0, 0, 0, 0],
    [0, 32, 0, 0, 0, 0, 0, 0, 2, 2],
    [0, 0, 41, 1, 0, 0, 0, 0, 0, 1],
    [0, 1, 0, 34, 0, 0, 0, 0, 1, 0],
    [0, 1, 0, 0, 37, 0, 0, 1, 0, 1],
    [0, 0, 0, 0, 0, 37, 0, 0, 0, 3],
    [0, 1, 0, 0, 0, 0, 37, 0, 0, 0],
    [0, 0, 0, 0, 0, 1, 0, 31, 0, 0],
    [0, 3, 0, 1, 0, 2, 0, 2, 23, 0],
    [0, 0, 0, 0, 0, 0, 0, 1, 0, 34],
]


def _replay_digits(dense):
    """Cells 35-39 of the same notebook: ``GroupElasticNet(solver="cv_grpnet", family="multinomial")`` on scikit-learn's digits
    (1437 x 64 training rows, 10 classes), refit at the CV-selected lambda (``100/100 ... [dev:84.7%]``), test confusion matrix.
    The train/test split draws from the notebook's global numpy stream, so the earlier cells' draws are repeated first (the
    splits and the two fold orders; the fits themselves draw nothing)."""
    np.random.seed(42)
    d = sklearn_datasets.load_diabetes()
    train_test_split(d.data, d.target, test_size=0.2)
    np.random.choice(353, 353, replace=False)            # fold order of the Gaussian cv_grpnet (cv.py:219-221)
    d = sklearn_datasets.load_breast_cancer()
    train_test_split(d.data, d.target, test_size=0.2)
    np.random.choice(455, 455, replace=False)            # fold order of the binomial cv_grpnet
    d = sklearn_datasets.load_digits()
    X_train, X_test, y_train, y_test = train_test_split(d.data, d.target, test_size=0.2)
    X_train, X_test = np.asfortranarray(X_train), np.asfortranarray(X_test)
    assert X_train.shape == (1437, 64) and X_test.shape == (360, 64)   # cell 35 output
    y2 = np.eye(10)[y_train]                                            # OneHotEncoder(sparse_output=False)
    Xd = dense(X_train)
    glm = ad.glm.multinomial(y2)
    cv = ad.cv_grpnet(Xd, glm, n_concurrent=1)
    fit = cv.fit(Xd, glm)
    eta = ad.diagnostic.predict(dense(X_test), fit.betas[-1], np.array([fit.intercepts[-1]])).squeeze()
    yhat = np.argmax(eta, axis=-1)                                      # argmax of softmax, sklearn.py:185-212
    conf = [[int(np.sum((y_test == a) & (yhat == b))) for b in range(10)] for a in range(10)]
    return len(fit.lmdas), f"{100 * fit.devs[-1]:.1f}", conf


def test_oracle_reproduces_reference_multinomial_cv(oracle):
    n_lmdas, dev, conf = _replay_digits(oracle.dense)
    assert (n_lmdas, dev) == (100, "84.7")
    assert conf == REF_DIGITS_CONFUSION


@pytest.mark.gpu
def test_hip_reproduces_reference_multinomial_cv(hip):
    n_lmdas, dev, conf = _replay_digits(ad.matrix.dense)
    assert (n_lmdas, dev) == (100, "84.7")
    assert conf == REF_DIGITS_CONFUSION
