"""``adelie_amd.sklearn.GroupElasticNet`` — the estimator of reference ``adelie/sklearn.py:43-250``.

The known answers are the ones docs/sphinx/user_guide/notebooks/sklearn_api.ipynb prints for this very class (see
test_reference_known_answers.py for the cell numbers): here the notebook's calls are replayed through the estimator
itself, on the CPU oracle (``-m "not gpu"``) and on the HIP path (``-m gpu``)."""
import numpy as np
import pytest

import adelie_amd as ad
from adelie_amd.sklearn import GroupElasticNet

sklearn_datasets = pytest.importorskip("sklearn.datasets")
from sklearn.model_selection import train_test_split  # noqa: E402

REF_SCORE = 0.45673736460682524            # sklearn_api.ipynb cell 21
REF_CONFUSION = [[28, 16], [0, 70]]        # cell 29


def test_validation_and_unfitted_errors():
    with pytest.raises(ValueError, match="Unknown solver: foo"):
        GroupElasticNet(solver="foo").fit(None, None)
    with pytest.raises(ValueError, match="Unknown family: bar"):
        GroupElasticNet(family="bar").fit(None, None)
    m = GroupElasticNet()
    with pytest.raises(RuntimeError, match="has not been fitted"):
        m.predict(np.zeros((2, 2)))
    with pytest.raises(RuntimeError, match="has not been fitted"):
        GroupElasticNet(family="binomial").predict_proba(np.zeros((2, 2)))
    assert m.get_params() == {"family": "gaussian", "solver": "grpnet"}   # BaseEstimator plumbing (clone, grid search)
    m.set_params(family="poisson")
    assert m.family == "poisson"


def _replay(dense):
    out = {}
    np.random.seed(42)
    data = sklearn_datasets.load_diabetes()
    X_train, X_test, y_train, y_test = train_test_split(data.data, data.target, test_size=0.2)
    X_train, X_test = np.asfortranarray(X_train), np.asfortranarray(X_test)
    m = GroupElasticNet()                                                       # cells 8-12
    m.fit(dense(X_train), y_train)
    out["path"] = (m.coef_.shape, m.intercept_.shape, m.lambda_.shape, m.predict(dense(X_test)).shape)
    with pytest.raises(ValueError, match="only available"):
        m.predict_proba(dense(X_test))
    m = GroupElasticNet(solver="cv_grpnet")                                     # cells 15-21
    m.fit(dense(X_train), y_train, min_ratio=1e-3)
    out["cv_shapes"] = (m.coef_.shape, m.intercept_.shape, m.lambda_.shape)
    out["score"] = m.score(dense(X_test), y_test)
    data = sklearn_datasets.load_breast_cancer()                                # cells 25-29
    X_train, X_test, y_train, y_test = train_test_split(data.data, data.target, test_size=0.2)
    X_train, X_test = np.asfortranarray(X_train), np.asfortranarray(X_test)
    m = GroupElasticNet(solver="cv_grpnet", family="binomial")
    m.fit(dense(X_train), y_train.astype(np.float64))
    pr = m.predict_proba(dense(X_test))
    yhat = m.predict(dense(X_test))
    out["proba_ok"] = pr.shape == (114, 2) and np.allclose(pr.sum(axis=1), 1) and np.array_equal(yhat, np.argmax(pr, axis=1))
    out["confusion"] = [[int(np.sum((y_test == a) & (yhat == b))) for b in (0, 1)] for a in (0, 1)]
    return out


def _check(out):
    assert out["path"] == ((100, 10), (100,), (100,), (100, 89))
    assert out["cv_shapes"] == ((1, 10), (1,), (1,))
    assert abs(out["score"] - REF_SCORE) < 1e-9, out["score"]
    assert out["proba_ok"]
    assert out["confusion"] == REF_CONFUSION


def test_estimator_on_oracle_reproduces_reference_notebook(oracle):
    _check(_replay(oracle.dense))


@pytest.mark.gpu
def test_estimator_on_hip_reproduces_reference_notebook(hip):
    _check(_replay(ad.matrix.dense))


@pytest.mark.gpu
def test_estimator_accepts_numpy_and_multi_families(hip):
    rng = np.random.RandomState(0)
    n, p, K = 200, 30, 3
    X = np.asfortranarray(rng.normal(size=(n, p)))
    Y = X[:, :3] @ rng.normal(size=(3, K)) + 0.3 * rng.normal(size=(n, K))
    m = GroupElasticNet(family="multigaussian").fit(X, Y, progress_bar=False)    # ndarray in, as in the notebook
    L = len(m.lambda_)
    assert m.coef_.shape == (L, p * K) and m.intercept_.shape == (L, K)
    assert m.predict(X).shape == (L, n, K)
    lab = np.argmax(Y, axis=1)
    m = GroupElasticNet(family="multinomial").fit(X, np.eye(K)[lab], progress_bar=False)
    pr = m.predict_proba(X)
    assert pr.shape == (len(m.lambda_), n, K) and np.allclose(pr.sum(axis=-1), 1)
    acc = np.mean(m.predict(X)[-1] == lab)
    assert acc > 0.8, acc
    cnt = rng.poisson(np.exp(0.3 * X[:, 0])).astype(float)
    m = GroupElasticNet(family="poisson").fit(X, cnt, progress_bar=False)
    assert m.predict(X).shape == (len(m.lambda_), n)
