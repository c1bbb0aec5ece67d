"""The oracle's design operations against numpy — the checks of the reference's run_naive
(tests/test_matrix.py:251-409; tolerances 1e-14 relative f64 / 1e-4 f32, :258)."""
import numpy as np
import pytest
from scipy.sparse import random as sprandom

from matrix_checks import run_naive


@pytest.mark.parametrize("dtype,order", [(np.float64, "F"), (np.float64, "C"), (np.float32, "F"), (np.float32, "C")])
@pytest.mark.parametrize("n,p", [(2, 2), (100, 20), (20, 100), (257, 37)])
def test_oracle_dense(oracle, n, p, dtype, order):
    rng = np.random.RandomState(0)
    X = np.asarray(rng.normal(size=(n, p)), dtype=dtype, order=order)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        run_naive(oracle.dense(X, n_threads=3), X, dtype)


def test_oracle_snp(oracle):
    rng = np.random.RandomState(1)
    n, p = 300, 17
    calldata = rng.choice([0, 1, 2, -9], size=(n, p), p=[0.6, 0.25, 0.05, 0.1]).astype(np.int8)
    Xs = oracle.snp_calldata(calldata)
    import adelie_amd as ad
    imp = ad.matrix.compute_impute(calldata)
    Xd = np.where(calldata < 0, imp[None], calldata).astype(np.float64)
    run_naive(Xs, np.asfortranarray(Xd), np.float64)
