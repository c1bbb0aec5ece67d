"""HIP design operations through the C ABI against numpy: the reference's run_naive check list
(tests/test_matrix.py:251-409,529-557,711-752) on dense f32/f64, C/F order, ragged sizes, adopted device tensors and
the 2-bit SNP design."""
import warnings

import numpy as np
import pytest

import adelie_amd as ad
from matrix_checks import run_naive

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,order", [(np.float64, "F"), (np.float64, "C"), (np.float32, "F"), (np.float32, "C")])
@pytest.mark.parametrize("n,p", [(2, 2), (100, 20), (20, 100), (257, 37), (1025, 3), (5000, 70)])
def test_dense(hip, n, p, dtype, order):
    rng = np.random.RandomState(0)
    X = np.asarray(rng.normal(size=(n, p)), dtype=dtype, order=order)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        run_naive(ad.matrix.dense(X), X, dtype)


@pytest.mark.parametrize("n", [64, 257, 4099])  # 257 / 4099: columns not 16-byte aligned -> scalar-load variants
def test_adopted_device_tensor(hip, n):
    import torch

    p = 33
    t = torch.randn((p, n), dtype=torch.float64, device="cuda").t()  # (n, p) column-major view
    X = t.cpu().numpy()
    run_naive(ad.matrix.dense(t), np.asfortranarray(X), np.float64)
    tc = torch.randn((n, p), dtype=torch.float32, device="cuda")  # row-major: transposed copy on device
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        run_naive(ad.matrix.dense(tc), tc.cpu().numpy(), np.float32)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,p", [(300, 17), (1023, 40), (8, 3)])
def test_snp_calldata(hip, n, p, dtype):
    rng = np.random.RandomState(1)
    calldata = rng.choice([0, 1, 2, -9], size=(n, p), p=[0.6, 0.25, 0.05, 0.1]).astype(np.int8)
    imp = ad.matrix.compute_impute(calldata)
    Xd = np.asfortranarray(np.where(calldata < 0, imp[None], calldata).astype(dtype))
    run_naive(ad.matrix.snp_calldata(calldata, dtype=dtype), Xd, dtype)


def test_snp_from_snpdat_file(hip, tmp_path):
    rng = np.random.RandomState(2)
    n, p = 700, 21
    calldata = rng.choice([0, 1, 2, -9], size=(n, p), p=[0.6, 0.25, 0.05, 0.1]).astype(np.int8)
    calldata[:, 3] = 0  # an all-zero column
    calldata[:, 4] = -9  # an all-missing column
    h = ad.io.snp_unphased(str(tmp_path / "x.snpdat"))
    h.write(calldata)
    h.read()
    X = ad.matrix.snp_unphased(h)
    imp = h.impute
    Xd = np.asfortranarray(np.where(calldata < 0, imp[None], calldata).astype(np.float64))
    run_naive(X, Xd, np.float64)


def test_inconsistent_inputs_raise(hip):
    X = ad.matrix.dense(np.asfortranarray(np.zeros((10, 4))))
    with pytest.raises(RuntimeError, match="cmul"):
        X.cmul(4, np.zeros(10), np.zeros(10))
    with pytest.raises(RuntimeError, match="bmul"):
        X.bmul(2, 3, np.zeros(10), np.zeros(10), np.zeros(3))
    with pytest.raises(RuntimeError, match="mul"):
        X.mul(np.zeros(9), np.zeros(9), np.zeros(4))
