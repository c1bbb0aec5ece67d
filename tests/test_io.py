""".snpdat codec (reference tests/test_io.py:15-70): write -> read -> properties -> to_dense round trip."""
import numpy as np
import pytest

import adelie_amd as ad


@pytest.mark.parametrize("n,p", [(1, 1), (200, 13), (300, 5), (1000, 3)])
@pytest.mark.parametrize("read_mode", ["file", "mmap"])
def test_snp_unphased_roundtrip(tmp_path, n, p, read_mode):
    rng = np.random.RandomState(0)
    calldata = rng.choice([0, 1, 2, -9], size=(n, p), p=[0.5, 0.3, 0.1, 0.1]).astype(np.int8)
    f = str(tmp_path / "t.snpdat")
    h = ad.io.snp_unphased(f, read_mode=read_mode)
    assert not h.is_read
    w, _ = h.write(calldata)
    r = h.read()
    assert w == r and h.is_read
    assert h.rows == n and h.snps == p and h.cols == p
    assert h.endian == (not np.little_endian)
    assert np.array_equal(h.nnz, np.sum(calldata != 0, axis=0))
    assert np.array_equal(h.nnm, np.sum(calldata >= 0, axis=0))
    valid = calldata >= 0
    assert np.allclose(h.impute, np.where(valid, calldata, 0).sum(0) / np.maximum(valid.sum(0), 1))
    dense = h.to_dense()
    assert dense.dtype == np.int8 and dense.flags.f_contiguous
    assert np.array_equal(dense, np.where(calldata < 0, -9, calldata))


def test_snp_unphased_rejects_bad_values(tmp_path):
    h = ad.io.snp_unphased(str(tmp_path / "b.snpdat"))
    with pytest.raises(RuntimeError, match="greater than"):
        h.write(np.array([[3]], dtype=np.int8))
    with pytest.raises(RuntimeError, match="not read"):
        h.rows
