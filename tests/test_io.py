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


@pytest.mark.parametrize("n,p", [(1, 1), (7, 3), (64, 5), (1001, 17)])
def test_bed_codec_roundtrip(tmp_path, n, p):
    """PLINK 1 .bed host codec (the checker of matrix.snp_bed): magic, record stride ceil(n/4), field code points
    (00 = 2 copies of A1, 01 = missing, 10 = 1, 11 = 0, low bits first), zero padding of the last byte."""
    rng = np.random.RandomState(n + p)
    cd = rng.choice(np.array([0, 1, 2, -9], dtype=np.int8), size=(n, p), p=[0.6, 0.2, 0.1, 0.1])
    path = str(tmp_path / "x.bed")
    nbytes = ad.io.write_bed(path, cd)
    assert nbytes == 3 + ((n + 3) // 4) * p
    raw = np.fromfile(path, dtype=np.uint8)
    assert bytes(raw[:3]) == bytes([0x6C, 0x1B, 0x01])
    # first sample of the first SNP sits in the two low bits of the first record byte
    want = {2: 0b00, -9: 0b01, 1: 0b10, 0: 0b11}[int(cd[0, 0])]
    assert (raw[3] & 3) == want
    back = ad.io.read_bed(path, n)
    assert back.dtype == np.int8 and back.shape == (n, p)
    np.testing.assert_array_equal(back, cd)


def test_plink_fileset_dims(tmp_path):
    rng = np.random.RandomState(3)
    cd = rng.choice(np.array([0, 1, 2, -9], dtype=np.int8), size=(37, 11), p=[0.6, 0.2, 0.1, 0.1])
    prefix = str(tmp_path / "toy")
    ad.io.write_plink(prefix, cd)
    assert ad.io.plink_dims(prefix) == (37, 11)
    back = ad.io.read_bed(prefix + ".bed", 37, 11)
    assert np.array_equal(back, cd)
