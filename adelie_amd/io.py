"""``.snpdat`` IO — mirrors ``adelie.io.snp_unphased`` (reference ``adelie/io.py:114-194``,
``adelie_core/io/io_snp_unphased.hpp:136-275``, ``.ipp:10-303``, ``io_snp_base.ipp:21-85``).

File layout (little endian, reference ``io_snp_unphased.ipp:117-135,225-274``)::

    [u8 is_big_endian][u64 n][u64 p][u64 nnz[p]][u64 nnm[p]][f64 impute[p]][u64 outer[p+1]]
    per column j at outer[j]:  [u64 off_cat0, off_cat1, off_cat2]   (offsets relative to the column start)
      per category c in (0 = missing, 1, 2) at off_catc:  [u32 n_chunks]
        per non-empty 256-row chunk:  [u32 chunk_idx][u8 nnz-1][u8 row_in_chunk * nnz]

The codec below is integer/byte work done once on the host in numpy; the device design keeps a dense
2-bit-per-call layout built from it (``adelie_hip_design_create_snp_unphased``).
"""
import os

import numpy as np

CHUNK = 256


def _encode(calldata, impute):
    calldata = np.asarray(calldata)
    n, p = calldata.shape
    cols = []
    nnz = np.zeros(p, dtype=np.uint64)
    nnm = np.zeros(p, dtype=np.uint64)
    for j in range(p):
        c = calldata[:, j]
        nnz[j] = np.count_nonzero(c != 0)
        nnm[j] = np.count_nonzero(c >= 0)
        cats = [np.flatnonzero(c < 0), np.flatnonzero(c == 1), np.flatnonzero(c == 2)]
        blobs = []
        for rows in cats:
            chunk_ids = rows // CHUNK
            uniq, starts, counts = np.unique(chunk_ids, return_index=True, return_counts=True)
            parts = [np.uint32(len(uniq)).tobytes()]
            for u, s, k in zip(uniq, starts, counts):
                parts.append(np.uint32(u).tobytes())
                parts.append(np.uint8(k - 1).tobytes())
                parts.append((rows[s:s + k] % CHUNK).astype(np.uint8).tobytes())
            blobs.append(b"".join(parts))
        offs = np.array([24, 24 + len(blobs[0]), 24 + len(blobs[0]) + len(blobs[1])], dtype=np.uint64)
        cols.append(offs.tobytes() + b"".join(blobs))
    header_size = 1 + 8 + 8 + 8 * p + 8 * p + 8 * p + 8 * (p + 1)
    outer = np.zeros(p + 1, dtype=np.uint64)
    outer[0] = header_size
    for j in range(p):
        outer[j + 1] = outer[j] + np.uint64(len(cols[j]))
    head = (np.uint8(0 if np.little_endian else 1).tobytes() + np.uint64(n).tobytes() + np.uint64(p).tobytes() + nnz.tobytes() + nnm.tobytes()
            + np.asarray(impute, dtype=np.float64).tobytes() + outer.tobytes())
    return head + b"".join(cols)


class snp_unphased:
    """IO handler for the ``.snpdat`` unphased format (reference ``adelie.io.snp_unphased``)."""

    def __init__(self, filename: str, read_mode: str = "file"):
        if read_mode not in ("file", "mmap"):
            raise RuntimeError("adelie_core: read_mode must be 'file' or 'mmap'.")
        self._filename = filename
        self._read_mode = read_mode
        self._buffer = None

    # -- properties (py_io.cpp:53-126) -----------------------------------------------------
    @property
    def is_read(self):
        return self._buffer is not None

    def _need(self):
        if not self.is_read:
            raise RuntimeError("adelie_core: File is not read yet. Call read() first.")

    @property
    def endian(self):
        self._need()
        return bool(self._buffer[0])

    @property
    def rows(self):
        self._need()
        return int(self._buffer[1:9].view(np.uint64)[0])

    @property
    def snps(self):
        self._need()
        return int(self._buffer[9:17].view(np.uint64)[0])

    cols = snps

    def _vec(self, k, dtype):
        p = self.snps
        off = 17 + 8 * p * k
        return self._buffer[off:off + 8 * p].view(dtype)

    @property
    def nnz(self):
        return self._vec(0, np.uint64).copy()

    @property
    def nnm(self):
        return self._vec(1, np.uint64).copy()

    @property
    def impute(self):
        return self._vec(2, np.float64).copy()

    @property
    def outer(self):
        p = self.snps
        off = 17 + 24 * p
        return self._buffer[off:off + 8 * (p + 1)].view(np.uint64)

    # -- read / write ----------------------------------------------------------------------
    def read(self):
        """Reads (or memory-maps) the whole file image; returns the number of bytes
        (reference ``io_snp_base.ipp:21-85``, incl. the endianness check)."""
        if self._read_mode == "mmap":
            buf = np.memmap(self._filename, dtype=np.uint8, mode="r")
        else:
            buf = np.fromfile(self._filename, dtype=np.uint8)
        if buf.size < 17:
            raise RuntimeError("adelie_core: file is too small to be a .snpdat file.")
        if bool(buf[0]) != (not np.little_endian):  # byte 0 = is_big_endian() of the writer
            raise RuntimeError("adelie_core: Endianness is inconsistent! Regenerate the file on this machine.")
        self._buffer = np.ascontiguousarray(buf)
        return int(buf.size)

    def write(self, calldata: np.ndarray, impute_method: str = "mean", impute: np.ndarray = None, n_threads: int = 1):
        """Writes an int8 ``(n, p)`` calldata matrix (negative = missing) (reference ``io_snp_unphased.ipp:70-303``).
        Returns ``(bytes_written, benchmark)``."""
        calldata = np.asarray(calldata, dtype=np.int8)
        if calldata.ndim != 2:
            raise RuntimeError("adelie_core: calldata must be 2-dimensional.")
        if np.any(calldata > 2):
            raise RuntimeError("adelie_core: Detected a value greater than 2.")
        if impute_method == "mean":
            valid = calldata >= 0
            impute = np.where(valid, calldata, 0).sum(axis=0) / np.maximum(valid.sum(axis=0), 1)
        elif impute_method == "user":
            if impute is None or len(impute) != calldata.shape[1]:
                raise RuntimeError("adelie_core: impute must be (p,) when impute_method == 'user'.")
        else:
            raise RuntimeError("adelie_core: Unexpected impute method: " + str(impute_method))
        blob = _encode(calldata, impute)
        with open(self._filename, "wb") as f:
            f.write(blob)
        return len(blob), {}

    def to_dense(self, n_threads: int = 1):
        """Dense int8 ``(n, p)`` F-ordered matrix with missing = -9 (reference ``io_snp_unphased.ipp:43-68``)."""
        self._need()
        n, p = self.rows, self.snps
        out = np.zeros((n, p), dtype=np.int8, order="F")
        buf = self._buffer
        outer = self.outer
        vals = (-9, 1, 2)
        for j in range(p):
            base = int(outer[j])
            offs = buf[base:base + 24].view(np.uint64)
            for c in range(3):
                pos = base + int(offs[c])
                n_chunks = int(buf[pos:pos + 4].view(np.uint32)[0])
                pos += 4
                for _ in range(n_chunks):
                    cidx = int(buf[pos:pos + 4].view(np.uint32)[0])
                    k = int(buf[pos + 4]) + 1
                    rows = buf[pos + 5:pos + 5 + k].astype(np.int64) + cidx * CHUNK
                    out[rows, j] = vals[c]
                    pos += 5 + k
        return out


# ---- PLINK 1 .bed (SNP-major) -------------------------------------------------------------------------------------------
_BED_MAGIC = bytes([0x6C, 0x1B, 0x01])
# count of allele A1 -> 2-bit field (low bits first in the byte): 2 -> 00, missing -> 01, 1 -> 10, 0 -> 11
_BED_FIELD = {2: 0b00, -9: 0b01, 1: 0b10, 0: 0b11}


def write_bed(path, calldata):
    """Writes an int8 ``(n, p)`` calldata matrix (A1 counts 0/1/2, negative = missing) as a PLINK 1 ``.bed`` image
    (SNP-major).  Returns the number of bytes written.  Host-side reference codec for ``matrix.snp_bed``."""
    cd = np.asarray(calldata)
    n, p = cd.shape
    field = np.full(cd.shape, _BED_FIELD[-9], dtype=np.uint8)
    for v, f in _BED_FIELD.items():
        if v >= 0:
            field[cd == v] = f
    stride = (n + 3) // 4
    pad = np.zeros((stride * 4, p), dtype=np.uint8)
    pad[:n] = field
    q = pad.reshape(stride, 4, p)
    rec = (q[:, 0] | (q[:, 1] << 2) | (q[:, 2] << 4) | (q[:, 3] << 6)).astype(np.uint8)  # (stride, p)
    image = _BED_MAGIC + np.ascontiguousarray(rec.T).tobytes()
    with open(path, "wb") as f:
        f.write(image)
    return len(image)


def read_bed(path, n, p=None):
    """Decodes a PLINK 1 ``.bed`` image into int8 ``(n, p)`` calldata (A1 counts, -9 = missing) on the host."""
    buf = np.fromfile(path, dtype=np.uint8) if isinstance(path, str) else np.frombuffer(path, dtype=np.uint8)
    if buf.size < 3 or bytes(buf[:3]) != _BED_MAGIC:
        raise RuntimeError("adelie_core: not a SNP-major PLINK .bed image.")
    stride = (n + 3) // 4
    if p is None:
        p = (buf.size - 3) // stride
    rec = buf[3:3 + stride * p].reshape(p, stride).T  # (stride, p)
    fields = np.stack([(rec >> (2 * k)) & 3 for k in range(4)], axis=1).reshape(stride * 4, p)[:n]
    lut = np.array([2, -9, 1, 0], dtype=np.int8)
    return np.asfortranarray(lut[fields])


def plink_dims(prefix):
    """``(n, p)`` of a PLINK 1 fileset ``prefix.{bed,bim,fam}``: one line per sample in ``.fam``, one per variant in ``.bim``."""
    def count(path):
        with open(path, "rb") as f:
            return sum(1 for line in f if line.strip())
    return count(prefix + ".fam"), count(prefix + ".bim")


def write_plink(prefix, calldata):
    """Writes ``prefix.bed`` (see :func:`write_bed`) with minimal ``.fam`` / ``.bim`` companions (test / demo helper)."""
    calldata = np.asarray(calldata)
    n, p = calldata.shape
    nbytes = write_bed(prefix + ".bed", calldata)
    with open(prefix + ".fam", "w") as f:
        for i in range(n):
            f.write(f"F{i} I{i} 0 0 0 -9\n")
    with open(prefix + ".bim", "w") as f:
        for j in range(p):
            f.write(f"1 rs{j} 0 {j + 1} A C\n")
    return nbytes
