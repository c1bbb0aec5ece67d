"""Design-matrix plugin surface — mirrors ``adelie.matrix`` for the grpnet hot path.

Reference: ``adelie/matrix.py:52-191`` (Python sugar ``@``, ``.T``), ``:549-680`` (``dense``),
``:1245-1298`` (``snp_unphased``) and the C++ virtual interface
``adelie_core/matrix/matrix_naive_base.hpp:18-144``.

A matrix object here is a handle to a design that is *resident in MI355X HBM*; every method is
one call through the C ABI of ``libadelie_hip.so`` (``include/adelie_hip.h``) with host numpy
vectors in/out, exactly the argument convention of the reference's pybind methods
(pre-allocated outputs written in place).
"""
import warnings

import numpy as np
from scipy.sparse import csc_matrix, csr_matrix

from . import _abi


class MatrixNaiveBase:
    """Common base of every design handle (role of ``MatrixNaiveBase{32,64}``)."""

    dtype = None


class MatrixNaiveBase64(MatrixNaiveBase):
    dtype = np.float64


class MatrixNaiveBase32(MatrixNaiveBase):
    dtype = np.float32


class MatrixCovBase:
    """Base of the covariance-method matrices (role of ``MatrixCovBase{32,64}``, ``matrix_cov_base.hpp:20-110``)."""

    dtype = None


class MatrixCovBase64(MatrixCovBase):
    dtype = np.float64


class MatrixCovBase32(MatrixCovBase):
    dtype = np.float32


def _as(v, dtype):
    return np.ascontiguousarray(v, dtype=dtype)


class PyMatrixNaiveTranspose:
    """``X.T @ v`` (reference ``matrix.py:52-76``)."""

    def __init__(self, mat):
        self._mat = mat
        self.T = mat

    def __matmul__(self, v):
        """``X.T @ v`` for a vector (-> ``(p,)``) or an ``(n, m)`` array (-> ``(p, m)``).  Several right-hand sides go
        to the device together when the design has the batched sweep (X read once per eight vectors)."""
        X = self._mat
        rhs = np.asarray(v, dtype=X.dtype)
        if rhs.ndim not in (1, 2):
            raise ValueError("Right argument must be either 1 or 2-dimensional.")
        n, p = X.shape
        cols = rhs.reshape(n, -1).T  # (m, n): one right-hand side per row
        if rhs.ndim == 2 and hasattr(X, "mul_batch"):
            res = np.asarray(X.mul_batch(np.ascontiguousarray(cols)))
        else:
            unit = np.ones(n, dtype=X.dtype)
            res = np.empty((cols.shape[0], p), dtype=X.dtype)
            for src, dst in zip(cols, res):
                X.mul(np.ascontiguousarray(src), unit, dst)
        return res[0] if rhs.ndim == 1 else res.T


class _NativeMatrix:
    """Methods shared by every native design handle; ``_backend`` is an ``_abi.Backend``."""

    def _init_native(self, backend, handle, n_threads):
        self._backend = backend
        self._handle = handle
        self._n_threads = n_threads
        self._rows = int(backend.fn("design_rows")(handle))
        self._cols = int(backend.fn("design_cols")(handle))

    @property
    def T(self):
        """``X.T @ v`` sugar (reference ``matrix.py:79-100``).  Made on demand: a stored wrapper would tie the design into a
        reference cycle, and its device memory (gigabytes) would wait for the cycle collector instead of going when the last
        user reference goes."""
        return PyMatrixNaiveTranspose(self)

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None:
            try:
                self._backend.fn("design_destroy")(h)
            except Exception:
                pass
            self._handle = None

    # -- MatrixNaiveBase virtuals (matrix_naive_base.hpp:18-144) -----------------------------
    def rows(self):
        return self._rows

    def cols(self):
        return self._cols

    def glm_path_losses(self, glm_kind, betas, intercepts, offsets, y, weights_a, weights_b):
        """Losses of ``eta_l = X beta_l + intercepts[l] + offsets`` under two weight vectors, all on the device
        (``adelie_hip_design_glm_path_losses``).  ``betas`` is a CSR ``(L, p)`` matrix.  Returns two ``(L,)`` arrays."""
        dt = self.dtype
        betas = betas.tocsr()
        L = betas.shape[0]
        indptr = np.ascontiguousarray(betas.indptr, dtype=np.int64)
        indices = np.ascontiguousarray(betas.indices, dtype=np.int64)
        values = np.ascontiguousarray(betas.data, dtype=dt)
        vecs = [np.ascontiguousarray(v, dtype=dt) for v in (intercepts, offsets, y, weights_a, weights_b)]
        out = np.empty(2 * L, dtype=np.float64)
        self._backend.check(self._backend.fn("design_glm_path_losses")(
            self._handle, int(glm_kind), L, indptr.ctypes.data, indices.ctypes.data, values.ctypes.data,
            *[v.ctypes.data for v in vecs], out.ctypes.data))
        return out[:L], out[L:]

    def multi_path_losses(self, glm_kind, K, betas, intercepts, offsets, y, weights_a, weights_b):
        """:meth:`glm_path_losses` for multi-response fits on this (base) design: ``betas`` is CSR ``(L, p*K)`` over the view
        columns ``feature*K + response``, ``intercepts`` ``(L, K)``, ``offsets`` / ``y`` ``(n, K)``
        (``adelie_hip_design_multi_path_losses``).  Returns two ``(L,)`` arrays."""
        dt = self.dtype
        betas = betas.tocsr()
        L = betas.shape[0]
        self._chk(betas.shape[1] == self._cols * K, "multi_path_losses() is given inconsistent inputs!")
        indptr = np.ascontiguousarray(betas.indptr, dtype=np.int64)
        indices = np.ascontiguousarray(betas.indices, dtype=np.int64)
        values = np.ascontiguousarray(betas.data, dtype=dt)
        icpt = np.ascontiguousarray(np.asarray(intercepts, dtype=dt).reshape(L, K))
        off = np.ascontiguousarray(np.asarray(offsets, dtype=dt).reshape(self._rows, K))
        yy = np.ascontiguousarray(np.asarray(y, dtype=dt).reshape(self._rows, K))
        wa, wb = (np.ascontiguousarray(v, dtype=dt) for v in (weights_a, weights_b))
        out = np.empty(2 * L, dtype=np.float64)
        self._backend.check(self._backend.fn("design_multi_path_losses")(
            self._handle, int(glm_kind), int(K), L, indptr.ctypes.data, indices.ctypes.data, values.ctypes.data,
            icpt.ctypes.data, off.ctypes.data, yy.ctypes.data, wa.ctypes.data, wb.ctypes.data, out.ctypes.data))
        return out[:L], out[L:]

    def alias(self):
        """A second handle on the same resident matrix with its own stream (``adelie_hip_design_alias``): lets independent
        solves run concurrently from different threads.  Keeps this design alive."""
        handle = _abi.C.c_void_p()
        self._backend.check(self._backend.fn("design_alias")(self._handle, handle))
        out = _wrap(self._backend, handle, self.dtype, self._n_threads, kind=getattr(self, "_kind", None))
        out._alias_of = self
        if hasattr(self, "_scipy"):
            out._scipy, out._device = self._scipy, getattr(self, "_device", 0)
            if getattr(self, "_std", None) is not None:
                out._std = self._std
        return out

    def batch_stats(self):
        """Shared sweeps of concurrent solves on this design and its aliases since it was created
        (``adelie_hip_design_batch_stats``): ``{"launches", "vectors", "ms"}``."""
        out = (_abi.C.c_double * 3)()
        self._backend.check(self._backend.fn("design_batch_stats")(self._handle, out))
        return {"launches": int(out[0]), "vectors": int(out[1]), "ms": float(out[2])}

    def impute(self):
        """The ``(p,)`` impute values of an SNP design (what a missing call contributes)."""
        out = np.empty(self._cols, dtype=np.float64)
        self._backend.check(self._backend.fn("design_impute")(self._handle, out.ctypes.data))
        return out

    def __getitem__(self, key):
        """``mat[rows]`` / ``mat[:, cols]`` / ``mat[rows, cols]`` -> :func:`subset` (reference ``matrix.py:102-191``)."""
        if isinstance(key, tuple):
            if len(key) != 2:
                raise IndexError("too many indices for a 2-dimensional matrix.")
            rk, ck = key
        else:
            rk, ck = key, slice(None)

        def idx(k, size):
            if isinstance(k, slice):
                return None if k == slice(None) else np.arange(size)[k]
            return np.atleast_1d(np.asarray(k))

        r, c = idx(rk, self._rows), idx(ck, self._cols)
        out = self
        if c is not None:
            out = subset(out, c, axis=1, n_threads=self._n_threads)
        if r is not None:
            out = subset(out, r, axis=0, n_threads=self._n_threads)
        return out

    @property
    def ndim(self):
        return 2

    @property
    def shape(self):
        return (self._rows, self._cols)

    def _chk(self, cond, msg):
        if not cond:
            raise RuntimeError(msg)

    def cmul(self, j, v, weights):
        self._chk(0 <= j < self._cols and len(v) == self._rows and len(weights) == self._rows,
                  "cmul() is given inconsistent inputs!")
        v = _as(v, self.dtype)
        w = _as(weights, self.dtype)
        out = _abi.C.c_double()
        self._backend.check(self._backend.fn("design_cmul")(self._handle, j, v.ctypes.data, w.ctypes.data, out))
        return self.dtype(out.value)

    cmul_safe = cmul

    def ctmul(self, j, v, out):
        self._chk(0 <= j < self._cols and len(out) == self._rows, "ctmul() is given inconsistent inputs!")
        o = self._out(out)
        self._backend.check(self._backend.fn("design_ctmul")(self._handle, j, float(v), o.ctypes.data))
        self._writeback(o, out)

    def bmul(self, j, q, v, weights, out):
        self._chk(0 <= j <= self._cols - q and len(v) == self._rows and len(weights) == self._rows and len(out) == q,
                  "bmul() is given inconsistent inputs!")
        v = _as(v, self.dtype)
        w = _as(weights, self.dtype)
        o = self._out(out)
        self._backend.check(self._backend.fn("design_bmul")(self._handle, j, q, v.ctypes.data, w.ctypes.data, o.ctypes.data))
        self._writeback(o, out)

    bmul_safe = bmul

    def btmul(self, j, q, v, out):
        self._chk(0 <= j <= self._cols - q and len(v) == q and len(out) == self._rows,
                  "btmul() is given inconsistent inputs!")
        v = _as(v, self.dtype)
        o = self._out(out)
        self._backend.check(self._backend.fn("design_btmul")(self._handle, j, q, v.ctypes.data, o.ctypes.data))
        self._writeback(o, out)

    def mul(self, v, weights, out):
        self._chk(len(v) == self._rows and len(weights) == self._rows and len(out) == self._cols,
                  "mul() is given inconsistent inputs!")
        v = _as(v, self.dtype)
        w = _as(weights, self.dtype)
        o = self._out(out)
        self._backend.check(self._backend.fn("design_mul")(self._handle, v.ctypes.data, w.ctypes.data, o.ctypes.data))
        self._writeback(o, out)

    def mul_batch(self, V):
        """``V @ X`` for an (L, n) array of vectors in one call (used by :func:`adelie_amd.diagnostic.gradients`; the
        reference loops ``X.mul`` per vector, ``diagnostic.py:377-386``)."""
        V = np.ascontiguousarray(V, dtype=self.dtype)
        self._chk(V.ndim == 2 and V.shape[1] == self._rows, "mul_batch() is given inconsistent inputs!")
        out = np.empty((V.shape[0], self._cols), dtype=self.dtype)
        self._backend.check(self._backend.fn("design_mul_batch")(self._handle, V.ctypes.data, V.shape[0], out.ctypes.data))
        return out

    def cov(self, j, q, sqrt_weights, out):
        self._chk(0 <= j <= self._cols - q and len(sqrt_weights) == self._rows and out.shape == (q, q),
                  "cov() is given inconsistent inputs!")
        sw = _as(sqrt_weights, self.dtype)
        o = np.empty(q * q, dtype=self.dtype)
        self._backend.check(self._backend.fn("design_cov")(self._handle, j, q, sw.ctypes.data, o.ctypes.data))
        out[...] = o.reshape(q, q, order="F")

    def sq_mul(self, weights, out):
        self._chk(len(weights) == self._rows and len(out) == self._cols, "sq_mul() is given inconsistent inputs!")
        w = _as(weights, self.dtype)
        o = self._out(out)
        self._backend.check(self._backend.fn("design_sq_mul")(self._handle, w.ctypes.data, o.ctypes.data))
        self._writeback(o, out)

    def sp_tmul(self, v, out):
        v = csr_matrix(v)
        L = v.shape[0]
        self._chk(v.shape[1] == self._cols and out.shape == (L, self._rows), "sp_tmul() is given inconsistent inputs!")
        indptr = np.ascontiguousarray(v.indptr, dtype=np.int64)
        indices = np.ascontiguousarray(v.indices, dtype=np.int64)
        values = _as(v.data, self.dtype)
        o = np.empty((L, self._rows), dtype=self.dtype)
        self._backend.check(self._backend.fn("design_sp_tmul")(
            self._handle, L, indptr.ctypes.data, indices.ctypes.data, values.ctypes.data, o.ctypes.data))
        out[...] = o

    def mean(self, weights, out):
        """Default ``MatrixNaiveBase::mean`` (matrix_naive_base.hpp:113-122): ``out = w^T X``."""
        ones = np.ones(self._rows, dtype=self.dtype)
        self.mul(ones, weights, out)

    def var(self, centers, weights, out):
        """Default ``MatrixNaiveBase::var`` (matrix_naive_base.hpp:124-133)."""
        sum_w = np.sum(weights)
        m = np.empty(self._cols, dtype=self.dtype)
        self.mean(weights, m)
        self.sq_mul(weights, out)
        centers = np.asarray(centers, dtype=self.dtype)
        out += centers * (centers * sum_w - 2 * m)

    # -- python sugar (matrix.py:79-191) ---------------------------------------------------
    def __matmul__(self, v):
        """``X @ v``: ``(p,)`` -> ``(n,)``, dense ``(p, m)`` -> ``(n, m)``; a scipy CSR / CSC right-hand side is applied by
        the sparse kernel (``sp_tmul`` on its transpose) in one call."""
        n, p = self.shape
        if isinstance(v, (csr_matrix, csc_matrix)):
            vt = v.tocsr().transpose()
            res = np.empty((vt.shape[0], n), dtype=self.dtype)
            self.sp_tmul(vt, res)
            return res.T
        rhs = np.asarray(v, dtype=self.dtype)
        if rhs.ndim not in (1, 2):
            raise ValueError("Right argument must be either 1 or 2-dimensional.")
        coefs = rhs.reshape(p, -1).T  # (m, p): one coefficient vector per row
        res = np.zeros((coefs.shape[0], n), dtype=self.dtype)
        for src, dst in zip(coefs, res):
            self.btmul(0, p, np.ascontiguousarray(src), dst)
        return res[0] if rhs.ndim == 1 else res.T

    # -- helpers ---------------------------------------------------------------------------
    def _out(self, out):
        if isinstance(out, np.ndarray) and out.dtype == self.dtype and out.flags.c_contiguous:
            return out
        return np.ascontiguousarray(out, dtype=self.dtype)

    @staticmethod
    def _writeback(o, out):
        if o is not out:
            out[...] = o


def _make_class(dtype):
    base = MatrixNaiveBase64 if np.dtype(dtype) == np.float64 else MatrixNaiveBase32

    class _matrix(_NativeMatrix, base):
        pass

    _matrix.dtype = base.dtype
    return _matrix


def _wrap(backend, handle, dtype, n_threads, keep=None, kind=None):
    obj = _make_class(dtype)()
    obj._init_native(backend, handle, n_threads)
    obj._keep = keep
    obj._kind = kind  # "dense" (cv_grpnet batches the sweeps of concurrent folds on these) or "snp" (2-bit calls)
    return obj


class _MultiView(_NativeMatrix):
    """``[1 (x) I_K, X (x) I_K]`` (or ``X (x) I_K``) over a resident base design — what the reference assembles from
    ``matrix.kronecker_eye`` and ``matrix.concatenate`` for multi-response fits (``state.py:1100-1125``).

    The native handle (``adelie_hip_design_create_multi``) is what ``grpnet_solve`` runs on; the matrix plugin methods are
    served here through the *base* design's device kernels, one response at a time
    (``matrix_naive_kronecker_eye.ipp``: column ``j`` = (feature ``j // K``, response ``j % K``), vectors ``(n, K)`` C-order).
    """

    def _init_view(self, base, K, icpt):
        self._base, self._K, self._icpt = base, int(K), 1 if icpt else 0

    def _split(self, j):
        u, l = divmod(int(j), self._K)
        return u - self._icpt, l  # base column (-1: the column of ones), response

    def _mat(self, v):
        return np.asarray(v, dtype=self.dtype).reshape(-1, self._K)

    def cmul(self, j, v, weights):
        self._chk(0 <= j < self._cols and len(v) == self._rows and len(weights) == self._rows,
                  "cmul() is given inconsistent inputs!")
        f, l = self._split(j)
        vl, wl = self._mat(v)[:, l], self._mat(weights)[:, l]
        if f < 0:
            return self.dtype(np.sum(vl * wl))
        return self._base.cmul(f, vl, wl)

    cmul_safe = cmul

    def ctmul(self, j, v, out):
        self._chk(0 <= j < self._cols and len(out) == self._rows, "ctmul() is given inconsistent inputs!")
        f, l = self._split(j)
        O = out.reshape(-1, self._K)
        if f < 0:
            O[:, l] += v
            return
        t = np.zeros(O.shape[0], dtype=self.dtype)
        self._base.ctmul(f, v, t)
        O[:, l] += t

    def bmul(self, j, q, v, weights, out):
        self._chk(0 <= j <= self._cols - q and len(v) == self._rows and len(weights) == self._rows and len(out) == q,
                  "bmul() is given inconsistent inputs!")
        for t in range(q):
            out[t] = self.cmul(j + t, v, weights)

    bmul_safe = bmul

    def btmul(self, j, q, v, out):
        self._chk(0 <= j <= self._cols - q and len(v) == q and len(out) == self._rows,
                  "btmul() is given inconsistent inputs!")
        K, I = self._K, self._icpt
        full = np.zeros(self._cols, dtype=self.dtype)
        full[j:j + q] = v
        B = full.reshape(-1, K)  # (p + I, K)
        O = out.reshape(-1, K)
        pb = self._base.cols()
        for l in range(K):
            if I:
                O[:, l] += B[0, l]
            if np.any(B[I:, l]):
                t = np.zeros(O.shape[0], dtype=self.dtype)
                self._base.btmul(0, pb, np.ascontiguousarray(B[I:, l]), t)
                O[:, l] += t

    def mul(self, v, weights, out):
        self._chk(len(v) == self._rows and len(weights) == self._rows and len(out) == self._cols,
                  "mul() is given inconsistent inputs!")
        K, I = self._K, self._icpt
        V, W = self._mat(v), self._mat(weights)
        O = np.empty((self._base.cols() + I, K), dtype=self.dtype)
        t = np.empty(self._base.cols(), dtype=self.dtype)
        for l in range(K):
            vl, wl = np.ascontiguousarray(V[:, l]), np.ascontiguousarray(W[:, l])
            if I:
                O[0, l] = np.sum(vl * wl)
            self._base.mul(vl, wl, t)
            O[I:, l] = t
        out[...] = O.ravel()

    def cov(self, j, q, sqrt_weights, out):
        self._chk(0 <= j <= self._cols - q and len(sqrt_weights) == self._rows and out.shape == (q, q),
                  "cov() is given inconsistent inputs!")
        SW = self._mat(sqrt_weights)
        out[...] = 0
        fs = [self._split(j + t) for t in range(q)]
        for l in range(self._K):
            idx = [t for t in range(q) if fs[t][1] == l]
            if not idx:
                continue
            swl = np.ascontiguousarray(SW[:, l])
            cols = [fs[t][0] for t in idx]
            if min(cols) < 0:
                self._chk(max(cols) < 0, "cov() across the intercept block and the feature block.")
                for t in idx:
                    out[t, t] = np.sum(swl ** 2)
                continue
            f0, nf = min(cols), max(cols) - min(cols) + 1
            c = np.empty((nf, nf), dtype=self.dtype, order="F")
            self._base.cov(f0, nf, swl, c)
            for a, ta in enumerate(idx):
                for b, tb in enumerate(idx):
                    out[ta, tb] = c[cols[a] - f0, cols[b] - f0]

    def sq_mul(self, weights, out):
        self._chk(len(weights) == self._rows and len(out) == self._cols, "sq_mul() is given inconsistent inputs!")
        K, I = self._K, self._icpt
        W = self._mat(weights)
        O = np.empty((self._base.cols() + I, K), dtype=self.dtype)
        t = np.empty(self._base.cols(), dtype=self.dtype)
        for l in range(K):
            wl = np.ascontiguousarray(W[:, l])
            if I:
                O[0, l] = np.sum(wl)
            self._base.sq_mul(wl, t)
            O[I:, l] = t
        out[...] = O.ravel()

    def sp_tmul(self, v, out):
        v = csr_matrix(v)
        L = v.shape[0]
        self._chk(v.shape[1] == self._cols and out.shape == (L, self._rows), "sp_tmul() is given inconsistent inputs!")
        K, I = self._K, self._icpt
        nb = self._base.rows()
        O = np.zeros((L, nb, K), dtype=self.dtype)
        vc = v.tocsc()
        t = np.empty((L, nb), dtype=self.dtype)
        for l in range(K):
            if I:
                O[:, :, l] += np.asarray(vc[:, l].todense()).reshape(L, 1)
            self._base.sp_tmul(vc[:, I * K + l::K].tocsr(), t)
            O[:, :, l] += t
        out[...] = O.reshape(L, -1)

    def alias(self):
        raise NotImplementedError("adelie_amd: alias() of a multi-response view; alias the base design instead.")


class _StdView(_NativeMatrix):
    """``(Z - 1 c^T) diag(s)^-1`` over a resident dense or 2-bit SNP design ``Z``, nothing materialised — the reference's lazy
    ``MatrixNaiveStandardize`` (``matrix_naive_standardize.ipp``).  The native handle (``adelie_hip_design_create_standardized``)
    serves ``grpnet`` (the solver composes the base design's kernels with the centring / scaling corrections and runs its
    full-Gram engines); the matrix operations below are the base design's plus the same rank-one corrections in numpy, as the
    reference's wrapper does around its wrapped matrix."""

    _is_std_view = True

    def _init_view(self, base, centers, scales):
        self._base = base
        self._c = np.array(centers, dtype=self.dtype)
        self._s = np.array(scales, dtype=self.dtype)
        self._centers, self._scales = self._c.copy(), self._s.copy()

    def _materialize(self):
        return _derived(self._base, None, None, self._c, self._s, self._n_threads)

    def alias(self):
        out = _std_view(self._base.alias(), self._c, self._s, self._n_threads)
        out._alias_of = self
        return out

    def cmul(self, j, v, weights):
        self._chk(0 <= j < self._cols and len(v) == self._rows and len(weights) == self._rows,
                  "cmul() is given inconsistent inputs!")
        vw = np.sum(np.asarray(v, dtype=np.float64) * np.asarray(weights, dtype=np.float64))
        return self.dtype((self._base.cmul(j, v, weights) - self._c[j] * vw) / self._s[j])

    cmul_safe = cmul

    def ctmul(self, j, v, out):
        self._chk(0 <= j < self._cols and len(out) == self._rows, "ctmul() is given inconsistent inputs!")
        t = v / self._s[j]
        self._base.ctmul(j, t, out)
        out -= self.dtype(t * self._c[j])

    def bmul(self, j, q, v, weights, out):
        self._chk(0 <= j <= self._cols - q and len(v) == self._rows and len(weights) == self._rows and len(out) == q,
                  "bmul() is given inconsistent inputs!")
        raw = np.empty(q, dtype=self.dtype)
        self._base.bmul(j, q, v, weights, raw)
        vw = np.sum(np.asarray(v, dtype=np.float64) * np.asarray(weights, dtype=np.float64))
        out[...] = (raw - self._c[j:j + q] * vw) / self._s[j:j + q]

    bmul_safe = bmul

    def btmul(self, j, q, v, out):
        self._chk(0 <= j <= self._cols - q and len(v) == q and len(out) == self._rows,
                  "btmul() is given inconsistent inputs!")
        t = np.asarray(v, dtype=self.dtype) / self._s[j:j + q]
        self._base.btmul(j, q, t, out)
        out -= self.dtype(np.dot(t, self._c[j:j + q]))

    def mul(self, v, weights, out):
        self.bmul(0, self._cols, v, weights, out)

    def mul_batch(self, V):
        V = np.ascontiguousarray(V, dtype=self.dtype)
        raw = self._base.mul_batch(V)
        return ((raw - np.sum(V, axis=1, dtype=np.float64)[:, None] * self._c[None]) / self._s[None]).astype(self.dtype)

    def cov(self, j, q, sqrt_weights, out):
        self._chk(0 <= j <= self._cols - q and len(sqrt_weights) == self._rows and out.shape == (q, q),
                  "cov() is given inconsistent inputs!")
        raw = np.empty((q, q), dtype=self.dtype, order="F")
        self._base.cov(j, q, sqrt_weights, raw)
        w = np.asarray(sqrt_weights, dtype=self.dtype) ** 2
        m = np.empty(q, dtype=self.dtype)
        self._base.bmul(j, q, np.ones(self._rows, dtype=self.dtype), w, m)
        c, s = self._c[j:j + q], self._s[j:j + q]
        k = np.outer(c, m)  # (k + k.T, c c^T and s s^T are symmetric bit for bit, and so is the result)
        out[...] = (raw - (k + k.T) + np.outer(c, c) * np.sum(w, dtype=np.float64)) / np.outer(s, s)

    def sq_mul(self, weights, out):
        self._chk(len(weights) == self._rows and len(out) == self._cols, "sq_mul() is given inconsistent inputs!")
        w = np.asarray(weights, dtype=self.dtype)
        sq = np.empty(self._cols, dtype=self.dtype)
        m = np.empty(self._cols, dtype=self.dtype)
        self._base.sq_mul(w, sq)
        self._base.mul(np.ones(self._rows, dtype=self.dtype), w, m)
        out[...] = (sq - 2 * self._c * m + self._c ** 2 * np.sum(w, dtype=np.float64)) / self._s ** 2

    def sp_tmul(self, v, out):
        v = v.tocsr()
        self._chk(v.shape[1] == self._cols and out.shape == (v.shape[0], self._rows), "sp_tmul() is given inconsistent inputs!")
        scaled = v.multiply(1 / self._s[None]).tocsr().astype(self.dtype)
        self._base.sp_tmul(scaled, out)
        out -= np.asarray(scaled @ self._c, dtype=self.dtype).reshape(-1, 1)

    def glm_path_losses(self, glm_kind, betas, intercepts, offsets, y, weights_a, weights_b):
        # eta = Z (beta / s) + (intercept - sum_j beta_j c_j / s_j): the base design's device path with transformed inputs
        scaled = betas.tocsr().multiply(1 / self._s[None]).tocsr()
        shift = np.asarray(scaled @ self._c).reshape(-1)
        return self._base.glm_path_losses(glm_kind, scaled, np.asarray(intercepts) - shift, offsets, y, weights_a, weights_b)

    def multi_path_losses(self, glm_kind, K, betas, intercepts, offsets, y, weights_a, weights_b):
        betas = betas.tocsr()
        scaled = betas.multiply(1 / np.repeat(self._s, K)[None]).tocsr()
        sel = csr_matrix((np.ones(self._cols * K), (np.arange(self._cols * K), np.arange(self._cols * K) % K)),
                         shape=(self._cols * K, K))
        shift = np.asarray((scaled.multiply(np.repeat(self._c, K)[None]).tocsr() @ sel).todense())
        L = betas.shape[0]
        return self._base.multi_path_losses(glm_kind, K, scaled, np.asarray(intercepts).reshape(L, K) - shift, offsets, y,
                                            weights_a, weights_b)

    def impute(self):
        raise RuntimeError("adelie_amd: impute() belongs to the SNP design under the standardized view (view._base).")


def _std_view(base, centers, scales, n_threads):
    backend = base._backend
    ce = np.ascontiguousarray(centers, dtype=np.float64)
    sc = np.ascontiguousarray(scales, dtype=np.float64)
    handle = _abi.C.c_void_p()
    backend.check(backend.fn("design_create_standardized")(base._handle, ce.ctypes.data, sc.ctypes.data, handle))
    mixin = MatrixNaiveBase64 if np.dtype(base.dtype) == np.float64 else MatrixNaiveBase32

    class _view(_StdView, mixin):
        pass

    _view.dtype = mixin.dtype
    obj = _view()
    obj._init_native(backend, handle, n_threads)
    obj._init_view(base, ce, sc)
    obj._keep = base
    obj._kind = "std"
    return obj


def _multi_view(base, K, intercept):
    if not isinstance(base, _NativeMatrix) or isinstance(base, _MultiView):
        raise RuntimeError("adelie_amd: the multi-response view needs a resident (dense or 2-bit SNP) design as its base.")
    if int(K) < 1:
        raise RuntimeError("adelie_core: K must be >= 1.")
    if _is_kept_sparse(base):  # the K-wide kernels stream dense or 2-bit column slices
        base = _expanded(base)
    if isinstance(base, _StdView):
        base = base._materialize()
    # (a 2-bit SNP base stays 2-bit: the K-wide sweep, panel step and Gram kernels decode a column's calls once for all K
    # responses -- matrix_naive_kronecker_eye.ipp over matrix_naive_snp_unphased.ipp)
    backend = base._backend
    handle = _abi.C.c_void_p()
    backend.check(backend.fn("design_create_multi")(base._handle, int(K), 1 if intercept else 0, handle))
    mixin = MatrixNaiveBase64 if np.dtype(base.dtype) == np.float64 else MatrixNaiveBase32

    class _view(_MultiView, mixin):
        pass

    _view.dtype = mixin.dtype
    obj = _view()
    obj._init_native(backend, handle, base._n_threads)
    obj._init_view(base, K, intercept)
    obj._keep = base
    return obj


def kronecker_eye(mat, K: int, *, n_threads: int = 1):
    """``mat (x) I_K`` as a view of a resident design (reference ``adelie.matrix.kronecker_eye``,
    ``matrix_naive_kronecker_eye.ipp``).  An ``(n, 1)`` array of ones is recognised as the intercept block that
    :func:`concatenate` puts in front of ``kronecker_eye(X, K)``."""
    if isinstance(mat, np.ndarray):
        if mat.ndim == 2 and mat.shape[1] == 1 and np.all(mat == 1):
            return _OnesKron(mat.shape[0], int(K), mat.dtype.type)
        mat = dense(mat, method="naive", n_threads=n_threads)
    return _multi_view(mat, K, False)


class _OnesKron:
    """``1_n (x) I_K`` — only meaningful as the first block of :func:`concatenate`."""

    def __init__(self, n, K, dtype):
        self.n, self.K, self.dtype = n, K, dtype


def concatenate(mats, *, axis: int = 0, n_threads: int = 1):
    """Reference ``adelie.matrix.concatenate`` (``matrix.py:214-310``).

    * ``[kronecker_eye(ones((n, 1)), K), kronecker_eye(X, K)]`` along ``axis=1`` — the combination the multi-response
      solver builds (``state.py:1110-1120``) — becomes the multi-response view of ``X`` (nothing materialised);
    * resident designs (dense, SNP, sparse, derived; ndarrays are uploaded first) are copied side by side into one new
      dense design on the device (``adelie_hip_design_create_concat``); the reference keeps a list of views and dispatches
      every operation to the pieces (``matrix_naive_concatenate.ipp``)."""
    mats = list(mats)
    if (axis == 1 and len(mats) == 2 and isinstance(mats[0], _OnesKron) and isinstance(mats[1], _MultiView)
            and mats[1]._icpt == 0 and mats[0].K == mats[1]._K and mats[0].n == mats[1]._base.rows()):
        return _multi_view(mats[1]._base, mats[1]._K, True)
    if n_threads < 1:
        raise RuntimeError("adelie_core: n_threads must be >= 1.")
    if axis not in (0, 1):
        raise ValueError("axis must be 0 or 1.")
    if len(mats) == 0:
        raise RuntimeError("mats must be non-empty.")
    mats = [dense(m, n_threads=n_threads) if isinstance(m, np.ndarray) else m for m in mats]
    if all(_is_kept_sparse(m) and getattr(m, "_std", None) is None for m in mats):  # sparse pieces stay sparse (stacked on the host)
        import scipy.sparse as _sp

        if len({np.dtype(m.dtype) for m in mats}) != 1:
            raise RuntimeError("All matrices must have the same dtype.")
        if axis == 1 and len({m.rows() for m in mats}) != 1:
            raise RuntimeError("All matrices must have the same number of rows.")
        if axis == 0 and len({m.cols() for m in mats}) != 1:
            raise RuntimeError("All matrices must have the same number of columns.")
        stacked = (_sp.hstack if axis == 1 else _sp.vstack)([m._scipy for m in mats], format="csc")
        return sparse(stacked, n_threads=n_threads, device=getattr(mats[0], "_device", 0), resident="csc")
    mats = [_expanded(m) if _is_kept_sparse(m) else (m._materialize() if isinstance(m, _StdView) else m) for m in mats]
    for m in mats:
        if isinstance(m, (_MultiView, _OnesKron)) or not isinstance(m, _NativeMatrix):
            raise NotImplementedError(
                "adelie_amd.matrix.concatenate: the pieces must be resident designs (dense / snp / sparse / derived) or "
                "the multi-response pair [kronecker_eye(ones((n,1)), K), kronecker_eye(X, K)].")
    if len({np.dtype(m.dtype) for m in mats}) != 1:
        raise RuntimeError("All matrices must have the same dtype.")
    backend = mats[0]._backend
    handles = (_abi.C.c_void_p * len(mats))(*[m._handle for m in mats])
    handle = _abi.C.c_void_p()
    backend.check(backend.fn("design_create_concat")(handles, len(mats), int(axis), handle))
    # (2-bit designs side by side stay a 2-bit design; every other combination is copied into one dense design)
    stays_snp = axis == 1 and all(getattr(m, "_kind", None) == "snp" for m in mats)
    return _wrap(backend, handle, mats[0].dtype, n_threads, kind="snp" if stays_snp else "dense")


def dense(mat, *, method: str = "naive", copy: bool = False, n_threads: int = 1, device: int = 0):
    """Creates a dense design resident on an MI355X (reference ``adelie.matrix.dense``, ``matrix.py:549-680``).

    Parameters
    ----------
    mat : (n, p) ndarray or torch.Tensor
        float32/float64 matrix.  A numpy array is copied to HBM once.  A CUDA(ROCm) torch tensor is
        adopted in place (no copy); it must be F- or C-contiguous and is kept alive by the handle.
    method : str
        Only ``"naive"`` is on the hot path.
    n_threads : int
        Accepted for API parity (the reference's OpenMP thread count); must be >= 1.
    device : int
        HIP device ordinal for host inputs.
    """
    if method == "cov":
        return _cov_dense(_abi.hip_backend(), "design_create_cov_dense", mat, n_threads, device)
    if method != "naive":
        raise ValueError("method must be one of 'naive' or 'cov'.")
    if n_threads < 1:
        raise RuntimeError("adelie_core: n_threads must be >= 1.")
    backend = _abi.hip_backend()
    handle = _abi.C.c_void_p()

    if not isinstance(mat, np.ndarray) and type(mat).__module__.startswith("torch"):
        t = mat
        if t.dim() != 2 or not t.is_cuda:
            raise RuntimeError("torch input must be a 2-D tensor in device memory.")
        dtype = {"torch.float64": np.float64, "torch.float32": np.float32}[str(t.dtype)]
        n, p = t.shape
        if t.stride() == (1, n) or (p == 1 and t.stride(0) == 1):
            order = _abi.COL_MAJOR
        elif t.is_contiguous():
            order = _abi.ROW_MAJOR
            warnings.warn("Detected matrix to be C-contiguous. Performance may improve with F-contiguous matrix.")
        else:
            raise RuntimeError("torch input must be F- or C-contiguous.")
        backend.check(backend.fn("design_adopt_dense_dev")(
            t.data_ptr(), n, p, _abi.dtype_code(dtype), order, t.device.index or 0, handle))
        return _wrap(backend, handle, dtype, n_threads, keep=t, kind="dense")

    mat = np.asarray(mat)
    if mat.ndim != 2:
        raise RuntimeError("mat must be 2-dimensional.")
    dtype = mat.dtype
    code = _abi.dtype_code(dtype)
    if mat.flags.f_contiguous:
        order = _abi.COL_MAJOR
    else:
        order = _abi.ROW_MAJOR
        mat = np.ascontiguousarray(mat)
        warnings.warn("Detected matrix to be C-contiguous. Performance may improve with F-contiguous matrix.")
    backend.check(backend.fn("design_create_dense")(
        mat.ctypes.data, mat.shape[0], mat.shape[1], code, order, device, handle))
    return _wrap(backend, handle, dtype.type, n_threads, kind="dense")


# matrix.sparse(resident="auto"): kept sparse below this density or above this dense size, expanded to a dense design otherwise
_SPARSE_AUTO_DENSITY = 0.02
_SPARSE_AUTO_DENSE_BYTES = 32 << 30


def sparse(mat, *, method: str = "naive", copy: bool = False, n_threads: int = 1, device: int = 0, resident: str = "auto"):
    """Creates a design from a scipy sparse matrix (reference ``adelie.matrix.sparse``, ``matrix.py:1301-1385``).

    ``resident`` chooses what lives in HBM:

    * ``"csc"`` — the matrix stays sparse, as in the reference (``matrix_naive_sparse.ipp`` walks the CSC arrays per
      operation): the stored entries are uploaded column- and row-compressed (``adelie_hip_design_create_csc``, 24 bytes
      per entry in f64, 34 with the tile-major copy the library adds for its full sweeps on larger designs), every operation
      streams the entries it needs, and ``grpnet`` runs its full-Gram engines on it.  A
      design whose ``n * p`` values do not fit in device memory can run this way.  Duplicate entries are summed first.
    * ``"dense"`` — the entries are scattered once into a dense column-major array (``adelie_hip_design_create_sparse``)
      and the design then behaves as ``matrix.dense`` (panel engines, MFMA Gram builds): faster while ``n * p`` fits and
      the matrix is not very sparse.
    * ``"auto"`` (default) — ``"csc"`` when fewer than 2 % of the cells are stored or the dense form would exceed 32 GiB,
      ``"dense"`` otherwise.

    Same results either way up to the summation order of the dot products."""
    if not (isinstance(mat, csr_matrix) or isinstance(mat, csc_matrix)):
        raise TypeError("mat must be scipy.sparse.csr_matrix or scipy.sparse.csc_matrix.")
    if method == "cov":  # MatrixCovSparse (matrix_cov_sparse.ipp): a symmetric sparse (p, p) matrix, dense in HBM here
        if mat.shape[0] != mat.shape[1]:
            raise RuntimeError("adelie_core: mat must be (p, p).")
        return _cov_dense(_abi.hip_backend(), "design_create_cov_dense", np.asfortranarray(mat.toarray()), n_threads, device)
    if method != "naive":
        raise ValueError("method must be one of 'naive' or 'cov'.")
    if n_threads < 1:
        raise RuntimeError("adelie_core: n_threads must be >= 1.")
    if isinstance(mat, csr_matrix):
        warnings.warn("Converting to CSC format.")
        mat = mat.tocsc(copy=True)
    elif copy:
        mat = mat.copy()
    mat.prune()
    mat.sort_indices()
    dtype = np.dtype(mat.dtype).type
    if dtype not in (np.float32, np.float64):
        raise RuntimeError("mat must have dtype float32 or float64.")
    n, p = mat.shape
    if resident not in ("auto", "csc", "dense"):
        raise ValueError("resident must be one of 'auto', 'csc' or 'dense'.")
    if resident == "auto":
        cells = float(n) * float(p)
        resident = "csc" if (mat.nnz < _SPARSE_AUTO_DENSITY * cells
                             or cells * np.dtype(dtype).itemsize > _SPARSE_AUTO_DENSE_BYTES) else "dense"
    backend = _abi.hip_backend()
    handle = _abi.C.c_void_p()
    if resident == "csc":
        if not mat.has_canonical_format:
            mat = mat.copy()
            mat.sum_duplicates()
        rows = mat.tocsr()
        rows.sort_indices()
        c_ptr = np.ascontiguousarray(mat.indptr, dtype=np.int64)
        c_idx = np.ascontiguousarray(mat.indices, dtype=np.int32)
        c_val = np.ascontiguousarray(mat.data, dtype=dtype)
        r_ptr = np.ascontiguousarray(rows.indptr, dtype=np.int64)
        r_idx = np.ascontiguousarray(rows.indices, dtype=np.int32)
        r_val = np.ascontiguousarray(rows.data, dtype=dtype)
        backend.check(backend.fn("design_create_csc")(
            c_ptr.ctypes.data, c_idx.ctypes.data, c_val.ctypes.data, r_ptr.ctypes.data, r_idx.ctypes.data, r_val.ctypes.data,
            n, p, _abi.dtype_code(dtype), device, handle))
        out = _wrap(backend, handle, dtype, n_threads, kind="sparse")
        out._scipy = mat  # derived designs (subset / concatenate / standardize / the multi-response view) start from it
        out._device = device
        return out
    indptr = np.ascontiguousarray(mat.indptr, dtype=np.int64)
    indices = np.ascontiguousarray(mat.indices, dtype=np.int32)
    values = np.ascontiguousarray(mat.data, dtype=dtype)
    backend.check(backend.fn("design_create_sparse")(
        indptr.ctypes.data, indices.ctypes.data, values.ctypes.data, n, p, _abi.dtype_code(dtype), device, handle))
    return _wrap(backend, handle, dtype, n_threads, kind="dense")


def _is_kept_sparse(m):
    return isinstance(m, _NativeMatrix) and getattr(m, "_kind", None) == "sparse"


def _expanded(m):
    """The dense-resident form of a design that is kept sparse (for the operations that need dense column slices)."""
    out = sparse(m._scipy, n_threads=m._n_threads, device=getattr(m, "_device", 0), resident="dense")
    std = getattr(m, "_std", None)
    if std is not None:  # a standardized view: the same centres / scales on the expanded copy
        out = _derived(out, None, None, std[0], std[1], m._n_threads)
    return out


def snp_unphased(io, *, dtype=np.float64, n_threads: int = 1, device: int = 0):
    """Creates an SNP-unphased design resident on an MI355X as dense 2-bit calls
    (reference ``adelie.matrix.snp_unphased``, ``matrix.py:1245-1298``).

    ``io`` is an ``adelie_amd.io.snp_unphased`` handler that has been ``read()``.
    """
    if n_threads < 1:
        raise RuntimeError("adelie_core: n_threads must be >= 1.")
    if not io.is_read:
        io.read()
    backend = _abi.hip_backend()
    handle = _abi.C.c_void_p()
    buf = io._buffer
    backend.check(backend.fn("design_create_snp_unphased")(
        buf.ctypes.data, buf.size, _abi.dtype_code(dtype), device, handle))
    return _wrap(backend, handle, np.dtype(dtype).type, n_threads, keep=io, kind="snp")


def snp_calldata(calldata, impute=None, *, dtype=np.float64, n_threads: int = 1, device: int = 0):
    """SNP design straight from an int8 ``(n, p)`` calldata matrix (negative = missing), skipping the
    ``.snpdat`` file.  ``impute`` defaults to the column mean of the non-missing calls
    (reference ``io/utils.hpp:10-31``)."""
    if hasattr(calldata, "data_ptr"):
        # a torch int8 tensor already on the device, column-major (n, p): packed in place, no host round trip
        import torch

        if calldata.dtype != torch.int8 or calldata.dim() != 2 or calldata.stride() != (1, calldata.shape[0]):
            raise RuntimeError("adelie_core: device calldata must be an int8 (n, p) tensor with strides (1, n).")
        n, p = calldata.shape
        if impute is None:
            valid = calldata >= 0
            cnt = valid.sum(dim=0).clamp(min=1)
            impute = ((calldata * valid).sum(dim=0, dtype=torch.float64) / cnt).cpu().numpy()
        ptr = calldata.data_ptr()
        device = calldata.device.index if calldata.is_cuda else device
    else:
        calldata = np.asfortranarray(calldata, dtype=np.int8)
        n, p = calldata.shape
        if impute is None:
            impute = compute_impute(calldata)
        ptr = calldata.ctypes.data
    impute = np.ascontiguousarray(impute, dtype=np.float64)
    backend = _abi.hip_backend()
    handle = _abi.C.c_void_p()
    backend.check(backend.fn("design_create_snp_calldata")(
        ptr, n, p, impute.ctypes.data, _abi.dtype_code(dtype), device, handle))
    return _wrap(backend, handle, np.dtype(dtype).type, n_threads, kind="snp")


def snp_bed(bed, n: int, p: int = None, *, dtype=np.float64, n_threads: int = 1, device: int = 0):
    """SNP design from a PLINK 1 ``.bed`` file (SNP-major): ``bed`` is a path, a ``bytes``-like image or a ``uint8`` array.
    ``n`` (samples) comes from the ``.fam`` file; ``p`` defaults to what the image holds.  Calls are counts of allele A1,
    missing calls are mean-imputed; the 2-bit records are transcoded on the device (no host decode)."""
    if isinstance(bed, str):
        buf = np.fromfile(bed, dtype=np.uint8)
    elif isinstance(bed, np.ndarray):
        buf = np.ascontiguousarray(bed, dtype=np.uint8)
    else:
        buf = np.frombuffer(bed, dtype=np.uint8)
    stride = (int(n) + 3) // 4
    if p is None:
        p = (buf.size - 3) // stride
    backend = _abi.hip_backend()
    handle = _abi.C.c_void_p()
    backend.check(backend.fn("design_create_snp_bed")(
        buf.ctypes.data, buf.size, int(n), int(p), _abi.dtype_code(dtype), device, handle))
    return _wrap(backend, handle, np.dtype(dtype).type, n_threads, kind="snp")


def _derived(mat, rows, cols, centers, scales, n_threads):
    if isinstance(mat, _StdView):  # derived designs of a lazily standardized design start from its materialised form
        mat = mat._materialize()
    if not isinstance(mat, _NativeMatrix) or isinstance(mat, _MultiView):
        raise RuntimeError("adelie_amd: subset / standardize need a resident dense or SNP design.")
    if _is_kept_sparse(mat) and getattr(mat, "_std", None) is not None:
        mat = _expanded(mat)  # (derived designs of a standardized view are built from its expanded copy)
    if _is_kept_sparse(mat):
        if centers is not None and scales is not None and rows is None and cols is None:
            # standardize: the entries stay as they are, centring / scaling become rank-one corrections of every operation
            backend = mat._backend
            ce = np.ascontiguousarray(centers, dtype=np.float64)
            sc = np.ascontiguousarray(scales, dtype=np.float64)
            handle = _abi.C.c_void_p()
            backend.check(backend.fn("design_create_standardized")(mat._handle, ce.ctypes.data, sc.ctypes.data, handle))
            out = _wrap(backend, handle, mat.dtype, n_threads, keep=mat, kind="sparse")
            out._scipy, out._device, out._std = mat._scipy, getattr(mat, "_device", 0), (ce, sc)
            return out
        if centers is None and scales is None:  # a subset of a sparse matrix is sparse: composed on the host, kept sparse
            sub = mat._scipy
            if rows is not None:
                rows = np.asarray(rows, dtype=np.int64)
                if rows.size and (rows.min() < 0 or rows.max() >= sub.shape[0]):
                    raise RuntimeError("adelie_core: subset contains an out-of-range row index.")
                sub = sub[rows]
            if cols is not None:
                cols = np.asarray(cols, dtype=np.int64)
                if cols.size and (cols.min() < 0 or cols.max() >= sub.shape[1]):
                    raise RuntimeError("adelie_core: subset contains an out-of-range column index.")
                sub = sub[:, cols]
            return sparse(sub.tocsc(), n_threads=n_threads, device=getattr(mat, "_device", 0), resident="csc")
        mat = _expanded(mat)  # centring fills every cell: the standardized design is dense
    backend = mat._backend
    if not backend.has("design_create_derived"):
        raise NotImplementedError("this backend does not derive designs.")
    keep = []

    def arr(x, dt):
        if x is None:
            return None, 0
        a = np.ascontiguousarray(x, dtype=dt)
        keep.append(a)
        return a.ctypes.data, a.size

    r, nr = arr(rows, np.int64)
    c, nc = arr(cols, np.int64)
    ce, _ = arr(centers, np.float64)
    sc, _ = arr(scales, np.float64)
    handle = _abi.C.c_void_p()
    backend.check(backend.fn("design_create_derived")(mat._handle, r, nr, c, nc, ce, sc, handle))
    # (rows / columns of a 2-bit design are re-packed as a 2-bit design; everything else is a dense design)
    stays_snp = getattr(mat, "_kind", None) == "snp" and centers is None and scales is None
    return _wrap(backend, handle, mat.dtype, n_threads, kind="snp" if stays_snp else "dense")


def standardize(mat, centers=None, scales=None, ddof: int = 0, *, n_threads: int = 1, lazy="auto"):
    """``X = (Z - 1 c^T) diag(s)^-1`` (reference ``adelie.matrix.standardize``, ``matrix.py:1414-1533``,
    ``matrix_naive_standardize.ipp``).  ``centers`` / ``scales`` default to the column means and standard deviations of ``mat``
    under equal weights (``ddof`` as in the reference).  A numpy input gives a numpy result, as in the reference.  For a
    resident design, ``lazy`` chooses between the reference's wrapper and a copy:

    * ``lazy=True`` — a view that shares the resident matrix (``adelie_hip_design_create_standardized``): centring and scaling
      are applied as rank-one corrections around the base design's kernels, ``grpnet`` runs its full-Gram engines on it.  A
      standardized 2-bit SNP design stays 2 bits per call (the copy is 8 bytes per call: 32 times the memory);
    * ``lazy=False`` — a new resident dense design materialised by one kernel: twice the memory of a dense base, but the panel
      engines and MFMA Gram builds run on it (faster for dense bases that fit);
    * ``lazy="auto"`` (default) — a view over SNP designs and over designs kept sparse (always: centring would fill every
      cell), a copy of dense ones.

    ``_centers`` / ``_scales`` are attached to the result."""
    if isinstance(mat, (list, np.ndarray)):
        mat = np.array(mat, order="F", copy=True)
        if centers is None:
            centers = np.mean(mat, axis=0)
        mat -= np.asarray(centers)[None]
        if scales is None:
            n = mat.shape[0]
            scales = np.sqrt(np.sum(mat ** 2, axis=0) / (n - ddof))
        mat /= np.asarray(scales)[None]
        return mat
    dtype = mat.dtype
    n, p = mat.shape
    weights = np.full(n, 1 / n, dtype=dtype)
    if centers is None:
        centers = np.empty(p, dtype=dtype)
        mat.mean(weights, centers)
    if scales is None:
        v = np.empty(p, dtype=dtype)
        mat.var(centers, weights, v)
        scales = np.sqrt((n / (n - ddof)) * v)
    if lazy not in (True, False, "auto"):
        raise ValueError("lazy must be True, False or 'auto'.")
    plain = (isinstance(mat, _NativeMatrix) and not isinstance(mat, (_MultiView, _StdView)) and not _is_kept_sparse(mat)
             and mat._backend.has("design_create_standardized"))
    if plain and (lazy is True or (lazy == "auto" and getattr(mat, "_kind", None) == "snp")):
        if np.any(np.asarray(scales) == 0):
            raise RuntimeError("adelie_core: scales must be non-zero.")
        return _std_view(mat, centers, scales, n_threads)
    out = _derived(mat, None, None, centers, scales, n_threads)
    out._centers = np.array(centers, copy=True, dtype=dtype)
    out._scales = np.array(scales, copy=True, dtype=dtype)
    return out


def subset(mat, indices, *, axis: int = 0, n_threads: int = 1):
    """``mat[indices]`` (``axis=0``) or ``mat[:, indices]`` (``axis=1``) as a new design (reference ``adelie.matrix.subset``,
    ``matrix.py:1538-1640``, ``matrix_naive_subset.ipp``).  As the reference notes, to fit on a subset of the observations it
    is cheaper to zero their weights than to subset the rows."""
    if isinstance(mat, np.ndarray):
        return mat[indices] if axis == 0 else mat[:, indices]
    indices = np.asarray(indices)
    if indices.dtype == bool:
        indices = np.flatnonzero(indices)
    if axis not in (0, 1):
        raise RuntimeError("axis must be 0 or 1.")
    view = _slice_view(mat, indices, axis, n_threads)
    if view is not None:
        return view
    if axis == 0:
        return _derived(mat, indices, None, None, None, n_threads)
    if axis == 1:
        return _derived(mat, None, indices, None, None, n_threads)
    raise RuntimeError("axis must be 0 or 1.")


def _slice_view(mat, indices, axis, n_threads):
    """A contiguous ascending index range of a resident dense (either axis; rows from a 16-byte boundary) or 2-bit SNP design
    (columns) as a design that SHARES the matrix (``adelie_hip_design_create_slice``) -- the reference's ``subset`` is a lazy
    wrapper too (``matrix_naive_subset.ipp``); ``None`` when the range or the design does not allow it (the caller copies)."""
    if (not isinstance(mat, _NativeMatrix) or isinstance(mat, (_MultiView, _StdView)) or _is_kept_sparse(mat)
            or getattr(mat, "_kind", None) not in ("dense", "snp") or isinstance(mat, (MatrixCovBase64, MatrixCovBase32))):
        return None
    idx = np.asarray(indices)
    if idx.ndim != 1 or idx.size == 0 or not np.issubdtype(idx.dtype, np.integer):
        return None
    if idx.size > 1 and not np.all(np.diff(idx) == 1):
        return None
    i0, cnt = int(idx[0]), int(idx.size)
    size = mat.rows() if axis == 0 else mat.cols()
    if i0 < 0 or i0 + cnt > size:
        return None   # (the copying route raises the reference's out-of-range error)
    if axis == 0 and (mat._kind != "dense" or (i0 * np.dtype(mat.dtype).itemsize) % 16 != 0):
        return None
    backend = mat._backend
    if not backend.has("design_create_slice"):
        return None
    handle = _abi.C.c_void_p()
    r0, nr, c0, nc = (i0, cnt, 0, mat.cols()) if axis == 0 else (0, mat.rows(), i0, cnt)
    backend.check(backend.fn("design_create_slice")(mat._handle, r0, nr, c0, nc, handle))
    out = _wrap(backend, handle, mat.dtype, n_threads, keep=mat, kind=mat._kind)
    out._slice_of = mat
    return out


def snp_plink(prefix, *, dtype=np.float64, n_threads: int = 1, device: int = 0):
    """SNP design from a PLINK 1 fileset ``prefix.{bed,bim,fam}``: the dimensions come from the ``.fam`` / ``.bim`` line
    counts (``adelie_amd.io.plink_dims``), the ``.bed`` image goes to :func:`snp_bed`."""
    from . import io as _io

    n, p = _io.plink_dims(prefix)
    return snp_bed(prefix + ".bed", n, p, dtype=dtype, n_threads=n_threads, device=device)


def compute_impute(calldata):
    """Mean imputation (reference ``io/utils.hpp:10-31``): mean of the non-missing entries per column."""
    calldata = np.asarray(calldata)
    valid = calldata >= 0
    cnt = np.maximum(valid.sum(axis=0), 1)
    return (np.where(valid, calldata, 0).sum(axis=0) / cnt).astype(np.float64)


def from_plugin(mat, *, n_threads: int = 1, device: int = 0):
    """Brings a Python subclass of ``MatrixNaiveBase64/32`` onto the device.

    The reference lets users subclass its matrix base class in Python and calls the overridden ``cmul`` / ``ctmul`` / ... from
    the C++ solver, one interpreter round trip per coordinate visit (trampoline ``PyMatrixNaiveBase``,
    ``py_matrix.cpp:626-825``).  A design that lives behind Python callbacks cannot feed a GPU; with 288 GB of HBM the
    MI355X answer is to evaluate the user's matrix ONCE — column ``j`` is ``ctmul(j, 1, zeros(n))``, by definition of the
    interface (``matrix_naive_base.hpp:49-59``: ``out += v * X[:, j]``) — and to run the path on the resident dense copy, which
    is then exactly the matrix the user's methods describe."""
    out = dense(densify_plugin(mat), method="naive", n_threads=n_threads, device=device)
    out._plugin = mat
    return out


def densify_plugin(mat):
    """The ``(n, p)`` F-ordered array a user-defined matrix class describes, column by column through its ``ctmul``."""
    if not isinstance(mat, (MatrixNaiveBase64, MatrixNaiveBase32)):
        raise RuntimeError("X must be an ndarray, an adelie_amd.matrix design or a subclass of MatrixNaiveBase64 / MatrixNaiveBase32.")
    dtype = np.float64 if isinstance(mat, MatrixNaiveBase64) else np.float32
    n, p = int(mat.rows()), int(mat.cols())
    out = np.zeros((n, p), dtype=dtype, order="F")
    for j in range(p):
        mat.ctmul(j, dtype(1), out[:, j])
    return out


def as_design(X, *, n_threads: int = 1):
    """What ``grpnet`` / ``cv_grpnet`` / the state constructors accept as ``X``: an ndarray or device tensor (wrapped by
    :func:`dense`), a native design handle (returned as is), or a user-defined matrix class (:func:`from_plugin`)."""
    if isinstance(X, MatrixCovBase):
        raise RuntimeError("X is a covariance matrix (matrix.dense(method=\"cov\")): use gaussian_cov for the covariance method.")
    if hasattr(X, "_backend"):
        return X
    if isinstance(X, np.ndarray) or type(X).__module__.startswith("torch"):
        return dense(X, method="naive", n_threads=n_threads)
    return from_plugin(X, n_threads=n_threads)


class _CovMatrix:
    """A symmetric ``(p, p)`` matrix resident in HBM for the covariance method (``adelie.matrix.dense(method="cov")``,
    reference ``MatrixCovDense``, ``matrix_cov_dense.ipp:9-84``).  Methods are the ``MatrixCovBase`` virtuals, one C-ABI call
    each, with the reference's argument conventions (pre-allocated outputs written in place)."""

    def _init_native(self, backend, handle, n_threads, keep):
        self._backend, self._handle, self._n_threads, self._keep = backend, handle, n_threads, keep
        self._cols = int(backend.fn("design_cols")(handle))

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None:
            try:
                self._backend.fn("design_destroy")(h)
            except Exception:
                pass
            self._handle = None

    def rows(self):
        return self._cols

    def cols(self):
        return self._cols

    @property
    def ndim(self):
        return 2

    @property
    def shape(self):
        return (self._cols, self._cols)

    def _chk(self, cond, msg):
        if not cond:
            raise RuntimeError("adelie_core: " + msg)

    def bmul(self, subset, indices, values, out):
        """``out[j] = sum_i values[i] * A[indices[i], subset[j]]`` (``matrix_cov_dense.ipp:23-41``)."""
        r = self._cols
        subset, indices = (np.ascontiguousarray(x, dtype=np.int64) for x in (subset, indices))
        values = np.ascontiguousarray(values, dtype=self.dtype)
        s, i, v, o = len(subset), len(indices), len(values), len(out)
        self._chk(0 <= s <= r and 0 <= i <= r and i == v and o == s,
                  "bmul() is given inconsistent inputs! Invoked check_bmul(s=%d, i=%d, v=%d, o=%d, r=%d, c=%d)" % (s, i, v, o, r, r))
        buf = out if (isinstance(out, np.ndarray) and out.dtype == self.dtype and out.flags.c_contiguous) else np.empty(s, dtype=self.dtype)
        self._backend.check(self._backend.fn("design_cov_bmul")(
            self._handle, subset.ctypes.data, s, indices.ctypes.data, values.ctypes.data, i, buf.ctypes.data))
        if buf is not out:
            out[...] = buf

    def mul(self, indices, values, out):
        """``out = sum_i values[i] * A[indices[i], :]`` (``matrix_cov_dense.ipp:43-62``)."""
        r = self._cols
        indices = np.ascontiguousarray(indices, dtype=np.int64)
        values = np.ascontiguousarray(values, dtype=self.dtype)
        i, v, o = len(indices), len(values), len(out)
        self._chk(0 <= i <= r and i == v and o == r,
                  "mul() is given inconsistent inputs! Invoked check_mul(i=%d, v=%d, o=%d, r=%d, c=%d)" % (i, v, o, r, r))
        buf = out if (isinstance(out, np.ndarray) and out.dtype == self.dtype and out.flags.c_contiguous) else np.empty(r, dtype=self.dtype)
        self._backend.check(self._backend.fn("design_cov_mul")(self._handle, indices.ctypes.data, values.ctypes.data, i,
                                                                buf.ctypes.data))
        if buf is not out:
            out[...] = buf

    def to_dense(self, i, p, out):
        """``out = A[i:i+p, i:i+p]`` (``matrix_cov_dense.ipp:64-74``); ``out`` is ``(p, p)`` F-ordered."""
        r = self._cols
        self._chk(0 <= i <= r - p and out.shape == (p, p),
                  "to_dense() is given inconsistent inputs! Invoked check_to_dense(i=%d, p=%d, o_r=%d, o_c=%d, r=%d, c=%d)"
                  % (i, p, out.shape[0], out.shape[1] if out.ndim > 1 else -1, r, r))
        buf = np.empty((p, p), dtype=self.dtype, order="F")
        self._backend.check(self._backend.fn("design_cov_to_dense")(self._handle, int(i), int(p), buf.ctypes.data))
        out[...] = buf


def _cov_dense(backend, ctor, mat, n_threads, device):
    """``adelie.matrix.dense(mat, method="cov")`` (reference ``matrix.py:549-680``, cov branch): a dense symmetric matrix,
    copied to HBM once."""
    if n_threads < 1:
        raise RuntimeError("adelie_core: n_threads must be >= 1.")
    mat = np.asarray(mat)
    if mat.ndim != 2 or mat.shape[0] != mat.shape[1]:
        raise RuntimeError("adelie_core: mat must be (p, p).")
    code = _abi.dtype_code(mat.dtype)
    if mat.flags.f_contiguous:
        order = _abi.COL_MAJOR
    else:
        order = _abi.ROW_MAJOR
        mat = np.ascontiguousarray(mat)
    handle = _abi.C.c_void_p()
    backend.check(backend.fn(ctor)(mat.ctypes.data, mat.shape[0], code, order, device, handle))
    base = MatrixCovBase64 if mat.dtype == np.float64 else MatrixCovBase32
    cls = type("_cov_matrix", (_CovMatrix, base), {"dtype": base.dtype})
    obj = cls()
    obj._init_native(backend, handle, n_threads, mat)
    return obj


def lazy_cov(mat, *, copy: bool = False, n_threads: int = 1, device: int = 0):
    """``A = X^T X`` for the covariance method (reference ``adelie.matrix.lazy_cov``, ``matrix.py:1000-1065``,
    ``MatrixCovLazyCov``).  The reference computes the rows of ``A`` that the solver asks for on demand, because ``A`` may not
    fit in host memory; here ``A`` is formed once on the device by the MFMA Gram kernel from the resident design (dense ndarray,
    device tensor, or any dense / SNP ``adelie_amd.matrix`` design) and then behaves like ``dense(A, method="cov")``.
    ``copy`` is accepted for signature parity (the data matrix is copied to HBM in any case and released afterwards)."""
    if n_threads < 1:
        raise RuntimeError("adelie_core: n_threads must be >= 1.")
    X = mat if isinstance(mat, _NativeMatrix) else dense(mat, method="naive", n_threads=n_threads, device=device)
    backend = X._backend
    if not backend.has("design_create_cov_lazy"):
        raise NotImplementedError("adelie_amd: lazy_cov needs the device library.")
    handle = _abi.C.c_void_p()
    backend.check(backend.fn("design_create_cov_lazy")(X._handle, handle))
    base = MatrixCovBase64 if X.dtype == np.float64 else MatrixCovBase32
    cls = type("_cov_matrix", (_CovMatrix, base), {"dtype": base.dtype})
    obj = cls()
    obj._init_native(backend, handle, n_threads, None)
    return obj


def block_diag(mats, *, method: str = "naive", n_threads: int = 1, device: int = 0):
    """Block-diagonal matrix (reference ``adelie.matrix.block_diag``, ``matrix.py:198-300``).  ``method="cov"``
    (``MatrixCovBlockDiag``): the blocks are symmetric matrices — ndarrays or covariance matrices of this module — and the result
    is one resident ``(p, p)`` covariance matrix with them on its diagonal (dense in HBM; the reference keeps the list and
    dispatches per block).  ``method="naive"`` is not implemented."""
    if method == "naive":
        raise NotImplementedError("adelie_amd.matrix.block_diag: only method='cov' is implemented.")
    if method != "cov":
        raise ValueError("method must be one of 'naive' or 'cov'.")
    if len(mats) == 0:
        raise RuntimeError("adelie_core: mats must be non-empty.")
    blocks = []
    for m in mats:
        if isinstance(m, _CovMatrix):
            q = m.cols()
            out = np.empty((q, q), dtype=m.dtype, order="F")
            m.to_dense(0, q, out)
            blocks.append(out)
        else:
            m = np.asarray(m)
            if m.ndim != 2 or m.shape[0] != m.shape[1]:
                raise RuntimeError("adelie_core: every block must be (q, q).")
            blocks.append(m)
    dtype = blocks[0].dtype
    if any(b.dtype != dtype for b in blocks):
        raise RuntimeError("adelie_core: all blocks must have the same dtype.")
    p = sum(b.shape[0] for b in blocks)
    A = np.zeros((p, p), dtype=dtype, order="F")
    o = 0
    for b in blocks:
        q = b.shape[0]
        A[o:o + q, o:o + q] = b
        o += q
    return _cov_dense(_abi.hip_backend(), "design_create_cov_dense", A, n_threads, device)
