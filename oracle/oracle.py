"""Python handle on the CPU oracle (``oracle/liboracle.so``).  TEST INFRASTRUCTURE ONLY.

Binds ``grpnet_oracle.cpp`` through the same ctypes structures as the product (symbol prefix ``oracle_``),
so that a test can run ``adelie_amd.solver.grpnet`` twice on identical inputs — once on a design created
by ``adelie_amd.matrix.dense`` (HIP path) and once on a design created by ``oracle.oracle.dense`` (this
file) — and compare.  Nothing under ``adelie_amd/`` imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from adelie_amd import _abi
from adelie_amd import matrix as _matrix

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
_BACKEND = None


class OracleBackend(_abi.Backend):
    """The product's ctypes binding pointed at ``liboracle.so``: same structures and signatures, symbols ``oracle_*``."""

    def fn(self, name):
        return getattr(self.lib, "oracle_" + name)

    def has(self, name):
        return hasattr(self.lib, "oracle_" + name)


def build(force: bool = False):
    """Compiles the oracle with the committed Makefile (gcc only, no GPU needed)."""
    src = os.path.join(_HERE, "grpnet_oracle.cpp")
    hdr = os.path.join(_HERE, "..", "include", "adelie_hip.h")
    lin = os.path.join(_HERE, "linear_constraint.hpp")
    if (not force and os.path.exists(_LIB)
            and os.path.getmtime(_LIB) >= max(os.path.getmtime(src), os.path.getmtime(hdr), os.path.getmtime(lin))):
        return _LIB
    subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


def backend():
    global _BACKEND
    if _BACKEND is None:
        if not os.path.exists(_LIB):
            build()
        b = OracleBackend(_LIB)
        p, i64, dbl, vp, ci = C.POINTER, C.c_int64, C.c_double, C.c_void_p, C.c_int
        b.fn("design_create_dense").argtypes = [vp, i64, i64, ci, ci, ci, p(vp)]  # last int = n_threads
        b.fn("design_create_snp_calldata").argtypes = [vp, i64, i64, vp, ci, ci, p(vp)]
        b.lib.oracle_bcd_newton.argtypes = [i64, vp, vp, dbl, dbl, dbl, i64, vp, p(i64)]
        b.lib.oracle_search_pivot.argtypes = [i64, vp, vp, vp]
        b.lib.oracle_eigh.argtypes = [i64, vp, vp, vp]
        _BACKEND = b
    return _BACKEND


def dense(mat, *, n_threads: int = 1):
    """CPU counterpart of ``adelie_amd.matrix.dense`` (non-owning view of ``mat``, like the reference)."""
    b = backend()
    mat = np.asarray(mat)
    order = _abi.COL_MAJOR if mat.flags.f_contiguous else _abi.ROW_MAJOR
    if order == _abi.ROW_MAJOR:
        mat = np.ascontiguousarray(mat)
    h = C.c_void_p()
    b.check(b.fn("design_create_dense")(mat.ctypes.data, mat.shape[0], mat.shape[1], _abi.dtype_code(mat.dtype),
                                         order, n_threads, h))
    return _matrix._wrap(b, h, mat.dtype.type, n_threads, keep=mat)


def snp_calldata(calldata, impute=None, *, dtype=np.float64, n_threads: int = 1):
    b = backend()
    calldata = np.asfortranarray(calldata, dtype=np.int8)
    if impute is None:
        impute = _matrix.compute_impute(calldata)
    impute = np.ascontiguousarray(impute, dtype=np.float64)
    h = C.c_void_p()
    b.check(b.fn("design_create_snp_calldata")(calldata.ctypes.data, calldata.shape[0], calldata.shape[1],
                                                impute.ctypes.data, _abi.dtype_code(dtype), n_threads, h))
    return _matrix._wrap(b, h, np.dtype(dtype).type, n_threads, keep=calldata)


def bcd_newton(L, v, l1, l2, tol=1e-12, max_iters=1000):
    b = backend()
    L = np.ascontiguousarray(L, dtype=np.float64)
    v = np.ascontiguousarray(v, dtype=np.float64)
    x = np.empty_like(L)
    it = C.c_int64()
    b.lib.oracle_bcd_newton(len(L), L.ctypes.data, v.ctypes.data, l1, l2, tol, max_iters, x.ctypes.data, it)
    return x, it.value


def search_pivot(x, y):
    b = backend()
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    mses = np.empty_like(x)
    idx = b.lib.oracle_search_pivot(len(x), x.ctypes.data, y.ctypes.data, mses.ctypes.data)
    return idx, mses


def eigh(A):
    b = backend()
    A = np.asfortranarray(A, dtype=np.float64)
    q = A.shape[0]
    V = np.empty((q, q), order="F")
    D = np.empty(q)
    b.lib.oracle_eigh(q, A.ctypes.data, V.ctypes.data, D.ctypes.data)
    return D, V


def cov_dense(mat, *, n_threads: int = 1):
    """CPU counterpart of ``adelie_amd.matrix.dense(mat, method="cov")`` (non-owning view, like the reference)."""
    b = backend()
    p, vp, i64, ci = C.POINTER, C.c_void_p, C.c_int64, C.c_int
    b.fn("design_create_cov_dense").argtypes = [vp, i64, ci, ci, ci, p(vp)]  # last int = n_threads
    return _matrix._cov_dense(b, "design_create_cov_dense", np.asarray(mat), n_threads, n_threads)
