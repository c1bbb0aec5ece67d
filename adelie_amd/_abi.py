"""ctypes binding of ``include/adelie_hip.h`` (the C ABI of ``libadelie_hip.so``).

This is the only place where Python touches the native library.  It replaces what
``adelie.adelie_core`` (the pybind11 module, reference ``adelie/src/py_adelie_core.cpp:6-44``)
is for the reference: the boundary between the Python API layer and the native solver.

There is no CPU fallback: a missing ``libadelie_hip.so`` raises.
"""
import ctypes as C
import os

import numpy as np

F32, F64 = 0, 1
COL_MAJOR, ROW_MAJOR = 0, 1
SCREEN_STRONG, SCREEN_PIVOT = 0, 1
GLM_GAUSSIAN, GLM_BINOMIAL_LOGIT, GLM_GAUSSIAN_IRLS, GLM_MULTINOMIAL, GLM_POISSON, GLM_BINOMIAL_PROBIT = 0, 1, 2, 3, 4, 5
GLM_CALLBACK = 6  # a Python subclass of glm.GlmBase64/32, evaluated by host callbacks

# int poll(void* user, int final, int64_t n_solutions, const adelie_hip_result* live)
POLL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_void_p)
# adelie_hip_done_fn of adelie_hip_grpnet_solve_many (ABI 10): (k, rc, user), from solve k's own thread
DONE_FN = C.CFUNCTYPE(None, C.c_int32, C.c_int, C.c_void_p)
GLM_GRADIENT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)
GLM_HESSIAN_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
GLM_LOSS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_double))


CONS_SOLVE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double),
                           C.POINTER(C.c_double), C.c_double, C.c_double, C.POINTER(C.c_double))
CONS_GRADIENT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double))
CONS_SOLVE_ZERO_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double))
CONS_DUAL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_double))


class ConstraintCallbacks(C.Structure):
    """``adelie_hip_constraint_callbacks``."""

    _fields_ = [("user", C.c_void_p), ("solve", CONS_SOLVE_FN), ("gradient", CONS_GRADIENT_FN),
                ("solve_zero", CONS_SOLVE_ZERO_FN), ("dual", CONS_DUAL_FN)]


class LinearConstraint(C.Structure):
    """``adelie_hip_linear_constraint`` (ABI 5)."""

    _fields_ = [("m", C.c_int64), ("d", C.c_int64), ("A", C.c_void_p), ("lower", C.c_void_p), ("upper", C.c_void_p),
                ("vars", C.c_void_p), ("cfg", C.c_double * 7)]


class GlmCallbacks(C.Structure):
    """``adelie_hip_glm_callbacks``."""

    _fields_ = [("user", C.c_void_p), ("gradient", GLM_GRADIENT_FN), ("hessian", GLM_HESSIAN_FN), ("loss", GLM_LOSS_FN)]


class GrpnetArgs(C.Structure):
    """``adelie_hip_grpnet_args`` — field order must match include/adelie_hip.h exactly."""

    _fields_ = [
        ("G", C.c_int64),
        ("groups", C.c_void_p),
        ("group_sizes", C.c_void_p),
        ("alpha", C.c_double),
        ("penalty", C.c_void_p),
        ("weights", C.c_void_p),
        ("X_means", C.c_void_p),
        ("y_mean", C.c_double),
        ("y_var", C.c_double),
        ("resid_sum", C.c_double),
        ("rsq", C.c_double),
        ("glm_kind", C.c_int32),
        ("_pad0", C.c_int32),
        ("glm_y", C.c_void_p),
        ("glm_weights", C.c_void_p),
        ("offsets", C.c_void_p),
        ("eta", C.c_void_p),
        ("beta0", C.c_double),
        ("loss_null", C.c_double),
        ("loss_full", C.c_double),
        ("irls_max_iters", C.c_int64),
        ("irls_tol", C.c_double),
        ("setup_loss_null", C.c_int32),
        ("_pad1", C.c_int32),
        ("resid", C.c_void_p),
        ("grad", C.c_void_p),
        ("lmda_path", C.c_void_p),
        ("n_lmda_path", C.c_int64),
        ("lmda_max", C.c_double),
        ("min_ratio", C.c_double),
        ("lmda_path_size", C.c_int64),
        ("max_screen_size", C.c_int64),
        ("max_active_size", C.c_int64),
        ("pivot_subset_ratio", C.c_double),
        ("pivot_subset_min", C.c_int64),
        ("pivot_slack_ratio", C.c_double),
        ("screen_rule", C.c_int32),
        ("early_exit", C.c_int32),
        ("max_iters", C.c_int64),
        ("tol", C.c_double),
        ("adev_tol", C.c_double),
        ("ddev_tol", C.c_double),
        ("newton_tol", C.c_double),
        ("newton_max_iters", C.c_int64),
        ("setup_lmda_max", C.c_int32),
        ("setup_lmda_path", C.c_int32),
        ("intercept", C.c_int32),
        ("n_threads", C.c_int32),
        ("screen_set_size", C.c_int64),
        ("screen_set", C.c_void_p),
        ("screen_beta_size", C.c_int64),
        ("screen_beta", C.c_void_p),
        ("screen_is_active", C.c_void_p),
        ("active_set_size", C.c_int64),
        ("active_set", C.c_void_p),
        ("lmda", C.c_double),
        ("poll", POLL_FN),
        ("poll_user", C.c_void_p),
        ("glm_cb", C.POINTER(GlmCallbacks)),
        ("cov_v", C.c_void_p),
        ("rdev_tol", C.c_double),
        ("constraint_kind", C.c_void_p),
        ("constraint_a", C.c_void_p),
        ("constraint_b", C.c_void_p),
        ("constraint_mu", C.c_void_p),
        ("constraint_duals", C.c_void_p),
        ("constraint_cb", C.POINTER(ConstraintCallbacks)),
        ("constraint_native", C.c_void_p),
        ("constraint_va", C.c_void_p),
        ("constraint_vb", C.c_void_p),
        ("constraint_cfg", C.c_void_p),
        ("constraint_lin", C.c_void_p),
        ("constraint_vmu", C.c_void_p),
        ("penalty_l2", C.c_void_p),
        ("lmda_aug_ratios", C.POINTER(C.c_double)),
        ("n_lmda_aug", C.c_int64),
        ("lmda_aug_min", C.c_double),
    ]


# enum adelie_hip_vec / adelie_hip_scalar
V = dict(
    intercepts=0, devs=1, lmdas=2, lmda_path=3, screen_beta=4, grad=5, abs_grad=6, resid=7, eta=8,
    screen_X_means=9, screen_vars=10, screen_transforms=11,
    benchmark_screen=12, benchmark_fit_screen=13, benchmark_fit_active=14, benchmark_kkt=15,
    benchmark_invariance=16,
    betas_values=200, duals_values=201, constraint_mu=202, constraint_vmu=203,
)
I = dict(
    screen_set=100, screen_begins=101, screen_is_active=102, active_set=103, n_valid_solutions=104,
    active_sizes=105, screen_sizes=106, betas_indptr=107, betas_indices=108,
    duals_indptr=109, duals_indices=110, constraint_dev_groups=111,
)
S = dict(
    lmda_max=0, lmda=1, rsq=2, resid_sum=3, active_set_size=4, beta0=5, loss_null=6, loss_full=7,
    total_time=8,
    n_basil_iters=50, n_sweeps=51, n_cd_visits_screen=52, n_cd_visits_active=53, n_updates=54,
    n_irls_iters=55, n_new_screen_cols=56, n_cd_passes_screen=57, n_cd_passes_active=58,
    n_gram_col_reads=59, n_resid_col_reads=60, gram_flops=61, n_panel_blocks=62, n_panel_grams=63, n_panel_cols=64, n_irls_screen_cols=65, n_speculated=66, n_spec_rollbacks=67, n_sweeps_shared=68, n_update_cols=69, n_device_screens=70, n_host_screens=71, n_host_cons_visits=72, n_dev_cons_visits=73,
    t_sweep_ms=80, t_gram_ms=81, t_cd_ms=82, t_axpy_ms=83, n_sweep_launches=84, n_gram_launches=85,
    t_host_screen_ms=86, t_panel_step_ms=87, n_panel_step_launches=88, t_host_screen_wait_ms=89,
)

# every symbol include/adelie_hip.h declares (checked by tests/test_abi.py)
HIP_SYMBOLS = [
    "abi_version", "last_error", "device_count", "set_config",
    "design_create_dense", "design_create_sparse", "design_create_csc", "design_create_standardized", "design_adopt_dense_dev", "design_create_snp_unphased",
    "design_create_snp_calldata", "design_create_snp_bed", "design_alias", "design_create_slice", "design_create_multi", "design_create_derived", "design_create_concat", "design_impute", "design_destroy",
    "design_glm_path_losses", "design_multi_path_losses", "design_batch_stats", "design_rows", "design_cols", "design_dtype",
    "design_device", "design_stream",
    "design_cmul", "design_ctmul", "design_bmul", "design_btmul", "design_mul", "design_mul_batch", "design_cov",
    "design_sq_mul", "design_sp_tmul",
    "design_create_cov_dense", "design_create_cov_lazy", "design_cov_bmul", "design_cov_mul", "design_cov_to_dense", "gaussian_cov_solve",
    "grpnet_solve", "grpnet_solve_many", "result_destroy", "result_size", "result_copy", "result_scalar", "result_error", "result_sync",
    "bench_sweep",
]


def np_dtype(code):
    return np.float64 if code == F64 else np.float32


def dtype_code(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float64:
        return F64
    if dtype == np.float32:
        return F32
    raise RuntimeError("dtype must be either np.float32 or np.float64.")


# kept equal to ADELIE_HIP_ABI_VERSION in include/adelie_hip.h (tests/test_abi.py compares the two)
ABI_VERSION = 10


class Backend:
    """``libadelie_hip.so`` with the argument types of include/adelie_hip.h attached."""

    def __init__(self, path):
        self.path = path
        self.lib = C.CDLL(path)
        self._setup()
        # the structures below were written for exactly one layout of include/adelie_hip.h: a stale library that still
        # exports every symbol would misread the newer fields
        found = self.fn("abi_version")() if self.has("abi_version") else None
        if found != ABI_VERSION:
            raise RuntimeError(f"{path}: ABI version {found}, this package was written for {ABI_VERSION} "
                               "(ADELIE_HIP_ABI_VERSION, include/adelie_hip.h) -- rebuild the library.")

    def fn(self, name):
        return getattr(self.lib, "adelie_hip_" + name)

    def has(self, name):
        return hasattr(self.lib, "adelie_hip_" + name)

    def _setup(self):
        p, i64, dbl, vp, ci = C.POINTER, C.c_int64, C.c_double, C.c_void_p, C.c_int

        def sig(name, restype, argtypes):
            if self.has(name):
                f = self.fn(name)
                f.restype = restype
                f.argtypes = argtypes

        sig("abi_version", ci, [])
        sig("last_error", C.c_char_p, [])
        sig("device_count", ci, [])
        sig("set_config", ci, [C.c_char_p, dbl])
        sig("design_create_dense", ci, [vp, i64, i64, ci, ci, ci, p(vp)])
        sig("design_adopt_dense_dev", ci, [vp, i64, i64, ci, ci, ci, p(vp)])
        sig("design_create_sparse", ci, [vp, vp, vp, i64, i64, ci, ci, p(vp)])
        sig("design_create_csc", ci, [vp, vp, vp, vp, vp, vp, i64, i64, ci, ci, p(vp)])
        sig("design_create_standardized", ci, [vp, vp, vp, p(vp)])
        sig("design_create_snp_unphased", ci, [vp, i64, ci, ci, p(vp)])
        sig("design_create_snp_calldata", ci, [vp, i64, i64, vp, ci, ci, p(vp)])
        sig("design_create_snp_bed", ci, [vp, i64, i64, i64, ci, ci, p(vp)])
        sig("design_impute", ci, [vp, vp])
        sig("design_alias", ci, [vp, p(vp)])
        sig("design_create_slice", ci, [vp, i64, i64, i64, i64, p(vp)])
        sig("design_batch_stats", ci, [vp, p(dbl)])
        sig("design_create_multi", ci, [vp, i64, ci, p(vp)])
        sig("design_create_derived", ci, [vp, vp, i64, vp, i64, vp, vp, p(vp)])
        sig("design_create_concat", ci, [vp, i64, ci, p(vp)])
        sig("design_glm_path_losses", ci, [vp, ci, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp])
        sig("design_multi_path_losses", ci, [vp, ci, ci, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp])
        sig("design_destroy", ci, [vp])
        sig("design_rows", i64, [vp])
        sig("design_cols", i64, [vp])
        sig("design_dtype", ci, [vp])
        sig("design_device", ci, [vp])
        sig("design_stream", vp, [vp])
        sig("design_cmul", ci, [vp, i64, vp, vp, p(dbl)])
        sig("design_ctmul", ci, [vp, i64, dbl, vp])
        sig("design_bmul", ci, [vp, i64, i64, vp, vp, vp])
        sig("design_btmul", ci, [vp, i64, i64, vp, vp])
        sig("design_mul", ci, [vp, vp, vp, vp])
        sig("design_mul_batch", ci, [vp, vp, i64, vp])
        sig("design_cov", ci, [vp, i64, i64, vp, vp])
        sig("design_sq_mul", ci, [vp, vp, vp])
        sig("design_sp_tmul", ci, [vp, i64, vp, vp, vp, vp])
        sig("grpnet_solve", ci, [vp, p(GrpnetArgs), p(vp)])
        sig("grpnet_solve_many", ci, [p(vp), p(p(GrpnetArgs)), C.c_int32, p(vp), DONE_FN, vp])
        sig("gaussian_cov_solve", ci, [vp, p(GrpnetArgs), p(vp)])
        sig("design_create_cov_dense", ci, [vp, i64, ci, ci, ci, p(vp)])
        sig("design_create_cov_lazy", ci, [vp, p(vp)])
        sig("design_cov_bmul", ci, [vp, vp, i64, vp, vp, i64, vp])
        sig("design_cov_mul", ci, [vp, vp, vp, i64, vp])
        sig("design_cov_to_dense", ci, [vp, i64, i64, vp])
        sig("result_destroy", ci, [vp])
        sig("result_size", i64, [vp, ci])
        sig("result_copy", ci, [vp, ci, vp, i64])
        sig("result_scalar", dbl, [vp, ci])
        sig("result_error", C.c_char_p, [vp])
        sig("result_sync", ci, [vp])
        sig("bench_sweep", ci, [vp, i64, p(dbl)])

    def check(self, rc):
        if rc != 0:
            msg = self.fn("last_error")()
            raise RuntimeError(msg.decode() if msg else "native call failed")

    # -- result helpers ---------------------------------------------------------------------
    def result_vec(self, r, which, index=False):
        n = self.fn("result_size")(r, which)
        if n < 0:
            raise RuntimeError(f"unknown result vector {which}")
        out = np.empty(n, dtype=np.int64 if index else np.float64)
        if n:
            self.check(self.fn("result_copy")(r, which, out.ctypes.data, n))
        return out


_HIP = None


def hip_library_path():
    """In-tree ``libadelie_hip.so``; ``ADELIE_HIP_LIB`` names another build of the same sources (A/B runs of compile-time
    variants, ``scripts/ab.sh``) — still a HIP build of this library, never a fallback."""
    alt = os.environ.get("ADELIE_HIP_LIB")
    if alt:
        return alt if os.path.isabs(alt) else os.path.join(os.path.dirname(os.path.abspath(__file__)), alt)
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "libadelie_hip.so")


def hip_backend():
    """Load ``libadelie_hip.so``.  There is no CPU fallback: a missing library is an error."""
    global _HIP
    if _HIP is None:
        path = hip_library_path()
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  adelie_amd has no CPU fallback."
            )
        # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.  If torch is going to be
        # used next to this library (device tensors adopted by matrix.dense, torch.distributed in cv_grpnet),
        # it must initialise first so that libadelie_hip.so binds to the runtime that is already loaded.
        try:
            import torch  # noqa: F401
        except Exception:  # torch is optional plumbing
            pass
        _HIP = Backend(path)
    return _HIP


def ptr(a):
    return None if a is None else a.ctypes.data
