/*
 * adelie_hip.h — C ABI of libadelie_hip.so: the MI355X-native replacement for the
 * grpnet hot path of JamesYang007/adelie.
 *
 * The reference has no C ABI: its boundary is the pybind11 module `adelie.adelie_core`
 * (reference adelie/src/py_adelie_core.cpp:6-44).  Every entry point below names the
 * pybind class / method it stands in for (file:line relative to /root/reference).
 * Plain pointers and sizes only; no torch / Eigen / numpy types cross this boundary.
 *
 * Conventions
 *   - dtype: ADELIE_HIP_F32 / ADELIE_HIP_F64 is a property of the design matrix; every
 *     `const void*` / `void*` value array is in that dtype (value_t of the reference).
 *     Scalars cross as double.  Indices are int64 (Eigen::Index, state_base.hpp:33).
 *   - All array arguments are HOST pointers unless the name ends in `_dev`.
 *   - Return value 0 = ok; nonzero = construction / argument error, message via
 *     adelie_hip_last_error() (the reference throws adelie_core_error -> RuntimeError,
 *     state_base.ipp:15-92).  Errors raised INSIDE a solve do not fail the call: they are
 *     recorded in the result's error string and the partial path is returned, exactly as
 *     py_state.cpp:62-91 (`_solve`) does.
 */
#ifndef ADELIE_HIP_H
#define ADELIE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADELIE_HIP_ABI_VERSION 10

enum adelie_hip_dtype { ADELIE_HIP_F32 = 0, ADELIE_HIP_F64 = 1 };
enum adelie_hip_order { ADELIE_HIP_COL_MAJOR = 0, ADELIE_HIP_ROW_MAJOR = 1 };
enum adelie_hip_screen_rule { ADELIE_HIP_SCREEN_STRONG = 0, ADELIE_HIP_SCREEN_PIVOT = 1 };
/* glm_kind: which GlmBase implementation runs on device (glm_gaussian.ipp, glm_binomial.ipp) */
enum adelie_hip_glm_kind {
    ADELIE_HIP_GLM_GAUSSIAN = 0,        /* glm.gaussian(opt=True): StateGaussianNaive, no IRLS (solver.py:683-686) */
    ADELIE_HIP_GLM_BINOMIAL_LOGIT = 1,  /* glm.binomial(link="logit"): StateGlmNaive + IRLS */
    ADELIE_HIP_GLM_GAUSSIAN_IRLS = 2,   /* glm.gaussian(opt=False): Gaussian loss forced through StateGlmNaive */
    ADELIE_HIP_GLM_POISSON = 4,         /* glm.poisson (glm_poisson.ipp:14-58): StateGlmNaive + IRLS */
    ADELIE_HIP_GLM_BINOMIAL_PROBIT = 5, /* glm.binomial(link="probit") (glm_binomial.ipp:100-190) */
    ADELIE_HIP_GLM_CALLBACK = 6,        /* a GlmBase subclass written by the user (Python trampoline PyGlmBase, py_glm.cpp:8-92):
                                           StateGlmNaive + IRLS with gradient / hessian / loss evaluated by the host
                                           callbacks of adelie_hip_glm_callbacks on n-vectors, once per IRLS iteration */
    ADELIE_HIP_GLM_MULTINOMIAL = 3      /* glm.multinomial: StateMultiGlmNaive (solver_multiglm_naive.hpp) + IRLS; the design must
                                           be a multi-response view (adelie_hip_design_create_multi).  glm_y is (n, K) row-major,
                                           glm_weights is (n,), offsets / eta / resid are (n, K) row-major (glm_multinomial.ipp) */
};

/* Opaque handles. */
typedef struct adelie_hip_design adelie_hip_design; /* device-resident MatrixNaiveBase object */
typedef struct adelie_hip_result adelie_hip_result; /* solved state snapshot (the `state` copy _solve returns) */

/* ------------------------------------------------------------------------------------------
 * Library
 * ------------------------------------------------------------------------------------------ */
int         adelie_hip_abi_version(void);
/* Thread-local message of the last failing call on this thread. */
const char* adelie_hip_last_error(void);
/* Number of visible HIP devices (0 if none / runtime unavailable). */
int         adelie_hip_device_count(void);
/* Mirrors adelie.configs.set_configs (py_configs.cpp:6-49): names "hessian_min", "dbeta_tol".
 * One name has no counterpart upstream: "sweep_batch" (0/1, default 0).  When 1, solves that run concurrently (from
 * different host threads) on one dense matrix -- a design and its aliases, e.g. the folds of cv_grpnet -- share their
 * full-gradient sweeps: those that reach a sweep within a short window are answered by one pass over X.  The gradients are
 * then accumulated in another (fixed) order than the ordinary sweep's, i.e. they differ from it in the last bits.
 * Device memory: the working buffers of a finished solve are parked in a process-wide cache and handed to the next solve
 * instead of going through hipFree / hipMalloc; "pool_limit_mb" (default 6144; 0 = no caching) bounds what stays parked,
 * "pool_trim" (any value) returns the parked blocks to the driver. */
int         adelie_hip_set_config(const char* name, double value);

/* ------------------------------------------------------------------------------------------
 * Design matrix  == adelie.matrix.dense / MatrixNaiveDense{32,64}{C,F}
 *                   (matrix.py:549-680, matrix_naive_dense.ipp:9-21)
 * The reference holds a non-owning Eigen::Map of the caller's ndarray; here the matrix is
 * copied once into HBM (or adopted in place when it already lives there) and stays resident.
 * ------------------------------------------------------------------------------------------ */
/* Copy an (n,p) host matrix to device `device`. */
int adelie_hip_design_create_dense(const void* host, int64_t n, int64_t p, int dtype, int order,
                                   int device, adelie_hip_design** out);
/* Replaces MatrixNaiveSparse{32,64}F (adelie/matrix.py:1301-1385, matrix_naive_sparse.ipp): a CSC matrix (indptr p+1
 * int64, row indices int32, values of `dtype`; host pointers) becomes a resident dense design -- the entries are
 * scattered into zeroed column-major HBM once, every later operation is the dense kernel.  Duplicate entries add up. */
int adelie_hip_design_create_sparse(const int64_t* indptr, const int32_t* indices, const void* values, int64_t n, int64_t p,
                                    int dtype, int device, adelie_hip_design** out);
/* Replaces MatrixNaiveSparse{32,64}F with the matrix KEPT SPARSE in HBM (matrix_naive_sparse.ipp walks the CSC arrays per
 * operation, and so does this design): the stored entries are uploaded column-compressed (indptr p+1 int64, row indices int32
 * ascending and distinct inside a column, values of `dtype`) and row-compressed (row_indptr n+1, column indices, values: the
 * same entries, e.g. scipy's .tocsr()).  12 bytes per stored entry each way instead of n*p values (plus, for designs of more
 * than one tile of 16384 rows and at least a few entries per tile and column, a tile-major copy of 10 bytes per entry that
 * the library builds on the device for its full sweeps): a design whose dense form does not fit can run.  Gradients and Gram rows stream the CSC copy (one wavefront per column), residual updates and X beta
 * the CSR copy; grpnet_solve runs its full-Gram engines on Gaussian fits (the Gram is built once) and, under IRLS, the panel
 * engine over compressed columns, which also serves constrained fits (Gaussian fits with constraints are refused: the caller
 * expands the design).  REPRODUCIBILITY: the IRLS panel form applies the residual update of a block with hardware f64 atomics
 * over the stored entries (csc_panel_update_kernel); where two changed columns of a block share a row the order of the two
 * additions is not fixed, so two runs of a GLM path on a design kept sparse agree to rounding (1e-15 relative), not bit for bit
 * -- every other engine of this library is order-deterministic.  All MatrixNaiveBase operations below accept it; derived designs
 * are composed on the host. */
int adelie_hip_design_create_csc(const int64_t* indptr, const int32_t* indices, const void* values, const int64_t* row_indptr,
                                 const int32_t* row_indices, const void* row_values, int64_t n, int64_t p, int dtype, int device,
                                 adelie_hip_design** out);
/* Replaces MatrixNaiveStandardize (adelie/matrix.py:1414-1533, matrix_naive_standardize.ipp), the lazy wrapper: the view
 * (x_ij - centers[j]) / scales[j]  of `src` (dense, 2-bit SNP or kept sparse) with the resident matrix untouched and shared.
 * Over a design kept sparse every operation below applies the centring / scaling as a rank-one correction in its epilogue
 * (kernels_sparse.hip: centring would fill every cell).  Over a dense or 2-bit design the handle serves grpnet_solve only,
 * which composes the base design's kernels with those corrections and runs its full-Gram engines (a standardized 2-bit design
 * stays 2 bits per call instead of 8 bytes); its matrix operations are composed by the caller from the base design's
 * (adelie_amd.matrix does).  `src` must outlive the view; constraints and the multi-response view are not offered on it. */
int adelie_hip_design_create_standardized(adelie_hip_design* src, const double* centers, const double* scales,
                                          adelie_hip_design** out);
/* Adopt an (n,p) matrix that is ALREADY in device memory (e.g. a torch tensor's data_ptr);
 * not owned, must outlive the design. */
int adelie_hip_design_adopt_dense_dev(const void* dev_ptr, int64_t n, int64_t p, int dtype, int order,
                                      int device, adelie_hip_design** out);
/* == adelie.matrix.snp_unphased over an adelie.io.snp_unphased file image
 * (matrix.py:1245-1298, io_snp_unphased.ipp:10-41): `snpdat` is the whole .snpdat byte image;
 * it is decoded once into a dense 2-bit-per-call column-major device layout. */
int adelie_hip_design_create_snp_unphased(const void* snpdat, int64_t n_bytes, int dtype,
                                          int device, adelie_hip_design** out);
/* Same, from an int8 calldata matrix (n,p) column-major (values 0/1/2, negative = missing)
 * and per-column impute values (io_snp_unphased.ipp:70-303 semantics, without the file). */
int adelie_hip_design_create_snp_calldata(const int8_t* calldata, int64_t n, int64_t p,
                                          const double* impute, int dtype, int device,
                                          adelie_hip_design** out);
/* Same device layout straight from a PLINK 1 `.bed` image (SNP-major: magic 6c 1b 01, then p records of ceil(n/4)
 * bytes, 2 bits per sample, low bits first: 00 = two copies of allele A1, 10 = one, 11 = none, 01 = missing).  Calls become
 * A1 counts; missing calls are imputed with the column mean of the non-missing ones (the reference's default,
 * io/utils.hpp:10-31).  `bed` is host memory (the header is validated on the host).  SURVEY.md 8(f) rank 2: the on-disk format upstream of
 * adelie.io.snp_unphased in a genotype pipeline; both are 2 bits per call, so the record is transcoded on the device. */
int adelie_hip_design_create_snp_bed(const void* bed, int64_t n_bytes, int64_t n, int64_t p, int dtype, int device,
                                     adelie_hip_design** out);
/* A second handle on the same resident matrix with its own HIP stream and scratch space, so that independent solves
 * (the folds of cv_grpnet) can run concurrently from different host threads: one path leaves most of the chip idle
 * while its sequential block solves run, two or three paths interleave.  The alias must be destroyed before `src`. */
int adelie_hip_design_alias(adelie_hip_design* src, adelie_hip_design** out);
/* A rectangular slice  base[r0 : r0 + nr, c0 : c0 + nc]  as a design that SHARES the resident matrix (nothing is copied; own
 * stream and scratch; must be destroyed before `base`): what adelie.matrix.subset gives for a contiguous index range
 * (matrix_naive_subset.ipp wraps lazily as well; matrix.py:1538-1640).  Dense designs: any column range, row ranges that start on
 * a 16-byte boundary (r0 a multiple of 2 in f64, of 4 in f32); 2-bit SNP designs: column ranges over all rows.  Other designs
 * (sparse, views, covariance matrices) and other ranges are refused: the caller copies (adelie_hip_design_create_derived). */
int adelie_hip_design_create_slice(adelie_hip_design* base, int64_t r0, int64_t nr, int64_t c0, int64_t nc,
                                   adelie_hip_design** out);
/* Shared full-gradient sweeps of concurrent solves on this design and its aliases ("sweep_batch" config above), cumulative
 * since the design was created: out[0] = launches of the K-wide sweep kernel, out[1] = vectors they answered (= the ordinary
 * sweeps saved + launches), out[2] = their HIP-event time in ms on the batcher's stream.  Used by bench.py for the roofline of
 * cv_grpnet's dominant kernel (one launch streams the design once: n*p*sizeof(value) algorithmic bytes). */
int adelie_hip_design_batch_stats(adelie_hip_design* d, double* out);
/* A new design derived from a resident one (dense or SNP; a subset of an SNP design without centring / scaling is again a
 * 2-bit design, the selected calls re-packed four per byte; everything else is dense): rows `rows[0..n_rows)` (NULL: all), columns
 * `cols[0..n_cols)` (NULL: all), every resulting column j centred by centers[j] and divided by scales[j] (NULL: no centring /
 * scaling).  This is what adelie.matrix.subset (matrix_naive_subset.ipp) and adelie.matrix.standardize
 * (matrix_naive_standardize.ipp: X = (Z - 1 c^T) diag(s)^-1) describe; the reference wraps the parent lazily, here the result is
 * materialised in HBM by one kernel (matrix views).  The result does not reference `src`. */
int adelie_hip_design_create_derived(adelie_hip_design* src, const int64_t* rows, int64_t n_rows, const int64_t* cols,
                                     int64_t n_cols, const double* centers, const double* scales, adelie_hip_design** out);
/* Replaces MatrixNaiveCConcatenate / MatrixNaiveRConcatenate (adelie/matrix.py:214-310, matrix_naive_concatenate.ipp):
 * the k resident designs are copied side by side (axis 1: columns; axis 0: rows) into one new dense design; SNP sources
 * are decoded, except that 2-bit designs side by side (axis 1, all of them SNP) give a 2-bit design.  The sources stay valid and
 * independent.  The reference's error strings for mismatched shapes are kept. */
int adelie_hip_design_create_concat(adelie_hip_design* const* srcs, int64_t k, int axis, adelie_hip_design** out);
/* Multi-response view of a resident dense or 2-bit SNP design: the (n*K) x ((p + intercept)*K) matrix
 *     [ 1_n (x) I_K ,  X (x) I_K ]      (the first block only when `intercept` != 0)
 * that adelie/state.py:1100-1125 (_render_multi_inputs) builds from matrix.kronecker_eye / matrix.concatenate
 * (matrix_naive_kronecker_eye.ipp:27-47, matrix_naive_concatenate.ipp) and hands to StateMultiGaussianNaive.  Column j is
 * (extended feature j / K, response j % K); vectors over the rows are (n, K) row-major, as in the reference.  Nothing is
 * materialised: the view shares `base`'s matrix (which must outlive it) and owns a stream, scratch space and one column of
 * ones.  adelie_hip_grpnet_solve on the view runs the Gaussian naive solver with every kernel reading a column of X once
 * for all K responses (a 2-bit base stays 2-bit: its calls are decoded once per column for all K).  The matrix-op entry
 * points (cmul ... sp_tmul) are not offered on the view: the Python layer reaches them through `base`. */
int adelie_hip_design_create_multi(adelie_hip_design* base, int64_t K, int intercept, adelie_hip_design** out);
/* Copies the (p,) impute vector of an SNP design (as double). */
int adelie_hip_design_impute(adelie_hip_design* d, double* out);
int adelie_hip_design_destroy(adelie_hip_design* d);

/* cv_grpnet post-processing on the device (adelie/cv.py:296-312, diagnostic.py:30-121): for the L
 * sparse coefficient rows (CSR, int64 indices) computes eta_l = X beta_l + intercepts[l] + offsets and returns
 *   out[l]     = sum_i weights_a[i] * (A(eta_l,i) - y_i eta_l,i)        (the GLM's loss, glm_gaussian.ipp / glm_binomial.ipp)
 *   out[L + l] = the same under weights_b
 * without moving the (L, n) linear predictors to the host.  values / intercepts / offsets / y / weights in the design's dtype. */
int adelie_hip_design_glm_path_losses(adelie_hip_design* d, int glm_kind, int64_t L, const int64_t* indptr,
                                      const int64_t* indices, const void* values, const void* intercepts,
                                      const void* offsets, const void* y, const void* weights_a, const void* weights_b,
                                      double* out);
/* The same for multi-response fits (glm.multigaussian: glm_kind GAUSSIAN; glm.multinomial: MULTINOMIAL) on the BASE design:
 * row l of the CSR is a coefficient vector over the view columns feature*K + response (intercepts split off, as
 * state.betas holds them), intercepts is (L,K), offsets and y are (n,K) row-major, the weights (n,).  Replaces
 * diagnostic.predict through kronecker_eye + glm.loss per lambda in adelie/cv.py:281-314. */
int adelie_hip_design_multi_path_losses(adelie_hip_design* d, int glm_kind, int K, int64_t L, const int64_t* indptr,
                                        const int64_t* indices, const void* values, const void* intercepts,
                                        const void* offsets, const void* y, const void* weights_a, const void* weights_b,
                                        double* out);

int64_t adelie_hip_design_rows(const adelie_hip_design* d);   /* MatrixNaiveBase::rows */
int64_t adelie_hip_design_cols(const adelie_hip_design* d);   /* MatrixNaiveBase::cols */
int     adelie_hip_design_dtype(const adelie_hip_design* d);
int     adelie_hip_design_device(const adelie_hip_design* d);
/* Raw device pointer / HIP stream of the design (for callers that keep vectors on device). */
void*   adelie_hip_design_stream(const adelie_hip_design* d);

/* The MatrixNaiveBase virtuals (matrix_naive_base.hpp:18-144), host vectors in/out.
 * Semantics follow matrix_naive_dense.ipp line by line:                                   */
/* cmul  (:23-34):  *out = X[:,j] . (v * weights) */
int adelie_hip_design_cmul(adelie_hip_design* d, int64_t j, const void* v, const void* weights, double* out);
/* ctmul (:49-59):  out += v * X[:,j] */
int adelie_hip_design_ctmul(adelie_hip_design* d, int64_t j, double v, void* out);
/* bmul  (:61-80):  out = (v * weights)^T X[:, j:j+q] */
int adelie_hip_design_bmul(adelie_hip_design* d, int64_t j, int64_t q, const void* v, const void* weights, void* out);
/* btmul (:106-123): out += v^T X[:, j:j+q]^T */
int adelie_hip_design_btmul(adelie_hip_design* d, int64_t j, int64_t q, const void* v, void* out);
/* mul   (:125-146): out = (v * weights)^T X */
int adelie_hip_design_mul(adelie_hip_design* d, const void* v, const void* weights, void* out);
/* L sweeps in one call: out[l,:] = V[l,:]^T X for l < L; V is (L,n) and out (L,p), both row-major.  Replaces the loop of
 * X.mul calls in adelie/diagnostic.py:377-386 (gradients); the design (dense or 2-bit SNP) is streamed once per eight
 * vectors. */
int adelie_hip_design_mul_batch(adelie_hip_design* d, const void* V, int64_t L, void* out);
/* cov   (:162-197): out = X[:, j:j+q]^T diag(sqrt_weights^2) X[:, j:j+q], (q,q) column-major */
int adelie_hip_design_cov(adelie_hip_design* d, int64_t j, int64_t q, const void* sqrt_weights, void* out);
/* sq_mul (:199-217): out = weights^T X^2 */
int adelie_hip_design_sq_mul(adelie_hip_design* d, const void* weights, void* out);
/* sp_tmul (:219-256): out = V X^T with V an (L,p) CSR matrix, out (L,n) row-major */
int adelie_hip_design_sp_tmul(adelie_hip_design* d, int64_t L, const int64_t* indptr, const int64_t* indices,
                              const void* values, void* out);

/* ------------------------------------------------------------------------------------------
 * grpnet path solver
 *   == StateGaussianNaive{32,64}(...).solve(pb, exit_cond)   py_state.cpp:1068-1226
 *   == StateGlmNaive{32,64}(...).solve(glm, pb, exit_cond)   py_state.cpp:1556-1720
 * The struct carries exactly the constructor keyword arguments (py_state.cpp:1068-1154,
 * state_glm_naive.hpp:90-135); the Python sentinels are resolved by the caller the same way
 * adelie/state.py:1007-1045 resolves them (setup_lmda_max / setup_lmda_path / setup_loss_null).
 * ------------------------------------------------------------------------------------------ */
/* Polled once per saved lambda (the reference's exit_cond granularity, solver_base.hpp:581,679)
 * with `final`=1, and once per coordinate-descent fit with `final`=0 (the reference polls
 * PyErr_CheckSignals per CD sweep, py_state.cpp:70-74).  Return nonzero to stop:
 *   final=1 -> behaves as exit_cond() == True;  final=0 -> raises "interrupted" into error. */
typedef int (*adelie_hip_poll_fn)(void* user, int final, int64_t n_solutions, const adelie_hip_result* live);
/* `live` is the state being solved (the reference hands exit_cond the live C++ state, py_state.cpp:62-91): inside the callback
 * every adelie_hip_result_* accessor works on it -- lmdas, devs, intercepts, betas so far, screen / active sets, lmda, the
 * counters.  The device-resident invariants (grad, resid, eta, screen_beta, screen_X_means, screen_vars) are copied to the
 * host on request only: call adelie_hip_result_sync(live) first.  `live` must not be destroyed or kept by the callback. */
int adelie_hip_result_sync(const adelie_hip_result* live);

/* Host callbacks of a user-defined single-response GLM (glm_kind == ADELIE_HIP_GLM_CALLBACK).  They replace the virtuals of
 * GlmBase that the solver calls (glm_base.hpp:19-93; call sites solver_glm_naive.hpp:153,199-231,336-339,439-449): all arrays
 * are HOST pointers to n values of the design's dtype; return nonzero to abort the solve (recorded in the result's error
 * string).  `hessian` fills BOTH hess (GlmBase::hessian) and inv_hess_grad (GlmBase::inv_hessian_gradient, which a subclass may
 * override, glm_base.ipp:23-37) for the given eta and grad = gradient(eta).  loss_full crosses as the scalar in the args. */
typedef struct adelie_hip_glm_callbacks {
    void* user;
    int (*gradient)(void* user, const void* eta, void* grad);
    int (*hessian)(void* user, const void* eta, const void* grad, void* hess, void* inv_hess_grad);
    int (*loss)(void* user, const void* eta, double* loss);
} adelie_hip_glm_callbacks;

/* Host callbacks of the constraint objects the solver cannot run as a closed form on the device (constraint kind
 * ADELIE_HIP_CONSTRAINT_HOST): constraints on groups of several coefficients (the reference's ConstraintBox /
 * ConstraintOneSided proximal-Newton solvers, ConstraintLinear) and user-defined subclasses of ConstraintBase.  They stand
 * for the virtuals of ConstraintBase the solver calls (constraint_base.hpp:40-160; call sites
 * solver_gaussian_pin_naive.hpp:419-458, solver_base.hpp:62-93,158-222), exactly as the reference's PyConstraintBase
 * trampoline does (py_constraint.cpp).  `g` is the group, `d` its number of coefficients; all arrays are HOST doubles;
 * return nonzero to abort the solve.
 *   solve:      x (d) in/out in the coordinates of the group's eigenbasis Q (d x d, column-major), quad (d), linear (d)
 *   gradient:   out (d) = the constraint's term of the group's gradient at its current multipliers
 *   solve_zero: sets the multipliers that best explain v at x = 0; *norm = || v - gradient ||_2
 *   dual:       mu_out (m) = the object's multipliers, m = constraint_duals[g] */
typedef struct adelie_hip_constraint_callbacks {
    void* user;
    int (*solve)(void* user, int64_t g, int64_t d, double* x, const double* quad, const double* linear, double l1, double l2,
                 const double* Q);
    int (*gradient)(void* user, int64_t g, int64_t d, const double* x, double* out);
    int (*solve_zero)(void* user, int64_t g, int64_t d, const double* v, double* norm);
    int (*dual)(void* user, int64_t g, int64_t m, double* mu_out);
} adelie_hip_constraint_callbacks;
enum adelie_hip_constraint_kind {
    ADELIE_HIP_CONSTRAINT_NONE = 0,
    ADELIE_HIP_CONSTRAINT_BOX_1D = 1,       /* closed form on the device */
    ADELIE_HIP_CONSTRAINT_ONE_SIDED_1D = 2, /* closed form on the device */
    ADELIE_HIP_CONSTRAINT_HOST = 3          /* object on the caller's side, reached through adelie_hip_constraint_callbacks */
};

/* ABI 5.  Description of a `linear` constraint object (reference ConstraintLinear, adelie/constraint.py:137-306,
 * constraint_linear.ipp:142-217):  lower <= A z <= upper  on the group's d coefficients, m rows, with the settings of the
 * reference's proximal-Newton solver.  Like constraint_native / _va / _vb / _cfg it is read by the CPU checker (oracle/) only,
 * which runs such a group with its own restatement of that solver; libadelie_hip.so visits the group through constraint_cb. */
typedef struct adelie_hip_linear_constraint {
    int64_t        m, d;
    const double*  A;       /* (m, d) row-major */
    const double*  lower;   /* (m,) <= 0 (-1e100 and below: no bound) */
    const double*  upper;   /* (m,) >= 0 */
    const double*  vars;    /* (m,) squared row norms of A */
    double         cfg[7];  /* max_iters, tol, nnls_max_iters, nnls_tol, pinball_max_iters, pinball_tol, slack */
} adelie_hip_linear_constraint;

typedef struct adelie_hip_grpnet_args {
    /* ---- problem (static) ---- */
    int64_t        G;                 /* number of groups */
    const int64_t* groups;            /* (G,) start column of each group */
    const int64_t* group_sizes;       /* (G,) */
    double         alpha;
    const void*    penalty;           /* (G,) value_t */
    /* Gaussian (glm_kind == GAUSSIAN, the `opt` path of solver.py:683-686) */
    const void*    weights;           /* (n,) value_t, sums to 1 */
    const void*    X_means;           /* (p,) value_t */
    double         y_mean;
    double         y_var;
    double         resid_sum;
    double         rsq;
    /* GLM (glm_kind != GAUSSIAN): the GlmBase object is rebuilt on device from (y, weights) */
    int32_t        glm_kind;
    int32_t        _pad0;
    const void*    glm_y;             /* (n,) value_t */
    const void*    glm_weights;       /* (n,) value_t */
    const void*    offsets;           /* (n,) value_t */
    const void*    eta;               /* (n,) value_t */
    double         beta0;
    double         loss_null;         /* ignored if setup_loss_null */
    double         loss_full;
    int64_t        irls_max_iters;
    double         irls_tol;
    int32_t        setup_loss_null;
    int32_t        _pad1;
    /* shared dynamic inputs */
    const void*    resid;             /* (n,) value_t */
    const void*    grad;              /* (p,) value_t */
    /* ---- lambda path ---- */
    const void*    lmda_path;         /* (n_lmda_path,) value_t, may be NULL when setup_lmda_path */
    int64_t        n_lmda_path;
    double         lmda_max;          /* ignored if setup_lmda_max */
    double         min_ratio;
    int64_t        lmda_path_size;
    /* ---- configuration ---- */
    int64_t        max_screen_size;
    int64_t        max_active_size;
    double         pivot_subset_ratio;
    int64_t        pivot_subset_min;
    double         pivot_slack_ratio;
    int32_t        screen_rule;       /* adelie_hip_screen_rule */
    int32_t        early_exit;
    int64_t        max_iters;
    double         tol;
    double         adev_tol;
    double         ddev_tol;
    double         newton_tol;
    int64_t        newton_max_iters;
    int32_t        setup_lmda_max;
    int32_t        setup_lmda_path;
    int32_t        intercept;
    int32_t        n_threads;         /* accepted for API parity; host-side only */
    /* ---- warm-start invariants ---- */
    int64_t        screen_set_size;
    const int64_t* screen_set;        /* (s,) */
    int64_t        screen_beta_size;
    const void*    screen_beta;       /* (bs,) value_t */
    const int8_t*  screen_is_active;  /* (s,) */
    int64_t        active_set_size;
    const int64_t* active_set;        /* (G,) first active_set_size entries valid */
    double         lmda;              /* +inf for a cold start (solver.py:857) */
    /* ---- callbacks ---- */
    adelie_hip_poll_fn poll;          /* may be NULL */
    void*          poll_user;
    const adelie_hip_glm_callbacks* glm_cb; /* required iff glm_kind == ADELIE_HIP_GLM_CALLBACK */
    /* ---- covariance method (adelie_hip_gaussian_cov_solve only) ---- */
    const void*    cov_v;             /* (p,) value_t: the linear term v of 1/2 b'Ab - v'b */
    double         rdev_tol;          /* early exit on the relative change of the deviance (solver_gaussian_cov.hpp:183-201) */
    /* ---- per-group constraints (`constraints` of StateBase, state_base.hpp:60; adelie_core/constraint/) ----
     * Groups of ONE value with a box / one-sided constraint run as closed forms on the device (constraint_box.ipp:51-96,
     * constraint_one_sided.ipp:12-49); every other constraint object (several coefficients, linear, user-defined classes) is
     * kind 3: its group is visited on the host between two panel steps through constraint_cb (see above).  Accepted on every
     * state kind (naive, multi-response view, covariance method).
     *   kind 0: unconstrained;
     *   kind 1: box        constraint_a[i] <= beta_i <= constraint_b[i]   (a <= 0 <= b, infinities allowed); dual = mu_+ - mu_-
     *   kind 2: one-sided  constraint_a[i] * beta_i <= constraint_b[i]    (a = +-1, b >= 0);               dual = mu >= 0
     *   kind 3: host object; constraint_a / _b / _mu of the group are ignored.
     * NULL kind (or all zeros): no constraints. */
    const int32_t* constraint_kind;   /* (G,) */
    const void*    constraint_a;      /* (G,) value_t */
    const void*    constraint_b;      /* (G,) value_t */
    const void*    constraint_mu;     /* (G,) value_t or NULL: the multipliers the constraint objects hold on entry (warm start) */
    /* ABI 4 */
    const int64_t* constraint_duals;  /* (G,) number of multipliers of each group's constraint (ConstraintBase::duals; 0 = no
                                         constraint); NULL: one per constrained group.  Their running sum is `dual_groups`. */
    const adelie_hip_constraint_callbacks* constraint_cb; /* required when any kind is ADELIE_HIP_CONSTRAINT_HOST */
    /* What the CPU checker (oracle/) needs to run a kind-3 box / one-sided constraint with its OWN restatement of the
     * reference's solver instead of calling back (ABI 7: libadelie_hip.so solves 4 / 5 on the device from them too): constraint_native[g] = 0 (call back),
     * 4 (box: constraint_va = lower, constraint_vb = upper) or 5 (one-sided: va = sgn, vb = b), per COEFFICIENT (p,) arrays of
     * value_t indexed like the design's columns, and the solver settings (max_iters, tol, pinball_max_iters, pinball_tol,
     * slack) as 5 doubles per group. */
    const int32_t* constraint_native; /* (G,) or NULL */
    const void*    constraint_va;     /* (p,) value_t or NULL */
    const void*    constraint_vb;     /* (p,) value_t or NULL */
    const double*  constraint_cfg;    /* (G, 5) row-major or NULL */
    /* ABI 5: constraint_native[g] = 6 (linear): constraint_lin[g] describes the object (NULL entries elsewhere) */
    const adelie_hip_linear_constraint* const* constraint_lin; /* (G,) or NULL */
    /* ABI 7: kind-3 groups whose constraint_native is 4 (box) or 5 (one-sided) and that hold <= 64 coefficients are solved ON
     * THE DEVICE from constraint_va / _vb / _cfg (kernels_cons.hip: the reference's proximal-Newton dual solver in one
     * wavefront; no callback is made for them).  constraint_vmu: the multipliers those objects hold on entry, per COEFFICIENT
     * (p,) value_t, or NULL for zeros; what they hold on return is the result vector ADELIE_HIP_V_CONSTRAINT_VMU. */
    const void*    constraint_vmu;
    /* ABI 8: separate factors for the quadratic part of the penalty, (G,) value_t or NULL (= penalty): the objective's penalty
     * term becomes  lmda * sum_g (alpha * penalty[g] * |b_g| + (1 - alpha) / 2 * penalty_l2[g] * b_g^2).  Groups of one
     * coefficient only.  The reference has no such argument; it is what an elastic net on the standardized view
     * (Z - 1 c') diag(s)^-1 of a resident design needs to run on Z's own columns (penalty * |s|, penalty * s^2: every
     * coordinate update, screening score and KKT test is the same number in both coordinate systems, adelie_amd/solver.py). */
    const void*    penalty_l2;
    /* ABI 10: the lambda grid of a CV fold, assembled by the solve itself (adelie/cv.py:255-264 does it between two grpnet calls:
     * a first one with lmda_path_size = 0 for the fold's own lmda_max, then the path).  With lmda_path given
     * (setup_lmda_path = 0) and n_lmda_aug > 0 the solve, once its lmda_max is known, adds  lmda_max * lmda_aug_ratios[i]  for
     * every i whose product exceeds lmda_aug_min to the given path (all of them kept, sorted descending, cast to value_t) --
     * the numbers `state.lmda_max * np.logspace(0, log10(min_ratio), L)`, `> full_lmdas[0]`, `np.sort(np.concatenate(...))[::-1]`
     * produce; the first grpnet call (three sweeps over X and a state's worth of host work per fold) is not needed.  NULL / 0
     * for any other solve. */
    const double*  lmda_aug_ratios;
    int64_t        n_lmda_aug;
    double         lmda_aug_min;
} adelie_hip_grpnet_args;

/* Runs the whole path.  `*out` is always set on return code 0 (even when the solve recorded
 * an error string): the caller owns it and frees it with adelie_hip_result_destroy. */
int adelie_hip_grpnet_solve(adelie_hip_design* X, const adelie_hip_grpnet_args* args, adelie_hip_result** out);
/* ABI 10.  Replaces the fold loop of adelie.cv.cv_grpnet (adelie/cv.py:239-314, a Python `for fold in range(n_folds)` upstream)
 * for the solves it runs: `count` INDEPENDENT paths -- the folds' fits: the same matrix, their own weights / lambda grids /
 * warm starts in args[k] -- run concurrently, solve k on X[k], one host thread per solve below this call.  The handles are a
 * design and its aliases on one device (adelie_hip_design_alias: own stream and scratch each; with the "sweep_batch" config
 * on, the solves in flight share their full-gradient sweeps), or replicas on several devices of the node (each solve runs on
 * its handle's device; no collective is involved).  The same handle must not appear twice.  Callbacks in args[k] (poll, GLM,
 * constraints) are invoked from the solve's own thread, possibly several at once: a binding that needs a lock (the Python
 * binding: the interpreter lock, taken by ctypes) takes it there.  out[k] is what adelie_hip_grpnet_solve would have returned
 * for (X[k], args[k]) or NULL where that call failed; returns 0 when every solve returned 0, else 1 with the first failing
 * solve's message in adelie_hip_last_error.  `on_done` (may be NULL) is called from solve k's own thread as soon as that solve
 * has returned and out[k] is set -- rc = its return code -- while the other solves are still running: the caller's per-fold work
 * behind a path (cv.py:281-314: coefficients at the full-data grid, predictions, losses) overlaps with them instead of queueing
 * up behind the slowest fold. */
typedef void (*adelie_hip_done_fn)(int32_t k, int rc, void* user);
int adelie_hip_grpnet_solve_many(adelie_hip_design* const* X, const adelie_hip_grpnet_args* const* args, int32_t count,
                                 adelie_hip_result** out, adelie_hip_done_fn on_done, void* user);

/* ------------------------------------------------------------------------------------------
 * Covariance method  
 *   == adelie.matrix.dense(method="cov") / MatrixCovDense{32,64}{C,F}   (matrix_cov_dense.ipp:9-84)
 *   == StateGaussianCov{32,64}(...).solve(pb, exit_cond)                (py_state.cpp, state_gaussian_cov.hpp:40-145,
 *                                                                        solver_gaussian_cov.hpp:362-457)
 * minimises 1/2 b'Ab - v'b + penalty along a lambda path from summary statistics: A is a symmetric positive semi-definite
 * (p, p) matrix resident in HBM, the solver keeps grad = v - A b and never sees individual-level data.  The args struct is
 * the one of grpnet_solve; of it the covariance state reads the problem, path, configuration and warm-start fields,
 * `grad`, `rsq`, `cov_v` and `rdev_tol` (weights / X_means / resid / GLM fields are ignored; there is no intercept).
 * Result accessors as above: devs holds rsq (the unnormalised decrease of the loss), intercepts are zero.
 * ------------------------------------------------------------------------------------------ */
int adelie_hip_design_create_cov_dense(const void* host, int64_t p, int dtype, int order, int device, adelie_hip_design** out);
/* == adelie.matrix.lazy_cov(mat) / MatrixCovLazyCov{32,64}{C,F} (matrix_cov_lazy_cov.ipp): A = X^T X of a resident naive design
 * (dense or 2-bit SNP).  The reference computes rows of A on demand and caches them because A may not fit in host memory;
 * here the whole (p, p) matrix is formed once by the MFMA Gram kernel and kept in HBM (p = 100 000 is 80 GB of the 288),
 * after which it is an ordinary covariance-method design.  X is not retained. */
int adelie_hip_design_create_cov_lazy(adelie_hip_design* X, adelie_hip_design** out);
/* MatrixCovBase::bmul (matrix_cov_dense.ipp:23-41): out[j] = sum_i values[i] * A(indices[i], subset[j]) */
int adelie_hip_design_cov_bmul(adelie_hip_design* A, const int64_t* subset, int64_t n_subset, const int64_t* indices,
                               const void* values, int64_t n_indices, void* out);
/* MatrixCovBase::mul (:43-62): out = sum_i values[i] * A[indices[i], :] */
int adelie_hip_design_cov_mul(adelie_hip_design* A, const int64_t* indices, const void* values, int64_t n_indices, void* out);
/* MatrixCovBase::to_dense (:64-74): out = A[i:i+q, i:i+q], (q, q) column-major */
int adelie_hip_design_cov_to_dense(adelie_hip_design* A, int64_t i, int64_t q, void* out);
int adelie_hip_gaussian_cov_solve(adelie_hip_design* A, const adelie_hip_grpnet_args* args, adelie_hip_result** out);
int adelie_hip_result_destroy(adelie_hip_result* r);

/* ---- result accessors: the read-only properties of py_state.cpp:763-1040,1156-1217 ---- */
enum adelie_hip_vec {
    /* value vectors (copied out as double) */
    ADELIE_HIP_V_INTERCEPTS = 0, ADELIE_HIP_V_DEVS, ADELIE_HIP_V_LMDAS, ADELIE_HIP_V_LMDA_PATH,
    ADELIE_HIP_V_SCREEN_BETA, ADELIE_HIP_V_GRAD, ADELIE_HIP_V_ABS_GRAD, ADELIE_HIP_V_RESID,
    ADELIE_HIP_V_ETA, ADELIE_HIP_V_SCREEN_X_MEANS, ADELIE_HIP_V_SCREEN_VARS,
    ADELIE_HIP_V_SCREEN_TRANSFORMS, /* concatenated column-major (q,q) blocks in screen order */
    ADELIE_HIP_V_BENCHMARK_SCREEN, ADELIE_HIP_V_BENCHMARK_FIT_SCREEN, ADELIE_HIP_V_BENCHMARK_FIT_ACTIVE,
    ADELIE_HIP_V_BENCHMARK_KKT, ADELIE_HIP_V_BENCHMARK_INVARIANCE,
    /* index vectors (copied out as int64) */
    ADELIE_HIP_I_SCREEN_SET = 100, ADELIE_HIP_I_SCREEN_BEGINS, ADELIE_HIP_I_SCREEN_IS_ACTIVE,
    ADELIE_HIP_I_ACTIVE_SET, ADELIE_HIP_I_N_VALID_SOLUTIONS, ADELIE_HIP_I_ACTIVE_SIZES,
    ADELIE_HIP_I_SCREEN_SIZES, ADELIE_HIP_I_BETAS_INDPTR, ADELIE_HIP_I_BETAS_INDICES,
    /* duals, CSR (L, n_duals) (state_base.hpp `duals`, filled by sparsify_dual, solver_base.hpp:158-222); n_duals = number
     * of constrained groups here (one multiplier per singleton constraint) */
    ADELIE_HIP_I_DUALS_INDPTR, ADELIE_HIP_I_DUALS_INDICES,
    /* ABI 9: the groups whose box / one-sided constraint objects were solved ON THE DEVICE during this solve (ascending group
     * indices; their multipliers are the entries of ADELIE_HIP_V_CONSTRAINT_VMU) -- the binding reads this list instead of
     * re-deriving the library's rule, and the objects of every other group stayed live on the caller's side */
    ADELIE_HIP_I_CONSTRAINT_DEV_GROUPS,
    /* betas values (double), CSR (L, p) like convert_sparse_to_dense's input, py_state.cpp:9-60 */
    ADELIE_HIP_V_BETAS_VALUES = 200,
    ADELIE_HIP_V_DUALS_VALUES,
    /* (G,) the multiplier each constraint object is left holding when the solve returns (0 for unconstrained groups) */
    ADELIE_HIP_V_CONSTRAINT_MU,
    /* ABI 7: (p,) per coefficient, the multipliers of the box / one-sided objects that were solved on the device (0 elsewhere) */
    ADELIE_HIP_V_CONSTRAINT_VMU
};
enum adelie_hip_scalar {
    ADELIE_HIP_S_LMDA_MAX = 0, ADELIE_HIP_S_LMDA, ADELIE_HIP_S_RSQ, ADELIE_HIP_S_RESID_SUM,
    ADELIE_HIP_S_ACTIVE_SET_SIZE, ADELIE_HIP_S_BETA0, ADELIE_HIP_S_LOSS_NULL, ADELIE_HIP_S_LOSS_FULL,
    ADELIE_HIP_S_TOTAL_TIME,
    /* instrumentation used by bench.py for the algorithmic-byte count */
    ADELIE_HIP_S_N_BASIL_ITERS = 50, ADELIE_HIP_S_N_SWEEPS, ADELIE_HIP_S_N_CD_VISITS_SCREEN,
    ADELIE_HIP_S_N_CD_VISITS_ACTIVE, ADELIE_HIP_S_N_UPDATES, ADELIE_HIP_S_N_IRLS_ITERS,
    ADELIE_HIP_S_N_NEW_SCREEN_COLS, ADELIE_HIP_S_N_CD_PASSES_SCREEN, ADELIE_HIP_S_N_CD_PASSES_ACTIVE,
    ADELIE_HIP_S_N_GRAM_COL_READS, ADELIE_HIP_S_N_RESID_COL_READS, ADELIE_HIP_S_GRAM_FLOPS,
    ADELIE_HIP_S_N_PANEL_BLOCKS, ADELIE_HIP_S_N_PANEL_GRAMS, /* block visits / diagonal blocks built by the panel engine */
    ADELIE_HIP_S_N_PANEL_COLS, /* design columns streamed by the panel steps (gradient + residual update) */
    ADELIE_HIP_S_N_IRLS_SCREEN_COLS, /* sum over IRLS iterations of the screened columns (their means and variances are
                                        recomputed under every iteration's weights: the IRLS term of SURVEY.md 8d's B_path) */
    ADELIE_HIP_S_N_SPECULATED,       /* fits whose first active-set pass was enqueued behind the previous lambda's sweep */
    ADELIE_HIP_S_N_SPEC_ROLLBACKS,   /* ... of which were taken back (KKT failure, early exit, live-state read) */
    ADELIE_HIP_S_N_SWEEPS_SHARED,    /* full-gradient sweeps of this solve that were answered by a launch shared with other
                                        solvers on the same design (cv_grpnet folds in flight): X is streamed once for all */
    ADELIE_HIP_S_N_UPDATE_COLS,      /* design columns streamed by the residual updates of the panel steps (n_updates counts
                                        groups on grouped problems; this counts their columns) */
    ADELIE_HIP_S_N_DEVICE_SCREENS,   /* screening steps (solver_base.hpp:273-403) whose decision was taken on the device
                                        (kernels_screen.hip) ... */
    ADELIE_HIP_S_N_HOST_SCREENS,     /* ... and by the host routine (first iteration, host constraint objects, G > 2^18) */
    ADELIE_HIP_S_N_HOST_CONS_VISITS, /* ABI 7: visits of constrained groups made on the host through the callbacks ... */
    ADELIE_HIP_S_N_DEV_CONS_VISITS,  /* ... and by the device kernel (box / one-sided objects, kernels_cons.hip) */
    /* HIP-event time (ms) of the device phases on the design's stream, summed over the solve, and launch counts */
    ADELIE_HIP_S_T_SWEEP_MS = 80, ADELIE_HIP_S_T_GRAM_MS, ADELIE_HIP_S_T_CD_MS, ADELIE_HIP_S_T_AXPY_MS,
    ADELIE_HIP_S_N_SWEEP_LAUNCHES, ADELIE_HIP_S_N_GRAM_LAUNCHES, ADELIE_HIP_S_T_HOST_SCREEN_MS,
    /* per-launch timing of the panel step kernel; only collected when ADELIE_HIP_TIME_PANEL=1 (adds two events per launch) */
    ADELIE_HIP_S_T_PANEL_STEP_MS, ADELIE_HIP_S_N_PANEL_STEP_LAUNCHES,
    /* the part of T_HOST_SCREEN_MS the host spent waiting for the device (stream / event synchronisation) rather than
     * computing the screening rule, the appends and the launches of the new groups' variances */
    ADELIE_HIP_S_T_HOST_SCREEN_WAIT_MS
};
int64_t     adelie_hip_result_size(const adelie_hip_result* r, int which);
/* Copies min(size, cap) elements: value vectors as double, index vectors as int64. */
int         adelie_hip_result_copy(const adelie_hip_result* r, int which, void* out, int64_t cap);
double      adelie_hip_result_scalar(const adelie_hip_result* r, int which);
const char* adelie_hip_result_error(const adelie_hip_result* r);

/* ------------------------------------------------------------------------------------------
 * Kernel-level timing hook used by bench.py (HIP events on the design's own stream):
 * runs `reps` launches of the dominant kernel (the full gradient sweep grad = X^T(w*r) - rs*X_means
 * with fused per-group abs_grad) on resident buffers and returns the mean milliseconds per launch.
 * ------------------------------------------------------------------------------------------ */
int adelie_hip_bench_sweep(adelie_hip_design* d, int64_t reps, double* ms_per_launch);

#ifdef __cplusplus
}
#endif
#endif /* ADELIE_HIP_H */
