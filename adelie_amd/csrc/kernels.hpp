// kernels.hpp — launch interface of the gfx950 kernels (implemented in kernels_*.hip).
// Everything here works on DEVICE pointers and enqueues on the given HIP stream; no host sync.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace ahip {

// Dense column-major design in HBM: element (i,j) at X[i + j*ld].
template <class T>
struct DenseView {
    const T* X;
    int64_t n, p, ld;
};
// SNP design: 2-bit calls, column-major, 4 calls per byte (row i of column j in byte (i>>2) of
// the column, bits 2*(i&3)); code 0 -> 0, 1 -> 1, 2 -> 2, 3 -> missing (impute[j]).
struct SnpView {
    const uint8_t* bits;
    int64_t n, p, ldb; // ldb = bytes per column (padded)
};

// Sparse design kept sparse (kernels_sparse.hip): the same matrix column-compressed (cptr / cidx / cval, row indices ascending
// inside a column) and row-compressed (rptr / rcol / rval).
template <class T>
struct CscView {
    const int64_t* cptr;
    const int32_t* cidx;
    const T* cval;
    const int64_t* rptr;
    const int32_t* rcol;
    const T* rval;
    int64_t n, p, nnz;
    // row blocks of rb rows for the sweeps (nb > 1): bptr[c * (nb + 1) + b] = first entry of column c with row >= b * rb
    const int64_t* bptr;
    int nb;
    int64_t rb;
    // standardized view (both null: the plain matrix): the design is (x_ij - center[j]) * inv_scale[j], entries untouched
    const T* center;
    const T* inv_scale;
    // tile-major copy for the full sweeps (nt > 0): the entries of row tile t (th rows) stand together, by column;
    // tptr[t * (p + 1) + c] = first entry of (tile t, column c), trow = row inside the tile, tval = value
    const int64_t* tptr;
    const uint16_t* trow;
    const T* tval;
    int nt;
    int64_t th;
};

// ---- vector helpers -------------------------------------------------------------------------
// out[i] = a[i] * b[i]
template <class T> void launch_vmul(const T* a, const T* b, T* out, int64_t n, hipStream_t s);
template <class T> void launch_fill(T* out, T value, int64_t n, hipStream_t s);

// ---- sweep: out[c] = sum_i X[i, col(c)] * v[i]  (col(c) = cols ? cols[c] : c0 + c) --------------
// epilogue: out[c] -= sub_scale[0] * sub_vec[col(c)] when sub_vec != nullptr (sub_scale is a DEVICE scalar)
// square: use X^2 instead of X (sq_mul).  `work` must hold sweep_work_elems(n, ncols) elements.
template <class T>
void launch_sweep(const DenseView<T>& X, const T* v, T* out, int64_t c0, int64_t ncols, const int32_t* cols,
                  const T* sub_scale, const T* sub_vec, bool square, T* work, hipStream_t s);
template <class T>
void launch_sweep_snp(const SnpView& X, const T* impute, const T* v, T* out, int64_t c0, int64_t ncols,
                      const int32_t* cols, const T* sub_scale, const T* sub_vec, bool square, T* work, hipStream_t s);
int64_t sweep_work_elems(int64_t n, int64_t ncols);
// sparse design: one wavefront per (column, row block) over its stored entries; `work` holds sweep_work_elems_csc(X.nb, ncols)
int64_t sweep_work_elems_csc(int nb, int64_t ncols);
template <class T>
void launch_sweep_csc(const CscView<T>& X, const T* v, T* out, int64_t c0, int64_t ncols, const int32_t* cols,
                      const T* sub_scale, const T* sub_vec, bool square, T* work, hipStream_t s);
// Pieces of the standardized view (x_ij - center[j]) * inv_scale[j] over ANY base design (kernels_sparse.hip): the solver composes
// the base design's raw sweep / Gram / axpy with them.
// out[0] = sum_i v[i], fixed order; `out` holds kVecSumScratch elements (the result, then the chunk sums)
constexpr int kVecSumScratch = 8 + 256;
template <class T> void launch_vec_sum(const T* v, int64_t n, T* out, hipStream_t s);
template <class T>
void launch_std_sweep_epilogue(const T* center, const T* inv_scale, const T* raw, const T* raw_plain, const T* vsum, bool square,
                               T* out, int64_t c0, int64_t ncols, const int32_t* cols, const T* sub_scale, const T* sub_vec,
                               hipStream_t s);
template <class T>
void launch_std_gram_fix(const T* center, const T* inv_scale, T* C, int64_t ldc, int32_t M, int32_t pos0, int32_t N,
                         const int32_t* vcol, const T* m, const T* wsum, const T* xm, bool centered, hipStream_t s);
template <class T>
void launch_std_scale_coef(const T* center, const T* inv_scale, const int32_t* cols, const T* coef, const int32_t* count_dev,
                           int32_t count, T* coef2, T* kappa, hipStream_t s);
template <class T> void launch_vec_shift(T* out, int64_t n, const T* kappa, T sign, const int32_t* count_dev, hipStream_t s);
// panel engine on a standardized view: a block's gradient from the raw sums, a batch of diagonal blocks from the raw X' W X
template <class T>
void launch_std_fix_gblk(T* gblk, const int32_t* cols, int nb, const T* center, const T* inv_scale, const T* rsum_dev,
                         const T* xm_view_or_null, hipStream_t s);
struct SyrkBatch;
template <class T>
void launch_std_block_fix(T* D0, const SyrkBatch& sb, const int32_t* cols_base, int ldb, const T* center, const T* inv_scale,
                          const T* xm_view, const T* wsum_dev, bool centered, hipStream_t s);
// tile-major copy of a sparse design (see CscView): from the per-column tile pointers colptr[c * (nt + 1) + t] (what
// launch_csc_block_ptr gives with rb = th) and the tile-major offsets tptr, the entries are copied to their places
template <class T>
void launch_csc_tile_scatter(const int64_t* colptr, const int32_t* cidx, const T* cval, int64_t p, int nt, int64_t th,
                             const int64_t* tptr, uint16_t* trow, T* tval, hipStream_t s);
constexpr int64_t kCscTileBytes = 128 * 1024; // a tile of v in LDS
// row-block layout of a sparse design with n rows (blocks whose slice of an n-vector is about 1 MB, at most 64 of them) and
// the per-column block pointers
void csc_block_layout(int64_t n, size_t value_size, int* nb, int64_t* rb);
void launch_csc_block_ptr(const int64_t* cptr, const int32_t* cidx, int64_t p, int nb, int64_t rb, int64_t* bptr, hipStream_t s);
// launch_gram on a sparse design; `work` holds gram_work_elems_csc(n, M, N, X.nb) elements
int64_t gram_work_elems_csc(int64_t n, int64_t M, int64_t N, int nb);
template <class T>
void launch_gram_csc(const CscView<T>& X, const T* w, const int32_t* mcols, int32_t M, int32_t m_pos0, const int32_t* ncols,
                     int32_t N, int32_t n_pos0, const T* xm_by_col, bool center, T* C, int64_t ldc, T* work, hipStream_t s);
// launch_axpy_cols on a sparse design (cols must be distinct); `delta_zeroed`: p + 8 elements, the first p all zero on entry and on exit
template <class T>
void launch_axpy_cols_csc(const CscView<T>& X, const int32_t* cols, const T* coef, const int32_t* count_dev, int32_t count,
                          T sign, T* out, T* delta_zeroed, hipStream_t s);
// launch_sp_tmul on a sparse design; `work` holds sp_tmul_work_elems_csc(p) elements
int64_t sp_tmul_work_elems_csc(int64_t p);
template <class T>
void launch_sp_tmul_csc(const CscView<T>& X, int64_t L, const int64_t* indptr, const int64_t* indices, const T* values, T* out,
                        T* work, hipStream_t s);

// ---- panel axpy: out[i] += sign * sum_k coef[k] * X[i, cols[k]],  k < *count_dev (or count if count_dev null)
template <class T>
void launch_axpy_cols(const DenseView<T>& X, const int32_t* cols, const T* coef, const int32_t* count_dev,
                      int32_t count, T sign, T* out, hipStream_t s);
template <class T>
void launch_axpy_cols_snp(const SnpView& X, const T* impute, const int32_t* cols, const T* coef,
                          const int32_t* count_dev, int32_t count, T sign, T* out, hipStream_t s);
// batched variant for sp_tmul: out (L,n) row-major, CSR (indptr,indices,values) on device
template <class T>
void launch_sp_tmul(const DenseView<T>& X, int64_t L, const int64_t* indptr, const int64_t* indices, const T* values,
                    T* out, hipStream_t s);
template <class T>
void launch_sp_tmul_snp(const SnpView& X, const T* impute, int64_t L, const int64_t* indptr, const int64_t* indices,
                        const T* values, T* out, hipStream_t s);

// ---- Gram (MFMA): for a in [0,M), b in [0,N):
//   C[rowpos[a] + colpos[b]*ldc] = C[colpos[b] + rowpos[a]*ldc]
//        = sum_i w[i] X[i,mcols[a]] X[i,ncols[b]]  - (center ? xm[mcols[a]]*xm[ncols[b]] : 0)
// where rowpos[a] = m_pos0 + a, colpos[b] = n_pos0 + b.  `work` holds gram_work_elems(...) elements.
template <class T>
void launch_gram(const DenseView<T>& X, const T* w, const int32_t* mcols, int32_t M, int32_t m_pos0,
                 const int32_t* ncols, int32_t N, int32_t n_pos0, const T* xm_by_col, bool center, T* C, int64_t ldc,
                 T* work, hipStream_t s);
template <class T>
void launch_gram_snp(const SnpView& X, const T* impute, const T* w, const int32_t* mcols, int32_t M, int32_t m_pos0,
                     const int32_t* ncols, int32_t N, int32_t n_pos0, const T* xm_by_col, bool center, T* C,
                     int64_t ldc, T* work, hipStream_t s);
int64_t gram_work_elems(int64_t n, int64_t M, int64_t N);
// symmetric diagonal block of M <= 128 columns: C[a + b*ldc] = C[b + a*ldc] = sum_i w_i X[i,cols[a]] X[i,cols[b]] (- xm xm^T);
// only the lower-triangle MFMA tiles are computed.  `work` holds syrk_work_elems(n, M) elements.
// Several diagonal blocks per launch (syrk_batch_kernel): block y of the batch has the nb[y] columns cols_base[off[y]...] and
// goes to C_base + dst[y] (leading dimension ldc).  `work` holds syrk_batch_work_elems(n, count) elements.
struct SyrkBatch {
    static constexpr int MAX = 16;
    int32_t off[MAX];
    int32_t nb[MAX];
    int64_t dst[MAX];
    int32_t count;
};
int64_t syrk_batch_work_elems(int64_t n, int count);
// the panel engine over compressed columns (kernels_sparse.hip): the panel step (phase (A) as launch_panel_step; the block's gradient straight
// into gblk, centring included: no reduce launch) and the diagonal blocks of a batch (both triangles, leading dimension ldb)
template <class T>
void launch_panel_step_csc(const CscView<T>& X, const T* w, T* r, const int32_t* dcol, const T* dlt, const int32_t* nz_dev,
                           const int32_t* cols, int nb, const T* rsum_dev, const T* xm_by_col, T* gblk, hipStream_t s);
template <class T>
void launch_block_gram_csc(const CscView<T>& X, const T* w, const int32_t* cols_base, const SyrkBatch& sb, const T* xm_by_col,
                           bool center, T* D0, int ldb, hipStream_t s);
// Several rectangular blocks of at most 128 x 128 per launch (gram_batch_kernel; the cross blocks of the look-ahead passes):
// block y = rows cols_base[moff[y] ...] (m[y] of them) x columns cols_base[noff[y] ...] (nn[y]), X_rows^T W X_cols -
// xm_rows xm_cols^T into C_base + dst[y] (leading dimension ldc).  `work` holds gram_batch_work_elems(n, count) elements.
struct GramBatch {
    static constexpr int MAX = 16;
    int32_t moff[MAX], m[MAX], noff[MAX], nn[MAX];
    int64_t dst[MAX];
    int32_t count;
};
int64_t gram_batch_work_elems(int64_t n, int count);
template <class T>
void launch_gram_batch(const DenseView<T>& X, const T* w, const int32_t* cols_base, const GramBatch& b, const T* xm_by_col,
                       bool center, T* C_base, int64_t ldc, T* work, hipStream_t s);
template <class T>
void launch_gram_batch_snp(const SnpView& X, const T* impute, const T* w, const int32_t* cols_base, const GramBatch& b,
                           const T* xm_by_col, bool center, T* C_base, int64_t ldc, T* work, hipStream_t s);
// The new rows of a panel block (kernels_strip.hip): strip y = the m[y] <= 64 columns cols_base[voff[y] ...] against the c0n[y]
// columns cols_base[c0off[y] ...] of the previous block and the c1n[y] columns cols_base[c1off[y] ...] of the block itself
// (c0n + c1n <= 256; the new columns are the last m[y] of the own block, i.e. its members [row0, row0 + m)):
//   X_base[dstX[y] + (row0 + r) + c ldc]            = v_r^T W x_c - xm xm     c <  c0n   (rows of the cross block)
//   D_base[dstD[y] + (row0 + r) + c' ldc] and mirror = v_r^T W x_c' - xm xm   c' < c1n   (rows / columns of the diagonal block)
struct StripBatch {
    static constexpr int MAX = 8;
    int32_t voff[MAX], m[MAX], c0off[MAX], c0n[MAX], c1off[MAX], c1n[MAX], row0[MAX];
    int64_t dstX[MAX], dstD[MAX];
    int32_t count;
};
int strip_row_tiles(int m);                 // 16-row tiles the strip kernel is instantiated for (0: more rows than it takes)
int64_t strip_work_elems(int64_t n, int count, int m_max);
void set_strip_workgroups(int wgs);         // spread of the next strip builds launched by this host thread (default 512)
void set_strip_lds(bool on);                // f64: fragments through the wave-private LDS transpose (default) or straight from HBM
template <class T>
void launch_strip_batch(const DenseView<T>& X, const T* w, const int32_t* cols_base, const StripBatch& b, const T* xm_by_col,
                        bool center, T* D_base, T* X_base, int64_t ldc, T* work, hipStream_t s);
void set_small_gram_workgroups(int wgs); // kernels_gram.hip: spread of the next small builds launched by this host thread
template <class T>
void launch_syrk_batch(const DenseView<T>& X, const T* w, const int32_t* cols_base, const SyrkBatch& b, const T* xm_by_col,
                       bool center, T* C_base, int64_t ldc, T* work, hipStream_t s);
// xm_build != nullptr (blocks of 33-64 columns on a 2-bit design, syrk_batch_snp_brings_means): the build computes the weighted
// means of its own columns on the side, leaves them in xm_build (by design column) and centres with THEM instead of xm_by_col
template <class T>
void launch_syrk_batch_snp(const SnpView& X, const T* impute, const T* w, const int32_t* cols_base, const SyrkBatch& b,
                           const T* xm_by_col, bool center, T* C_base, int64_t ldc, T* work, hipStream_t s, T* xm_build = nullptr);
bool syrk_batch_snp_brings_means(const SyrkBatch& b);
template <class T>
void launch_syrk(const DenseView<T>& X, const T* w, const int32_t* cols, int32_t M, const T* xm_by_col, bool center, T* C,
                   int64_t ldc, T* work, hipStream_t s);
template <class T>
void launch_syrk_snp(const SnpView& X, const T* impute, const T* w, const int32_t* cols, int32_t M, const T* xm_by_col,
                       bool center, T* C, int64_t ldc, T* work, hipStream_t s);
int64_t syrk_work_elems(int64_t n, int64_t M);

// ---- abs_grad (solver_base.hpp:20-110) --------------------------------------------------------
// abs_grad[g] = || grad[groups[g] : +gs] - regul_g * beta_slot ||,  regul_g = (1-alpha)*lmda*penalty[g] for screen
// groups (slot[g] >= 0 gives the value offset into screen_beta), plain norm otherwise.
template <class T>
void launch_abs_grad(const T* grad, const int64_t* groups, const int64_t* group_sizes, int64_t G, const int32_t* slot,
                     const T* screen_beta, const T* penalty, T one_minus_alpha_lmda, T* abs_grad, hipStream_t s);
// the same when some groups of one coefficient carry box constraints clo[g] <= beta <= chi[g] (+-inf: none); cmu: the
// multipliers of the screen values, mu_out (G,): every group's multiplier afterwards
template <class T>
void launch_abs_grad_cons(const T* grad, const int64_t* groups, const int64_t* group_sizes, int64_t G, const int32_t* slot, const T* screen_beta,
                          const T* penalty, T one_minus_alpha_lmda, const T* clo, const T* chi, const T* cmu, T* abs_grad,
                          T* mu_out, hipStream_t s);

// ---- coordinate descent (pin solver) ----------------------------------------------------------
enum CdStatus : int32_t {
    CD_OK = 0, CD_MAX_CDS = 1, CD_MAX_ACTIVE = 2, CD_NEWTON = 3,
    // constraint objects solved on the device (kernels_cons.hip): the errors of constraint/utils.hpp:24-243 and its sub-solvers
    CD_CONS_PN = 4, CD_CONS_QP_BOX = 5, CD_CONS_QP_NNQP = 6, CD_CONS_UNEXPECTED = 7
};
// adelie_hip_grpnet_args::constraint_native
constexpr int32_t ADELIE_HIP_NATIVE_BOX = 4, ADELIE_HIP_NATIVE_ONE_SIDED = 5, ADELIE_HIP_NATIVE_LINEAR = 6;

template <class T>
struct CdScalars { // one instance in device memory, read back by the host after the kernel
    T rsq;
    T resid_sum;
    int64_t iters;
    int64_t n_visits_screen, n_visits_active, n_updates, n_passes_screen, n_passes_active;
    int32_t active_size;
    int32_t status;
    int32_t n_delta; // number of (col, delta) entries produced for the residual update
    int32_t _pad;
    int64_t dbg[8];  // cycle counters of the kernel's phases (only filled when built with -DAHIP_CD_PROFILE)
};

template <class T>
struct CdParams {
    int32_t nv, ns;          // screen values / screen groups
    const int32_t* sbegin;   // [ns]
    const int32_t* ssize;    // [ns]
    const T* spen;           // [ns] penalty per screen group
    const T* spen2 = nullptr; // [ns] or nullptr (= spen): factors of the quadratic part (adelie_hip_grpnet_args::penalty_l2, scalars only)
    const T* C;              // centred Gram, screen-value order
    int64_t ldc;
    const T* vars;           // [nv]
    const T* xmean;          // [nv]
    const T* V;              // concatenated (q,q) col-major eigenvector blocks
    const int64_t* voff;     // [ns]
    T* beta;                 // [nv] in/out
    T* g;                    // [nv] in/out: gradient x_a^T W r - xmean_a * resid_sum at the current beta
    int8_t* is_active;       // [ns]
    int32_t* active_set;     // [>= ns]
    T lmda, alpha, tol, newton_tol, dbeta_tol;
    int32_t newton_max_iters, max_active_size, intercept, all_scalar;
    int64_t max_iters;
    CdScalars<T>* sc;        // in: rsq, resid_sum, active_size; out: everything
    // residual-update list produced at exit: delta = beta - beta0 per changed value
    const T* beta0;          // [nv] snapshot of beta at fit entry
    const int32_t* vcol;     // [nv] design column of each screen value
    int32_t* dcols;          // [nv] out
    T* dvals;                // [nv] out
    int32_t max_group_size;
};
template <class T> void launch_cd(const CdParams<T>& p, hipStream_t s);

// ---- block Gauss-Seidel form of a lasso CD pass (kernels_cd_block.hip) ---------------------------
template <class T>
struct CdBlkState { // device-resident scalars carried from block to block and read by the host once per pass
    T rsq, resid_sum, cm;
    int64_t n_updates;
    int32_t active_size, status, nz, _pad;
};
template <class T>
struct CdBlkParams {
    int32_t nv;
    const T* C;
    int64_t ldc;
    const T* vars;
    const T* xmean;
    const T* spen;
    const T* spen2 = nullptr; // factors of the quadratic part, or nullptr (= spen): adelie_hip_grpnet_args::penalty_l2
    T* beta;
    T* g;
    int8_t* is_active;
    int32_t* active_set;
    T l1, l2;
    int32_t max_active_size;
    T* Dbuf;          // 2 * BLK * BLK
    T* dlt;           // BLK
    int32_t* didx;    // BLK
    CdBlkState<T>* st;
    const int32_t* list; // visiting list of the pass (nullptr: screen order 0..count-1)
    int32_t count;
    int32_t mark;
    int32_t bsz;         // visits per block, <= cd_block_size() (the D slot keeps leading dimension cd_block_size())
    // end-of-pass report straight into host-mapped memory (no stream synchronisation on the host side): the solve of block
    // `report_j` copies the state to host_st and then publishes `report_seq` in host_seq (system-scope release)
    CdBlkState<T>* host_st;
    int32_t* host_seq;
    int32_t report_j, report_seq;
    // panel (residual-based) variant, kernels_cd_panel.hip: gradient of the block from the panel step, diagonal block
    // from the cache, changed design columns out for the residual update
    const T* gblk;       // [BLK] gradient of the block's coordinates (list order)
    const T* Dptr;       // BLK x BLK block (ld = BLK)
    const int32_t* vcol; // screen value -> design column
    int32_t* dcol;       // [BLK] out: design columns of the changed coordinates (same order as dlt)
    // look-ahead form of the panel passes (solver.hip::run_panel_passes): gblk was computed from a residual that does not
    // contain the previous block's changes yet; the solve first subtracts Cprev[:, ppos[m]] * pdlt[m], m < *pnz, where
    // Cprev = X_b^T W X_prev - xbar_b xbar_prev^T (BLK x BLK, ld BLK, rows = this block).  Outputs for the next block /
    // the next panel step: block-local positions of the changed coordinates, their count, resid_sum after this block.
    const T* Cprev;
    const T* pdlt;
    const int32_t* ppos;
    const int32_t* pnz;
    int32_t* dpos;
    int32_t* nz_out;
    T* rsum_out;
    // `part` != nullptr: the gradient of the block is not in gblk yet but still in the slice partials of the panel step that
    // ran in the PREVIOUS launch (part[c * part_ld + k], k < part_n): the solve sums them itself in its prologue — in a fixed
    // order — and applies the intercept term  - part_rsum[0] * xbar_c  (what panel_reduce_kernel does as a launch of its own)
    const T* part;
    int64_t part_ld;
    int32_t part_n;
    const T* part_rsum; // nullptr: no intercept term
    // dense form of the changes, for the one-round-trip prologue of the fused look-ahead launch (blk_solve_la_body): every
    // solve of a look-ahead pass leaves  dd[i] = new - old  of its BLK coordinates (0 where nothing changed / beyond the block),
    // the next one reads the previous block's through pdd and forms the correction as Cprev * pdd without index loads
    const T* pdd;
    T* dd;
    // one-coefficient constraints (blk_solve_body<.., CONS = true>): bounds per screen value (-inf / +inf where there is none)
    // and, out, the multiplier mu_+ - mu_- of every constrained coordinate the block visited
    const T* clo;
    const T* chi;
    T* cmu;
};
// One visit of a group whose box / one-sided constraint object is solved on the device (kernels_cons.hip): the group is a block
// of its own; pointers are already offset to the group (its screen begin `b`, its first design column `col0`).
template <class T>
struct ConsVisitParams {
    int32_t q, ss, b, col0, native;       // group size, screen position, screen begin, first design column, NATIVE_BOX / _ONE_SIDED
    const T* gsrc;                        // [q] gradient of the group (panel reduce output, or the Gram engines' g + b)
    T* beta;                              // [q] screen_beta + b
    const T* vars;                        // [q] eigenvalues
    const T* V;                           // [q * q] eigenbasis, column-major (ignored for q = 1)
    const T* sxm;                         // [q] screen_X_means + b, or nullptr
    int8_t* is_active;                    // (whole array, indexed by ss)
    int32_t* active_set;
    CdBlkState<T>* st;
    double l1, l2, dbeta_tol;
    int32_t mark, first_of_pass, gram, max_active_size;
    int32_t* dcol; T* dlt;                // [q] out: the changes for the next residual update / Gram update
    const T* va; const T* vb;             // [q] box: lower, upper; one-sided: sgn, b
    T* mu;                                // [q] in/out: the object's multipliers
    double cfg[5];                        // max_iters, tol, pinball / nnqp max_iters, its tol, slack
    CdBlkState<T>* host_st; int32_t* host_seq; int32_t report_seq; // end-of-pass report (report_seq = 0: none)
    int64_t* n_visits;                    // device counter or nullptr
};
size_t cons_visit_lds(int q);
template <class T>
void launch_grp_cons_visit(const ConsVisitParams<T>& p, hipStream_t s);
// abs_grad (and, outside the screen set, the solve_zero multipliers) of the `count` groups in `list` with device constraint objects
template <class T>
void launch_cons_abs_grad(const int32_t* list, int count, const int32_t* native, const int64_t* groups, const int64_t* gsizes,
                          const int32_t* slot, const T* grad, const T* beta, const T* penalty, T regul_scale,
                          const T* va, const T* vb, T* mu, T* abs_grad, hipStream_t s);

// group (q > 1) variant: a block = consecutive groups of the visiting list with <= 128 values in total
template <class T>
struct CdGrpBlkParams {
    int32_t nv;
    const T* C;
    int64_t ldc;
    const T* vars;
    const T* xmean;
    T* beta;
    T* g;
    int8_t* is_active;
    int32_t* active_set;
    T l1, l2, newton_tol, dbeta_tol;
    int32_t newton_max_iters, max_active_size, mark;
    const T* V;
    const int64_t* voff;
    const T* spen;
    const int32_t* sbegin;
    const int32_t* ssize;
    const int32_t* blk_g0; // [nblk+1] first list position of each block
    const int32_t* list;   // visiting list of screen-group indices (nullptr: 0..)
    int32_t nblk;
    T* Dbuf;
    T* dlt;
    int32_t* didx;
    CdBlkState<T>* st;
    // panel (residual-based) variant: see CdBlkParams
    const T* gblk;       // [128] gradient of the block's values (block-local order)
    const T* Dptr;       // 128 x 128 slot, block-local order
    const int32_t* vcol; // screen value -> design column
    int32_t* dcol;       // [128] out: design columns of the changed values
    CdBlkState<T>* host_st;
    int32_t* host_seq;
    int32_t report_j, report_seq;
    int64_t* dbg;        // 8 cycle counters, only written by builds with -DAHIP_GRP_PROFILE
    // look-ahead form, see CdBlkParams (positions are block-local value indices)
    const T* Cprev;
    const T* pdlt;
    const int32_t* ppos;
    const int32_t* pnz;
    int32_t* dpos;
    int32_t* nz_out;
    T* rsum_out;
    // rot != 0 (panel variant only): Dptr holds the block in the eigen-coordinates of its groups, R^T D R with
    // R = blockdiag(V_g) (launch_grp_block_rotate, applied once per build): the sequential loop then works on rotated
    // gradients / coefficients throughout and the per-visit rotations of pin_naive:123-157 happen once per block and value,
    // lane-parallel, in the prologue and the epilogue of the solve
    int32_t rot;
    // rot != 0: per-block layout descriptors of the pass (launch_grp_layout, GDESC_* below: the value / group tables that
    // block_layout would otherwise derive through a chain of dependent global loads in every solve's prologue), and the
    // look-ahead correction in dense form: dd (out) / pdd (previous block's, in) = change of every value of the block by
    // block-local position (0 where unchanged or beyond the block), so that the correction is Cprev * pdd with no index loads
    const int32_t* desc;
    const T* pdd;
    T* dd;
    // fused look-ahead launch, rot != 0: the block's gradient is still in the slice partials the previous launch's step left
    // (slice-major, part[k * 128 + c], k < part_n) and the solve sums them itself in the second round trip of its prologue,
    // in a fixed order, then applies the intercept term  - part_rsum[0] * xbar  (as CdBlkParams::part)
    const T* part;
    int32_t part_n;
    const T* part_rsum;
    // fused look-ahead launch: the LAST step workgroup to finish sums the slice partials of the next block (slice-major) in a
    // fixed order, applies  - tail_rsum[0] * tail_xm[col]  and leaves the block's gradient in tail_g — in the shadow of the
    // solve, which outlasts the step in a group launch — instead of a panel_reduce launch between two fused launches.
    // tail_counter: two int32 that are 0 between launches (arrivals; finished tail workgroups: the last of those resets both); nullptr: off.
    int32_t* tail_counter;
    T* tail_g;
    const T* tail_rsum;
    const T* tail_xm; // nullptr: no intercept term
    // one-coefficient constraints of groups of size one (see CdBlkParams): per screen value, +-inf where there is none
    const T* clo;
    const T* chi;
    T* cmu;
    // Fused look-ahead launch, rot != 0: the look-ahead correction of the NEXT block is formed by THIS solve's otherwise idle
    // waves — they fetch the cross block Cnext = C_{j+1,j} while the visiting wave runs, multiply it with this block's dense
    // changes when those exist and leave the 128 sums in corr_out; the next solve then reads corr_in (1 KB) instead of pulling
    // 128 KB of cross block through its own CU at the head of its prologue.  Same products, same order of summation.
    const T* Cnext;   // nullptr: the next solve forms the correction itself from Cprev / pdd
    T* corr_out;
    const T* corr_in; // nullptr: as before
};
// layout descriptor of one block (int32 words)
constexpr int GDESC_VMAP = 0;    // [128] screen-value index of block value i
constexpr int GDESC_VGRP = 128;  // [128] group (of the block) of value i
constexpr int GDESC_VSS = 256;   // [128] screen-group index of value i
constexpr int GDESC_GOFF = 384;  // [129] first value of group k; [ng] = nval
constexpr int GDESC_GQ = 520;    // [128] size of group k
constexpr int GDESC_GSS = 648;   // [128] screen-group index of group k
constexpr int GDESC_NG = 776, GDESC_NVAL = 777;
constexpr int GDESC_VOFF = 784;  // [128] offset of the eigenbasis of value i's group in CdGrpBlkParams::V
constexpr int GDESC_VPOS = 912;  // [128] position of value i inside its group
constexpr int GDESC_VQ = 1040;   // [128] size of value i's group
constexpr int GDESC_STRIDE = 1168;
// descriptors of blocks 0..nblk-1 of the pass described by p (blk_g0, list, sbegin, ssize) into desc
template <class T> void launch_grp_layout(const CdGrpBlkParams<T>& p, int nblk, int32_t* desc, hipStream_t s);
// Dptr <- R^T Dsrc R for one 128 x 128 slot (ld 128; Dsrc == Dptr: in place): `ng` groups, group k = block values [goff[k], goff[k+1]) with eigenbasis
// (q, q) column-major at V + voff[k] (ignored for q == 1).  `scratch` holds 128 * 128 elements, private to the stream.
struct GrpRotArgs {
    int32_t ng;
    int32_t goff[129];
    int64_t voff[128];
};
template <class T> void launch_grp_block_rotate(T* Dptr, const T* Dsrc, const T* V, const GrpRotArgs& a, T* scratch, hipStream_t s);
template <class T> void launch_cd_group_block_pass(const CdGrpBlkParams<T>& p, hipStream_t s);
template <class T> void launch_cd_group_block_range(const CdGrpBlkParams<T>& p, int j0, int j1, hipStream_t s);
template <class T> void launch_cd_group_block_update(const CdGrpBlkParams<T>& p, int j, hipStream_t s);
// the visits of block j against p.gblk / p.Dptr (one workgroup)
template <class T> void launch_cd_group_panel_solve(const CdGrpBlkParams<T>& p, int j, hipStream_t s);
int cd_block_size();
// enqueues one whole pass (gather, then solve/update per block); the host reads st afterwards
template <class T> void launch_cd_block_pass(const CdBlkParams<T>& p, hipStream_t s);
// ---- panel form (kernels_cd_panel.hip): residual-based block passes ------------------------------------------------
// step: r -= sum_{m < *nz_dev} dlt[m] X[:, dcol[m]]; then part[c][slice] = X[slice, cols[c]] . (w*r)[slice] for c < nb.
// Returns the number of row slices.  `part` holds panel_part_elems(n) elements.  `slice_major`: part[slice * 128 + c] instead,
// the layout the stand-alone solve sums itself (CdBlkParams::part with part_ld == 0: no panel_reduce launch in between).
template <class T>
int launch_panel_step(const DenseView<T>& X, const T* w, T* r, const int32_t* dcol, const T* dlt, const int32_t* nz_dev,
                      const int32_t* cols, int nb, T* part, hipStream_t s, bool slice_major = false);
// `tail` (2-bit designs, the one-word-per-lane kernel only; *tailed says whether it was used): the step sums its own partials
// -- the last eight workgroups to finish take eight columns each once every workgroup's partials are out -- and leaves the
// block's gradient in tail->g (what panel_reduce would: - rsum[0] * xm[cols[c]] applied): no reduce launch behind the step.
// `counter` counts finished workgroups monotonically over the launches of one solver; `base` = its value before this launch.
// Means mode (xm_col != nullptr; IRLS with an intercept): phase (B) also accumulates sum_i x_ic w_i, the CURRENT weighted mean of
// every column it reads; the tail centres the gradient with it and leaves it where the solve and later steps look (by design
// column in xm_col, by screen value in sxm[list ? list[pos0 + c] : pos0 + c]) -- no sweep over the screen columns per IRLS
// iteration for the means.
template <class T>
struct StepTail {
    int32_t* counter;
    int32_t base;
    T* g;
    const T* rsum;
    const T* xm; // by design column, or nullptr (no intercept term)
    T* xm_col;           // means mode: out, by design column
    T* sxm;              // means mode: out, by screen value
    const int32_t* list; // the pass's visiting list (nullptr: screen order)
    int32_t pos0;        // list position of the block's first coordinate
};
// whether launch_panel_step_snp would take the kernel that can run a StepTail on this design
bool panel_step_snp_has_tail(const SnpView& X);
template <class T>
int launch_panel_step_snp(const SnpView& X, const T* impute, const T* w, T* r, const int32_t* dcol, const T* dlt,
                          const int32_t* nz_dev, const int32_t* cols, int nb, T* part, hipStream_t s, bool slice_major = false,
                          const StepTail<T>* tail = nullptr, bool* tailed = nullptr);
int64_t panel_part_elems(int64_t n);
// Opening of a look-ahead pass whose block-0 gradient already exists: g[c] = grad[cols[c]] (c < nb) out of the full gradient
// the invariance sweep left for the same residual, and both look-ahead residual-sum slots <- rsum_src[0] (nullptr: skip).
template <class T>
void launch_la_open_from_grad(const T* grad, const int32_t* cols, int nb, T* g, const T* rsum_src, T* rsum_out, hipStream_t s);
// gblk[c] = sum_slices part[c][.] - (xm_by_col ? rsum_dev[0] * xm_by_col[cols[c]] : 0)
template <class T>
void launch_panel_reduce(const T* part, int nslices, int nb, const int32_t* cols, const T* rsum_dev, const T* xm_by_col,
                         T* gblk, hipStream_t s);
template <class T>
void launch_panel_reduce_ld(const T* part, int64_t part_ld, int nslices, int nb, const int32_t* cols, const T* rsum_dev,
                            const T* xm_by_col, T* gblk, hipStream_t s);
// fused look-ahead step: the solve of block j (sp, as launch_cd_panel_solve) and a panel step in ONE launch; returns the
// leading dimension of `part` (>= number of row slices; use launch_panel_reduce_ld).  `part` holds 2*panel_part_elems(n).
// `tr`: slice-major partials part[k * 128 + c] (for a solve that sums them itself, CdBlkParams::part with part_ld == 0);
// the return value is then the number of partials per column.
template <class T>
int launch_panel_fused(const CdBlkParams<T>& sp, int j, const DenseView<T>& X, const T* w, T* r, const int32_t* dcol,
                       const T* dlt, const int32_t* nz_dev, const int32_t* cols, int nb, T* part, bool tr, hipStream_t s);
template <class T>
int launch_panel_fused_snp(const CdBlkParams<T>& sp, int j, const SnpView& X, const T* impute, const T* w, T* r,
                           const int32_t* dcol, const T* dlt, const int32_t* nz_dev, const int32_t* cols, int nb, T* part,
                           bool tr, hipStream_t s);
template <class T>
int launch_panel_fused_grp(const CdGrpBlkParams<T>& sp, int j, const DenseView<T>& X, const T* w, T* r, const int32_t* dcol,
                           const T* dlt, const int32_t* nz_dev, const int32_t* cols, int nb, T* part, bool tr, hipStream_t s);
template <class T>
int launch_panel_fused_grp_snp(const CdGrpBlkParams<T>& sp, int j, const SnpView& X, const T* impute, const T* w, T* r,
                               const int32_t* dcol, const T* dlt, const int32_t* nz_dev, const int32_t* cols, int nb,
                               T* part, bool tr, hipStream_t s);
// the visits of block j of the pass (one workgroup); p.gblk / p.Dptr / p.vcol / p.dcol must be set
template <class T> void launch_cd_panel_solve(const CdBlkParams<T>& p, int j, hipStream_t s);
template <class T> void launch_center_vars(T* vars, const T* xm, int cnt, bool center, hipStream_t s);
void launch_gather_i32(const int32_t* src, const int32_t* idx, int cnt, int32_t* out, hipStream_t s);
template <class T>
void launch_cd_compact(const T* beta, const T* beta0, const int32_t* vcol, int nv, int32_t* dcols, T* dvals,
                       int32_t* n_delta, hipStream_t s);

// ---- multi-response view (kernels_multi.hip) -------------------------------------------------------------------------
// The design [1 (x) I_K, X (x) I_K] (first block iff icpt) over a dense base X (nb x pb), never materialised.  Column
// j = u*K + l is (extended feature u, response l); u < icpt is the column of ones, u - icpt the base column otherwise.
// Row vectors (residual, weights) live on the device RESPONSE-MAJOR: element (i, l) at [l*nb + i] (K contiguous vectors of
// length nb), so that every kernel reads a slice of a column of X once and applies it to all K responses with unit-stride
// vector accesses; the C ABI's (n, K) row-major vectors are transposed once on the way in and out.
template <class T>
struct MultiView {
    const T* X;
    int64_t nb, pb, ld;
    const T* ones; // nb ones
    int32_t K, icpt;
    // 2-bit SNP base (matrix_naive_kronecker_eye.ipp over matrix_naive_snp_unphased.ipp): bits != nullptr, X unused — the
    // K-wide kernels decode the calls of a column once for all K responses (accessors.hpp::SnpOnesAcc)
    const uint8_t* bits = nullptr;
    int64_t ldb = 0;
    const T* impute = nullptr;
};
// out[u*K + l] = sum_i xcol(u)[i] * v[l*nb + i]  for every u in [0, pb + icpt): X is read ONCE for all K responses.
template <class T> void launch_multi_sweep(const MultiView<T>& X, const T* v, T* out, T* work, hipStream_t s);
template <class T> int64_t multi_sweep_work_elems(const MultiView<T>& X);
// the same over a 2-bit SNP design: out[u*K + l] = x_u . v[l*n + :]; work sized as for the view {nb = n, pb = p, icpt = 0, K}
template <class T>
void launch_multi_sweep_snp(const SnpView& X, const T* impute, int K, const T* v, T* out, T* work, hipStream_t s);
// panel step on view columns (same contract as launch_panel_step; `part` holds multi_panel_part_elems(nb) elements):
//   r -= sum_{m < *nz_dev} dlt[m] X'[:, dcol[m]];   part[c][slice] = X'[slice, cols[c]] . (w*r)[slice]  for c < nb_cols
// entries that share an extended feature (the K responses of a group) are served by one load of the column slice.
template <class T>
int launch_multi_panel_step(const MultiView<T>& X, const T* w, T* r, const int32_t* dcol, const T* dlt,
                            const int32_t* nz_dev, const int32_t* cols, int nb_cols, T* part, hipStream_t s);
int64_t multi_panel_part_elems(int64_t nb);
// out[:, response] += sign * coef[m] * xcol(feature)  for m < *cnt_dev   (rollback path)
template <class T>
void launch_multi_axpy_cols(const MultiView<T>& X, const int32_t* cols, const T* coef, const int32_t* cnt_dev, T sign,
                            T* out, hipStream_t s);
// symmetric M x M block over extended features ucols (MFMA syrk of kernels_gram.hip with the ones-aware accessor)
template <class T>
void launch_syrk_multi(const MultiView<T>& X, const T* w, const int32_t* ucols, int32_t M, T* C, int64_t ldc, T* work,
                       hipStream_t s);
// D[a + b*ldd] = (resp[a] == resp[b] && (lsel < 0 || resp[a] == lsel)) ? C[slot[a] + slot[b]*ldc] : (lsel <= 0 ? 0 : keep)
// for a, b < nv: expands a Gram block over extended features into the block over view columns (X (x) I_K has no entries
// between different responses).  lsel < 0: one weight vector for all responses; otherwise the call for response lsel.
template <class T>
void launch_multi_expand(const T* C, int64_t ldc, const int32_t* slot, const int32_t* resp, int nv, int lsel, T* D,
                         int64_t ldd, hipStream_t s);
// distinct extended features of a block's view columns (order of first appearance) -> ulist; per column its slot in
// ulist and its response.  nv <= 128.
void launch_multi_block_lists(const int32_t* cols, int nv, int K, int32_t* ulist, int32_t* slot, int32_t* resp,
                              hipStream_t s);
// fused look-ahead launch on the view: group solve of block j + multi-response step; returns the leading dimension of part
template <class T>
int launch_multi_panel_fused(const CdGrpBlkParams<T>& sp, int j, const MultiView<T>& X, const T* w, T* r,
                             const int32_t* dcol, const T* dlt, const int32_t* nz_dev, const int32_t* cols, int nb_cols,
                             T* part, hipStream_t s);
template <class T>
void launch_gram_multi(const MultiView<T>& X, const T* w, const int32_t* mcols, int32_t M, const int32_t* ncols, int32_t N,
                       T* C, int64_t ldc, T* work, hipStream_t s);
// C[a + b*ldc] = (resp_r[a] == resp_c[b]) ? G[slot_r[a] + slot_c[b]*ldg] : 0   for a < nr, b < nc
template <class T>
void launch_multi_expand_cross(const T* G, int64_t ldg, const int32_t* slot_r, const int32_t* resp_r, int nr,
                               const int32_t* slot_c, const int32_t* resp_c, int nc, T* C, int64_t ldc, hipStream_t s);
template <class T>
void launch_batch_pick(const T* src, int64_t nfeat, int K, int slot, const T* sub_scale, const T* sub_vec, T* out,
                       hipStream_t s);
// (n, K) row-major <-> response-major
template <class T> void launch_multi_to_major(const T* src, int64_t nb, int K, T* dst, hipStream_t s);
template <class T> void launch_multi_from_major(const T* src, int64_t nb, int K, T* dst, hipStream_t s);

// ---- GLM elementwise (solver_glm_naive.hpp:336-348, 439-449; glm_*.ipp) --------------------------
template <class T>
struct IrlsScalars {
    T hess_sum, y_mean, y_var, resid_sum, conv, loss;
};
// step 1: hess/irls_resid from (eta, resid); partial sums
// (implemented in kernels_glm.hip; declared there to keep this header small)

// ---- misc -------------------------------------------------------------------------------------
// copy a (rows x cols) block between column-major matrices with different leading dimensions
// covariance-method matrix (kernels_cov.hip)
template <class T>
void launch_cov_gather(const T* S, int64_t lda, int tr, const int32_t* vcol, int32_t nv, int32_t pos0, int32_t N, T* C,
                       int64_t ldc, hipStream_t s);
template <class T>
void launch_cov_bmul(const T* S, int64_t lda, int tr, const int64_t* subset, int64_t ns, const int64_t* indices,
                     const T* values, int64_t ni, T* out, hipStream_t s);
template <class T>
void launch_cov_mul(const T* S, int64_t lda, int64_t p, const int64_t* indices, const T* values, int64_t ni, T* out,
                    hipStream_t s);
template <class T>
void launch_cov_grad(const T* S, int64_t lda, int64_t p, const T* v, const int32_t* cols, const T* coef, const int32_t* cnt,
                     T* grad, hipStream_t s);
template <class T> void launch_copy2d(const T* src, int64_t lds, T* dst, int64_t ldd, int64_t rows, int64_t cols, hipStream_t s);
// vars[pos0 + a] = max(C[(pos0+a)*(ldc+1)], 0) for a < cnt   (gs == 1 groups)
// Appending new screen groups to the device mirrors (solver.hip::device_append_screen): ONE packed upload + one scatter
// launch instead of seven to ten small copies per lambda.  Packed image (8-byte aligned throughout):
//   T      pen[Ng], beta[Nv], (lo[Nv], hi[Nv], mu[Nv] when `cons`)
//   int32  begin[Ng], size[Ng], isact[Ng], group[Ng], vcol[Nv]
template <class T>
struct AppendDst {
    T* spen; T* beta; T* clo; T* chi; T* cmu;
    int32_t* sbegin; int32_t* ssize; int8_t* isact; int32_t* slot; int32_t* vcol;
    int32_t ns_old, nv_old, Ng, Nv, cons;
};
template <class T>
struct AppendImage { // offsets (bytes) of the arrays inside the packed image
    size_t pen, beta, lo, hi, mu, begin, size, isact, group, vcol, total;
    __host__ __device__ AppendImage(int Ng, int Nv, bool cons) {
        size_t o = 0;
        pen = o; o += sizeof(T) * size_t(Ng);
        beta = o; o += sizeof(T) * size_t(Nv);
        lo = hi = mu = o;
        if (cons) { lo = o; o += sizeof(T) * size_t(Nv); hi = o; o += sizeof(T) * size_t(Nv); mu = o; o += sizeof(T) * size_t(Nv); }
        o = (o + 7) & ~size_t(7);
        begin = o; o += 4 * size_t(Ng);
        size = o; o += 4 * size_t(Ng);
        isact = o; o += 4 * size_t(Ng);
        group = o; o += 4 * size_t(Ng);
        vcol = o; o += 4 * size_t(Nv);
        total = (o + 7) & ~size_t(7);
    }
};
template <class T> void launch_screen_append(const void* image_dev, const AppendDst<T>& d, hipStream_t s);
// Eigen-decomposition of new screen groups' Gram blocks on the device (kernels_eig.hip): group `i` of the launch reads the
// (q, q) block at src_base + src (leading dimension ld), writes its eigenvalues (ascending, clamped at 0) to
// vars[vars_pos .. + q) and its eigenvectors (column-major, in the columns) to V[v_off .. + q*q).  q == 1: the variance only.
struct EigDesc { int64_t src; int64_t vars_pos; int64_t v_off; int32_t ld; int32_t q; };
constexpr int kEigMaxQ = 96; // 2 * q * q doubles of LDS
template <class T>
void launch_grp_eig(const T* src_base, const EigDesc* desc_dev, int count, int max_q, T* vars, T* V, hipStream_t s);
template <class T> void launch_diag_vars(const T* C, int64_t ldc, int32_t pos0, int32_t cnt, T* vars, hipStream_t s);
// the diagonals of the blocks of a build batch into vars (by screen position; list == nullptr: positions base + off + i)
template <class T>
void launch_block_diag_vars(const T* D0, const SyrkBatch& sb, int ldb, int32_t base, const int32_t* list, T* vars, hipStream_t s);
// transpose row-major (n,p) into column-major with leading dimension ld
template <class T> void launch_transpose(const T* src, int64_t n, int64_t p, T* dst, int64_t ld, hipStream_t s);
// rows / columns of a 2-bit design re-packed as a 2-bit design (dst: pout columns of ldb_dst bytes)
void launch_snp_subset(const SnpView& X, int64_t nout, int64_t pout, const int64_t* rows, const int64_t* cols, uint8_t* dst,
                       int64_t ldb_dst, hipStream_t s);
template <class T> void launch_gather_cols(const T* src, const int64_t* cols, int64_t pout, T* out, hipStream_t s);
// dst (zeroed, column-major, leading dimension ld) += the entries of a CSC matrix
template <class T>
void launch_csc_scatter(const int64_t* indptr, const int32_t* indices, const T* values, int64_t n, int64_t p, T* dst,
                        int64_t ld, hipStream_t s);
// new dense matrix from a resident design: optional row / column gather, optional per-column centre and scale
template <class T>
void launch_derive_dense(const DenseView<T>& X, int64_t nout, int64_t pout, const int64_t* rows, const int64_t* cols,
                         const T* centers, const T* scales, T* dst, int64_t ldd, hipStream_t s);
template <class T>
void launch_derive_dense_snp(const SnpView& X, const T* impute, int64_t nout, int64_t pout, const int64_t* rows,
                             const int64_t* cols, const T* centers, const T* scales, T* dst, int64_t ldd, hipStream_t s);
// pack int8 calldata (n,p col-major) into 2-bit codes
void launch_pack_snp(const int8_t* calldata, int64_t n, int64_t p, uint8_t* bits, int64_t ldb, hipStream_t s);
// PLINK .bed records (p records of stride_in bytes, device memory) -> 2-bit codes; column means of the non-missing calls
void launch_bed_transcode(const uint8_t* bed, int64_t n, int64_t p, int64_t stride_in, uint8_t* bits, int64_t ldb,
                          hipStream_t s);
template <class T> void launch_snp_impute(const uint8_t* bits, int64_t n, int64_t p, int64_t ldb, T* impute, hipStream_t s);

} // namespace ahip
