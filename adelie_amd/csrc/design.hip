// design.hip — C ABI: library info, design-matrix handles and the MatrixNaiveBase operations.
// (include/adelie_hip.h documents which reference method each entry point replaces.)
#include <atomic>
#include "common.hpp"

#include <mutex>

using namespace ahip;

namespace {

thread_local std::string g_last_error;

inline int fail(const std::exception& e) {
    g_last_error = e.what();
    return 1;
}

template <class T>
T* scratch(DevBuf<char>& b, size_t n) {
    return reinterpret_cast<T*>(b.reserve((n ? n : 1) * sizeof(T)));
}

void set_device(const adelie_hip_design* d) { AHIP_CHECK(hipSetDevice(d->device)); }

void check_device(int device) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0)
        throw make_core_error("no HIP device is visible: adelie_hip has no CPU fallback.");
    if (device < 0 || device >= cnt) throw make_core_error("device ordinal out of range.");
    AHIP_CHECK(hipSetDevice(device));
}

template <class T>
void create_dense_t(adelie_hip_design* d, const void* src, bool src_on_device, int order) {
    constexpr int64_t kAlign = 32; // elements: columns start 128/256-byte aligned
    const int64_t n = d->n, p = d->p;
    if (src_on_device && order == ADELIE_HIP_COL_MAJOR) {
        d->X = const_cast<void*>(src);
        d->ld = n;
        d->owned = false;
        return;
    }
    const int64_t ld = ((n + kAlign - 1) / kAlign) * kAlign;
    T* X = nullptr;
    AHIP_CHECK(hipMalloc(reinterpret_cast<void**>(&X), size_t(ld) * size_t(p > 0 ? p : 1) * sizeof(T)));
    d->X = X;
    d->ld = ld;
    d->owned = true;
    if (order == ADELIE_HIP_COL_MAJOR) {
        AHIP_CHECK(hipMemcpy2DAsync(X, size_t(ld) * sizeof(T), src, size_t(n) * sizeof(T), size_t(n) * sizeof(T), size_t(p),
                                    hipMemcpyHostToDevice, d->stream));
        // zero the padding rows so that vector loads past n never see NaN payloads
        if (ld > n)
            AHIP_CHECK(hipMemset2DAsync(X + n, size_t(ld) * sizeof(T), 0, size_t(ld - n) * sizeof(T), size_t(p), d->stream));
    } else {
        // row-major source: stage (if on host) and transpose on device
        const T* rsrc = static_cast<const T*>(src);
        T* tmp = nullptr;
        if (!src_on_device) {
            AHIP_CHECK(hipMalloc(reinterpret_cast<void**>(&tmp), size_t(n) * size_t(p) * sizeof(T)));
            AHIP_CHECK(hipMemcpyAsync(tmp, src, size_t(n) * size_t(p) * sizeof(T), hipMemcpyHostToDevice, d->stream));
            rsrc = tmp;
        }
        AHIP_CHECK(hipMemsetAsync(X, 0, size_t(ld) * size_t(p) * sizeof(T), d->stream));
        launch_transpose<T>(rsrc, n, p, X, ld, d->stream);
        AHIP_CHECK(hipStreamSynchronize(d->stream));
        if (tmp) (void)hipFree(tmp);
    }
    AHIP_CHECK(hipStreamSynchronize(d->stream));
}

template <class T>
void create_sparse_t(adelie_hip_design* d, const int64_t* indptr, const int32_t* indices, const void* values) {
    constexpr int64_t kAlign = 32;
    const int64_t n = d->n, p = d->p;
    const int64_t nnz = indptr[p];
    const int64_t ld = ((n + kAlign - 1) / kAlign) * kAlign;
    T* X = nullptr;
    AHIP_CHECK(hipMalloc(reinterpret_cast<void**>(&X), size_t(ld) * size_t(p) * sizeof(T)));
    d->X = X;
    d->ld = ld;
    d->owned = true;
    AHIP_CHECK(hipMemsetAsync(X, 0, size_t(ld) * size_t(p) * sizeof(T), d->stream));
    int64_t* dptr = nullptr;
    int32_t* didx = nullptr;
    T* dval = nullptr;
    auto release = [&] { (void)hipFree(dptr); (void)hipFree(didx); (void)hipFree(dval); };
    try {
        AHIP_CHECK(hipMalloc(reinterpret_cast<void**>(&dptr), size_t(p + 1) * sizeof(int64_t)));
        AHIP_CHECK(hipMalloc(reinterpret_cast<void**>(&didx), size_t(std::max<int64_t>(nnz, 1)) * sizeof(int32_t)));
        AHIP_CHECK(hipMalloc(reinterpret_cast<void**>(&dval), size_t(std::max<int64_t>(nnz, 1)) * sizeof(T)));
        AHIP_CHECK(hipMemcpyAsync(dptr, indptr, size_t(p + 1) * sizeof(int64_t), hipMemcpyHostToDevice, d->stream));
        AHIP_CHECK(hipMemcpyAsync(didx, indices, size_t(nnz) * sizeof(int32_t), hipMemcpyHostToDevice, d->stream));
        AHIP_CHECK(hipMemcpyAsync(dval, values, size_t(nnz) * sizeof(T), hipMemcpyHostToDevice, d->stream));
        launch_csc_scatter<T>(dptr, didx, dval, n, p, X, ld, d->stream);
        AHIP_CHECK(hipStreamSynchronize(d->stream));
    } catch (...) {
        release();
        throw;
    }
    release();
}

// Designs alive in this process.  When the last one goes, the device blocks that finished solves parked for re-use (DevPool,
// up to 6 GB) are released: nothing of this library keeps device memory that another library in the process might want once
// the caller holds no design any more.
std::atomic<int>& live_designs() { static std::atomic<int> c{0}; return c; }

adelie_hip_design* new_design(int64_t n, int64_t p, int dtype, int device) {
    if (n <= 0 || p <= 0) throw make_core_error("matrix must have positive dimensions.");
    if (dtype != ADELIE_HIP_F32 && dtype != ADELIE_HIP_F64) throw make_core_error("dtype must be F32 or F64.");
    if (p >= (int64_t(1) << 31)) throw make_core_error("number of columns must fit in int32.");
    check_device(device);
    auto* d = new adelie_hip_design();
    d->dtype = dtype;
    d->device = device;
    d->n = n;
    d->p = p;
    AHIP_CHECK(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
    live_designs().fetch_add(1);
    return d;
}

void create_snp_from_calldata(adelie_hip_design* d, const int8_t* calldata, const double* impute) {
    const int64_t n = d->n, p = d->p;
    d->kind = 1;
    d->ldb = (((n + 3) / 4 + 127) / 128) * 128;
    AHIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d->bits), size_t(d->ldb) * size_t(p)));
    int8_t* tmp = nullptr;
    // stage the calldata in panels of columns to bound the temporary
    const int64_t panel = std::max<int64_t>(1, (int64_t(1) << 28) / n);
    AHIP_CHECK(hipMalloc(reinterpret_cast<void**>(&tmp), size_t(n) * size_t(std::min(panel, p))));
    for (int64_t j0 = 0; j0 < p; j0 += panel) {
        const int64_t pc = std::min(panel, p - j0);
        // hipMemcpyDefault: the calldata may live on the host or (generated in place) on this device
        AHIP_CHECK(hipMemcpyAsync(tmp, calldata + j0 * n, size_t(n) * size_t(pc), hipMemcpyDefault, d->stream));
        launch_pack_snp(tmp, n, pc, d->bits + j0 * d->ldb, d->ldb, d->stream);
        AHIP_CHECK(hipStreamSynchronize(d->stream));
    }
    (void)hipFree(tmp);
    if (d->dtype == ADELIE_HIP_F64) {
        AHIP_CHECK(hipMalloc(&d->impute, size_t(p) * sizeof(double)));
        AHIP_CHECK(hipMemcpy(d->impute, impute, size_t(p) * sizeof(double), hipMemcpyHostToDevice));
    } else {
        std::vector<float> f(impute, impute + p);
        AHIP_CHECK(hipMalloc(&d->impute, size_t(p) * sizeof(float)));
        AHIP_CHECK(hipMemcpy(d->impute, f.data(), size_t(p) * sizeof(float), hipMemcpyHostToDevice));
    }
}

// ---- host-vector matrix operations ----------------------------------------------------------------------
template <class T>
void op_sweep(adelie_hip_design* d, int64_t c0, int64_t ncols, const T* v, const T* w, T* out, bool square) {
    set_device(d);
    hipStream_t s = d->stream;
    const int64_t n = d->n;
    T* dv = scratch<T>(d->s_n1, n);
    T* dw = scratch<T>(d->s_n2, n);
    T* dout = scratch<T>(d->s_p1, ncols);
    T* work = scratch<T>(d->s_work, sweep_work_elems(n, ncols));
    AHIP_CHECK(hipMemcpyAsync(dv, v, n * sizeof(T), hipMemcpyHostToDevice, s));
    if (w) {
        AHIP_CHECK(hipMemcpyAsync(dw, w, n * sizeof(T), hipMemcpyHostToDevice, s));
        launch_vmul<T>(dv, dw, dv, n, s);
    }
    if (d->kind == 0)
        launch_sweep<T>(d->dense<T>(), dv, dout, c0, ncols, nullptr, nullptr, nullptr, square, work, s);
    else if (d->kind == 3)
        launch_sweep_csc<T>(d->csc<T>(), dv, dout, c0, ncols, nullptr, nullptr, nullptr, square,
                            scratch<T>(d->s_misc, size_t(sweep_work_elems_csc(d->sp_parts(), ncols))), s);
    else
        launch_sweep_snp<T>(d->snp(), static_cast<const T*>(d->impute), dv, dout, c0, ncols, nullptr, nullptr, nullptr,
                            square, work, s);
    AHIP_CHECK(hipMemcpyAsync(out, dout, ncols * sizeof(T), hipMemcpyDeviceToHost, s));
    AHIP_CHECK(hipStreamSynchronize(s));
}

// L sweeps at once (diagnostic.gradients, reference adelie/diagnostic.py:320-387: one X.mul per residual vector): the
// vectors go through the K-wide sweep eight at a time, so the design (dense or 2-bit SNP) is streamed once per eight vectors.
template <class T>
void op_mul_batch(adelie_hip_design* d, const T* V, int64_t L, T* out) {
    set_device(d);
    hipStream_t s = d->stream;
    const int64_t n = d->n, p = d->p;
    constexpr int64_t KB = 8;
    const bool is_dense = d->kind == 0;
    if (d->kind == 3) { // sparse: one pass over the stored entries per vector
        T* dv1 = scratch<T>(d->s_n1, size_t(n));
        T* dout1 = scratch<T>(d->s_p1, size_t(p));
        for (int64_t l = 0; l < L; ++l) {
            AHIP_CHECK(hipMemcpyAsync(dv1, V + l * n, size_t(n) * sizeof(T), hipMemcpyHostToDevice, s));
            launch_sweep_csc<T>(d->csc<T>(), dv1, dout1, 0, p, nullptr, nullptr, nullptr, false,
                                scratch<T>(d->s_misc, size_t(sweep_work_elems_csc(d->sp_parts(), p))), s);
            AHIP_CHECK(hipMemcpyAsync(out + l * p, dout1, size_t(p) * sizeof(T), hipMemcpyDeviceToHost, s));
            AHIP_CHECK(hipStreamSynchronize(s));
        }
        return;
    }
    T* dv = scratch<T>(d->s_n1, size_t(KB) * size_t(n));
    T* dout = scratch<T>(d->s_p1, size_t(KB) * size_t(p));
    // one work buffer for both kernels (a lone last vector goes through the single-vector sweep); the K-wide sweep's
    // work size only depends on (n, p, K)
    const MultiView<T> shape{nullptr, n, p, n, nullptr, int32_t(KB), 0};
    T* work = scratch<T>(d->s_work, std::max<size_t>(size_t(sweep_work_elems(n, p)), size_t(multi_sweep_work_elems<T>(shape))));
    std::vector<T> hout(size_t(KB) * size_t(p));
    for (int64_t l0 = 0; l0 < L; l0 += KB) {
        const int64_t K = std::min(KB, L - l0);
        AHIP_CHECK(hipMemcpyAsync(dv, V + l0 * n, size_t(K) * size_t(n) * sizeof(T), hipMemcpyHostToDevice, s));
        if (K == 1) {
            if (is_dense) launch_sweep<T>(d->dense<T>(), dv, dout, 0, p, nullptr, nullptr, nullptr, false, work, s);
            else launch_sweep_snp<T>(d->snp(), static_cast<const T*>(d->impute), dv, dout, 0, p, nullptr, nullptr, nullptr, false,
                                     work, s);
            AHIP_CHECK(hipMemcpyAsync(out + l0 * p, dout, size_t(p) * sizeof(T), hipMemcpyDeviceToHost, s));
            AHIP_CHECK(hipStreamSynchronize(s));
            continue;
        }
        if (is_dense) {
            const DenseView<T> X = d->dense<T>();
            launch_multi_sweep<T>(MultiView<T>{X.X, n, p, X.ld, nullptr, int32_t(K), 0}, dv, dout, work, s);
        } else {
            launch_multi_sweep_snp<T>(d->snp(), static_cast<const T*>(d->impute), int(K), dv, dout, work, s);
        }
        AHIP_CHECK(hipMemcpyAsync(hout.data(), dout, size_t(K) * size_t(p) * sizeof(T), hipMemcpyDeviceToHost, s));
        AHIP_CHECK(hipStreamSynchronize(s));
        for (int64_t l = 0; l < K; ++l) {
            T* o = out + (l0 + l) * p;
            for (int64_t u = 0; u < p; ++u) o[u] = hout[size_t(u) * size_t(K) + size_t(l)];
        }
    }
}

template <class T>
void op_axpy(adelie_hip_design* d, int64_t j, int64_t q, const T* coef, T* out) {
    set_device(d);
    hipStream_t s = d->stream;
    const int64_t n = d->n;
    T* dout = scratch<T>(d->s_n1, n);
    T* dcoef = scratch<T>(d->s_p1, q);
    int32_t* dcols = scratch<int32_t>(d->s_idx1, q);
    std::vector<int32_t> cols(q);
    for (int64_t k = 0; k < q; ++k) cols[k] = int32_t(j + k);
    AHIP_CHECK(hipMemcpyAsync(dout, out, n * sizeof(T), hipMemcpyHostToDevice, s));
    AHIP_CHECK(hipMemcpyAsync(dcoef, coef, q * sizeof(T), hipMemcpyHostToDevice, s));
    AHIP_CHECK(hipMemcpyAsync(dcols, cols.data(), q * sizeof(int32_t), hipMemcpyHostToDevice, s));
    if (d->kind == 0) {
        launch_axpy_cols<T>(d->dense<T>(), dcols, dcoef, nullptr, int32_t(q), T(1), dout, s);
    } else if (d->kind == 3) {
        T* delta = scratch<T>(d->s_misc, size_t(d->p) + 8);
        AHIP_CHECK(hipMemsetAsync(delta, 0, (size_t(d->p) + 8) * sizeof(T), s));
        launch_axpy_cols_csc<T>(d->csc<T>(), dcols, dcoef, nullptr, int32_t(q), T(1), dout, delta, s);
    } else {
        launch_axpy_cols_snp<T>(d->snp(), static_cast<const T*>(d->impute), dcols, dcoef, nullptr, int32_t(q), T(1), dout, s);
    }
    AHIP_CHECK(hipMemcpyAsync(out, dout, n * sizeof(T), hipMemcpyDeviceToHost, s));
    AHIP_CHECK(hipStreamSynchronize(s));
}

template <class T>
void op_cov(adelie_hip_design* d, int64_t j, int64_t q, const T* sw, T* out) {
    set_device(d);
    hipStream_t s = d->stream;
    const int64_t n = d->n;
    T* dw = scratch<T>(d->s_n1, n);
    T* dC = scratch<T>(d->s_p1, q * q);
    T* work = scratch<T>(d->s_work, d->kind == 3 ? gram_work_elems_csc(n, q, q, d->sp_parts()) : gram_work_elems(n, q, q));
    int32_t* dcols = scratch<int32_t>(d->s_idx1, q);
    std::vector<int32_t> cols(q);
    for (int64_t k = 0; k < q; ++k) cols[k] = int32_t(j + k);
    AHIP_CHECK(hipMemcpyAsync(dw, sw, n * sizeof(T), hipMemcpyHostToDevice, s));
    launch_vmul<T>(dw, dw, dw, n, s); // weights = sqrt_weights^2
    AHIP_CHECK(hipMemcpyAsync(dcols, cols.data(), q * sizeof(int32_t), hipMemcpyHostToDevice, s));
    if (d->kind == 0)
        launch_gram<T>(d->dense<T>(), dw, dcols, int32_t(q), 0, dcols, int32_t(q), 0, nullptr, false, dC, q, work, s);
    else if (d->kind == 3)
        launch_gram_csc<T>(d->csc<T>(), dw, dcols, int32_t(q), 0, dcols, int32_t(q), 0, nullptr, false, dC, q, work, s);
    else
        launch_gram_snp<T>(d->snp(), static_cast<const T*>(d->impute), dw, dcols, int32_t(q), 0, dcols, int32_t(q), 0,
                           nullptr, false, dC, q, work, s);
    AHIP_CHECK(hipMemcpyAsync(out, dC, q * q * sizeof(T), hipMemcpyDeviceToHost, s));
    AHIP_CHECK(hipStreamSynchronize(s));
}

// A = X^T X in column panels of 2048 columns (the Gram kernel's K-split partials stay bounded); unit weights, no centring
template <class T>
static void cov_lazy_t(adelie_hip_design* X, adelie_hip_design* A) {
    constexpr int64_t kAlign = 32, PANEL = 2048;
    const int64_t n = X->n, p = X->p;
    const int64_t ld = ((p + kAlign - 1) / kAlign) * kAlign;
    hipStream_t s = A->stream;
    T* C = nullptr;
    AHIP_CHECK(hipMalloc(reinterpret_cast<void**>(&C), size_t(ld) * size_t(p) * sizeof(T)));
    A->X = C;
    A->ld = ld;
    A->owned = true;
    AHIP_CHECK(hipMemsetAsync(C, 0, size_t(ld) * size_t(p) * sizeof(T), s));
    DevBuf<T> ones, work;
    DevBuf<int32_t> cols;
    std::vector<T> h1(size_t(n), T(1));
    std::vector<int32_t> hc(static_cast<size_t>(p));
    for (int64_t j = 0; j < p; ++j) hc[size_t(j)] = int32_t(j);
    ones.reserve(size_t(n));
    cols.reserve(size_t(p));
    ones.upload(h1.data(), size_t(n), s);
    cols.upload(hc.data(), size_t(p), s);
    work.reserve(size_t(X->kind == 3 ? gram_work_elems_csc(n, p, std::min<int64_t>(p, PANEL), X->sp_parts())
                                     : gram_work_elems(n, p, std::min<int64_t>(p, PANEL))));
    for (int64_t c0 = 0; c0 < p; c0 += PANEL) {
        const int64_t nc = std::min<int64_t>(PANEL, p - c0);
        if (X->kind == 0)
            launch_gram<T>(X->dense<T>(), ones.p, cols.p, int32_t(p), 0, cols.p + c0, int32_t(nc), int32_t(c0), nullptr, false, C, ld,
                           work.p, s);
        else if (X->kind == 3)
            launch_gram_csc<T>(X->csc<T>(), ones.p, cols.p, int32_t(p), 0, cols.p + c0, int32_t(nc), int32_t(c0), nullptr, false, C,
                               ld, work.p, s);
        else
            launch_gram_snp<T>(X->snp(), static_cast<const T*>(X->impute), ones.p, cols.p, int32_t(p), 0, cols.p + c0, int32_t(nc),
                               int32_t(c0), nullptr, false, C, ld, work.p, s);
    }
    AHIP_CHECK(hipStreamSynchronize(s));
}

template <class T>
void op_sp_tmul(adelie_hip_design* d, int64_t L, const int64_t* indptr, const int64_t* indices, const T* values, T* out) {
    set_device(d);
    hipStream_t s = d->stream;
    const int64_t n = d->n;
    const int64_t nnz = indptr[L];
    int64_t* dptr = scratch<int64_t>(d->s_idx1, L + 1);
    int64_t* dind = scratch<int64_t>(d->s_idx2, nnz);
    T* dval = scratch<T>(d->s_p1, nnz);
    AHIP_CHECK(hipMemcpyAsync(dptr, indptr, (L + 1) * sizeof(int64_t), hipMemcpyHostToDevice, s));
    if (nnz) {
        AHIP_CHECK(hipMemcpyAsync(dind, indices, nnz * sizeof(int64_t), hipMemcpyHostToDevice, s));
        AHIP_CHECK(hipMemcpyAsync(dval, values, nnz * sizeof(T), hipMemcpyHostToDevice, s));
    }
    // bound the device output panel to ~1 GiB
    const int64_t Lp = std::max<int64_t>(1, (int64_t(1) << 30) / int64_t(n * sizeof(T)));
    T* dout = scratch<T>(d->s_misc, size_t(std::min(Lp, L)) * n);
    for (int64_t l0 = 0; l0 < L; l0 += Lp) {
        const int64_t lc = std::min(Lp, L - l0);
        if (d->kind == 0) launch_sp_tmul<T>(d->dense<T>(), lc, dptr + l0, dind, dval, dout, s);
        else if (d->kind == 3)
            launch_sp_tmul_csc<T>(d->csc<T>(), lc, dptr + l0, dind, dval, dout, scratch<T>(d->s_work, sp_tmul_work_elems_csc(d->p)), s);
        else launch_sp_tmul_snp<T>(d->snp(), static_cast<const T*>(d->impute), lc, dptr + l0, dind, dval, dout, s);
        AHIP_CHECK(hipMemcpyAsync(out + l0 * n, dout, size_t(lc) * n * sizeof(T), hipMemcpyDeviceToHost, s));
        AHIP_CHECK(hipStreamSynchronize(s));
    }
}

} // namespace
namespace ahip {
template <class T>
void launch_glm_loss2(int kind, const T* y, const T* wa, const T* wb, const T* base, T b0, const T* off, int64_t n, T* sums,
                      hipStream_t s);
template <class T>
void launch_multi_loss2(int kind, const T* y, const T* wa, const T* wb, const T* base, const T* b0, const T* off, int64_t nb,
                        int K, T* sums, hipStream_t s);
}
namespace {

// Per-row losses of eta_l = X beta_l + intercepts[l] + offsets under two weight vectors, without moving eta to the host.
template <class T>
void op_path_losses(adelie_hip_design* d, int kind, int64_t L, const int64_t* indptr, const int64_t* indices,
                    const T* values, const T* intercepts, const T* offsets, const T* y, const T* wa, const T* wb,
                    double* out) {
    set_device(d);
    hipStream_t s = d->stream;
    const int64_t n = d->n;
    const int64_t nnz = indptr[L];
    int64_t* dptr = scratch<int64_t>(d->s_idx1, L + 1);
    int64_t* dind = scratch<int64_t>(d->s_idx2, nnz);
    T* dval = scratch<T>(d->s_p1, nnz);
    AHIP_CHECK(hipMemcpyAsync(dptr, indptr, (L + 1) * sizeof(int64_t), hipMemcpyHostToDevice, s));
    if (nnz) {
        AHIP_CHECK(hipMemcpyAsync(dind, indices, nnz * sizeof(int64_t), hipMemcpyHostToDevice, s));
        AHIP_CHECK(hipMemcpyAsync(dval, values, nnz * sizeof(T), hipMemcpyHostToDevice, s));
    }
    // n-vectors y, wa, wb, offsets and the reduction scratch share one buffer
    const size_t SUMS = 16 + 4 * 256;
    T* vec = scratch<T>(d->s_n1, size_t(4) * n + SUMS + size_t(2) * L);
    T *dy = vec, *dwa = vec + n, *dwb = vec + 2 * n, *doff = vec + 3 * n, *sums = vec + 4 * n, *dres = sums + SUMS;
    AHIP_CHECK(hipMemcpyAsync(dy, y, n * sizeof(T), hipMemcpyHostToDevice, s));
    AHIP_CHECK(hipMemcpyAsync(dwa, wa, n * sizeof(T), hipMemcpyHostToDevice, s));
    AHIP_CHECK(hipMemcpyAsync(dwb, wb, n * sizeof(T), hipMemcpyHostToDevice, s));
    AHIP_CHECK(hipMemcpyAsync(doff, offsets, n * sizeof(T), hipMemcpyHostToDevice, s));
    const int64_t Lp = std::max<int64_t>(1, (int64_t(1) << 30) / int64_t(n * sizeof(T)));
    T* dout = scratch<T>(d->s_misc, size_t(std::min(Lp, L)) * n);
    for (int64_t l0 = 0; l0 < L; l0 += Lp) {
        const int64_t lc = std::min(Lp, L - l0);
        if (d->kind == 0) launch_sp_tmul<T>(d->dense<T>(), lc, dptr + l0, dind, dval, dout, s);
        else if (d->kind == 3)
            launch_sp_tmul_csc<T>(d->csc<T>(), lc, dptr + l0, dind, dval, dout, scratch<T>(d->s_work, sp_tmul_work_elems_csc(d->p)), s);
        else launch_sp_tmul_snp<T>(d->snp(), static_cast<const T*>(d->impute), lc, dptr + l0, dind, dval, dout, s);
        for (int64_t l = 0; l < lc; ++l) {
            launch_glm_loss2<T>(kind, dy, dwa, dwb, dout + l * n, intercepts[l0 + l], doff, n, sums, s);
            AHIP_CHECK(hipMemcpyAsync(dres + 2 * (l0 + l), sums, 2 * sizeof(T), hipMemcpyDeviceToDevice, s));
        }
    }
    std::vector<T> res(size_t(2) * L);
    AHIP_CHECK(hipMemcpyAsync(res.data(), dres, res.size() * sizeof(T), hipMemcpyDeviceToHost, s));
    AHIP_CHECK(hipStreamSynchronize(s));
    for (int64_t l = 0; l < L; ++l) {
        out[l] = double(res[2 * l]);
        out[L + l] = double(res[2 * l + 1]);
    }
}

// The same for multi-response fits on the base design `d`: row l of the CSR is a coefficient vector over the view columns
// (feature*K + response, intercept columns already split off), intercepts (L, K), offsets / y (n, K) row-major.  Each row is
// split into its K per-response rows over the base features, so that one sp_tmul yields eta_l response-major.
template <class T>
void op_multi_path_losses(adelie_hip_design* d, int kind, int K, int64_t L, const int64_t* indptr, const int64_t* indices,
                          const T* values, const T* intercepts, const T* offsets, const T* y, const T* wa, const T* wb,
                          double* out) {
    set_device(d);
    hipStream_t s = d->stream;
    const int64_t n = d->n, p = d->p;
    const int64_t nnz = indptr[L];
    // per-response CSR rows (counting sort by response inside every row; the order of the features is kept)
    const size_t nz = size_t(nnz);
    std::vector<int64_t> ptr(size_t(L) * K + 1, 0), ind(nz);
    std::vector<T> val(nz);
    for (int64_t l = 0; l < L; ++l)
        for (int64_t e = indptr[l]; e < indptr[l + 1]; ++e) {
            if (indices[e] < 0 || indices[e] >= p * K) throw make_core_error("multi_path_losses: column index out of range.");
            ++ptr[size_t(l) * K + size_t(indices[e] % K) + 1];
        }
    for (size_t r = 0; r < size_t(L) * K; ++r) ptr[r + 1] += ptr[r];
    {
        std::vector<int64_t> fill(ptr.begin(), ptr.end() - 1);
        for (int64_t l = 0; l < L; ++l)
            for (int64_t e = indptr[l]; e < indptr[l + 1]; ++e) {
                const size_t r = size_t(l) * K + size_t(indices[e] % K);
                ind[size_t(fill[r])] = indices[e] / K;
                val[size_t(fill[r])] = values[e];
                ++fill[r];
            }
    }
    // (n, K) row-major -> response-major
    std::vector<T> ym(size_t(n) * K), om(size_t(n) * K);
    for (int64_t i = 0; i < n; ++i)
        for (int k = 0; k < K; ++k) {
            ym[size_t(k) * n + i] = y[i * K + k];
            om[size_t(k) * n + i] = offsets[i * K + k];
        }
    int64_t* dptr = scratch<int64_t>(d->s_idx1, size_t(L) * K + 1);
    int64_t* dind = scratch<int64_t>(d->s_idx2, nnz);
    T* dval = scratch<T>(d->s_p1, size_t(nnz) + size_t(L) * K);
    T* dicpt = dval + nnz;
    AHIP_CHECK(hipMemcpyAsync(dptr, ptr.data(), ptr.size() * sizeof(int64_t), hipMemcpyHostToDevice, s));
    if (nnz) {
        AHIP_CHECK(hipMemcpyAsync(dind, ind.data(), nnz * sizeof(int64_t), hipMemcpyHostToDevice, s));
        AHIP_CHECK(hipMemcpyAsync(dval, val.data(), nnz * sizeof(T), hipMemcpyHostToDevice, s));
    }
    AHIP_CHECK(hipMemcpyAsync(dicpt, intercepts, size_t(L) * K * sizeof(T), hipMemcpyHostToDevice, s));
    const size_t SUMS = 16 + 4 * 256;
    const size_t nK = size_t(n) * K;
    T* vec = scratch<T>(d->s_n1, 2 * nK + 2 * size_t(n) + SUMS + size_t(2) * L);
    T *dy = vec, *doff = vec + nK, *dwa = vec + 2 * nK, *dwb = dwa + n, *sums = dwb + n, *dres = sums + SUMS;
    AHIP_CHECK(hipMemcpyAsync(dy, ym.data(), nK * sizeof(T), hipMemcpyHostToDevice, s));
    AHIP_CHECK(hipMemcpyAsync(doff, om.data(), nK * sizeof(T), hipMemcpyHostToDevice, s));
    AHIP_CHECK(hipMemcpyAsync(dwa, wa, n * sizeof(T), hipMemcpyHostToDevice, s));
    AHIP_CHECK(hipMemcpyAsync(dwb, wb, n * sizeof(T), hipMemcpyHostToDevice, s));
    const int64_t Lp = std::max<int64_t>(1, (int64_t(1) << 30) / int64_t(nK * sizeof(T)));
    T* dout = scratch<T>(d->s_misc, size_t(std::min(Lp, L)) * nK);
    for (int64_t l0 = 0; l0 < L; l0 += Lp) {
        const int64_t lc = std::min(Lp, L - l0);
        if (d->kind == 0) launch_sp_tmul<T>(d->dense<T>(), lc * K, dptr + l0 * K, dind, dval, dout, s);
        else if (d->kind == 3)
            launch_sp_tmul_csc<T>(d->csc<T>(), lc * K, dptr + l0 * K, dind, dval, dout, scratch<T>(d->s_work, sp_tmul_work_elems_csc(d->p)), s);
        else launch_sp_tmul_snp<T>(d->snp(), static_cast<const T*>(d->impute), lc * K, dptr + l0 * K, dind, dval, dout, s);
        for (int64_t l = 0; l < lc; ++l) {
            launch_multi_loss2<T>(kind, dy, dwa, dwb, dout + size_t(l) * nK, dicpt + (l0 + l) * K, doff, n, K, sums, s);
            AHIP_CHECK(hipMemcpyAsync(dres + 2 * (l0 + l), sums, 2 * sizeof(T), hipMemcpyDeviceToDevice, s));
        }
    }
    std::vector<T> res(size_t(2) * L);
    AHIP_CHECK(hipMemcpyAsync(res.data(), dres, res.size() * sizeof(T), hipMemcpyDeviceToHost, s));
    AHIP_CHECK(hipStreamSynchronize(s));
    for (int64_t l = 0; l < L; ++l) {
        out[l] = double(res[2 * l]);
        out[L + l] = double(res[2 * l + 1]);
    }
}

// the multi-response view (kind 2) only serves grpnet_solve; its matrix ops go through the base design
void no_view(const adelie_hip_design* d) {
    if (d && d->std_center && d->kind != 3)
        throw make_core_error("the matrix operations of a standardized view over a dense or SNP design are composed by the caller "
                              "(adelie_amd.matrix); the handle only serves grpnet_solve.");
    if (d && d->kind == 2)
        throw make_core_error("this entry point is not offered on a multi-response view; use the base design.");
    if (d && d->cov)
        throw make_core_error("this entry point takes a design matrix, not a covariance matrix (matrix.dense(method=\"cov\")).");
}
void need_cov(const adelie_hip_design* d) {
    if (!d) throw make_core_error("null argument.");
    if (!d->cov) throw make_core_error("A must be a covariance matrix (matrix.dense(method=\"cov\")).");
}

void check_col(const adelie_hip_design* d, int64_t j, int64_t q, const char* what) {
    if (j < 0 || q < 0 || j + q > d->p) throw make_core_error(std::string(what) + "() is given inconsistent inputs!");
}

// .snpdat decoder (io_snp_unphased.ipp:10-68; layout in adelie_amd/io.py)
template <class U>
U read_as(const uint8_t* p) {
    U v;
    std::memcpy(&v, p, sizeof(U));
    return v;
}

} // namespace

#define ABI_TRY try {
#define ABI_CATCH                      \
    }                                  \
    catch (const std::exception& e) {  \
        return fail(e);                \
    }                                  \
    return 0;

#define DTYPE_DISPATCH(d, call64, call32)               \
    if ((d)->dtype == ADELIE_HIP_F64) { using T = double; (void)sizeof(T); call64; } \
    else { using T = float; (void)sizeof(T); call32; }

namespace {
template <class T>
void derive_t(adelie_hip_design* src, adelie_hip_design* d, const int64_t* rows, int64_t nrows, const int64_t* cols,
              int64_t ncols, const double* centers, const double* scales) {
    constexpr int64_t kAlign = 32;
    const int64_t nout = d->n, pout = d->p;
    if (src->kind == 1 && !centers && !scales) { // a subset of a 2-bit design stays 2 bits per call
        hipStream_t s = d->stream;
        d->kind = 1;
        d->ldb = (((nout + 3) / 4 + 127) / 128) * 128;
        AHIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d->bits), size_t(d->ldb) * size_t(pout)));
        AHIP_CHECK(hipMalloc(&d->impute, size_t(pout) * sizeof(T)));
        int64_t *drows = nullptr, *dcols = nullptr;
        if (rows) {
            drows = scratch<int64_t>(d->s_idx1, size_t(nrows));
            AHIP_CHECK(hipMemcpyAsync(drows, rows, size_t(nrows) * sizeof(int64_t), hipMemcpyHostToDevice, s));
        }
        if (cols) {
            dcols = scratch<int64_t>(d->s_idx2, size_t(ncols));
            AHIP_CHECK(hipMemcpyAsync(dcols, cols, size_t(ncols) * sizeof(int64_t), hipMemcpyHostToDevice, s));
        }
        AHIP_CHECK(hipStreamSynchronize(src->stream));
        launch_snp_subset(src->snp(), nout, pout, drows, dcols, d->bits, d->ldb, s);
        launch_gather_cols<T>(static_cast<const T*>(src->impute), dcols, pout, static_cast<T*>(d->impute), s);
        AHIP_CHECK(hipStreamSynchronize(s));
        return;
    }
    const int64_t ld = ((nout + kAlign - 1) / kAlign) * kAlign;
    T* X = nullptr;
    AHIP_CHECK(hipMalloc(reinterpret_cast<void**>(&X), size_t(ld) * size_t(pout) * sizeof(T)));
    d->X = X;
    d->ld = ld;
    d->owned = true;
    d->kind = 0;
    hipStream_t s = d->stream;
    AHIP_CHECK(hipMemsetAsync(X, 0, size_t(ld) * size_t(pout) * sizeof(T), s));
    int64_t *drows = nullptr, *dcols = nullptr;
    T *dc = nullptr, *ds = nullptr;
    std::vector<T> hc, hs;
    if (rows) {
        drows = scratch<int64_t>(d->s_idx1, size_t(nrows));
        AHIP_CHECK(hipMemcpyAsync(drows, rows, size_t(nrows) * sizeof(int64_t), hipMemcpyHostToDevice, s));
    }
    if (cols) {
        dcols = scratch<int64_t>(d->s_idx2, size_t(ncols));
        AHIP_CHECK(hipMemcpyAsync(dcols, cols, size_t(ncols) * sizeof(int64_t), hipMemcpyHostToDevice, s));
    }
    if (centers) {
        hc.assign(centers, centers + pout);
        dc = scratch<T>(d->s_p1, size_t(pout));
        AHIP_CHECK(hipMemcpyAsync(dc, hc.data(), size_t(pout) * sizeof(T), hipMemcpyHostToDevice, s));
    }
    if (scales) {
        hs.assign(scales, scales + pout);
        ds = scratch<T>(d->s_misc, size_t(pout));
        AHIP_CHECK(hipMemcpyAsync(ds, hs.data(), size_t(pout) * sizeof(T), hipMemcpyHostToDevice, s));
    }
    // the source's stream may still be writing it (e.g. its own creation): order after it
    AHIP_CHECK(hipStreamSynchronize(src->stream));
    if (src->kind == 3) throw make_core_error("derived designs of a sparse design are composed on the host (adelie_amd.matrix).");
    if (src->kind == 0) launch_derive_dense<T>(src->dense<T>(), nout, pout, drows, dcols, dc, ds, X, ld, s);
    else launch_derive_dense_snp<T>(src->snp(), static_cast<const T*>(src->impute), nout, pout, drows, dcols, dc, ds, X, ld, s);
    AHIP_CHECK(hipStreamSynchronize(s));
}
// concatenation of resident designs along the columns (axis 1) or the rows (axis 0): every source is copied (SNP sources
// decoded) into its slice of a new dense array
template <class T>
void concat_t(adelie_hip_design* const* srcs, int64_t k, int axis, adelie_hip_design* d) {
    constexpr int64_t kAlign = 32;
    const int64_t nout = d->n, pout = d->p;
    bool all_snp = axis == 1;
    for (int64_t m = 0; m < k && all_snp; ++m) all_snp = srcs[m]->kind == 1 && !srcs[m]->std_center && srcs[m]->ldb == srcs[0]->ldb;
    if (all_snp) { // 2-bit designs side by side stay a 2-bit design: their columns are copied as they are
        hipStream_t s = d->stream;
        d->kind = 1;
        d->ldb = srcs[0]->ldb;
        AHIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d->bits), size_t(d->ldb) * size_t(pout)));
        AHIP_CHECK(hipMalloc(&d->impute, size_t(pout) * sizeof(T)));
        int64_t off = 0;
        for (int64_t m = 0; m < k; ++m) {
            adelie_hip_design* src = srcs[m];
            AHIP_CHECK(hipStreamSynchronize(src->stream));
            AHIP_CHECK(hipMemcpyAsync(d->bits + off * d->ldb, src->bits, size_t(src->ldb) * size_t(src->p), hipMemcpyDeviceToDevice, s));
            AHIP_CHECK(hipMemcpyAsync(static_cast<T*>(d->impute) + off, src->impute, size_t(src->p) * sizeof(T), hipMemcpyDeviceToDevice, s));
            off += src->p;
        }
        AHIP_CHECK(hipStreamSynchronize(s));
        return;
    }
    const int64_t ld = ((nout + kAlign - 1) / kAlign) * kAlign;
    T* X = nullptr;
    AHIP_CHECK(hipMalloc(reinterpret_cast<void**>(&X), size_t(ld) * size_t(pout) * sizeof(T)));
    d->X = X;
    d->ld = ld;
    d->owned = true;
    d->kind = 0;
    hipStream_t s = d->stream;
    AHIP_CHECK(hipMemsetAsync(X, 0, size_t(ld) * size_t(pout) * sizeof(T), s));
    int64_t off = 0;
    for (int64_t m = 0; m < k; ++m) {
        adelie_hip_design* src = srcs[m];
        AHIP_CHECK(hipStreamSynchronize(src->stream));
        T* dst = axis == 1 ? X + off * ld : X + off;
        if (src->kind == 3) throw make_core_error("concatenations with a sparse design are composed on the host (adelie_amd.matrix).");
        if (src->kind == 0) launch_derive_dense<T>(src->dense<T>(), src->n, src->p, nullptr, nullptr, nullptr, nullptr, dst, ld, s);
        else launch_derive_dense_snp<T>(src->snp(), static_cast<const T*>(src->impute), src->n, src->p, nullptr, nullptr,
                                        nullptr, nullptr, dst, ld, s);
        off += axis == 1 ? src->p : src->n;
    }
    AHIP_CHECK(hipStreamSynchronize(s));
}
} // namespace

void adelie_hip_internal_free_batcher(void* b); // solver.hip
void adelie_hip_internal_batch_stats(void* b, double* out);

namespace {
template <class T>
void op_cov_bmul(adelie_hip_design* d, const int64_t* subset, int64_t ns, const int64_t* indices, const T* values, int64_t ni,
                 T* out) {
    set_device(d);
    hipStream_t s = d->stream;
    int64_t* dsub = scratch<int64_t>(d->s_idx1, ns + 1);
    int64_t* dind = scratch<int64_t>(d->s_idx2, ni + 1);
    T* dval = scratch<T>(d->s_n1, ni + 1);
    T* dout = scratch<T>(d->s_p1, ns + 1);
    AHIP_CHECK(hipMemcpyAsync(dsub, subset, size_t(ns) * sizeof(int64_t), hipMemcpyHostToDevice, s));
    AHIP_CHECK(hipMemcpyAsync(dind, indices, size_t(ni) * sizeof(int64_t), hipMemcpyHostToDevice, s));
    AHIP_CHECK(hipMemcpyAsync(dval, values, size_t(ni) * sizeof(T), hipMemcpyHostToDevice, s));
    launch_cov_bmul<T>(static_cast<const T*>(d->X), d->ld, d->cov == 2, dsub, ns, dind, dval, ni, dout, s);
    AHIP_CHECK(hipMemcpyAsync(out, dout, size_t(ns) * sizeof(T), hipMemcpyDeviceToHost, s));
    AHIP_CHECK(hipStreamSynchronize(s));
}
template <class T>
void op_cov_mul(adelie_hip_design* d, const int64_t* indices, const T* values, int64_t ni, T* out) {
    set_device(d);
    hipStream_t s = d->stream;
    int64_t* dind = scratch<int64_t>(d->s_idx2, ni + 1);
    T* dval = scratch<T>(d->s_n1, ni + 1);
    T* dout = scratch<T>(d->s_p1, d->p);
    AHIP_CHECK(hipMemcpyAsync(dind, indices, size_t(ni) * sizeof(int64_t), hipMemcpyHostToDevice, s));
    AHIP_CHECK(hipMemcpyAsync(dval, values, size_t(ni) * sizeof(T), hipMemcpyHostToDevice, s));
    launch_cov_mul<T>(static_cast<const T*>(d->X), d->ld, d->p, dind, dval, ni, dout, s);
    AHIP_CHECK(hipMemcpyAsync(out, dout, size_t(d->p) * sizeof(T), hipMemcpyDeviceToHost, s));
    AHIP_CHECK(hipStreamSynchronize(s));
}
template <class T>
void op_cov_to_dense(adelie_hip_design* d, int64_t i, int64_t q, T* out) {
    set_device(d);
    const T* S = static_cast<const T*>(d->X);
    AHIP_CHECK(hipMemcpy2DAsync(out, size_t(q) * sizeof(T), S + i + i * d->ld, size_t(d->ld) * sizeof(T), size_t(q) * sizeof(T),
                                size_t(q), hipMemcpyDeviceToHost, d->stream));
    AHIP_CHECK(hipStreamSynchronize(d->stream));
    if (d->cov == 2) // the stored block is the transpose
        for (int64_t b = 0; b < q; ++b)
            for (int64_t a = b + 1; a < q; ++a) std::swap(out[a + b * q], out[b + a * q]);
}
} // namespace

extern "C" {

int adelie_hip_abi_version(void) { return ADELIE_HIP_ABI_VERSION; }
const char* adelie_hip_last_error(void) { return g_last_error.c_str(); }
int adelie_hip_device_count(void) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) return 0;
    return cnt;
}

int adelie_hip_design_create_dense(const void* host, int64_t n, int64_t p, int dtype, int order, int device,
                                   adelie_hip_design** out) {
    ABI_TRY
    if (!host || !out) throw make_core_error("null argument.");
    adelie_hip_design* d = new_design(n, p, dtype, device);
    try {
        if (dtype == ADELIE_HIP_F64) create_dense_t<double>(d, host, false, order);
        else create_dense_t<float>(d, host, false, order);
    } catch (...) {
        adelie_hip_design_destroy(d);
        throw;
    }
    *out = d;
    ABI_CATCH
}

// adelie.matrix.dense(method="cov"): the (p, p) buffer is uploaded as it lies; a row-major input is therefore stored as A^T
// (cov == 2), which the cov_* operations account for and which is the same matrix when A is symmetric, as the method assumes
int adelie_hip_design_create_cov_dense(const void* host, int64_t p, int dtype, int order, int device, adelie_hip_design** out) {
    ABI_TRY
    if (!host || !out) throw make_core_error("null argument.");
    if (p <= 0) throw make_core_error("mat must be (p, p).");
    adelie_hip_design* d = new_design(p, p, dtype, device);
    try {
        if (dtype == ADELIE_HIP_F64) create_dense_t<double>(d, host, false, ADELIE_HIP_COL_MAJOR);
        else create_dense_t<float>(d, host, false, ADELIE_HIP_COL_MAJOR);
        d->cov = (order == ADELIE_HIP_ROW_MAJOR) ? 2 : 1;
    } catch (...) {
        adelie_hip_design_destroy(d);
        throw;
    }
    *out = d;
    ABI_CATCH
}


int adelie_hip_design_create_cov_lazy(adelie_hip_design* X, adelie_hip_design** out) {
    ABI_TRY
    if (!X || !out) throw make_core_error("null argument.");
    if (X->cov) throw make_core_error("mat must be a naive (n, p) matrix, not a covariance matrix.");
    if (X->kind != 0 && X->kind != 1) throw make_core_error("lazy_cov takes a dense or SNP design.");
    if (X->p > (int64_t(1) << 31) - 1) throw make_core_error("too many columns.");
    set_device(X);
    AHIP_CHECK(hipStreamSynchronize(X->stream));
    adelie_hip_design* d = new_design(X->p, X->p, X->dtype, X->device);
    try {
        if (X->dtype == ADELIE_HIP_F64) cov_lazy_t<double>(X, d);
        else cov_lazy_t<float>(X, d);
        d->cov = 1;
    } catch (...) {
        adelie_hip_design_destroy(d);
        throw;
    }
    *out = d;
    ABI_CATCH
}

int adelie_hip_design_cov_bmul(adelie_hip_design* d, const int64_t* subset, int64_t ns, const int64_t* indices,
                               const void* values, int64_t ni, void* out) {
    ABI_TRY
    need_cov(d);
    if (ns < 0 || ni < 0 || ns > d->p || ni > d->p) throw make_core_error("bmul() is given inconsistent inputs!");
    for (int64_t k = 0; k < ns; ++k)
        if (subset[k] < 0 || subset[k] >= d->p) throw make_core_error("bmul(): subset index out of range.");
    for (int64_t k = 0; k < ni; ++k)
        if (indices[k] < 0 || indices[k] >= d->p) throw make_core_error("bmul(): index out of range.");
    DTYPE_DISPATCH(d, op_cov_bmul<double>(d, subset, ns, indices, (const double*)values, ni, (double*)out),
                   op_cov_bmul<float>(d, subset, ns, indices, (const float*)values, ni, (float*)out))
    ABI_CATCH
}
int adelie_hip_design_cov_mul(adelie_hip_design* d, const int64_t* indices, const void* values, int64_t ni, void* out) {
    ABI_TRY
    need_cov(d);
    if (ni < 0 || ni > d->p) throw make_core_error("mul() is given inconsistent inputs!");
    for (int64_t k = 0; k < ni; ++k)
        if (indices[k] < 0 || indices[k] >= d->p) throw make_core_error("mul(): index out of range.");
    DTYPE_DISPATCH(d, op_cov_mul<double>(d, indices, (const double*)values, ni, (double*)out),
                   op_cov_mul<float>(d, indices, (const float*)values, ni, (float*)out))
    ABI_CATCH
}
int adelie_hip_design_cov_to_dense(adelie_hip_design* d, int64_t i, int64_t q, void* out) {
    ABI_TRY
    need_cov(d);
    if (i < 0 || q < 0 || i + q > d->p) throw make_core_error("to_dense() is given inconsistent inputs!");
    if (q > 0) { DTYPE_DISPATCH(d, op_cov_to_dense<double>(d, i, q, (double*)out), op_cov_to_dense<float>(d, i, q, (float*)out)) }
    ABI_CATCH
}

int adelie_hip_design_create_sparse(const int64_t* indptr, const int32_t* indices, const void* values, int64_t n, int64_t p,
                                    int dtype, int device, adelie_hip_design** out) {
    ABI_TRY
    if (!indptr || !out) throw make_core_error("null argument.");
    if (n <= 0 || p <= 0) throw make_core_error("matrix must have positive dimensions.");
    if (indptr[0] != 0) throw make_core_error("sparse(): indptr must start at 0.");
    for (int64_t j = 0; j < p; ++j)
        if (indptr[j + 1] < indptr[j]) throw make_core_error("sparse(): indptr must be non-decreasing.");
    const int64_t nnz = indptr[p];
    if (nnz > 0 && (!indices || !values)) throw make_core_error("null argument.");
    for (int64_t k = 0; k < nnz; ++k)
        if (indices[k] < 0 || indices[k] >= n) throw make_core_error("sparse(): row index out of range.");
    adelie_hip_design* d = new_design(n, p, dtype, device);
    try {
        if (dtype == ADELIE_HIP_F64) create_sparse_t<double>(d, indptr, indices, values);
        else create_sparse_t<float>(d, indptr, indices, values);
    } catch (...) {
        adelie_hip_design_destroy(d);
        throw;
    }
    *out = d;
    ABI_CATCH
}

int adelie_hip_design_create_csc(const int64_t* indptr, const int32_t* indices, const void* values, const int64_t* row_indptr,
                                 const int32_t* row_indices, const void* row_values, int64_t n, int64_t p, int dtype, int device,
                                 adelie_hip_design** out) {
    ABI_TRY
    if (!indptr || !row_indptr || !out) throw make_core_error("null argument.");
    if (n <= 0 || p <= 0) throw make_core_error("matrix must have positive dimensions.");
    if (n >= (int64_t(1) << 31)) throw make_core_error("number of rows must fit in int32.");
    if (indptr[0] != 0 || row_indptr[0] != 0) throw make_core_error("sparse(): indptr must start at 0.");
    for (int64_t j = 0; j < p; ++j)
        if (indptr[j + 1] < indptr[j]) throw make_core_error("sparse(): indptr must be non-decreasing.");
    for (int64_t i = 0; i < n; ++i)
        if (row_indptr[i + 1] < row_indptr[i]) throw make_core_error("sparse(): indptr must be non-decreasing.");
    const int64_t nnz = indptr[p];
    if (row_indptr[n] != nnz) throw make_core_error("sparse(): the two compressed forms must hold the same entries.");
    if (nnz > 0 && (!indices || !values || !row_indices || !row_values)) throw make_core_error("null argument.");
    for (int64_t j = 0; j < p; ++j) // rows ascending and distinct inside a column (scipy: sort_indices + sum_duplicates)
        for (int64_t k = indptr[j]; k < indptr[j + 1]; ++k) {
            if (indices[k] < 0 || indices[k] >= n) throw make_core_error("sparse(): row index out of range.");
            if (k > indptr[j] && indices[k] <= indices[k - 1]) throw make_core_error("sparse(): row indices must be sorted and distinct inside a column.");
        }
    if (p >= (int64_t(1) << 31)) throw make_core_error("number of columns must fit in int32.");
    {
        // The two forms must hold the SAME entries: gradients read the column form, residual updates the row form, and an
        // inconsistent pair would make them disagree silently.  Columns ascending and distinct inside a row, the same number of
        // entries per column in both forms, and an order-independent checksum over (row, column, value bits).
        const size_t vsz = dtype == ADELIE_HIP_F64 ? sizeof(double) : sizeof(float);
        auto mix = [](uint64_t r, uint64_t c, uint64_t v) {
            uint64_t h = (r * 0x9E3779B97F4A7C15ull) ^ (c * 0xC2B2AE3D27D4EB4Full) ^ (v * 0x165667B19E3779F9ull);
            h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
            return h;
        };
        auto bits = [&](const void* vals, int64_t k) {
            uint64_t b = 0;
            std::memcpy(&b, static_cast<const char*>(vals) + size_t(k) * vsz, vsz);
            return b;
        };
        std::vector<int64_t> per_col(static_cast<size_t>(p), 0);
        uint64_t sum_r = 0, sum_c = 0;
        for (int64_t i = 0; i < n; ++i)
            for (int64_t k = row_indptr[i]; k < row_indptr[i + 1]; ++k) {
                const int32_t c = row_indices[k];
                if (c < 0 || c >= p) throw make_core_error("sparse(): column index out of range.");
                if (k > row_indptr[i] && c <= row_indices[k - 1]) throw make_core_error("sparse(): column indices must be sorted and distinct inside a row.");
                ++per_col[size_t(c)];
                sum_r += mix(uint64_t(i), uint64_t(c), bits(row_values, k));
            }
        for (int64_t j = 0; j < p; ++j) {
            if (per_col[size_t(j)] != indptr[j + 1] - indptr[j]) throw make_core_error("sparse(): the two compressed forms must hold the same entries.");
            for (int64_t k = indptr[j]; k < indptr[j + 1]; ++k) sum_c += mix(uint64_t(indices[k]), uint64_t(j), bits(values, k));
        }
        if (sum_r != sum_c) throw make_core_error("sparse(): the two compressed forms must hold the same entries.");
    }
    adelie_hip_design* d = new_design(n, p, dtype, device);
    try {
        d->kind = 3;
        d->nnz = nnz;
        const size_t vs = dtype == ADELIE_HIP_F64 ? sizeof(double) : sizeof(float);
        const size_t nz1 = size_t(std::max<int64_t>(nnz, 1));
        AHIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d->cptr), size_t(p + 1) * sizeof(int64_t)));
        AHIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d->cidx), nz1 * sizeof(int32_t)));
        AHIP_CHECK(hipMalloc(&d->cval, nz1 * vs));
        AHIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d->rptr), size_t(n + 1) * sizeof(int64_t)));
        AHIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d->rcol), nz1 * sizeof(int32_t)));
        AHIP_CHECK(hipMalloc(&d->rval, nz1 * vs));
        hipStream_t s = d->stream;
        AHIP_CHECK(hipMemcpyAsync(d->cptr, indptr, size_t(p + 1) * sizeof(int64_t), hipMemcpyHostToDevice, s));
        AHIP_CHECK(hipMemcpyAsync(d->rptr, row_indptr, size_t(n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, s));
        if (nnz) {
            AHIP_CHECK(hipMemcpyAsync(d->cidx, indices, size_t(nnz) * sizeof(int32_t), hipMemcpyHostToDevice, s));
            AHIP_CHECK(hipMemcpyAsync(d->cval, values, size_t(nnz) * vs, hipMemcpyHostToDevice, s));
            AHIP_CHECK(hipMemcpyAsync(d->rcol, row_indices, size_t(nnz) * sizeof(int32_t), hipMemcpyHostToDevice, s));
            AHIP_CHECK(hipMemcpyAsync(d->rval, row_values, size_t(nnz) * vs, hipMemcpyHostToDevice, s));
        }
        csc_block_layout(n, vs, &d->sp_nb, &d->sp_rb);
        if (d->sp_nb > 1) {
            AHIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d->bptr), size_t(p) * size_t(d->sp_nb + 1) * sizeof(int64_t)));
            launch_csc_block_ptr(d->cptr, d->cidx, p, d->sp_nb, d->sp_rb, d->bptr, s);
        }
        // Tile-major copy for the full sweeps (csc_tile_sweep_kernel): tiles of kCscTileBytes of an n-vector.  Built on the
        // device from the per-column tile pointers; the tile-major offsets are a prefix sum over (tile, column) on the host.
        {
            const int64_t th = kCscTileBytes / int64_t(vs);
            const int64_t nt = (n + th - 1) / th;
            // (the tile of v is 128 KB of dynamic LDS — gfx950 has 160 KB per workgroup, the only target of this library; a launch
            // the device refuses raises in raw_sweep)
            const bool worth = nt > 1 && nt <= 4096 && nnz >= (int64_t(1) << 16) && nt * (p + 1) * 8 <= (int64_t(1) << 30) &&
                               nnz / (nt * p) >= 4; // (segments of a few entries at least: below, the pointers outweigh the entries)
            if (worth) {
                int64_t* colptr = nullptr;
                AHIP_CHECK(hipMalloc(reinterpret_cast<void**>(&colptr), size_t(p) * size_t(nt + 1) * sizeof(int64_t)));
                try {
                    launch_csc_block_ptr(d->cptr, d->cidx, p, int(nt), th, colptr, s);
                    std::vector<int64_t> cp(size_t(p) * size_t(nt + 1)), tp(size_t(nt) * size_t(p + 1));
                    AHIP_CHECK(hipMemcpyAsync(cp.data(), colptr, cp.size() * sizeof(int64_t), hipMemcpyDeviceToHost, s));
                    AHIP_CHECK(hipStreamSynchronize(s));
                    int64_t run = 0;
                    for (int64_t t = 0; t < nt; ++t) {
                        int64_t* row = tp.data() + size_t(t) * size_t(p + 1);
                        for (int64_t c = 0; c < p; ++c) {
                            row[c] = run;
                            run += cp[size_t(c) * size_t(nt + 1) + size_t(t) + 1] - cp[size_t(c) * size_t(nt + 1) + size_t(t)];
                        }
                        row[p] = run;
                    }
                    AHIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d->tptr), tp.size() * sizeof(int64_t)));
                    AHIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d->trow), nz1 * sizeof(uint16_t)));
                    AHIP_CHECK(hipMalloc(&d->tval, nz1 * vs));
                    AHIP_CHECK(hipMemcpyAsync(d->tptr, tp.data(), tp.size() * sizeof(int64_t), hipMemcpyHostToDevice, s));
                    if (dtype == ADELIE_HIP_F64)
                        launch_csc_tile_scatter<double>(colptr, d->cidx, static_cast<const double*>(d->cval), p, int(nt), th, d->tptr,
                                                        d->trow, static_cast<double*>(d->tval), s);
                    else
                        launch_csc_tile_scatter<float>(colptr, d->cidx, static_cast<const float*>(d->cval), p, int(nt), th, d->tptr,
                                                       d->trow, static_cast<float*>(d->tval), s);
                    AHIP_CHECK(hipStreamSynchronize(s));
                    d->sp_nt = int(nt);
                    d->sp_th = th;
                } catch (...) {
                    (void)hipFree(colptr);
                    throw;
                }
                (void)hipFree(colptr);
            }
        }
        AHIP_CHECK(hipStreamSynchronize(s));
    } catch (...) {
        adelie_hip_design_destroy(d);
        throw;
    }
    *out = d;
    ABI_CATCH
}

int adelie_hip_design_create_standardized(adelie_hip_design* src, const double* centers, const double* scales,
                                          adelie_hip_design** out) {
    ABI_TRY
    if (!src || !centers || !scales || !out) throw make_core_error("null argument.");
    if (src->kind == 2 || src->cov) no_view(src);
    if (src->std_center) throw make_core_error("the design is a standardized view already.");
    for (int64_t j = 0; j < src->p; ++j)
        if (!(scales[j] != 0.0)) throw make_core_error("scales must be non-zero.");
    adelie_hip_design* d = nullptr;
    if (adelie_hip_design_alias(src, &d)) throw make_core_error(g_last_error);
    try {
        const int64_t p = src->p;
        const size_t vs = src->dtype == ADELIE_HIP_F64 ? sizeof(double) : sizeof(float);
        AHIP_CHECK(hipMalloc(&d->std_center, size_t(p) * vs));
        d->std_owned = true;
        AHIP_CHECK(hipMalloc(&d->std_iscale, size_t(p) * vs));
        if (src->dtype == ADELIE_HIP_F64) {
            std::vector<double> is(static_cast<size_t>(p));
            for (int64_t j = 0; j < p; ++j) is[size_t(j)] = 1.0 / scales[j];
            AHIP_CHECK(hipMemcpy(d->std_center, centers, size_t(p) * vs, hipMemcpyHostToDevice));
            AHIP_CHECK(hipMemcpy(d->std_iscale, is.data(), size_t(p) * vs, hipMemcpyHostToDevice));
        } else {
            std::vector<float> ce(static_cast<size_t>(p)), is(static_cast<size_t>(p));
            for (int64_t j = 0; j < p; ++j) { ce[size_t(j)] = float(centers[j]); is[size_t(j)] = float(1.0 / scales[j]); }
            AHIP_CHECK(hipMemcpy(d->std_center, ce.data(), size_t(p) * vs, hipMemcpyHostToDevice));
            AHIP_CHECK(hipMemcpy(d->std_iscale, is.data(), size_t(p) * vs, hipMemcpyHostToDevice));
        }
    } catch (...) {
        adelie_hip_design_destroy(d);
        throw;
    }
    *out = d;
    ABI_CATCH
}

int adelie_hip_design_adopt_dense_dev(const void* dev_ptr, int64_t n, int64_t p, int dtype, int order, int device,
                                      adelie_hip_design** out) {
    ABI_TRY
    if (!dev_ptr || !out) throw make_core_error("null argument.");
    adelie_hip_design* d = new_design(n, p, dtype, device);
    try {
        if (dtype == ADELIE_HIP_F64) create_dense_t<double>(d, dev_ptr, true, order);
        else create_dense_t<float>(d, dev_ptr, true, order);
    } catch (...) {
        adelie_hip_design_destroy(d);
        throw;
    }
    *out = d;
    ABI_CATCH
}

int adelie_hip_design_create_snp_calldata(const int8_t* calldata, int64_t n, int64_t p, const double* impute, int dtype,
                                          int device, adelie_hip_design** out) {
    ABI_TRY
    if (!calldata || !impute || !out) throw make_core_error("null argument.");
    adelie_hip_design* d = new_design(n, p, dtype, device);
    try {
        create_snp_from_calldata(d, calldata, impute);
    } catch (...) {
        adelie_hip_design_destroy(d);
        throw;
    }
    *out = d;
    ABI_CATCH
}

// Decodes one column of a .snpdat image into int8 calls (missing = -9; `col` zeroed by the caller).  Every read is checked
// against the end of the column BEFORE it happens (io_snp_unphased.ipp:117-135,225-274 is the layout).
static void decode_snpdat_column(const uint8_t* buf, uint64_t base, uint64_t endc, uint64_t n, int8_t* col) {
    for (int c = 0; c < 3; ++c) {
        const uint64_t off = read_as<uint64_t>(buf + base + 8 * c);
        if (off > endc - base || endc - base - off < 4) throw make_core_error("corrupt .snpdat category offset.");
        uint64_t pos = base + off;
        const uint32_t n_chunks = read_as<uint32_t>(buf + pos);
        pos += 4;
        const int8_t val = c == 0 ? int8_t(-9) : int8_t(c);
        for (uint32_t k = 0; k < n_chunks; ++k) {
            if (endc - pos < 5) throw make_core_error("corrupt .snpdat chunk (overrun).");
            const uint32_t cidx = read_as<uint32_t>(buf + pos);
            const uint32_t cnt = uint32_t(buf[pos + 4]) + 1;
            pos += 5;
            if (endc - pos < cnt) throw make_core_error("corrupt .snpdat chunk (overrun).");
            for (uint32_t t = 0; t < cnt; ++t) {
                const uint64_t row = uint64_t(cidx) * 256 + buf[pos + t];
                if (row >= n) throw make_core_error("corrupt .snpdat chunk (row out of range).");
                col[row] = val;
            }
            pos += cnt;
        }
    }
}

int adelie_hip_design_create_snp_unphased(const void* snpdat, int64_t n_bytes, int dtype, int device,
                                          adelie_hip_design** out) {
    ABI_TRY
    if (!snpdat || !out) throw make_core_error("null argument.");
    const uint8_t* buf = static_cast<const uint8_t*>(snpdat);
    if (n_bytes < 17) throw make_core_error("buffer is too small to be a .snpdat image.");
    const uint64_t n = read_as<uint64_t>(buf + 1), p = read_as<uint64_t>(buf + 9);
    // Nothing below is sized from the header before the header is bounded.  Every column costs 32 header bytes (nnz, nnm,
    // impute, outer) plus its own 24-byte offset table, which bounds p by the image size; the rows are only bounded by what
    // the format can address, so a plausibility cap stands in (2^32 rows; the largest biobanks have < 2^24).  The host side
    // then never holds more than one panel of decoded columns (256 MB, or one column), whatever the header claims, and the
    // packed matrix is allocated on the device BEFORE any decoding: an image whose counts were damaged fails there.
    if (p == 0 || n == 0) throw make_core_error("corrupt .snpdat header (row / column counts).");
    if (p > uint64_t(n_bytes) / 32 || n > (uint64_t(1) << 32)) throw make_core_error("corrupt .snpdat header (row / column counts).");
    const size_t hdr = 17 + 8 * p * 3 + 8 * (p + 1);
    if (size_t(n_bytes) < hdr) throw make_core_error("truncated .snpdat header.");
    const uint8_t* impute_p = buf + 17 + 16 * p;
    const uint8_t* outer_p = buf + 17 + 24 * p;
    std::vector<double> impute(p);
    std::memcpy(impute.data(), impute_p, 8 * p);
    for (uint64_t j = 0; j < p; ++j) { // the column index, before anything is allocated
        const uint64_t base = read_as<uint64_t>(outer_p + 8 * j), endc = read_as<uint64_t>(outer_p + 8 * (j + 1));
        if (endc > uint64_t(n_bytes) || base < hdr || base > endc || endc - base < 24)
            throw make_core_error("corrupt .snpdat column index.");
    }
    adelie_hip_design* d = new_design(int64_t(n), int64_t(p), dtype, device);
    int8_t* tmp = nullptr;
    try {
        d->kind = 1;
        d->ldb = int64_t((((n + 3) / 4 + 127) / 128) * 128);
        AHIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d->bits), size_t(d->ldb) * size_t(p)));
        // decode -> int8 panel on the host -> 2-bit on the device, a panel of columns at a time
        const uint64_t panel = std::min<uint64_t>(p, std::max<uint64_t>(1, (uint64_t(1) << 28) / n));
        AHIP_CHECK(hipMalloc(reinterpret_cast<void**>(&tmp), size_t(n) * size_t(panel)));
        std::vector<int8_t> calls(size_t(n) * size_t(panel));
        for (uint64_t j0 = 0; j0 < p; j0 += panel) {
            const uint64_t pc = std::min(panel, p - j0);
            std::fill(calls.begin(), calls.begin() + size_t(n) * size_t(pc), int8_t(0));
            for (uint64_t j = j0; j < j0 + pc; ++j)
                decode_snpdat_column(buf, read_as<uint64_t>(outer_p + 8 * j), read_as<uint64_t>(outer_p + 8 * (j + 1)), n,
                                     calls.data() + size_t(j - j0) * size_t(n));
            AHIP_CHECK(hipMemcpyAsync(tmp, calls.data(), size_t(n) * size_t(pc), hipMemcpyHostToDevice, d->stream));
            launch_pack_snp(tmp, int64_t(n), int64_t(pc), d->bits + int64_t(j0) * d->ldb, d->ldb, d->stream);
            AHIP_CHECK(hipStreamSynchronize(d->stream));
        }
        (void)hipFree(tmp);
        tmp = nullptr;
        if (d->dtype == ADELIE_HIP_F64) {
            AHIP_CHECK(hipMalloc(&d->impute, size_t(p) * sizeof(double)));
            AHIP_CHECK(hipMemcpy(d->impute, impute.data(), size_t(p) * sizeof(double), hipMemcpyHostToDevice));
        } else {
            std::vector<float> f(impute.begin(), impute.end());
            AHIP_CHECK(hipMalloc(&d->impute, size_t(p) * sizeof(float)));
            AHIP_CHECK(hipMemcpy(d->impute, f.data(), size_t(p) * sizeof(float), hipMemcpyHostToDevice));
        }
    } catch (...) {
        if (tmp) (void)hipFree(tmp);
        adelie_hip_design_destroy(d);
        throw;
    }
    *out = d;
    ABI_CATCH
}

int adelie_hip_design_create_snp_bed(const void* bed, int64_t n_bytes, int64_t n, int64_t p, int dtype, int device,
                                     adelie_hip_design** out) {
    ABI_TRY
    if (!bed || !out) throw make_core_error("null argument.");
    if (n <= 0 || p <= 0) throw make_core_error("n and p must be positive.");
    const uint8_t* buf = static_cast<const uint8_t*>(bed);
    const int64_t stride = (n + 3) / 4;
    if (n_bytes < 3 || buf[0] != 0x6c || buf[1] != 0x1b) throw make_core_error("not a PLINK .bed image (bad magic).");
    if (buf[2] != 0x01) throw make_core_error("only SNP-major .bed files (third byte 0x01) are supported.");
    if (n_bytes < 3 + stride * p) throw make_core_error("truncated .bed image: expected 3 + ceil(n/4)*p bytes.");
    adelie_hip_design* d = new_design(n, p, dtype, device);
    try {
        d->kind = 1;
        d->ldb = (((n + 3) / 4 + 127) / 128) * 128;
        AHIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d->bits), size_t(d->ldb) * size_t(p)));
        // stage the records in panels of SNPs to bound the temporary
        const int64_t panel = std::max<int64_t>(1, (int64_t(1) << 28) / stride);
        uint8_t* tmp = nullptr;
        AHIP_CHECK(hipMalloc(reinterpret_cast<void**>(&tmp), size_t(stride) * size_t(std::min(panel, p))));
        for (int64_t j0 = 0; j0 < p; j0 += panel) {
            const int64_t pc = std::min(panel, p - j0);
            AHIP_CHECK(hipMemcpyAsync(tmp, buf + 3 + j0 * stride, size_t(stride) * size_t(pc), hipMemcpyDefault, d->stream));
            launch_bed_transcode(tmp, n, pc, stride, d->bits + j0 * d->ldb, d->ldb, d->stream);
            AHIP_CHECK(hipStreamSynchronize(d->stream));
        }
        (void)hipFree(tmp);
        if (d->dtype == ADELIE_HIP_F64) {
            AHIP_CHECK(hipMalloc(&d->impute, size_t(p) * sizeof(double)));
            launch_snp_impute<double>(d->bits, n, p, d->ldb, static_cast<double*>(d->impute), d->stream);
        } else {
            AHIP_CHECK(hipMalloc(&d->impute, size_t(p) * sizeof(float)));
            launch_snp_impute<float>(d->bits, n, p, d->ldb, static_cast<float*>(d->impute), d->stream);
        }
        AHIP_CHECK(hipStreamSynchronize(d->stream));
    } catch (...) {
        adelie_hip_design_destroy(d);
        throw;
    }
    *out = d;
    ABI_CATCH
}

int adelie_hip_design_batch_stats(adelie_hip_design* d, double* out) {
    ABI_TRY
    if (!d || !out) throw make_core_error("null argument.");
    adelie_hip_design* owner = d->batch_owner ? d->batch_owner : d;
    adelie_hip_internal_batch_stats(owner->batcher, out);
    ABI_CATCH
}

int adelie_hip_design_alias(adelie_hip_design* src, adelie_hip_design** out) {
    ABI_TRY
    if (!src || !out) throw make_core_error("null argument.");
    if (src->kind == 2 || src->cov) no_view(src);
    adelie_hip_design* d = new_design(src->n, src->p, src->dtype, src->device); // own stream, own scratch
    d->kind = src->kind;
    d->X = src->X;
    d->ld = src->ld;
    d->owned = false;
    d->bits = src->bits;
    d->ldb = src->ldb;
    d->impute = src->impute;
    d->cptr = src->cptr; d->cidx = src->cidx; d->cval = src->cval;
    d->rptr = src->rptr; d->rcol = src->rcol; d->rval = src->rval;
    d->nnz = src->nnz;
    d->bptr = src->bptr; d->sp_nb = src->sp_nb; d->sp_rb = src->sp_rb;
    d->tptr = src->tptr; d->trow = src->trow; d->tval = src->tval; d->sp_nt = src->sp_nt; d->sp_th = src->sp_th;
    d->std_center = src->std_center; d->std_iscale = src->std_iscale; // (not owned: std_owned stays false)
    d->alias = true;
    d->batch_owner = src->batch_owner ? src->batch_owner : src;
    *out = d;
    ABI_CATCH
}

int adelie_hip_design_create_slice(adelie_hip_design* base, int64_t r0, int64_t nr, int64_t c0, int64_t nc,
                                   adelie_hip_design** out) {
    ABI_TRY
    if (!base || !out) throw make_core_error("null argument.");
    if ((base->kind != 0 && base->kind != 1) || base->cov || base->std_center)
        throw make_core_error("only dense and 2-bit SNP designs are sliced in place.");
    if (r0 < 0 || nr < 1 || c0 < 0 || nc < 1 || r0 + nr > base->n || c0 + nc > base->p)
        throw make_core_error("slice out of range.");
    const int64_t es = base->dtype == ADELIE_HIP_F64 ? 8 : 4;
    if (base->kind == 1 && (r0 != 0 || nr != base->n)) throw make_core_error("a 2-bit design is sliced by columns only.");
    if (base->kind == 0 && (r0 * es) % 16 != 0) throw make_core_error("a row slice must start on a 16-byte boundary.");
    adelie_hip_design* d = new_design(nr, nc, base->dtype, base->device); // own stream, own scratch
    d->kind = base->kind;
    d->owned = false;
    d->alias = true; // never frees what it points into
    if (base->kind == 0) {
        d->X = static_cast<char*>(base->X) + (c0 * base->ld + r0) * es;
        d->ld = base->ld;
    } else {
        d->bits = base->bits + c0 * base->ldb;
        d->ldb = base->ldb;
        d->impute = static_cast<char*>(base->impute) + c0 * es;
    }
    *out = d;
    ABI_CATCH
}

int adelie_hip_design_create_multi(adelie_hip_design* base, int64_t K, int intercept, adelie_hip_design** out) {
    ABI_TRY
    if (!base || !out) throw make_core_error("null argument.");
    if ((base->kind != 0 && base->kind != 1) || base->cov || base->std_center)
        throw make_core_error("the multi-response view needs a dense or 2-bit SNP base design.");
    if (K < 1) throw make_core_error("K must be >= 1.");
    const int64_t icpt = intercept ? 1 : 0;
    if ((base->p + icpt) * K > int64_t(0x7fffffff) || base->n * K > (int64_t(1) << 40))
        throw make_core_error("the multi-response view is too large.");
    adelie_hip_design* d = new_design(base->n * K, (base->p + icpt) * K, base->dtype, base->device); // own stream + scratch
    d->kind = 2;
    d->X = base->X;
    d->ld = base->ld;
    d->bits = base->bits; // (2-bit base: the K-wide kernels decode the calls themselves, MultiView::bits)
    d->ldb = base->ldb;
    d->impute = base->impute;
    d->owned = false;
    d->alias = true;
    d->mK = K;
    d->micpt = int(icpt);
    d->nb = base->n;
    d->pb = base->p;
    {   // the column of ones (padded like a design column so that vector loads past nb stay in bounds)
        const size_t esz = base->dtype == ADELIE_HIP_F64 ? sizeof(double) : sizeof(float);
        const int64_t len = ((base->n + 31) / 32) * 32;
        AHIP_CHECK(hipMalloc(&d->ones, size_t(len) * esz));
        AHIP_CHECK(hipMemsetAsync(d->ones, 0, size_t(len) * esz, d->stream));
        DTYPE_DISPATCH(d, launch_fill<T>((T*)d->ones, T(1), base->n, d->stream), launch_fill<T>((T*)d->ones, T(1), base->n, d->stream))
        AHIP_CHECK(hipStreamSynchronize(d->stream));
    }
    *out = d;
    ABI_CATCH
}

int adelie_hip_design_create_derived(adelie_hip_design* src, const int64_t* rows, int64_t n_rows, const int64_t* cols,
                                     int64_t n_cols, const double* centers, const double* scales, adelie_hip_design** out) {
    ABI_TRY
    if (!src || !out) throw make_core_error("null argument.");
    no_view(src);
    const int64_t nout = rows ? n_rows : src->n, pout = cols ? n_cols : src->p;
    if (rows)
        for (int64_t i = 0; i < n_rows; ++i)
            if (rows[i] < 0 || rows[i] >= src->n) throw make_core_error("subset contains an out-of-range row index.");
    if (cols)
        for (int64_t j = 0; j < n_cols; ++j)
            if (cols[j] < 0 || cols[j] >= src->p) throw make_core_error("subset contains an out-of-range column index.");
    if (scales)
        for (int64_t j = 0; j < pout; ++j)
            if (!(scales[j] != 0.0)) throw make_core_error("scales must be non-zero.");
    adelie_hip_design* d = new_design(nout, pout, src->dtype, src->device);
    try {
        DTYPE_DISPATCH(src, derive_t<T>(src, d, rows, n_rows, cols, n_cols, centers, scales),
                       derive_t<T>(src, d, rows, n_rows, cols, n_cols, centers, scales))
    } catch (...) {
        adelie_hip_design_destroy(d);
        throw;
    }
    *out = d;
    ABI_CATCH
}

int adelie_hip_design_create_concat(adelie_hip_design* const* srcs, int64_t k, int axis, adelie_hip_design** out) {
    ABI_TRY
    if (!srcs || !out || k <= 0) throw make_core_error("null argument.");
    if (axis != 0 && axis != 1) throw make_core_error("axis must be 0 or 1.");
    int64_t n = 0, p = 0;
    for (int64_t m = 0; m < k; ++m) {
        if (!srcs[m]) throw make_core_error("null argument.");
        no_view(srcs[m]);
        if (srcs[m]->dtype != srcs[0]->dtype || srcs[m]->device != srcs[0]->device)
            throw make_core_error("concatenate(): the matrices must share dtype and device.");
        if (axis == 1) {
            if (srcs[m]->n != srcs[0]->n) throw make_core_error("All matrices must have the same number of rows.");
            p += srcs[m]->p;
            n = srcs[0]->n;
        } else {
            if (srcs[m]->p != srcs[0]->p) throw make_core_error("All matrices must have the same number of columns.");
            n += srcs[m]->n;
            p = srcs[0]->p;
        }
    }
    adelie_hip_design* d = new_design(n, p, srcs[0]->dtype, srcs[0]->device);
    try {
        DTYPE_DISPATCH(srcs[0], concat_t<T>(srcs, k, axis, d), concat_t<T>(srcs, k, axis, d))
    } catch (...) {
        adelie_hip_design_destroy(d);
        throw;
    }
    *out = d;
    ABI_CATCH
}

int adelie_hip_design_impute(adelie_hip_design* d, double* out) {
    ABI_TRY
    no_view(d);
    if (!d || !out) throw make_core_error("null argument.");
    if (d->kind != 1) throw make_core_error("impute is only defined for SNP designs.");
    set_device(d);
    if (d->dtype == ADELIE_HIP_F64) {
        AHIP_CHECK(hipMemcpy(out, d->impute, size_t(d->p) * sizeof(double), hipMemcpyDeviceToHost));
    } else {
        std::vector<float> f(size_t(d->p));
        AHIP_CHECK(hipMemcpy(f.data(), d->impute, size_t(d->p) * sizeof(float), hipMemcpyDeviceToHost));
        for (int64_t j = 0; j < d->p; ++j) out[j] = f[size_t(j)];
    }
    ABI_CATCH
}

int adelie_hip_design_destroy(adelie_hip_design* d) {
    if (!d) return 0;
    (void)hipSetDevice(d->device);
    if (d->stream) {
        (void)hipStreamSynchronize(d->stream);
        (void)hipStreamDestroy(d->stream);
    }
    if (d->owned && d->X && !d->alias) (void)hipFree(d->X);
    if (d->bits && !d->alias) (void)hipFree(d->bits);
    if (d->impute && !d->alias) (void)hipFree(d->impute);
    if (!d->alias) {
        (void)hipFree(d->cptr); (void)hipFree(d->cidx); (void)hipFree(d->cval);
        (void)hipFree(d->rptr); (void)hipFree(d->rcol); (void)hipFree(d->rval);
        (void)hipFree(d->bptr);
        (void)hipFree(d->tptr); (void)hipFree(d->trow); (void)hipFree(d->tval);
    }
    if (d->std_owned) { (void)hipFree(d->std_center); (void)hipFree(d->std_iscale); }
    if (d->ones) (void)hipFree(d->ones);
    if (d->batcher) adelie_hip_internal_free_batcher(d->batcher);
    delete d;
    if (live_designs().fetch_sub(1) == 1) DevPool::trim();
    return 0;
}

int64_t adelie_hip_design_rows(const adelie_hip_design* d) { return d->n; }
int64_t adelie_hip_design_cols(const adelie_hip_design* d) { return d->p; }
int adelie_hip_design_dtype(const adelie_hip_design* d) { return d->dtype; }
int adelie_hip_design_device(const adelie_hip_design* d) { return d->device; }
void* adelie_hip_design_stream(const adelie_hip_design* d) { return d->stream; }

int adelie_hip_design_cmul(adelie_hip_design* d, int64_t j, const void* v, const void* weights, double* out) {
    ABI_TRY
    no_view(d);
    check_col(d, j, 1, "cmul");
    DTYPE_DISPATCH(d, { T o; op_sweep<T>(d, j, 1, (const T*)v, (const T*)weights, &o, false); *out = o; },
                   { T o; op_sweep<T>(d, j, 1, (const T*)v, (const T*)weights, &o, false); *out = o; })
    ABI_CATCH
}
int adelie_hip_design_ctmul(adelie_hip_design* d, int64_t j, double v, void* out) {
    ABI_TRY
    no_view(d);
    check_col(d, j, 1, "ctmul");
    DTYPE_DISPATCH(d, { T c = T(v); op_axpy<T>(d, j, 1, &c, (T*)out); }, { T c = T(v); op_axpy<T>(d, j, 1, &c, (T*)out); })
    ABI_CATCH
}
int adelie_hip_design_bmul(adelie_hip_design* d, int64_t j, int64_t q, const void* v, const void* weights, void* out) {
    ABI_TRY
    no_view(d);
    check_col(d, j, q, "bmul");
    DTYPE_DISPATCH(d, op_sweep<T>(d, j, q, (const T*)v, (const T*)weights, (T*)out, false),
                   op_sweep<T>(d, j, q, (const T*)v, (const T*)weights, (T*)out, false))
    ABI_CATCH
}
int adelie_hip_design_btmul(adelie_hip_design* d, int64_t j, int64_t q, const void* v, void* out) {
    ABI_TRY
    no_view(d);
    check_col(d, j, q, "btmul");
    DTYPE_DISPATCH(d, op_axpy<T>(d, j, q, (const T*)v, (T*)out), op_axpy<T>(d, j, q, (const T*)v, (T*)out))
    ABI_CATCH
}
int adelie_hip_design_mul(adelie_hip_design* d, const void* v, const void* weights, void* out) {
    ABI_TRY
    no_view(d);
    DTYPE_DISPATCH(d, op_sweep<T>(d, 0, d->p, (const T*)v, (const T*)weights, (T*)out, false),
                   op_sweep<T>(d, 0, d->p, (const T*)v, (const T*)weights, (T*)out, false))
    ABI_CATCH
}
int adelie_hip_design_mul_batch(adelie_hip_design* d, const void* V, int64_t L, void* out) {
    ABI_TRY
    no_view(d);
    if (L < 0 || (L > 0 && (!V || !out))) throw make_core_error("mul_batch() is given inconsistent inputs!");
    if (L == 0) return 0;
    DTYPE_DISPATCH(d, op_mul_batch<T>(d, (const T*)V, L, (T*)out), op_mul_batch<T>(d, (const T*)V, L, (T*)out))
    ABI_CATCH
}
int adelie_hip_design_cov(adelie_hip_design* d, int64_t j, int64_t q, const void* sqrt_weights, void* out) {
    ABI_TRY
    no_view(d);
    check_col(d, j, q, "cov");
    DTYPE_DISPATCH(d, op_cov<T>(d, j, q, (const T*)sqrt_weights, (T*)out), op_cov<T>(d, j, q, (const T*)sqrt_weights, (T*)out))
    ABI_CATCH
}
int adelie_hip_design_sq_mul(adelie_hip_design* d, const void* weights, void* out) {
    ABI_TRY
    no_view(d);
    DTYPE_DISPATCH(d, op_sweep<T>(d, 0, d->p, (const T*)weights, (const T*)nullptr, (T*)out, true),
                   op_sweep<T>(d, 0, d->p, (const T*)weights, (const T*)nullptr, (T*)out, true))
    ABI_CATCH
}
int adelie_hip_design_sp_tmul(adelie_hip_design* d, int64_t L, const int64_t* indptr, const int64_t* indices,
                              const void* values, void* out) {
    ABI_TRY
    no_view(d);
    if (L < 0) throw make_core_error("sp_tmul() is given inconsistent inputs!");
    if (L == 0) return 0;
    DTYPE_DISPATCH(d, op_sp_tmul<T>(d, L, indptr, indices, (const T*)values, (T*)out),
                   op_sp_tmul<T>(d, L, indptr, indices, (const T*)values, (T*)out))
    ABI_CATCH
}

int adelie_hip_design_glm_path_losses(adelie_hip_design* d, int glm_kind, int64_t L, const int64_t* indptr,
                                      const int64_t* indices, const void* values, const void* intercepts,
                                      const void* offsets, const void* y, const void* weights_a, const void* weights_b,
                                      double* out) {
    ABI_TRY
    no_view(d);
    if (!d || !indptr || !intercepts || !offsets || !y || !weights_a || !weights_b || !out)
        throw make_core_error("null argument.");
    if (L < 0) throw make_core_error("L must be >= 0.");
    if (L > 0) {
        DTYPE_DISPATCH(d, op_path_losses<T>(d, glm_kind, L, indptr, indices, (const T*)values, (const T*)intercepts,
                                            (const T*)offsets, (const T*)y, (const T*)weights_a, (const T*)weights_b, out),
                       op_path_losses<T>(d, glm_kind, L, indptr, indices, (const T*)values, (const T*)intercepts,
                                         (const T*)offsets, (const T*)y, (const T*)weights_a, (const T*)weights_b, out))
    }
    ABI_CATCH
}

int adelie_hip_design_multi_path_losses(adelie_hip_design* d, int glm_kind, int K, int64_t L, const int64_t* indptr,
                                        const int64_t* indices, const void* values, const void* intercepts,
                                        const void* offsets, const void* y, const void* weights_a, const void* weights_b,
                                        double* out) {
    ABI_TRY
    no_view(d);
    if (!d || !indptr || !intercepts || !offsets || !y || !weights_a || !weights_b || !out)
        throw make_core_error("null argument.");
    if (L < 0) throw make_core_error("L must be >= 0.");
    if (K < 1) throw make_core_error("K must be >= 1.");
    if (glm_kind != ADELIE_HIP_GLM_GAUSSIAN && glm_kind != ADELIE_HIP_GLM_MULTINOMIAL)
        throw make_core_error("multi_path_losses: glm_kind must be GAUSSIAN (multigaussian) or MULTINOMIAL.");
    if (L > 0) {
        DTYPE_DISPATCH(d, op_multi_path_losses<T>(d, glm_kind, K, L, indptr, indices, (const T*)values, (const T*)intercepts,
                                                  (const T*)offsets, (const T*)y, (const T*)weights_a, (const T*)weights_b, out),
                       op_multi_path_losses<T>(d, glm_kind, K, L, indptr, indices, (const T*)values, (const T*)intercepts,
                                               (const T*)offsets, (const T*)y, (const T*)weights_a, (const T*)weights_b, out))
    }
    ABI_CATCH
}

} // extern "C"

namespace ahip {
void set_last_error(const std::string& s) { g_last_error = s; }
} // namespace ahip
