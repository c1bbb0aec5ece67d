// common.hpp — host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/adelie_hip.h"
#include "kernels.hpp"

namespace ahip {

// util/exceptions.hpp:8-55 — same prefixes so that the Python layer's error-vs-warning split keeps working
struct core_error : std::runtime_error {
    using std::runtime_error::runtime_error;
};
inline core_error make_core_error(const std::string& m) { return core_error("adelie_core: " + m); }
inline core_error make_solver_error(const std::string& m) { return core_error("adelie_core solver: " + m); }

#define AHIP_CHECK(expr)                                                                                   \
    do {                                                                                                   \
        hipError_t _e = (expr);                                                                            \
        if (_e != hipSuccess)                                                                              \
            throw ::ahip::core_error(std::string("adelie_hip: HIP error '") + hipGetErrorString(_e) +     \
                                     "' at " #expr);                                                       \
    } while (0)

// wall-clock spent in hipMalloc / hipFree by the DevBufs of this process (tuning aid, printed under ADELIE_HIP_TRACE_ENQ)
struct DevAllocStats {
    static double& seconds() { static double s = 0; return s; }
    static long& calls() { static long c = 0; return c; }
};

// Owning device buffer (grow-only).
template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) {
            const auto t0 = std::chrono::steady_clock::now();
            (void)hipFree(p);
            DevAllocStats::seconds() += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            ++DevAllocStats::calls();
        }
        p = nullptr;
        cap = 0;
    }
    // ensure capacity >= n elements; contents are NOT preserved
    T* reserve(size_t n) {
        if (n > cap) {
            release();
            size_t want = n < 16 ? 16 : n;
            const auto t0 = std::chrono::steady_clock::now();
            AHIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p), want * sizeof(T)));
            DevAllocStats::seconds() += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            ++DevAllocStats::calls();
            cap = want;
        }
        return p;
    }
    // ensure capacity, preserving the first `keep` elements
    T* grow(size_t n, size_t keep, hipStream_t s) {
        if (n <= cap) return p;
        size_t want = cap * 2 > n ? cap * 2 : n;
        if (want < 16) want = 16;
        T* q = nullptr;
        AHIP_CHECK(hipMalloc(reinterpret_cast<void**>(&q), want * sizeof(T)));
        if (p && keep) AHIP_CHECK(hipMemcpyAsync(q, p, keep * sizeof(T), hipMemcpyDeviceToDevice, s));
        AHIP_CHECK(hipStreamSynchronize(s));
        if (p) (void)hipFree(p);
        p = q;
        cap = want;
        return p;
    }
    void upload(const T* h, size_t n, hipStream_t s, size_t off = 0) {
        if (n) AHIP_CHECK(hipMemcpyAsync(p + off, h, n * sizeof(T), hipMemcpyHostToDevice, s));
    }
    void download(T* h, size_t n, hipStream_t s, size_t off = 0) const {
        if (n) AHIP_CHECK(hipMemcpyAsync(h, p + off, n * sizeof(T), hipMemcpyDeviceToHost, s));
    }
};

} // namespace ahip

// ---- the opaque handles of include/adelie_hip.h ------------------------------------------------------------
struct adelie_hip_design {
    int dtype = ADELIE_HIP_F64;
    int device = 0;
    int kind = 0; // 0 dense, 1 snp (2-bit), 2 multi-response view of a dense design (adelie_hip_design_create_multi)
    // covariance-method matrix A (adelie_hip_design_create_cov_dense): a dense (p, p) design with n == p that only
    // adelie_hip_gaussian_cov_solve and the cov_* operations accept; cov == 2: the stored matrix is A^T (row-major input)
    int cov = 0;
    int64_t n = 0, p = 0;
    // dense
    void* X = nullptr;
    int64_t ld = 0;
    bool owned = false;
    // snp
    uint8_t* bits = nullptr;
    int64_t ldb = 0;
    void* impute = nullptr; // (p,) value_t on device
    bool alias = false;     // shares X / bits / impute with another design (adelie_hip_design_alias): never frees them
    // multi-response view: [1 (x) I_K, X (x) I_K] over the base's X (nb x pb); n = nb*K, p = (pb + micpt)*K
    int64_t mK = 0, nb = 0, pb = 0;
    int micpt = 0;
    void* ones = nullptr; // nb ones (owned)
    // Sweep batching across solvers that run concurrently on one resident matrix (cv_grpnet folds on alias handles): the
    // batcher object lives with the design the aliases were made from (solver.hip::SweepBatcher, created on first use).
    adelie_hip_design* batch_owner = nullptr; // nullptr: this design itself
    void* batcher = nullptr;
    hipStream_t stream = nullptr;
    // scratch for the host-vector matrix ops (value_t typed, grow-only)
    ahip::DevBuf<char> s_n1, s_n2, s_p1, s_work, s_misc, s_idx1, s_idx2;

    template <class T> ahip::DenseView<T> dense() const { return ahip::DenseView<T>{static_cast<const T*>(X), n, p, ld}; }
    ahip::SnpView snp() const { return ahip::SnpView{bits, n, p, ldb}; }
    template <class T> ahip::MultiView<T> multi() const {
        return ahip::MultiView<T>{static_cast<const T*>(X), nb, pb, ld, static_cast<const T*>(ones), int32_t(mK), int32_t(micpt)};
    }
};
