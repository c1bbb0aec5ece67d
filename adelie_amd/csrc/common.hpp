// common.hpp — host-side helpers shared by the C-ABI translation units.
#pragma once
#include <cstdlib>
#include <algorithm>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/adelie_hip.h"
#include "kernels.hpp"

namespace ahip {

// ---- every environment hook of libadelie_hip.so, in one place ------------------------------------------------------------------
// Eleven variables.  They are read when a solve starts (Hooks::from_env, one call per solve): the tests force the multi-CU engines
// at sizes the CPU checker finishes in seconds and compare variants between two solves of ONE process, so reading them once at
// library load would freeze the first test's setting.  None changes results beyond rounding.  (Python side: ADELIE_HIP_LIB picks
// another build of this library, ADELIE_HIP_SWEEP_BATCH=0 keeps concurrent CV folds from sharing their sweeps.)
//   ADELIE_HIP_CD_BLOCK_MIN_NV=n  screen sets of at least n coefficients use the panel engine (default 128)       [test hook]
//   ADELIE_HIP_PANEL_BSZ=32|64|128  visits per lasso panel block (default 128, 64 under IRLS)                      [test hook]
//   ADELIE_HIP_LOOKAHEAD=0        Gaussian passes in the sequential form (step -> reduce -> solve) that IRLS and
//                                 constrained problems always use                                                   [test hook]
//   ADELIE_HIP_SPECULATE=0        no speculative first active-set pass behind the invariance sweep                  [A/B, bit-identity test]
//   ADELIE_HIP_IRLS_REUSE=theta   IRLS diagonal-block / Gram reuse threshold (default 0.1, 0 = rebuild per iteration) [documented deviation]
//   ADELIE_HIP_CONS_HOST=1        box / one-sided constraint objects visited on the host instead of kernels_cons.hip  [A/B, tests]
//   ADELIE_HIP_GROUP_NEXT_CORR=0  group look-ahead: the next block's correction formed by its own solve              [A/B]
//   ADELIE_HIP_SPARSE_PANEL=0     IRLS on a design kept sparse on the full-Gram engines instead of the panel engine    [A/B, tests]
//   ADELIE_HIP_STD_PANEL=0        standardized dense / 2-bit views on their full-Gram engines instead of the panel engines [A/B, tests]
//   ADELIE_HIP_TIME_PANEL=1       per-launch HIP events around the panel step (bench.py's roofline leg)
//   ADELIE_HIP_TRACE=1|2          1: per-pass trace on stderr; 2: + enqueue / allocation / build timings
struct Hooks {
    long long cd_block_min_nv = -1; // -1: unset
    int panel_bsz = 0;
    int sparse_panel = -1;
    int std_panel = -1;      // ADELIE_HIP_STD_PANEL=0: a standardized dense / 2-bit view stays on its full-Gram engines            [A/B hook]   // ADELIE_HIP_SPARSE_PANEL=0: IRLS on a design kept sparse stays on the full-Gram engines      [A/B hook]
    int lookahead = -1, speculate = -1;
    double irls_reuse = -1;
    bool time_panel = false;
    int group_next_corr = -1; // ADELIE_HIP_GROUP_NEXT_CORR=0: every group solve forms its own look-ahead correction (A/B)
    bool cons_host = false; // ADELIE_HIP_CONS_HOST=1: box / one-sided objects on several coefficients visited on the host (A/B, tests)
    int trace = 0;
    int solve_sums = -1;    // ADELIE_HIP_SOLVE_SUMS=0: the sequential panel form keeps its panel_reduce launch per block             [A/B hook]
    int step_tail = -1;     // ADELIE_HIP_STEP_TAIL=0: 2-bit designs keep the panel_reduce launch behind every sequential step        [A/B hook]
    int step_means = -1;    // ADELIE_HIP_STEP_MEANS=1: IRLS takes the column means from the panel steps (no mean sweep per iteration)   [A/B hook]
    static Hooks from_env() {
        Hooks h;
        if (const char* e = std::getenv("ADELIE_HIP_CD_BLOCK_MIN_NV")) h.cd_block_min_nv = std::atoll(e);
        if (const char* e = std::getenv("ADELIE_HIP_PANEL_BSZ")) {
            const int v = std::atoi(e);
            h.panel_bsz = (v == 32 || v == 64 || v == 128) ? v : 0;
        }
        if (const char* e = std::getenv("ADELIE_HIP_LOOKAHEAD")) h.lookahead = std::atoi(e) != 0;
        if (const char* e = std::getenv("ADELIE_HIP_SPECULATE")) h.speculate = std::atoi(e) != 0;
        if (const char* e = std::getenv("ADELIE_HIP_IRLS_REUSE")) h.irls_reuse = std::max(0.0, std::atof(e));
        h.time_panel = std::getenv("ADELIE_HIP_TIME_PANEL") != nullptr;
        if (const char* e = std::getenv("ADELIE_HIP_CONS_HOST")) h.cons_host = std::atoi(e) != 0;
        if (const char* e = std::getenv("ADELIE_HIP_GROUP_NEXT_CORR")) h.group_next_corr = std::atoi(e) != 0;
        if (const char* e = std::getenv("ADELIE_HIP_SPARSE_PANEL")) h.sparse_panel = std::atoi(e) != 0;
        if (const char* e = std::getenv("ADELIE_HIP_STD_PANEL")) h.std_panel = std::atoi(e) != 0;
        if (const char* e = std::getenv("ADELIE_HIP_TRACE")) h.trace = std::max(1, std::atoi(e));
        if (const char* e = std::getenv("ADELIE_HIP_SOLVE_SUMS")) h.solve_sums = std::atoi(e) != 0;
        if (const char* e = std::getenv("ADELIE_HIP_STEP_MEANS")) h.step_means = std::atoi(e) != 0;
        if (const char* e = std::getenv("ADELIE_HIP_STEP_TAIL")) h.step_tail = std::atoi(e) != 0;
        return h;
    }
};


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

// Process-wide caches of what a solve would otherwise create and destroy every time: pinned host blocks (the staging arena,
// the host-mapped pass report: hipHostMalloc / hipHostFree pin and unpin pages, ~1 ms for the 8 MB arena) and non-blocking
// streams.  Same contract as DevPool below: an owner hands a block / stream back only after it synchronised what used it.
struct HostPool {
    struct Entry { size_t bytes; unsigned flags; void* p; };
    static std::mutex& mu() { static std::mutex* m = new std::mutex; return *m; }
    static std::vector<Entry>& parked() { static auto* v = new std::vector<Entry>; return *v; }
    static void* take(size_t bytes, unsigned flags) {
        {
            std::lock_guard<std::mutex> lk(mu());
            auto& v = parked();
            for (size_t i = 0; i < v.size(); ++i)
                if (v[i].bytes == bytes && v[i].flags == flags) {
                    void* p = v[i].p;
                    v[i] = v.back();
                    v.pop_back();
                    return p;
                }
        }
        void* hp = nullptr;
        if (hipHostMalloc(&hp, bytes, flags) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        return hp;
    }
    static void give(void* p, size_t bytes, unsigned flags) {
        if (!p) return;
        {
            std::lock_guard<std::mutex> lk(mu());
            if (parked().size() < 64) {
                parked().push_back(Entry{bytes, flags, p});
                return;
            }
        }
        (void)hipHostFree(p);
    }
};
struct StreamPool {
    struct Entry { int dev; hipStream_t s; };
    static std::mutex& mu() { static std::mutex* m = new std::mutex; return *m; }
    static std::vector<Entry>& parked() { static auto* v = new std::vector<Entry>; return *v; }
    static hipStream_t take() {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) throw core_error("adelie_hip: hipGetDevice failed");
        {
            std::lock_guard<std::mutex> lk(mu());
            auto& v = parked();
            for (size_t i = 0; i < v.size(); ++i)
                if (v[i].dev == dev) {
                    hipStream_t s = v[i].s;
                    v[i] = v.back();
                    v.pop_back();
                    return s;
                }
        }
        hipStream_t s = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess)
            throw core_error("adelie_hip: hipStreamCreateWithFlags failed");
        return s;
    }
    // `s` is idle (the owner synchronised it)
    static void give(hipStream_t s) {
        if (!s) return;
        int dev = 0;
        hipDevice_t sd;
        // the stream's own device (see DevPool::give); older runtimes without the query fall back to the current device
        const bool known = hipStreamGetDevice(s, &sd) == hipSuccess ? (dev = int(sd), true) : (hipGetDevice(&dev) == hipSuccess);
        if (known) {
            std::lock_guard<std::mutex> lk(mu());
            if (parked().size() < 64) {
                parked().push_back(Entry{dev, s});
                return;
            }
        }
        (void)hipStreamDestroy(s);
    }
};

// Pinned staging arena for the many small host<->device copies of a path (screen-set appends, per-fit scalars, coefficient
// downloads: ~15 per lambda).  hipMemcpyAsync on pageable host memory costs the calling thread ~20 us per copy on this
// platform (staging + an internal wait); from / to pinned memory it is an enqueue.  A solve installs its arena for the
// calling thread (Staging::Scope) and DevBuf::upload / download on the arena's stream go through it: uploads snapshot the
// host data into a slot, downloads land in a slot and reach their destination at the next Staging::flush() — which the
// owner calls after every host wait that covers the staged copies (stream synchronisation, or an event recorded behind
// them).  Slots are recycled after a full stream synchronisation (reset()) or up to a mark whose event was waited for (release()).  When the arena is exhausted the
// copies fall back to the pageable path (correct, slower).
struct Staging {
    char* base = nullptr;
    size_t cap = 0, head = 0, tail = 0; // ring: slots live in [tail, head) (modulo cap); head == tail: empty
    hipStream_t stream = nullptr;
    struct Pend { void* dst; const void* src; size_t bytes; };
    std::vector<Pend> pend;
    size_t n_fallback = 0; // copies on this arena's stream that did not fit and went the pageable way
    Staging() = default;
    Staging(const Staging&) = delete;
    Staging& operator=(const Staging&) = delete;
    ~Staging() { HostPool::give(base, cap, hipHostMallocDefault); } // (the owner synchronised its stream)
    void init(size_t bytes, hipStream_t s) {
        stream = s;
        if (base) return;
        if (void* hp = HostPool::take(bytes, hipHostMallocDefault)) {
            base = static_cast<char*>(hp);
            cap = bytes;
        }
    }
    void* take(size_t bytes) {
        if (!base) return nullptr;
        bytes = (bytes + 63) & ~size_t(63);
        size_t at;
        if (head >= tail) { // free: [head, cap) and [0, tail)
            if (head + bytes <= cap) at = head;
            else if (bytes < tail) at = 0;
            else return nullptr;
        } else {            // free: [head, tail)
            if (head + bytes < tail) at = head;
            else return nullptr;
        }
        head = at + bytes;
        return base + at;
    }
    // every staged download is complete (the owner waited for the stream, or for an event recorded behind them)
    void flush() {
        for (const Pend& q : pend) std::memcpy(q.dst, q.src, q.bytes);
        pend.clear();
    }
    // `m = mark()` taken when an event was recorded; after waiting for that event the slots handed out before it are free
    size_t mark() const { return head; }
    void release(size_t m) { tail = m; }
    // after a full stream synchronisation
    void reset() {
        flush();
        head = tail = 0;
    }
    static Staging*& current() { static thread_local Staging* cur = nullptr; return cur; }
    struct Scope {
        Staging* prev;
        explicit Scope(Staging* s) : prev(current()) { current() = s; }
        ~Scope() { current() = prev; }
    };
};

// Process-wide cache of device allocations.  Every solve builds ~100 DevBufs and drops them again (2.7 ms of hipMalloc /
// hipFree per headline path, 17 solves per 8-fold CV), and hipFree synchronises the device.  Released buffers are parked here by
// (device, size) and handed to the next request of a similar size; consecutive solves on one design ask for the same sizes,
// so after the first solve nothing is allocated.  Contents are NOT zeroed (neither does hipMalloc promise that).  A buffer is
// only parked by its owner's destructor / release(), i.e. after the owner's streams were synchronised.  At most
// `limit_bytes()` stay parked (default 6 GB; adelie_hip_set_config("pool_limit_mb", x); 0 disables the cache and
// "pool_trim" frees what is parked).
struct DevPool {
    struct Entry { size_t bytes; void* p; int dev; };
    static std::mutex& mu() { static std::mutex* m = new std::mutex; return *m; }
    static std::vector<Entry>& parked() { static auto* v = new std::vector<Entry>; return *v; } // (never destroyed: no HIP calls at exit)
    static size_t& parked_bytes() { static size_t b = 0; return b; }
    static size_t& limit_bytes() { static size_t b = size_t(6) << 30; return b; }
    static void* take(size_t bytes) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return nullptr;
        std::lock_guard<std::mutex> lk(mu());
        auto& v = parked();
        size_t best = v.size();
        for (size_t i = 0; i < v.size(); ++i)
            if (v[i].dev == dev && v[i].bytes >= bytes && v[i].bytes <= bytes + bytes / 8 + 4096 &&
                (best == v.size() || v[i].bytes < v[best].bytes))
                best = i;
        if (best == v.size()) return nullptr;
        void* p = v[best].p;
        parked_bytes() -= v[best].bytes;
        v[best] = v.back();
        v.pop_back();
        return p;
    }
    // returns false when the buffer was not parked (the caller frees it)
    static bool give(void* p, size_t bytes) {
        // the device the block LIVES on, not the calling thread's current one: a result may be destroyed from any thread
        // (C-ABI users, a garbage collector), and a block parked under the wrong device would be handed to a later solve there
        hipPointerAttribute_t at{};
        if (hipPointerGetAttributes(&at, p) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        const int dev = at.device;
        std::lock_guard<std::mutex> lk(mu());
        if (parked_bytes() + bytes > limit_bytes() || parked().size() >= 4096) return false;
        parked().push_back(Entry{bytes, p, dev});
        parked_bytes() += bytes;
        return true;
    }
    static void trim() {
        std::vector<Entry> v;
        {
            std::lock_guard<std::mutex> lk(mu());
            v.swap(parked());
            parked_bytes() = 0;
        }
        for (const Entry& e : v) (void)hipFree(e.p);
    }
};

// Device blocks a DevBuf outgrew in the middle of a solve.  hipFree waits for the whole device, i.e. it drains the queue the
// host has built up (≈ 5 growth events per headline path); with a list installed for the calling thread (run<T>) the old
// block is kept until the end of the solve instead and parked in DevPool once the solve's streams are idle, where the next
// solve's same growth step finds it.
struct DeferredFrees {
    struct Blk { void* p; size_t bytes; };
    std::vector<Blk> v;
    static DeferredFrees*& current() { static thread_local DeferredFrees* cur = nullptr; return cur; }
    struct Scope {
        DeferredFrees* prev;
        explicit Scope(DeferredFrees* d) : prev(current()) { current() = d; }
        ~Scope() { current() = prev; }
    };
    // nothing on the device uses the blocks any more
    void drain() {
        for (const Blk& b : v)
            if (!DevPool::give(b.p, b.bytes)) {
                const auto t0 = std::chrono::steady_clock::now();
                (void)hipFree(b.p);
                DevAllocStats::seconds() += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                ++DevAllocStats::calls();
            }
        v.clear();
    }
    ~DeferredFrees() { drain(); }
};

// Owning device buffer (grow-only).
template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    // The destructor parks the block in the process-wide cache: the owner has synchronised its streams by then.  A release in
    // the middle of a solve (growth) goes through hipFree, which waits for whatever may still be using the old block.
    ~DevBuf() { release(true); }
    void release(bool park = false) {
        if (p && !park && DeferredFrees::current()) {
            DeferredFrees::current()->v.push_back(DeferredFrees::Blk{p, cap * sizeof(T)});
        } else if (p && !(park && DevPool::give(p, cap * sizeof(T)))) {
            const auto t0 = std::chrono::steady_clock::now();
            (void)hipFree(p);
            DevAllocStats::seconds() += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            ++DevAllocStats::calls();
        }
        p = nullptr;
        cap = 0;
    }
    // allocation through the process-wide cache; `cap` is set to what the block really holds
    static T* alloc(size_t want, size_t& got) {
        // sizes are rounded up so that requests that differ a little between solves (screen sets of slightly different
        // lengths) still find their parked block
        size_t bytes = want * sizeof(T);
        const size_t gran = bytes < (size_t(1) << 16) ? 4096 : (bytes < (size_t(1) << 24) ? (size_t(1) << 16) : (size_t(1) << 21));
        bytes = (bytes + gran - 1) / gran * gran;
        void* q = DevPool::take(bytes);
        if (!q) {
            const auto t0 = std::chrono::steady_clock::now();
            hipError_t e = hipMalloc(&q, bytes);
            if (e != hipSuccess) { // out of memory with blocks parked: give them back and try once more
                (void)hipGetLastError();
                DevPool::trim();
                AHIP_CHECK(hipMalloc(&q, bytes));
            }
            DevAllocStats::seconds() += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            ++DevAllocStats::calls();
        }
        got = bytes / sizeof(T);
        return static_cast<T*>(q);
    }
    // ensure capacity >= n elements; contents are NOT preserved
    T* reserve(size_t n) {
        if (n > cap) {
            release();
            size_t want = n < 16 ? 16 : n;
            p = alloc(want, cap);
        }
        return p;
    }
    // ensure capacity, preserving the first `keep` elements
    T* grow(size_t n, size_t keep, hipStream_t s) {
        if (n <= cap) return p;
        size_t want = cap * 2 > n ? cap * 2 : n;
        if (want < 16) want = 16;
        size_t got = 0;
        T* q = alloc(want, got);
        if (p && keep) AHIP_CHECK(hipMemcpyAsync(q, p, keep * sizeof(T), hipMemcpyDeviceToDevice, s));
        if (p && DeferredFrees::current()) { // the copy is ordered on `s`; the old block lives until the end of the solve
            DeferredFrees::current()->v.push_back(DeferredFrees::Blk{p, cap * sizeof(T)});
        } else if (p) {
            AHIP_CHECK(hipStreamSynchronize(s));
            (void)hipFree(p);
        }
        p = q;
        cap = got;
        return p;
    }
    void upload(const T* h, size_t n, hipStream_t s, size_t off = 0) {
        if (!n) return;
        Staging* sg = Staging::current();
        if (sg && sg->stream == s) {
            if (void* slot = sg->take(n * sizeof(T))) {
                std::memcpy(slot, h, n * sizeof(T));
                AHIP_CHECK(hipMemcpyAsync(p + off, slot, n * sizeof(T), hipMemcpyHostToDevice, s));
                return;
            }
            ++sg->n_fallback;
        }
        AHIP_CHECK(hipMemcpyAsync(p + off, h, n * sizeof(T), hipMemcpyHostToDevice, s));
    }
    // NOTE: with a staging arena installed the data reaches `h` at the owner's next Staging::flush() (Solver::sync()).
    void download(T* h, size_t n, hipStream_t s, size_t off = 0) const {
        if (!n) return;
        Staging* sg = Staging::current();
        if (sg && sg->stream == s) {
            if (void* slot = sg->take(n * sizeof(T))) {
                AHIP_CHECK(hipMemcpyAsync(slot, p + off, n * sizeof(T), hipMemcpyDeviceToHost, s));
                sg->pend.push_back(Staging::Pend{h, slot, n * sizeof(T)});
                return;
            }
        }
        AHIP_CHECK(hipMemcpyAsync(h, p + off, n * sizeof(T), hipMemcpyDeviceToHost, s));
    }
};

} // namespace ahip

// ---- the opaque handles of include/adelie_hip.h ------------------------------------------------------------
struct adelie_hip_design {
    int dtype = ADELIE_HIP_F64;
    int device = 0;
    int kind = 0; // 0 dense, 1 snp (2-bit), 2 multi-response view of a dense or 2-bit design (adelie_hip_design_create_multi),
                  // 3 sparse kept sparse (adelie_hip_design_create_csc: CSC + CSR copies of the entries, kernels_sparse.hip)
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
    bool alias = false;     // shares X / bits / impute / the sparse arrays with another design (adelie_hip_design_alias): never frees them
    // sparse (kind 3): column-compressed and row-compressed copies of the stored entries
    int64_t* cptr = nullptr;
    int32_t* cidx = nullptr;
    void* cval = nullptr;
    int64_t* rptr = nullptr;
    int32_t* rcol = nullptr;
    void* rval = nullptr;
    int64_t nnz = 0;
    int64_t* bptr = nullptr; // per-column row-block pointers (p x (sp_nb + 1)) when sp_nb > 1
    int sp_nb = 1;
    int64_t sp_rb = 0;
    // standardized view of a sparse design (adelie_hip_design_create_csc_standardized): (p,) value_t each, owned unless alias
    // tile-major copy of the entries for the full sweeps (kernels_sparse.hip: csc_tile_sweep_kernel), sp_nt > 0
    int64_t* tptr = nullptr;
    uint16_t* trow = nullptr;
    void* tval = nullptr;
    int sp_nt = 0;
    int64_t sp_th = 0;
    int sp_parts() const { return sp_nb > sp_nt ? sp_nb : sp_nt; } // partial sums per column a sweep may leave
    void* std_center = nullptr;
    void* std_iscale = nullptr;
    bool std_owned = false;
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
    template <class T> ahip::CscView<T> csc() const {
        return ahip::CscView<T>{cptr, cidx, static_cast<const T*>(cval), rptr, rcol, static_cast<const T*>(rval), n, p, nnz,
                                bptr, sp_nb, sp_rb, static_cast<const T*>(std_center), static_cast<const T*>(std_iscale),
                                tptr, trow, static_cast<const T*>(tval), sp_nt, sp_th};
    }
    template <class T> ahip::MultiView<T> multi() const {
        return ahip::MultiView<T>{static_cast<const T*>(X), nb, pb, ld, static_cast<const T*>(ones), int32_t(mK), int32_t(micpt),
                                  bits, ldb, static_cast<const T*>(impute)};
    }
};
