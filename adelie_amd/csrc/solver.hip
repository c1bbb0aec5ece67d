// solver.hip — C ABI: the grpnet path driver (host side) over the device-resident state.
//
// Host logic restated from the reference (paths relative to adelie/src/include/adelie_core):
//   solve_core            solver/solver_base.hpp:435-687   (lambda_max bootstrap, path generation, BASIL loop)
//   screen / search_pivot solver/solver_base.hpp:273-403, optimization/search_pivot.hpp:7-62
//   kkt, early_exit       solver/solver_base.hpp:408-433, :241-263
//   update_screen_derived solver/solver_base.hpp:120-153, solver/solver_gaussian_naive.hpp:41-176
//   gaussian fit          solver/solver_gaussian_naive.hpp:209-349
//   glm (IRLS) fit        solver/solver_glm_naive.hpp:160-459
// These are O(G log G) scalar decisions per lambda and stay on the host; everything that touches an n- or
// p-vector or the design runs in the kernels of kernels_*.hip on the design's stream.  Per BASIL iteration the
// host receives: the CD kernel's scalar block, the screen coefficients (<= |S| values) and abs_grad (G values).
#include "common.hpp"

#include <algorithm>
#include <cstring>
#include <exception>
#include <type_traits>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <numeric>
#include <unordered_set>

namespace ahip {
void set_last_error(const std::string& s);
double g_hessian_min = 1e-24; // configs.hpp:6-21
double g_dbeta_tol = 1e-12;

// elementwise GLM kernels (kernels_glm.hip)
template <class T>
void launch_irls_prepare(int kind, const T* y, const T* w, const T* eta, const T* resid, const T* offsets, T hessian_min,
                         int64_t n, T* hess, T* irls_resid, T* irls_y, T* sums4, hipStream_t s, int K = 1);
template <class T>
void launch_irls_weights(const T* hess, T hess_sum, const T* irls_y, T shift, int64_t n, T* wts, T* irls_resid,
                         T* sums3, hipStream_t s);
template <class T>
void launch_irls_finish(int kind, const T* y, const T* w, const T* irls_y, const T* offsets, const T* irls_resid, T shift,
                        int64_t n, T* eta, T* resid, T* sums2, hipStream_t s, int K = 1);
template <class T>
void launch_glm_gradient(int kind, const T* y, const T* w, const T* eta, int64_t n, T* resid, hipStream_t s, int K = 1);
template <class T>
void launch_glm_loss(int kind, const T* y, const T* w, const T* eta, int64_t n, T* out1, hipStream_t s, int K = 1);
template <class T>
void launch_null_step(int kind, const T* y, const T* w, const T* eta, const T* resid, const T* offsets, T hessian_min,
                      int64_t n, T* sums2, hipStream_t s, int K = 1, const T* cb_hess = nullptr, const T* cb_z = nullptr);
template <class T>
void launch_set_eta(const T* offsets, T beta0, int64_t n, T* eta, hipStream_t s);
template <class T>
void launch_dot_diff(const T* a, const T* a0, const T* b, const T* b0, int64_t n, T* out1, hipStream_t s);
template <class T>
void launch_rel_change(const T* a, T* prev, int64_t n, T* out, hipStream_t s);
template <class T>
void launch_gather(const T* src, const int32_t* idx, int64_t cnt, T* dst, hipStream_t s);
template <class T>
void launch_scatter(const T* src, const int32_t* idx, int64_t cnt, T* dst, hipStream_t s); // dst[idx[a]] = src[a]
} // namespace ahip

using namespace ahip;
using idx = int64_t;

namespace {

struct Stopwatch {
    std::chrono::steady_clock::time_point t0;
    void start() { t0 = std::chrono::steady_clock::now(); }
    double elapsed() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

core_error max_cds_error(int l) {
    return make_solver_error("max coordinate descents reached at lambda index: " + std::to_string(l) + ".");
}
core_error max_screen_set_error() { return make_solver_error("maximum screen set size reached."); }

// cyclic Jacobi eigen-decomposition of a symmetric (q,q) matrix; stands in for Eigen::SelfAdjointEigenSolver
// (solver_gaussian_naive.hpp:113).  Eigenvalues ascending, V column-major with eigenvectors in columns.
void jacobi_eigh(int q, std::vector<double>& A, std::vector<double>& V, std::vector<double>& D) {
    V.assign(size_t(q) * q, 0.0);
    for (int i = 0; i < q; ++i) V[i + size_t(i) * q] = 1.0;
    auto a = [&](int i, int j) -> double& { return A[i + size_t(j) * q]; };
    auto v = [&](int i, int j) -> double& { return V[i + size_t(j) * q]; };
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0, dg = 0;
        for (int i = 0; i < q; ++i) {
            dg += a(i, i) * a(i, i);
            for (int j = i + 1; j < q; ++j) off += a(i, j) * a(i, j);
        }
        if (off == 0 || off <= 1e-32 * (dg + off)) break;
        for (int r = 0; r < q - 1; ++r)
            for (int c = r + 1; c < q; ++c) {
                const double arc = a(r, c);
                if (arc == 0.0) continue;
                const double theta = (a(c, c) - a(r, r)) / (2.0 * arc);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
                for (int k = 0; k < q; ++k) {
                    const double x = a(k, r), y = a(k, c);
                    a(k, r) = cs * x - sn * y;
                    a(k, c) = sn * x + cs * y;
                }
                for (int k = 0; k < q; ++k) {
                    const double x = a(r, k), y = a(c, k);
                    a(r, k) = cs * x - sn * y;
                    a(c, k) = sn * x + cs * y;
                }
                for (int k = 0; k < q; ++k) {
                    const double x = v(k, r), y = v(k, c);
                    v(k, r) = cs * x - sn * y;
                    v(k, c) = sn * x + cs * y;
                }
            }
    }
    std::vector<int> ord(q);
    std::iota(ord.begin(), ord.end(), 0);
    std::sort(ord.begin(), ord.end(), [&](int i, int j) { return a(i, i) < a(j, j); });
    std::vector<double> V2(size_t(q) * q);
    D.resize(q);
    for (int k = 0; k < q; ++k) {
        D[k] = a(ord[k], ord[k]);
        for (int i = 0; i < q; ++i) V2[i + size_t(k) * q] = v(i, ord[k]);
    }
    V.swap(V2);
}

// HIP-event timing of the device phases on the design's own stream (read by bench.py for the roofline object)
// (timing events are recycled through a process-wide free list: a headline path records ~2000 pairs)
struct TimingEvents {
    static std::mutex& mu() { static std::mutex* m = new std::mutex; return *m; }
    static std::vector<hipEvent_t>& idle() { static auto* v = new std::vector<hipEvent_t>; return *v; }
    static hipEvent_t take() {
        {
            std::lock_guard<std::mutex> lk(mu());
            auto& v = idle();
            if (!v.empty()) {
                hipEvent_t e = v.back();
                v.pop_back();
                return e;
            }
        }
        hipEvent_t e;
        AHIP_CHECK(hipEventCreate(&e));
        return e;
    }
    static void give(hipEvent_t e) {
        {
            std::lock_guard<std::mutex> lk(mu());
            if (idle().size() < 16384) {
                idle().push_back(e);
                return;
            }
        }
        (void)hipEventDestroy(e);
    }
};
struct KTimer {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    double ms = 0;
    int64_t launches = 0;
    void begin(hipStream_t st) {
        hipEvent_t a = TimingEvents::take(), b = TimingEvents::take();
        AHIP_CHECK(hipEventRecord(a, st));
        ev.emplace_back(a, b);
    }
    void end(hipStream_t st) { AHIP_CHECK(hipEventRecord(ev.back().second, st)); }
    std::vector<float> each; // per-launch milliseconds (debug / tuning)
    void collect() {
        for (auto& e : ev) {
            float t = 0;
            if (hipEventSynchronize(e.second) == hipSuccess && hipEventElapsedTime(&t, e.first, e.second) == hipSuccess) {
                ms += t;
                ++launches;
                each.push_back(t);
            }
            TimingEvents::give(e.first);
            TimingEvents::give(e.second);
        }
        ev.clear();
    }
    ~KTimer() { collect(); }
};

struct Counters {
    int64_t n_basil_iters = 0, n_sweeps = 0, n_cd_visits_screen = 0, n_cd_visits_active = 0, n_updates = 0,
            n_irls_iters = 0, n_new_screen_cols = 0, n_cd_passes_screen = 0, n_cd_passes_active = 0,
            n_gram_col_reads = 0, n_resid_col_reads = 0, n_panel_blocks = 0, n_panel_grams = 0, n_panel_cols = 0,
            n_irls_screen_cols = 0, n_sweeps_shared = 0, n_update_cols = 0;
    double gram_flops = 0;
};

template <class T>
struct FitOut {
    std::vector<idx> beta_idx;
    std::vector<T> beta_val;
    T intercept = 0, rsq = 0;
    double t_screen = 0, t_active = 0;
};

// ------------------------------------------------------------------------------------------------------------
// Sweep batching: solvers that run concurrently on one resident dense matrix (the folds of cv_grpnet, each from its own
// host thread on an alias handle) all spend most of their time in the same HBM-bound kernel, the full-gradient sweep X^T v.
// The K-wide sweep of kernels_multi.hip reads X once for K vectors, so the solvers that reach their sweep within a short
// window are answered by ONE pass over X.  A solver that arrives alone (or with batching off) takes its ordinary sweep.
// Results do not depend on who shares a batch: every vector's dot products are accumulated independently and in a fixed
// order; they differ from the ordinary sweep kernel's only by that order (last bits).
// ------------------------------------------------------------------------------------------------------------
int g_sweep_batch = 0; // adelie_hip_set_config("sweep_batch", 0/1)
std::mutex g_batcher_create_mutex;

struct SweepBatcher {
    static constexpr int KMAX = 8, NGEN = 4;
    std::mutex m;
    std::condition_variable cv;
    int registered = 0;
    struct Gen {
        int count = 0, K = 0, pending = 0; // pending: members of the launched batch that have not enqueued their pick yet
        hipEvent_t done = nullptr, in_ev[KMAX], out_ev[KMAX];
        bool out_set[KMAX], done_set = false, failed = false;
    };
    Gen gen[NGEN];
    uint64_t cur = 0, launched_upto = 0; // generation collecting arrivals; generations < launched_upto have been launched
    hipStream_t stream = nullptr;
    DevBuf<char> vbuf[NGEN], obuf[NGEN], work;
    // HIP-event timing of the shared sweeps on the batcher's stream (adelie_hip_design_batch_stats): launches, vectors
    // answered, milliseconds
    KTimer timer;
    int64_t n_vectors = 0;
    SweepBatcher() {
        AHIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        for (auto& g : gen) {
            AHIP_CHECK(hipEventCreateWithFlags(&g.done, hipEventDisableTiming));
            for (int k = 0; k < KMAX; ++k) {
                AHIP_CHECK(hipEventCreateWithFlags(&g.in_ev[k], hipEventDisableTiming));
                AHIP_CHECK(hipEventCreateWithFlags(&g.out_ev[k], hipEventDisableTiming));
                g.out_set[k] = false;
            }
        }
    }
    ~SweepBatcher() {
        if (stream) {
            (void)hipStreamSynchronize(stream);
            (void)hipStreamDestroy(stream);
        }
        for (auto& g : gen) {
            (void)hipEventDestroy(g.done);
            for (int k = 0; k < KMAX; ++k) {
                (void)hipEventDestroy(g.in_ev[k]);
                (void)hipEventDestroy(g.out_ev[k]);
            }
        }
    }
    void add() { std::lock_guard<std::mutex> lk(m); ++registered; }
    void remove() {
        { std::lock_guard<std::mutex> lk(m); --registered; }
        cv.notify_all();
    }
    // out[u] = X[:,u] . v - (sub_vec ? sub_scale[0] * sub_vec[u] : 0) on stream `ps`; returns false if the caller should run its
    // own sweep (it is the only solver around)
    template <class T>
    bool sweep(const DenseView<T>& X, const T* v, T* out, const T* sub_scale, const T* sub_vec, hipStream_t ps) {
        const int64_t n = X.n, p = X.p;
        std::unique_lock<std::mutex> lk(m);
        if (registered <= 1) return false;
        // join the generation that is collecting (a full one is launched by its leader before anybody can join again)
        while (gen[cur % NGEN].count >= KMAX || gen[cur % NGEN].pending > 0) cv.wait(lk);
        const uint64_t g = cur;
        Gen& G = gen[g % NGEN];
        const int slot = G.count++;
        char* vb = nullptr;
        try { // stage this solver's vector.  The mutex is held from the join above to the end of this block, so nobody has
              // joined after us yet: a failure here is undone by leaving the generation again, and nobody ever waits for us
            vb = vbuf[g % NGEN].reserve(size_t(KMAX) * size_t(n) * sizeof(T));
            if (G.done_set) AHIP_CHECK(hipStreamWaitEvent(ps, G.done, 0)); // the previous user of this buffer has been read
            AHIP_CHECK(hipMemcpyAsync(vb + size_t(slot) * size_t(n) * sizeof(T), v, size_t(n) * sizeof(T), hipMemcpyDeviceToDevice, ps));
            AHIP_CHECK(hipEventRecord(G.in_ev[slot], ps));
        } catch (...) {
            --G.count;
            lk.unlock();
            cv.notify_all();
            throw;
        }
        if (slot == 0) {
            // leader: give the others a window of about four sweep times (a sweep moves n*p values at ~7 TB/s), then launch
            // for whoever has arrived: waiting costs a lone solver at most that, sharing saves K - 1 sweeps
            // (8-fold CV, 100k x 10k: 1.05 s with a 0.25 ms window, 0.84 s with 0.8 ms, 0.70 s with 5 ms, no better beyond)
            const double sweep_us = double(n) * double(p) * double(sizeof(T)) / 7.0e6;
            const int window_us = int(std::min(5000.0, std::max(100.0, 4.0 * sweep_us)));
            const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(window_us);
            cv.wait_until(lk, deadline, [&] { return G.count >= std::min(registered, KMAX); });
            const int K = G.count;
            G.K = K;
            G.failed = false;
            ++cur; // later arrivals collect in the next generation
            try {
                T* ob = reinterpret_cast<T*>(obuf[g % NGEN].reserve(size_t(KMAX) * size_t(p) * sizeof(T)));
                for (int k = 0; k < K; ++k) AHIP_CHECK(hipStreamWaitEvent(stream, G.in_ev[k], 0));
                for (int k = 0; k < KMAX; ++k) // the previous readers of this output buffer are done
                    if (G.out_set[k]) { AHIP_CHECK(hipStreamWaitEvent(stream, G.out_ev[k], 0)); G.out_set[k] = false; }
                MultiView<T> mv{X.X, n, p, X.ld, nullptr, int32_t(K), 0};
                T* wk = reinterpret_cast<T*>(work.reserve(size_t(multi_sweep_work_elems<T>(MultiView<T>{X.X, n, p, X.ld, nullptr, KMAX, 0})) * sizeof(T)));
                timer.begin(stream);
                launch_multi_sweep<T>(mv, reinterpret_cast<const T*>(vb), ob, wk, stream);
                timer.end(stream);
                n_vectors += K;
                AHIP_CHECK(hipEventRecord(G.done, stream));
                G.done_set = true;
            } catch (...) {
                G.failed = true; // the members of this batch must not wait for a launch that did not happen
            }
            G.count = 0; // the slot bookkeeping of this generation index restarts when it comes round again
            G.pending = K;
            launched_upto = g + 1;
            lk.unlock();
            cv.notify_all();
            lk.lock();
        } else {
            cv.notify_all(); // the leader may be waiting for the last arrival
            cv.wait(lk, [&] { return launched_upto > g; });
        }
        // every member of a launched generation accounts for itself in `pending` exactly once, whatever happens to its pick:
        // a generation whose count of pending members never reaches zero would stall every solver once `cur` wraps to it
        const int K = G.K;
        std::exception_ptr err;
        if (G.failed) {
            err = std::make_exception_ptr(core_error("adelie_hip: the shared sweep of a batch of concurrent solves failed to launch."));
        } else {
            try {
                const T* ob = reinterpret_cast<const T*>(obuf[g % NGEN].p);
                AHIP_CHECK(hipStreamWaitEvent(ps, G.done, 0));
                launch_batch_pick<T>(ob, p, K, slot, sub_scale, sub_vec, out, ps);
                AHIP_CHECK(hipEventRecord(G.out_ev[slot], ps));
                G.out_set[slot] = true;
            } catch (...) {
                err = std::current_exception();
            }
        }
        --G.pending;
        lk.unlock();
        cv.notify_all();
        if (err) std::rethrow_exception(err);
        return true;
    }
};

SweepBatcher* batcher_of(adelie_hip_design* d) {
    adelie_hip_design* owner = d->batch_owner ? d->batch_owner : d;
    std::lock_guard<std::mutex> lk(g_batcher_create_mutex);
    if (!owner->batcher) owner->batcher = new SweepBatcher();
    return static_cast<SweepBatcher*>(owner->batcher);
}

// ------------------------------------------------------------------------------------------------------------
// Solver: host mirror of StateGaussianNaive / StateGlmNaive + the device-resident working set
// ------------------------------------------------------------------------------------------------------------
template <class T>
struct Solver {
#include "solver_builds.hpp"
#include "solver_screen.hpp"
#include "solver_panel.hpp"
#include "solver_fit.hpp"
#include "solver_path.hpp"
};

struct ResultBase {
    int device = -1; // the device the solve ran on
    virtual ~ResultBase() {}
    virtual void sync_live() = 0;
    virtual int64_t size(int which) const = 0;
    virtual int copy(int which, void* out, int64_t cap) const = 0;
    virtual double scalar(int which) const = 0;
    virtual const char* err() const = 0;
};

template <class V>
void cp_d(const V& v, double* out, int64_t cap) {
    const int64_t m = std::min<int64_t>(cap, int64_t(v.size()));
    for (int64_t i = 0; i < m; ++i) out[i] = double(v[i]);
}
template <class V>
void cp_i(const V& v, int64_t* out, int64_t cap) {
    const int64_t m = std::min<int64_t>(cap, int64_t(v.size()));
    for (int64_t i = 0; i < m; ++i) out[i] = int64_t(v[i]);
}

template <class T>
struct Result : ResultBase {
    Solver<T> s;
    void sync_live() override { s.download_invariants(); }
    int64_t size(int which) const override {
        switch (which) {
            case ADELIE_HIP_V_INTERCEPTS: return s.intercepts.size();
            case ADELIE_HIP_V_DEVS: return s.devs.size();
            case ADELIE_HIP_V_LMDAS: return s.lmdas.size();
            case ADELIE_HIP_V_LMDA_PATH: return s.lmda_path.size();
            case ADELIE_HIP_V_SCREEN_BETA: return s.screen_beta.size();
            case ADELIE_HIP_V_GRAD: return s.grad.size();
            case ADELIE_HIP_V_ABS_GRAD: return s.abs_grad.size();
            case ADELIE_HIP_V_RESID: return s.resid.size();
            case ADELIE_HIP_V_ETA: return s.eta.size();
            case ADELIE_HIP_V_SCREEN_X_MEANS: return s.screen_X_means.size();
            case ADELIE_HIP_V_SCREEN_VARS: return s.screen_vars.size();
            case ADELIE_HIP_V_SCREEN_TRANSFORMS: { int64_t t = 0; for (auto& v : s.screen_transforms) t += v.size(); return t; }
            case ADELIE_HIP_V_BENCHMARK_SCREEN: return s.benchmark_screen.size();
            case ADELIE_HIP_V_BENCHMARK_FIT_SCREEN: return s.benchmark_fit_screen.size();
            case ADELIE_HIP_V_BENCHMARK_FIT_ACTIVE: return s.benchmark_fit_active.size();
            case ADELIE_HIP_V_BENCHMARK_KKT: return s.benchmark_kkt.size();
            case ADELIE_HIP_V_BENCHMARK_INVARIANCE: return s.benchmark_invariance.size();
            case ADELIE_HIP_I_SCREEN_SET: return s.screen_set.size();
            case ADELIE_HIP_I_SCREEN_BEGINS: return s.screen_begins.size();
            case ADELIE_HIP_I_SCREEN_IS_ACTIVE: return s.screen_is_active.size();
            case ADELIE_HIP_I_ACTIVE_SET: return s.active_set.size();
            case ADELIE_HIP_I_N_VALID_SOLUTIONS: return s.n_valid_solutions.size();
            case ADELIE_HIP_I_ACTIVE_SIZES: return s.active_sizes.size();
            case ADELIE_HIP_I_SCREEN_SIZES: return s.screen_sizes.size();
            case ADELIE_HIP_I_BETAS_INDPTR: return s.betas_idx.size() + 1;
            case ADELIE_HIP_I_BETAS_INDICES:
            case ADELIE_HIP_V_BETAS_VALUES: { int64_t t = 0; for (auto& v : s.betas_idx) t += v.size(); return t; }
            case ADELIE_HIP_I_DUALS_INDPTR: return s.duals_idx.size() + 1;
            case ADELIE_HIP_I_DUALS_INDICES:
            case ADELIE_HIP_V_DUALS_VALUES: { int64_t t = 0; for (auto& v : s.duals_idx) t += v.size(); return t; }
            case ADELIE_HIP_V_CONSTRAINT_MU: return s.cons_on ? s.G : 0;
            case ADELIE_HIP_V_CONSTRAINT_VMU: return s.cons_dev ? s.p : 0;
            case ADELIE_HIP_I_CONSTRAINT_DEV_GROUPS: return s.cons_dev ? int64_t(s.devcons_list.size()) : 0;
        }
        return -1;
    }
    int copy(int which, void* out, int64_t cap) const override {
        double* d = (double*)out;
        int64_t* ii = (int64_t*)out;
        switch (which) {
            case ADELIE_HIP_V_INTERCEPTS: cp_d(s.intercepts, d, cap); return 0;
            case ADELIE_HIP_V_DEVS: cp_d(s.devs, d, cap); return 0;
            case ADELIE_HIP_V_LMDAS: cp_d(s.lmdas, d, cap); return 0;
            case ADELIE_HIP_V_LMDA_PATH: cp_d(s.lmda_path, d, cap); return 0;
            case ADELIE_HIP_V_SCREEN_BETA: cp_d(s.screen_beta, d, cap); return 0;
            case ADELIE_HIP_V_GRAD: cp_d(s.grad, d, cap); return 0;
            case ADELIE_HIP_V_ABS_GRAD: cp_d(s.abs_grad, d, cap); return 0;
            case ADELIE_HIP_V_RESID: cp_d(s.resid, d, cap); return 0;
            case ADELIE_HIP_V_ETA: cp_d(s.eta, d, cap); return 0;
            case ADELIE_HIP_V_SCREEN_X_MEANS: cp_d(s.screen_X_means, d, cap); return 0;
            case ADELIE_HIP_V_SCREEN_VARS: cp_d(s.screen_vars, d, cap); return 0;
            case ADELIE_HIP_V_SCREEN_TRANSFORMS: {
                int64_t k = 0;
                for (auto& v : s.screen_transforms)
                    for (auto x : v) { if (k < cap) d[k] = double(x); ++k; }
                return 0;
            }
            case ADELIE_HIP_V_BENCHMARK_SCREEN: cp_d(s.benchmark_screen, d, cap); return 0;
            case ADELIE_HIP_V_BENCHMARK_FIT_SCREEN: cp_d(s.benchmark_fit_screen, d, cap); return 0;
            case ADELIE_HIP_V_BENCHMARK_FIT_ACTIVE: cp_d(s.benchmark_fit_active, d, cap); return 0;
            case ADELIE_HIP_V_BENCHMARK_KKT: cp_d(s.benchmark_kkt, d, cap); return 0;
            case ADELIE_HIP_V_BENCHMARK_INVARIANCE: cp_d(s.benchmark_invariance, d, cap); return 0;
            case ADELIE_HIP_I_SCREEN_SET: cp_i(s.screen_set, ii, cap); return 0;
            case ADELIE_HIP_I_SCREEN_BEGINS: cp_i(s.screen_begins, ii, cap); return 0;
            case ADELIE_HIP_I_SCREEN_IS_ACTIVE: cp_i(s.screen_is_active, ii, cap); return 0;
            case ADELIE_HIP_I_ACTIVE_SET: cp_i(s.active_set, ii, cap); return 0;
            case ADELIE_HIP_I_N_VALID_SOLUTIONS: cp_i(s.n_valid_solutions, ii, cap); return 0;
            case ADELIE_HIP_I_ACTIVE_SIZES: cp_i(s.active_sizes, ii, cap); return 0;
            case ADELIE_HIP_I_SCREEN_SIZES: cp_i(s.screen_sizes, ii, cap); return 0;
            case ADELIE_HIP_I_BETAS_INDPTR: {
                int64_t acc = 0, k = 0;
                if (k < cap) ii[k] = 0;
                ++k;
                for (auto& v : s.betas_idx) { acc += v.size(); if (k < cap) ii[k] = acc; ++k; }
                return 0;
            }
            case ADELIE_HIP_I_BETAS_INDICES: {
                int64_t k = 0;
                for (auto& v : s.betas_idx) for (auto x : v) { if (k < cap) ii[k] = x; ++k; }
                return 0;
            }
            case ADELIE_HIP_V_BETAS_VALUES: {
                int64_t k = 0;
                for (auto& v : s.betas_val) for (auto x : v) { if (k < cap) d[k] = double(x); ++k; }
                return 0;
            }
            case ADELIE_HIP_I_DUALS_INDPTR: {
                int64_t acc = 0, k = 0;
                if (k < cap) ii[k] = 0;
                ++k;
                for (auto& v : s.duals_idx) { acc += v.size(); if (k < cap) ii[k] = acc; ++k; }
                return 0;
            }
            case ADELIE_HIP_I_DUALS_INDICES: {
                int64_t k = 0;
                for (auto& v : s.duals_idx) for (auto x : v) { if (k < cap) ii[k] = x; ++k; }
                return 0;
            }
            case ADELIE_HIP_I_CONSTRAINT_DEV_GROUPS: {
                if (!s.cons_dev) return 0;
                int64_t k = 0;
                for (int32_t g : s.devcons_list) { if (k < cap) ii[k] = g; ++k; }
                return 0;
            }
            case ADELIE_HIP_V_DUALS_VALUES: {
                int64_t k = 0;
                for (auto& v : s.duals_val) for (auto x : v) { if (k < cap) d[k] = double(x); ++k; }
                return 0;
            }
            case ADELIE_HIP_V_CONSTRAINT_MU: {
                if (!s.cons_on) return 0;
                for (idx g = 0; g < s.G && g < cap; ++g) d[g] = s.cons_kind[g] ? double(s.cons_dual_of(g)) : 0.0;
                return 0;
            }
            case ADELIE_HIP_V_CONSTRAINT_VMU: {
                if (!s.cons_dev) return 0;
                cp_d(s.cons_vmu, d, cap);
                return 0;
            }
        }
        return 1;
    }
    double scalar(int which) const override {
        switch (which) {
            case ADELIE_HIP_S_LMDA_MAX: return s.lmda_max;
            case ADELIE_HIP_S_LMDA: return s.lmda;
            case ADELIE_HIP_S_RSQ: return s.rsq;
            case ADELIE_HIP_S_RESID_SUM: return s.resid_sum;
            case ADELIE_HIP_S_ACTIVE_SET_SIZE: return double(s.active_set_size);
            case ADELIE_HIP_S_BETA0: return s.beta0;
            case ADELIE_HIP_S_LOSS_NULL: return s.loss_null;
            case ADELIE_HIP_S_LOSS_FULL: return s.loss_full;
            case ADELIE_HIP_S_TOTAL_TIME: return s.total_time;
            case ADELIE_HIP_S_N_BASIL_ITERS: return double(s.cnt.n_basil_iters);
            case ADELIE_HIP_S_N_SWEEPS: return double(s.cnt.n_sweeps);
            case ADELIE_HIP_S_N_CD_VISITS_SCREEN: return double(s.cnt.n_cd_visits_screen);
            case ADELIE_HIP_S_N_CD_VISITS_ACTIVE: return double(s.cnt.n_cd_visits_active);
            case ADELIE_HIP_S_N_UPDATES: return double(s.cnt.n_updates);
            case ADELIE_HIP_S_N_IRLS_ITERS: return double(s.cnt.n_irls_iters);
            case ADELIE_HIP_S_N_NEW_SCREEN_COLS: return double(s.cnt.n_new_screen_cols);
            case ADELIE_HIP_S_N_CD_PASSES_SCREEN: return double(s.cnt.n_cd_passes_screen);
            case ADELIE_HIP_S_N_CD_PASSES_ACTIVE: return double(s.cnt.n_cd_passes_active);
            case ADELIE_HIP_S_N_GRAM_COL_READS: return double(s.cnt.n_gram_col_reads);
            case ADELIE_HIP_S_N_RESID_COL_READS: return double(s.cnt.n_resid_col_reads);
            case ADELIE_HIP_S_GRAM_FLOPS: return s.cnt.gram_flops;
            case ADELIE_HIP_S_N_PANEL_BLOCKS: return double(s.cnt.n_panel_blocks);
            case ADELIE_HIP_S_N_PANEL_GRAMS: return double(s.cnt.n_panel_grams);
            case ADELIE_HIP_S_N_PANEL_COLS: return double(s.cnt.n_panel_cols);
            case ADELIE_HIP_S_N_IRLS_SCREEN_COLS: return double(s.cnt.n_irls_screen_cols);
            case ADELIE_HIP_S_N_SPECULATED: return double(s.n_spec);
            case ADELIE_HIP_S_N_SPEC_ROLLBACKS: return double(s.n_spec_rollback);
            case ADELIE_HIP_S_T_PANEL_STEP_MS: return s.t_step.ms;
            case ADELIE_HIP_S_N_PANEL_STEP_LAUNCHES: return double(s.t_step.launches);
            case ADELIE_HIP_S_T_SWEEP_MS: return s.t_sweep.ms;
            case ADELIE_HIP_S_T_GRAM_MS: return s.t_gram.ms;
            case ADELIE_HIP_S_T_CD_MS: return s.t_cd.ms;
            case ADELIE_HIP_S_T_AXPY_MS: return s.t_axpy.ms;
            case ADELIE_HIP_S_N_SWEEP_LAUNCHES: return double(s.t_sweep.launches);
            case ADELIE_HIP_S_N_GRAM_LAUNCHES: return double(s.t_gram.launches);
            case ADELIE_HIP_S_T_HOST_SCREEN_MS: return 1e3 * s.t_host_screen;
            case ADELIE_HIP_S_T_HOST_SCREEN_WAIT_MS: return 1e3 * s.t_host_screen_wait;
            case ADELIE_HIP_S_N_SWEEPS_SHARED: return double(s.cnt.n_sweeps_shared);
            case ADELIE_HIP_S_N_UPDATE_COLS: return double(s.cnt.n_update_cols);
            case ADELIE_HIP_S_N_DEVICE_SCREENS: return 0.0; /* (screening on the device was measured slower and removed in round 4) */
            case ADELIE_HIP_S_N_HOST_SCREENS: return double(s.n_host_screens);
            case ADELIE_HIP_S_N_HOST_CONS_VISITS: return double(s.n_host_cons_visits);
            case ADELIE_HIP_S_N_DEV_CONS_VISITS: return double(s.n_dev_cons_visits_final);
            default:
                if (which >= 900 && which < 908) return double(s.cd_dbg[which - 900]);
                if (which >= 910 && which < 918) return 1e3 * s.t_host[which - 910];
        }
        return std::numeric_limits<double>::quiet_NaN();
    }
    const char* err() const override { return s.error.c_str(); }
};

template <class T>
void run(adelie_hip_design* X, const adelie_hip_grpnet_args* a, adelie_hip_result* res);

} // namespace

struct adelie_hip_result {
    ResultBase* r = nullptr;
    ~adelie_hip_result() { delete r; }
};

namespace {

template <class T>
void run(adelie_hip_design* X, const adelie_hip_grpnet_args* a, adelie_hip_result* res) {
    auto* r = new Result<T>();
    res->r = r; // owned by `res` from here on (the poll callbacks read the live state through it)
    r->device = X->device;
    r->s.live = res;
    r->s.stage.init(size_t(8) << 20, X->stream);
    Staging::Scope stage_scope(&r->s.stage);
    DeferredFrees::Scope deferred_scope(&r->s.deferred);
    Stopwatch sw_build;
    sw_build.start();
    r->s.build(X, a);
    if (r->s.hooks.trace >= 2) std::fprintf(stderr, "[build] state set-up %.2f ms\n", sw_build.elapsed() * 1e3);
    Stopwatch sw;
    sw.start();
    struct BatchGuard { // registered for sweep batching exactly while the path runs
        SweepBatcher* b = nullptr;
        ~BatchGuard() { if (b) b->remove(); }
    } guard;
    if (g_sweep_batch && X->kind == 0) {
        guard.b = batcher_of(X);
        guard.b->add();
        r->s.batcher = guard.b;
    }
    try {
        r->s.solve();
    } catch (const std::exception& e) {
        r->s.error = e.what();
    }
    r->s.batcher = nullptr;
    if (guard.b) {
        guard.b->remove();
        guard.b = nullptr;
    }
    try {
        r->s.finalize();
    } catch (const std::exception& e) {
        if (r->s.error.empty()) r->s.error = e.what();
    }
    r->s.total_time = sw.elapsed();
    if (r->s.hooks.trace >= 2) std::fprintf(stderr, "[run] solve + finalize %.2f ms\n", r->s.total_time * 1e3);
    r->s.live = nullptr;
}

} // namespace

void adelie_hip_internal_free_batcher(void* b) { delete static_cast<SweepBatcher*>(b); }
void adelie_hip_internal_batch_stats(void* b, double* out) {
    out[0] = out[1] = out[2] = 0;
    if (!b) return;
    auto* sb = static_cast<SweepBatcher*>(b);
    std::lock_guard<std::mutex> lk(sb->m);
    sb->timer.collect();
    out[0] = double(sb->timer.launches);
    out[1] = double(sb->n_vectors);
    out[2] = sb->timer.ms;
}

extern "C" {

int adelie_hip_set_config(const char* name, double value) {
    const std::string nm(name ? name : "");
    if (nm == "hessian_min") g_hessian_min = value;
    else if (nm == "dbeta_tol") g_dbeta_tol = value;
    else if (nm == "sweep_batch") g_sweep_batch = value != 0.0 ? 1 : 0;
    else if (nm == "pool_limit_mb") {
        DevPool::limit_bytes() = value > 0 ? size_t(value) << 20 : 0;
        if (value <= 0) DevPool::trim();
    } else if (nm == "pool_trim") DevPool::trim();
    else {
        set_last_error("adelie_core: unknown config name.");
        return 1;
    }
    return 0;
}

static int solve_entry(adelie_hip_design* X, const adelie_hip_grpnet_args* args, adelie_hip_result** out, bool cov);
int adelie_hip_grpnet_solve(adelie_hip_design* X, const adelie_hip_grpnet_args* args, adelie_hip_result** out) {
    return solve_entry(X, args, out, false);
}
int adelie_hip_gaussian_cov_solve(adelie_hip_design* A, const adelie_hip_grpnet_args* args, adelie_hip_result** out) {
    return solve_entry(A, args, out, true);
}
static int solve_entry(adelie_hip_design* X, const adelie_hip_grpnet_args* args, adelie_hip_result** out, bool cov) {
    try {
        if (!X || !args || !out) throw make_core_error("null argument.");
        if (cov && !X->cov) throw make_core_error("A must be a covariance matrix (matrix.dense(method=\"cov\")).");
        if (!cov && X->cov) throw make_core_error("X is a covariance matrix: use gaussian_cov for the covariance method.");
        auto* res = new adelie_hip_result();
        try {
            if (X->dtype == ADELIE_HIP_F64) run<double>(X, args, res);
            else run<float>(X, args, res);
        } catch (...) {
            delete res;
            throw;
        }
        *out = res;
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return 1;
    }
    return 0;
}
int adelie_hip_grpnet_solve_many(adelie_hip_design* const* X, const adelie_hip_grpnet_args* const* args, int32_t count,
                                 adelie_hip_result** out, adelie_hip_done_fn on_done, void* user) {
    try {
        if (count < 0 || (count > 0 && (!X || !args || !out))) throw make_core_error("null argument.");
        for (int32_t k = 0; k < count; ++k) {
            out[k] = nullptr;
            if (!X[k] || !args[k]) throw make_core_error("null argument.");
            for (int32_t m = 0; m < k; ++m)
                if (X[m] == X[k]) throw make_core_error("solve_many: a design handle appears twice (concurrent solves need a handle each: adelie_hip_design_alias).");
        }
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return 1;
    }
    std::vector<std::string> msg(static_cast<size_t>(count));
    std::vector<int> rc(static_cast<size_t>(count), 0);
    auto one = [&](int32_t k) {
        rc[size_t(k)] = solve_entry(X[k], args[k], &out[k], false);
        if (rc[size_t(k)] != 0) msg[size_t(k)] = adelie_hip_last_error(); // (thread-local: read it on the solve's own thread)
        if (on_done) on_done(k, rc[size_t(k)], user);
    };
    if (count == 1) {
        one(0);
    } else {
        std::vector<std::thread> th;
        th.reserve(static_cast<size_t>(count));
        for (int32_t k = 0; k < count; ++k) th.emplace_back(one, k);
        for (auto& t : th) t.join();
    }
    for (int32_t k = 0; k < count; ++k)
        if (rc[size_t(k)] != 0) {
            set_last_error("solve " + std::to_string(k) + ": " + msg[size_t(k)]);
            return 1;
        }
    return 0;
}
int adelie_hip_result_destroy(adelie_hip_result* r) {
    if (r && r->r && r->r->device >= 0) (void)hipSetDevice(r->r->device); // (its buffers and streams are parked per device)
    delete r;
    return 0;
}
int64_t adelie_hip_result_size(const adelie_hip_result* r, int which) { return r->r->size(which); }
int adelie_hip_result_copy(const adelie_hip_result* r, int which, void* out, int64_t cap) { return r->r->copy(which, out, cap); }
double adelie_hip_result_scalar(const adelie_hip_result* r, int which) { return r->r->scalar(which); }
const char* adelie_hip_result_error(const adelie_hip_result* r) { return r->r->err(); }
int adelie_hip_result_sync(const adelie_hip_result* live) {
    try {
        if (!live || !live->r) throw make_core_error("null argument.");
        live->r->sync_live();
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return 1;
    }
    return 0;
}

// Times `reps` launches of the dominant kernel with HIP events on the design's own stream.
int adelie_hip_bench_sweep(adelie_hip_design* d, int64_t reps, double* ms_per_launch) {
    try {
        if (!d || !ms_per_launch || reps <= 0) throw make_core_error("bad arguments.");
        AHIP_CHECK(hipSetDevice(d->device));
        hipStream_t s = d->stream;
        auto body = [&](auto tag) {
            using T = decltype(tag);
            DevBuf<T> v, out, xm, work, sc;
            v.reserve(d->n); out.reserve(d->p); xm.reserve(d->p); sc.reserve(1);
            work.reserve(size_t(d->kind == 3 ? sweep_work_elems_csc(d->sp_parts(), d->p) : sweep_work_elems(d->n, d->p)));
            launch_fill<T>(v.p, T(1) / T(d->n), d->n, s);
            launch_fill<T>(xm.p, T(0.5), d->p, s);
            launch_fill<T>(sc.p, T(0.25), 1, s);
            auto once = [&]() {
                if (d->kind == 0) launch_sweep<T>(d->dense<T>(), v.p, out.p, 0, d->p, nullptr, sc.p, xm.p, false, work.p, s);
                else if (d->kind == 3) launch_sweep_csc<T>(d->csc<T>(), v.p, out.p, 0, d->p, nullptr, sc.p, xm.p, false, work.p, s);
                else launch_sweep_snp<T>(d->snp(), static_cast<const T*>(d->impute), v.p, out.p, 0, d->p, nullptr, sc.p, xm.p, false, work.p, s);
            };
            once();
            AHIP_CHECK(hipStreamSynchronize(s));
            hipEvent_t e0, e1;
            AHIP_CHECK(hipEventCreate(&e0));
            AHIP_CHECK(hipEventCreate(&e1));
            AHIP_CHECK(hipEventRecord(e0, s));
            for (int64_t i = 0; i < reps; ++i) once();
            AHIP_CHECK(hipEventRecord(e1, s));
            AHIP_CHECK(hipEventSynchronize(e1));
            float ms = 0;
            AHIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
            *ms_per_launch = double(ms) / double(reps);
        };
        if (d->dtype == ADELIE_HIP_F64) body(double{});
        else body(float{});
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return 1;
    }
    return 0;
}

} // extern "C"
