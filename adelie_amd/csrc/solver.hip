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
    adelie_hip_design* D = nullptr;
    hipStream_t st = nullptr;
    idx n, p, G;
    // ---- static inputs (host copies) ----
    std::vector<idx> groups, group_sizes;
    std::vector<T> penalty;
    T alpha, min_ratio;
    size_t lmda_path_size, max_screen_size, max_active_size;
    T pivot_subset_ratio;
    size_t pivot_subset_min;
    T pivot_slack_ratio;
    int screen_rule;
    size_t max_iters;
    T tol, adev_tol, ddev_tol, newton_tol;
    size_t newton_max_iters;
    bool early_exit_, setup_lmda_max, setup_lmda_path, intercept;
    int glm_kind;
    adelie_hip_poll_fn poll;
    void* poll_user;
    const adelie_hip_result* live = nullptr; // the handle poll() receives: the state being solved (py_state.cpp:62-91)
    // covariance method (StateGaussianCov, state_gaussian_cov.hpp:40-145): D holds A (p x p), there is no residual; the
    // invariant is grad = v - A beta and the Gram engines iterate on C = A[S, S]
    bool cov_mode = false;
    T rdev_tol = 0;
    DevBuf<T> d_covv, d_zero;
    // one-coefficient constraints (args constraint_*; ConstraintBox / ConstraintOneSided, adelie_core/constraint/): the host
    // keeps them as passed (kind, a, b) for the dual's convention, the device sees the unified form lo <= beta <= hi with
    // lo <= 0 <= hi (+-inf where there is no bound) and the signed multiplier mu_+ - mu_- (the term the constraint adds to
    // the coordinate's gradient; a one-sided constraint's dual is sgn times it)
    Hooks hooks;
    bool cons_on = false;
    std::vector<int32_t> cons_kind;
    std::vector<T> cons_a, cons_lo, cons_hi, cons_mu; // (G,)
    std::vector<idx> dual_groups;
    DevBuf<T> d_clo, d_chi, d_cmu;                    // per screen value
    DevBuf<T> d_clo_g, d_chi_g, d_mu_g;               // per group (abs_grad of groups outside the screen set: solve_zero)
    std::vector<std::vector<idx>> duals_idx;
    std::vector<std::vector<T>> duals_val;
    T cons_dual_of(idx g) const { return cons_kind[g] == 2 ? cons_a[g] * cons_mu[g] : cons_mu[g]; }
    // Constraint objects on the caller's side (kind ADELIE_HIP_CONSTRAINT_HOST: several coefficients, user-defined classes):
    // their group is a block of its own in every pass and is visited on the host between two panel steps (host_group_visit),
    // abs_grad and the duals ask the object through the callbacks (host_cons_abs_grad, update_solutions)
    bool cons_host = false;
    const adelie_hip_constraint_callbacks* cons_cb = nullptr;
    std::vector<idx> cons_m; // (G,) multipliers per group
    bool host_cons(idx g) const { return cons_host && cons_kind[g] == ADELIE_HIP_CONSTRAINT_HOST; }
    adelie_hip_glm_callbacks glm_cb{};       // glm_kind == CALLBACK: the user's GlmBase subclass, evaluated on the host
    std::vector<T> cb_eta, cb_grad, cb_hess, cb_z;
    idx max_gs = 1;
    bool all_scalar = true;
    // ---- dynamic host state ----
    T lmda_max;
    std::vector<T> lmda_path;
    std::vector<uint8_t> in_screen; // role of screen_hashset (state_base.hpp): membership bitmap over the G groups
    std::vector<int32_t> slot_host; // group -> screen value offset (-1: not screened), mirrored in d_slot
    std::vector<idx> screen_set, screen_begins;
    std::vector<T> screen_beta;
    std::vector<int8_t> screen_is_active;
    size_t active_set_size;
    std::vector<idx> active_set;
    std::vector<idx> active_order; // positions of active_set sorted by design column (kept across fits)
    T lmda;
    std::vector<T> grad, abs_grad, X_means, resid, eta;
    std::vector<T> screen_X_means, screen_vars;
    std::vector<std::vector<T>> screen_transforms;
    T y_mean = 0, y_var = 0, loss_null = 0, loss_full = 0, rsq = 0, resid_sum = 0, beta0 = 0;
    size_t irls_max_iters = 0;
    T irls_tol = 0;
    bool setup_loss_null = false;
    // outputs
    std::vector<std::vector<idx>> betas_idx;
    std::vector<std::vector<T>> betas_val;
    std::vector<T> intercepts, devs, lmdas;
    std::vector<double> benchmark_screen, benchmark_fit_screen, benchmark_fit_active, benchmark_kkt, benchmark_invariance;
    std::vector<int> n_valid_solutions, active_sizes, screen_sizes;
    Counters cnt;
    KTimer t_sweep, t_gram, t_cd, t_axpy, t_step;
    bool time_panel = false;
    int64_t cd_dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<std::pair<idx, idx>> gram_shapes;
    double t_host[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // wall-clock split of solve(): screen logic, append, gram+vars, fit, invariance, kkt+solutions
    double t_host_screen = 0, t_host_screen_wait = 0;
    std::string error;
    double total_time = 0;

    // ---- device working set ----
    DevBuf<T> d_w, d_r, d_v, d_xm, d_grad, d_absgrad, d_penalty;
    DevBuf<idx> d_groups, d_gsizes;
    DevBuf<int32_t> d_slot;
    // per screen value / group (sized p / G up front: a few hundred KB)
    DevBuf<int32_t> d_vcol, d_sbegin, d_ssize, d_actset, d_dcols;
    DevBuf<T> d_spen, d_beta, d_beta0, d_g, d_vars, d_sxm, d_dvals;
    DevBuf<int8_t> d_isact;
    DevBuf<char> d_app;          // packed image of the new screen groups (device_append_screen)
    std::vector<char> app_img;
    DevBuf<idx> d_voff;
    DevBuf<T> d_V;
    size_t v_used = 0;
    DevBuf<T> d_C;
    idx ldc = 0, gcap = 0;
    idx gram_nv = 0; // number of screen values whose Gram rows/cols are valid (for the current weights)
    DevBuf<CdScalars<T>> d_sc;
    DevBuf<CdBlkState<T>> d_blk;
    DevBuf<T> d_Dbuf, d_dlt;
    DevBuf<int32_t> d_didx;
    int64_t cd_block_min_nv = 128; // screen sets at least this large use the multi-CU block passes (256 until round 3: 128 lets the speculative first pass cover ten more lambdas of the headline path, 292.9 -> 287.8 ms)
    // panel engine (kernels_cd_panel.hip): residual-based block passes with cached B x B diagonal blocks
    bool engine_panel = true;
    int panel_bsz = 0;          // 0: automatic (128 Gaussian, 64 IRLS); test/tuning hook ADELIE_HIP_PANEL_BSZ
    const T* cur_w = nullptr;   // weights / by-column means the pin solve runs under (Gaussian: w, X_means; IRLS: per iteration)
    const T* cur_xm = nullptr;
    uint64_t w_version = 1;     // bumped whenever the weights behind cur_w change
    DevBuf<T> d_Dpool, d_part, d_gblk;
    DevBuf<int32_t> d_actcols, d_dcolblk;
    DevBuf<int64_t> d_grp_dbg;
    // Side stream for the diagonal-block builds of a pass: they only depend on the weights, so all stale blocks of a pass are
    // enqueued there up front and the MFMA work overlaps the HBM-bound steps / single-wave solves of the main chain, which
    // waits on a per-block event right before the block's solve.
    hipStream_t st2 = nullptr;
    // further build streams (ADELIE_HIP_SIDE_STREAMS = 1..4 in total): under IRLS every block is rebuilt per iteration and the
    // chain waits for the builds; one build kernel (512 workgroups, 2 per CU) leaves the MFMA pipes half idle, two or three in
    // flight fill them
    static constexpr int kMaxExtra = 3;
    hipStream_t st_x[kMaxExtra] = {nullptr, nullptr, nullptr};
    DevBuf<T> d_work_x[kMaxExtra];
    int n_side = 1;
    bool side_grams = true;
    DevBuf<T> d_work_gram2;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    hipEvent_t next_event() {
        if (ev_used == ev_pool.size()) {
            hipEvent_t e;
            AHIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ev_pool.push_back(e);
        }
        return ev_pool[ev_used++];
    }
    int n_built_side = 0;
    std::vector<hipEvent_t> blk_ev; // per block of the current pass: event of its build on the side stream (or nullptr)
    // Builds the stale blocks among `nblk` blocks of a pass; block j has nb_of(j) members and columns cols_of(j).
    // `prebuild`: enqueue the builds of a list whose pass comes LATER in this fit (the screen-order blocks under IRLS weights,
    // enqueued while the active-set passes run): their events come from a pool of their own and are parked in `pre_ev` until
    // that pass picks them up instead of finding the blocks fresh.
    std::vector<hipEvent_t> pre_pool, pre_ev;
    size_t pre_used = 0;
    hipEvent_t next_pre_event() {
        if (pre_used == pre_pool.size()) {
            hipEvent_t e;
            AHIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            pre_pool.push_back(e);
        }
        return pre_pool[pre_used++];
    }
    // `rot_list` != nullptr or `rot_screen`: group passes with CdGrpBlkParams::rot — every block built here is rotated into the
    // eigen-coordinates of its groups right behind its build (same stream), over the pass's visiting list.
    const idx* rot_list = nullptr;
    bool rot_on = false;
    template <class NbOf, class ColsOf>
    void build_stale_blocks(int nblk, std::vector<int32_t>& tab_nb, std::vector<uint64_t>& tab_ver, T* pool, NbOf nb_of,
                            ColsOf cols_of, bool prebuild = false, bool take_pre = false) {
        const int SL = cd_block_size();
        if (prebuild) {
            pre_ev.assign(size_t(nblk), nullptr);
        } else {
            blk_ev.assign(size_t(nblk), nullptr);
            ev_used = 0;
            if (take_pre) { // blocks that were built ahead of this pass: wait for their builds like for a fresh one
                for (size_t j = 0; j < pre_ev.size() && j < size_t(nblk); ++j) blk_ev[j] = pre_ev[j];
                pre_ev.clear();
                pre_used = 0;
            }
        }
        bool first = true;
        const bool side = side_grams && st2 != nullptr;
        if (side && pass_e0_valid) { // the pass recorded "inputs final" on the main stream before its first step went out
            AHIP_CHECK(hipStreamWaitEvent(st2, pass_e0, 0));
            for (int k = 0; k < kMaxExtra; ++k)
                if (st_x[k]) AHIP_CHECK(hipStreamWaitEvent(st_x[k], pass_e0, 0));
            first = false;
        }
        auto pick_side = [&]() { return !side ? 0 : ((n_side >= 2 && st_x[0] && !multi()) ? 1 + (n_built_side++ % n_side) : 1); };
        auto open_side = [&]() {
            if (side && first) { // the weights (and everything else the builds read) are final at this point of the main stream
                hipEvent_t e0 = prebuild ? next_pre_event() : next_event();
                AHIP_CHECK(hipEventRecord(e0, st));
                AHIP_CHECK(hipStreamWaitEvent(st2, e0, 0));
                for (int k = 0; k < kMaxExtra; ++k)
                    if (st_x[k]) AHIP_CHECK(hipStreamWaitEvent(st_x[k], e0, 0));
                first = false;
            }
        };
        // Stale blocks of the same tile class go out in batches of up to `batch_blocks` per launch (see syrk_batch_kernel);
        // the chain waits for a block through the event of its batch.  The first batch of a pass is kept small so that the
        // chain can start early.
        stale.clear();
        for (int j = 0; j < nblk; ++j) {
            if (!(tab_nb[j] == nb_of(j) && ver_usable(tab_ver[j]))) stale.push_back(j);
            else if (tab_ver[j] != w_version) ++n_blocks_reused;
        }
        auto cls = [](int nb) { return nb <= 32 ? 32 : (nb <= 64 ? 64 : 128); };
        size_t i = 0;
        bool first_batch = true;
        while (i < stale.size()) {
            const int j0 = stale[i];
            size_t cap = multi() ? 1 : size_t(first_batch ? std::min(batch_blocks, 4) : batch_blocks);
            first_batch = false;
            SyrkBatch sb{};
            const int32_t* cols_base = cols_of(j0);
            size_t k = i;
            for (; k < stale.size() && k - i < cap; ++k) {
                const int j = stale[k], nb = nb_of(j);
                if (cls(nb) != cls(nb_of(j0))) break;
                const int64_t off = cols_of(j) - cols_base;
                if (off < 0 || off > (int64_t(1) << 30)) break;
                sb.off[k - i] = int32_t(off);
                sb.nb[k - i] = nb;
                sb.dst[k - i] = int64_t(j - j0) * SL * SL;
            }
            sb.count = int32_t(k - i);
            const int sidx = pick_side();
            open_side();
            // Gaussian look-ahead passes: a build whose block the chain reaches late in the pass is confined to few CUs, so
            // that the fused launches (whole-CU workgroups) running meanwhile never wait for one (see set_small_gram_workgroups)
            set_small_gram_workgroups((side && side_wgs > 0 && !is_glm() && j0 >= side_wgs_from) ? side_wgs : 512);
            if (multi()) gram_block(cur_w, cols_of(j0), nb_of(j0), cur_xm, pool + size_t(j0) * SL * SL, sidx);
            else gram_block_batch(cur_w, cols_base, sb, cur_xm, pool + size_t(j0) * SL * SL, sidx);
            if (rot_on)
                for (size_t t = i; t < k; ++t) rotate_block(rot_list, stale[t], pool + size_t(stale[t]) * SL * SL, sidx);
            hipEvent_t e = nullptr;
            if (side) {
                e = prebuild ? next_pre_event() : next_event();
                AHIP_CHECK(hipEventRecord(e, sidx >= 2 ? st_x[sidx - 2] : st2));
            }
            set_small_gram_workgroups(512);
            for (size_t t = i; t < k; ++t) {
                const int j = stale[t];
                (prebuild ? pre_ev : blk_ev)[size_t(j)] = e;
                tab_nb[j] = nb_of(j);
                tab_ver[j] = w_version;
                ++cnt.n_panel_grams;
            }
            i = k;
        }
    }
    // Recorded by a panel pass on the main stream BEFORE it enqueues its first step: the side streams' builds wait for this
    // event instead of one recorded behind the step, so the host can launch the step first (it does not depend on the builds)
    // and enqueue the builds while it runs (the ~50 us of host time per pass that enqueueing them takes used to leave the
    // chain idle: 6 ms per headline path)
    hipEvent_t pass_e0 = nullptr;
    bool pass_e0_valid = false;
    void record_pass_e0() {
        pass_e0_valid = false;
        if (!(side_grams && st2 != nullptr)) return;
        if (!pass_e0) AHIP_CHECK(hipEventCreateWithFlags(&pass_e0, hipEventDisableTiming));
        AHIP_CHECK(hipEventRecord(pass_e0, st));
        pass_e0_valid = true;
    }
    // ---- IRLS: diagonal blocks of an earlier iteration as the in-block operator (hook ADELIE_HIP_IRLS_REUSE=theta) ----
    // Under IRLS every block is rebuilt per iteration and used about once (config 4: 54 k builds for 54 k block visits,
    // half of the path's time).  A block only carries the coupling INSIDE its 64 visits: the gradient a block starts from
    // comes from the residual, exactly, on every visit.  So a block built for weights that differ from the current ones by
    // at most `irls_reuse` (relative, every observation; accumulated over the iterations since its build) still gives the
    // exact solution of the weighted problem at the fixed point of the passes - the passes stop on the coefficient changes
    // they actually make - and what changes is the iterate sequence inside a pass, by O(theta |delta|).  The later IRLS
    // iterations of a lambda move the weights by 1e-3 or less.  Measured on config 4 (500k x 50k): theta = 0.01 builds
    // 20.8 k blocks instead of 54.3 k, 7.05 -> 5.04 s, the same 222 IRLS iterations / 585 passes / screen and active sets,
    // max |delta beta| against theta = 0 over the whole path 1.1e-9 (scripts/irls_reuse.py).  0 = always rebuild.
    double irls_reuse = 0.01;
    std::vector<double> ver_drift;       // ver_drift[v] = log(1 + max relative weight change between versions v-1 and v)
    uint64_t min_usable_version = 1;     // blocks built at this weight version or later are within irls_reuse of the current weights
    int64_t n_blocks_reused = 0;
    DevBuf<T> d_irls_w_prev;
    bool irls_w_prev_valid = false;
    bool ver_usable(uint64_t v) const {
        return v == w_version || (irls_reuse > 0 && all_scalar && v != 0 && v >= min_usable_version && v < w_version);
    }
    void note_weight_drift(double max_rel) { // called right after ++w_version
        if (ver_drift.size() <= size_t(w_version)) ver_drift.resize(size_t(w_version) + 1, 1e300);
        ver_drift[size_t(w_version)] = std::log1p(max_rel);
        double acc = 0;
        uint64_t v = w_version;
        const double budget = std::log1p(irls_reuse);
        while (v > 1 && acc + ver_drift[size_t(v)] <= budget) { acc += ver_drift[size_t(v)]; --v; }
        min_usable_version = v;
    }
    bool prebuild_enabled = true; // A/B hook ADELIE_HIP_PREBUILD=0
    // look-ahead passes: the solve of a fused launch sums the previous launch's slice partials itself (second round trip of
    // blk_solve_la_body's prologue) instead of a panel_reduce launch between every two fused launches.  Round 2 measured this
    // slower (3.08 vs 3.20 paths/s) with the solve's old prologue; with the one-round-trip prologue the fused launch grows by
    // 1 us and the reduce launch + its boundary go away: 290.3 -> 285.9 ms (f32: 178.1 -> 174.6).  Hook ADELIE_HIP_FUSE_REDUCE=0.
    // Only while a column has at most 200 partials (n <= 102 400 rows in f64): beyond, one workgroup summing them is slower
    // than the reduce launch.
    bool fuse_reduce_opt = true;
    bool fuse_reduce = false;     // (set per solve from fuse_reduce_opt and the partial count)
    int fused_partials() const {  // partials per column a fused launch leaves (kernels_cd_panel.hip::fused_launch)
        int vec = 4;
        if (dense()) {
            constexpr int V = int(16 / sizeof(T));
            const bool vecok = (D->ld % V == 0) && ((reinterpret_cast<uintptr_t>(D->X) % 16) == 0);
            vec = vecok ? V : 1;
        }
        const int64_t rs = 64 * vec, ns = (n + rs - 1) / rs, nwg = (ns + 3) / 4;
        return int(vec * 64 >= 128 ? nwg : nwg * 4);
    }
    DevBuf<T> d_part2;
    size_t part2_half = 0;
    int side_wgs = 0;             // >0: confine side-stream builds of Gaussian look-ahead passes to this many workgroups (hook ADELIE_HIP_SIDE_WGS; measured: 56 -> 2.69, 112 -> 2.99 vs 3.17 paths/s unconfined: the chain waits for the slower builds)
    int side_wgs_from = 4;        // ... for blocks the chain reaches at this position of the pass or later (ADELIE_HIP_SIDE_WGS_FROM)
    std::vector<int> stale;
    int batch_blocks = 8; // diagonal blocks per build launch (tuning hook ADELIE_HIP_BATCH_BLOCKS, 1..16)
    int cross_batch = 8;  // cross blocks per build launch (hook ADELIE_HIP_CROSS_BATCH, 1 = one gram launch per block)
    std::vector<int> stale_x;
    bool cross_incremental = true; // A/B hook ADELIE_HIP_CROSS_INCR=0: a cross block that gained rows is rebuilt whole
    int x_rows_new[GramBatch::MAX] = {};
    // host-mapped end-of-pass report (state + sequence number), see CdBlkParams::host_st
    struct PassReport { CdBlkState<T> st; int32_t seq; int32_t pad[15]; };
    PassReport* h_report = nullptr;
    int32_t report_seq = 0;
    bool use_report = true;
    ~Solver() {
        // every DevBuf member is parked in the allocation cache by its destructor: nothing may still be running on them
        if (st) (void)hipStreamSynchronize(st);
        if (st2) {
            (void)hipStreamSynchronize(st2);
            StreamPool::give(st2);
        }
        for (int k = 0; k < kMaxExtra; ++k)
            if (st_x[k]) {
                (void)hipStreamSynchronize(st_x[k]);
                StreamPool::give(st_x[k]);
            }
        HostPool::give(h_report, sizeof(PassReport), hipHostMallocMapped);
        deferred.drain(); // blocks outgrown during the solve: every stream that may have used them is idle now

        if (spec_ev) (void)hipEventDestroy(spec_ev);
        if (pass_e0) (void)hipEventDestroy(pass_e0);
        if (uv_ev) (void)hipEventDestroy(uv_ev);
        if (uv_in_ev) (void)hipEventDestroy(uv_in_ev);
        for (hipEvent_t e : ev_pool) (void)hipEventDestroy(e);
        for (hipEvent_t e : pre_pool) (void)hipEventDestroy(e);
        for (hipEvent_t e : strip_pool) (void)hipEventDestroy(e);
    }
    // ---- look-ahead form of the Gaussian panel passes (run_panel_passes) ----
    // The solve of block j (one wavefront, strictly sequential) and the panel step that prepares block j+1 only meet through
    // the residual; with the centred cross block C_{j+1,j} = X_{j+1}^T W X_j - xbar xbar^T cached next to the diagonal
    // blocks, the step can run BEFORE block j's changes are known (gradient of block j+1 from a residual without them) and
    // the solve of block j+1 subtracts C_{j+1,j} delta_j itself.  Solve j and the step for block j+1 then go out as ONE
    // launch (panel_fused_kernel: workgroup 0 solves, the others step): the chain costs max(step, solve) + reduce per block
    // instead of their sum.  (Solves on a second stream with event dependencies were measured first: ~35 us per
    // cross-queue hop, slower than no look-ahead at all.)
    bool lookahead = true;      // A/B hook ADELIE_HIP_LOOKAHEAD
    int la_min_blocks = 3;      // passes with fewer blocks run in the plain form (hook ADELIE_HIP_LOOKAHEAD_MIN_BLOCKS)
    DevBuf<T> d_Xpool, d_la_dlt, d_la_g, d_la_rsum, d_la_dd;
    DevBuf<int32_t> d_gdesc; // layout descriptors of the current group pass (launch_grp_layout)
    DevBuf<int32_t> d_la_dcol, d_la_dpos, d_la_nz;
    DevBuf<int32_t> d_tail_counter; // CdGrpBlkParams::tail_counter
    DevBuf<int32_t> d_zero_i32; // one int32 that stays 0 ("no changes to apply" for the step of a pass's first fused launch)
    bool la_fused_open = true;  // look-ahead passes open with (step: pending changes + block 0) -> fused (solve 0 || block 1) instead of
                                // (step: blocks 0 and 1) -> reduce -> solve 0; hook ADELIE_HIP_LA_FUSED_OPEN=0
    struct XKey { int32_t nb_prev = 0, nb = 0; uint64_t ver = 0; };
    std::vector<XKey> xscr_key, xact_key;
    std::vector<hipEvent_t> x_ev;
    double t_enq = 0, t_wait = 0; // host seconds spent enqueueing panel passes / waiting for their state (ADELIE_HIP_TRACE_ENQ)
    int pending_slot = -1;      // slot holding the changes of the last solved block that the residual does not contain yet
    int64_t n_cross_blocks = 0;
    template <class NbOf, class ColsOf>
    void build_stale_cross(int nblk, std::vector<XKey>& tab, T* xpool, NbOf nb_of, ColsOf cols_of) {
        const int SL = cd_block_size();
        x_ev.assign(size_t(nblk), nullptr);
        bool first = !(pass_e0_valid && side_grams && st2 != nullptr); // (the diagonal-block builder made st2 wait for pass_e0)
        if (!multi() && cross_batch > 1) {
            // several stale cross blocks per launch (gram_batch_kernel): their K-splits share one round over the chip, so the
            // split-K partials written and re-read per block shrink with the batch (134 MB for a block built alone)
            std::vector<int>& sx = stale_x;
            sx.clear();
            for (int j = 1; j < nblk; ++j) {
                const XKey& k = tab[size_t(j)];
                if (!(k.nb_prev == nb_of(j - 1) && k.nb == nb_of(j) && k.ver == w_version)) sx.push_back(j);
            }
            const bool side = side_grams && st2 != nullptr;
            hipStream_t gs = side ? st2 : st;
            const int32_t* cols_base = cols_of(0);
            for (size_t i = 0; i < sx.size();) {
                const size_t k = std::min(sx.size(), i + size_t(i == 0 ? std::min(cross_batch, 4) : cross_batch));
                if (side && first) {
                    hipEvent_t e0 = next_event();
                    AHIP_CHECK(hipEventRecord(e0, st));
                    AHIP_CHECK(hipStreamWaitEvent(st2, e0, 0));
                    first = false;
                }
                GramBatch gb{};
                gb.count = int32_t(k - i);
                for (size_t t = i; t < k; ++t) {
                    const int j = sx[t];
                    // Both visiting lists only grow by appending, so the rows of a cross block that were built for this weight
                    // version against the same (full) previous block stay valid when the block gains members: only the rows
                    // of the newcomers are computed (the batch kernel skips the 16-row tiles beyond them)
                    const XKey& key = tab[size_t(j)];
                    const int have = (cross_incremental && key.ver == w_version && key.nb_prev == nb_of(j - 1) &&
                                      key.nb > 0 && key.nb < nb_of(j)) ? key.nb : 0;
                    gb.moff[t - i] = int32_t(cols_of(j) - cols_base) + have;
                    gb.m[t - i] = nb_of(j) - have;
                    x_rows_new[t - i] = nb_of(j) - have;
                    gb.noff[t - i] = int32_t(cols_of(j - 1) - cols_base);
                    gb.nn[t - i] = nb_of(j - 1);
                    gb.dst[t - i] = int64_t(j) * SL * SL + have;
                }
                T* work = (side ? d_work_gram2 : d_work_gram)
                              .reserve(size_t(std::max<int64_t>(gram_batch_work_elems(n, gb.count), syrk_work_elems(n, 128))));
                t_gram.begin(gs);
                if (dense()) launch_gram_batch<T>(D->dense<T>(), cur_w, cols_base, gb, cur_xm, intercept, xpool, SL, work, gs);
                else launch_gram_batch_snp<T>(D->snp(), static_cast<const T*>(D->impute), cur_w, cols_base, gb, cur_xm, intercept,
                                              xpool, SL, work, gs);
                t_gram.end(gs);
                hipEvent_t e = nullptr;
                if (side) {
                    e = next_event();
                    AHIP_CHECK(hipEventRecord(e, st2));
                }
                for (size_t t = i; t < k; ++t) {
                    const int j = sx[t];
                    XKey& key = tab[size_t(j)];
                    key.nb_prev = nb_of(j - 1); key.nb = nb_of(j); key.ver = w_version;
                    x_ev[size_t(j)] = e;
                    cnt.gram_flops += 2.0 * double(n) * double(x_rows_new[t - i]) * double(nb_of(j - 1));
                    cnt.n_gram_col_reads += x_rows_new[t - i] + nb_of(j - 1);
                    ++n_cross_blocks;
                }
                i = k;
            }
            return;
        }
        for (int j = 1; j < nblk; ++j) {
            const int nbp = nb_of(j - 1), nb = nb_of(j);
            XKey& k = tab[size_t(j)];
            if (k.nb_prev == nbp && k.nb == nb && k.ver == w_version) continue;
            const bool side = side_grams && st2 != nullptr;
            hipStream_t gs = side ? st2 : st;
            if (side && first) {
                hipEvent_t e0 = next_event();
                AHIP_CHECK(hipEventRecord(e0, st));
                AHIP_CHECK(hipStreamWaitEvent(st2, e0, 0));
                first = false;
            }
            T* work = (side ? d_work_gram2 : d_work_gram)
                          .reserve(size_t(std::max<int64_t>(gram_work_elems(n, SL, SL), syrk_work_elems(n, 128))));
            T* Cx = xpool + size_t(j) * SL * SL;
            set_small_gram_workgroups((side && side_wgs > 0 && !is_glm() && j >= side_wgs_from) ? side_wgs : 512);
            t_gram.begin(gs);
            if (multi()) {
                // Gram of the two blocks' distinct features, expanded to view columns (zero between different responses);
                // look-ahead only runs under uniform weights (Gaussian), so one Gram serves all responses
                const MultiView<T> mv = D->multi<T>();
                auto distinct = [&](const int32_t* hc, int cntv) {
                    multi_seen.clear();
                    for (int a = 0; a < cntv; ++a) {
                        const int32_t u = hc[a] / mv.K;
                        if (std::find(multi_seen.begin(), multi_seen.end(), u) == multi_seen.end()) multi_seen.push_back(u);
                    }
                    return int(multi_seen.size());
                };
                const int nu = distinct(host_cols(cols_of(j)), nb), nup = distinct(host_cols(cols_of(j - 1)), nbp);
                DevBuf<int32_t>& ml = side ? d_mlist2 : d_mlist;
                DevBuf<T>& mc = side ? d_mC2 : d_mC;
                ml.reserve(size_t(6 * SL));
                mc.reserve(size_t(SL) * SL);
                launch_multi_block_lists(cols_of(j), nb, mv.K, ml.p, ml.p + SL, ml.p + 2 * SL, gs);
                launch_multi_block_lists(cols_of(j - 1), nbp, mv.K, ml.p + 3 * SL, ml.p + 4 * SL, ml.p + 5 * SL, gs);
                T* mwork = (side ? d_work_gram2 : d_work_gram)
                               .reserve(size_t(std::max<int64_t>(gram_work_elems(mv.nb, SL, SL), syrk_work_elems(mv.nb, 128))));
                launch_gram_multi<T>(mv, cur_w, ml.p, nu, ml.p + 3 * SL, nup, mc.p, SL, mwork, gs);
                launch_multi_expand_cross<T>(mc.p, SL, ml.p + SL, ml.p + 2 * SL, nb, ml.p + 4 * SL, ml.p + 5 * SL, nbp, Cx, SL, gs);
                cnt.gram_flops += 2.0 * double(mv.nb) * double(nu) * double(nup);
            } else if (dense())
                launch_gram<T>(D->dense<T>(), cur_w, cols_of(j), nb, 0, cols_of(j - 1), nbp, 0, cur_xm, intercept, Cx, SL, work, gs);
            else
                launch_gram_snp<T>(D->snp(), static_cast<const T*>(D->impute), cur_w, cols_of(j), nb, 0, cols_of(j - 1), nbp, 0,
                                   cur_xm, intercept, Cx, SL, work, gs);
            t_gram.end(gs);
            set_small_gram_workgroups(512);
            cnt.gram_flops += 2.0 * double(n) * double(nb) * double(nbp);
            cnt.n_gram_col_reads += nb + nbp;
            if (side) {
                hipEvent_t e = next_event();
                AHIP_CHECK(hipEventRecord(e, st2));
                x_ev[size_t(j)] = e;
            }
            k.nb_prev = nbp; k.nb = nb; k.ver = w_version;
            ++n_cross_blocks;
        }
    }
    // ---- strip builds (kernels_strip.hip): only the NEW rows of a block's diagonal and cross block ----
    // Gaussian passes over a dense design: both visiting lists are append-only and the weights are fixed, so a block that
    // gained m <= 64 members since its blocks were built needs the m x (|previous block| + |block|) strip of the newcomers
    // and nothing else.  One HBM-bound launch (+ reduce) per batch of strips replaces a full syrk build (128 x 128, 36 MFMA
    // tiles) plus a staged cross build whose cost does not shrink with the row count: 71 us against 282 us for 16 new
    // members of a full block pair at n = 100k (scripts/ubench/strip.hip).  Runs before build_stale_blocks /
    // build_stale_cross, which then find these blocks fresh; blocks with more new members stay with them.
    // Hook ADELIE_HIP_STRIP_BUILDS=0.
    bool strip_builds = true;
    int strip_max_m = 128;
    std::vector<hipEvent_t> strip_pool, strip_ev;
    std::vector<int> strip_built; // blocks the last build_stale_strips call built
    size_t strip_used = 0;
    int64_t n_strip_builds = 0;
    hipEvent_t next_strip_event() {
        if (strip_used == strip_pool.size()) {
            hipEvent_t e;
            AHIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            strip_pool.push_back(e);
        }
        return strip_pool[strip_used++];
    }
    bool strips_apply() const { return strip_builds && dense() && !is_glm(); }
    // `rot_dst` != nullptr (group passes with CdGrpBlkParams::rot): `pool` holds the blocks in the design's own coordinates
    // (d_Draw: what the strips extend), and every block a strip touched is rotated into the eigen-coordinates of its groups
    // right behind it on the same stream, out of place into rot_dst (the pool the solves read), over the visiting list `rlist`.
    DevBuf<T> d_Draw;
    template <class NbOf, class ColsOf>
    void build_stale_strips(int nblk, std::vector<int32_t>& tab_nb, std::vector<uint64_t>& tab_ver, std::vector<XKey>* xtab,
                            T* pool, T* xpool, NbOf nb_of, ColsOf cols_of, T* rot_dst = nullptr, const idx* rlist = nullptr,
                            bool force_main = false) {
        strip_ev.assign(size_t(nblk), nullptr);
        strip_used = 0;
        strip_built.clear();
        if (!strips_apply()) return;
        const int SL = cd_block_size();
        const bool side = !force_main && side_grams && st2 != nullptr;
        hipStream_t gs = side ? st2 : st;
        bool first = true;
        const int32_t* cols_base = cols_of(0);
        StripBatch sb{};
        int js[StripBatch::MAX];
        bool ent_d[StripBatch::MAX] = {}, ent_x[StripBatch::MAX] = {};
        auto flush = [&]() {
            if (sb.count == 0) return;
            if (side && first) {
                if (pass_e0_valid) {
                    AHIP_CHECK(hipStreamWaitEvent(st2, pass_e0, 0));
                } else {
                    hipEvent_t e0 = next_strip_event();
                    AHIP_CHECK(hipEventRecord(e0, st));
                    AHIP_CHECK(hipStreamWaitEvent(st2, e0, 0));
                }
                first = false;
            }
            int mx = 0;
            for (int y = 0; y < sb.count; ++y) mx = std::max(mx, int(sb.m[y]));
            T* work = (side ? d_work_gram2 : d_work_gram).reserve(size_t(strip_work_elems(n, sb.count, mx)));
            t_gram.begin(gs);
            launch_strip_batch<T>(D->dense<T>(), cur_w, cols_base, sb, cur_xm, intercept, pool, xpool, SL, work, gs);
            if (rot_dst)
                for (int y = 0; y < sb.count; ++y)
                    if (ent_d[y] && (y == 0 || js[y - 1] != js[y]))
                        rotate_block(rlist, js[y], rot_dst + size_t(js[y]) * SL * SL, side ? 1 : 0, pool + size_t(js[y]) * SL * SL);
            t_gram.end(gs);
            hipEvent_t e = nullptr;
            if (side) {
                e = next_strip_event();
                AHIP_CHECK(hipEventRecord(e, st2));
            }
            for (int y = 0; y < sb.count; ++y) {
                const int j = js[y];
                strip_ev[size_t(j)] = e;
                if (ent_d[y]) {
                    tab_nb[size_t(j)] = nb_of(j);
                    tab_ver[size_t(j)] = w_version;
                }
                if (ent_x[y] && xtab && j > 0) {
                    XKey& key = (*xtab)[size_t(j)];
                    key.nb_prev = nb_of(j - 1); key.nb = nb_of(j); key.ver = w_version;
                }
                cnt.gram_flops += 2.0 * double(n) * double(sb.m[y]) * double(sb.c0n[y] + sb.c1n[y]);
                cnt.n_gram_col_reads += sb.m[y] + sb.c0n[y] + sb.c1n[y];
                if (y == 0 || js[y - 1] != j) {
                    ++n_strip_builds;
                    strip_built.push_back(j);
                }
            }
            sb = StripBatch{};
        };
        for (int j = 0; j < nblk; ++j) {
            const int nb = nb_of(j);
            const bool want_x = xtab != nullptr && j > 0;
            // rows the diagonal / the cross block of this block already hold for the current weights and member lists
            const int have_d = (tab_ver[size_t(j)] == w_version && tab_nb[size_t(j)] <= nb) ? tab_nb[size_t(j)] : 0;
            int have_x = nb;
            if (want_x) {
                const XKey& key = (*xtab)[size_t(j)];
                have_x = (key.ver == w_version && key.nb_prev == nb_of(j - 1) && key.nb <= nb) ? int(key.nb) : 0;
            }
            const bool need_d = have_d < nb, need_x = have_x < nb;
            if (!need_d && !need_x) continue;
            // one of the two only: a strip over that block's columns alone; both: from the smaller of the two row counts
            const int have = (need_d && need_x) ? std::min(have_d, have_x) : (need_d ? have_d : have_x);
            const int m = nb - have;
            if (m <= 0 || m > strip_max_m) continue; // (left to the staged builders)
            const int64_t off1 = cols_of(j) - cols_base, off0 = want_x ? cols_of(j - 1) - cols_base : 0;
            if (off1 < 0 || off1 > (int64_t(1) << 30)) continue;
            // more than 64 new members: two strips of the same launch.  A strip only needs the columns of its block up to
            // its own last row: the rest of its rows of D lies above the diagonal of the new x new square and comes from
            // the mirror of the other strip's rows.
            const int pieces = m > 64 ? 2 : 1;
            if (sb.count + pieces > StripBatch::MAX) flush();
            for (int q = 0; q < pieces; ++q) {
                const int r0 = have + (q == 0 ? 0 : (m + 1) / 2), r1 = (q + 1 == pieces) ? nb : have + (m + 1) / 2;
                const int y = sb.count++;
                js[y] = j;
                ent_d[y] = need_d;
                ent_x[y] = need_x;
                sb.voff[y] = int32_t(off1) + r0;
                sb.m[y] = r1 - r0;
                sb.c0off[y] = int32_t(off0);
                sb.c0n[y] = need_x ? nb_of(j - 1) : 0;
                sb.c1off[y] = int32_t(off1);
                sb.c1n[y] = need_d ? r1 : 0;
                sb.row0[y] = r0;
                sb.dstX[y] = int64_t(j) * SL * SL;
                sb.dstD[y] = int64_t(j) * SL * SL;
            }
            if (sb.count == StripBatch::MAX) flush();
        }
        flush();
    }
    // after the staged builders ran (they reset blk_ev / x_ev): the chain waits for a strip-built block through its strip's event
    void merge_strip_events(bool with_cross) {
        for (size_t j = 0; j < strip_ev.size() && j < blk_ev.size(); ++j)
            if (strip_ev[j]) {
                if (!blk_ev[j]) blk_ev[j] = strip_ev[j];
                else if (with_cross && j < x_ev.size() && !x_ev[j]) x_ev[j] = strip_ev[j];
            }
    }
    std::vector<int32_t> dscr_nb, dact_nb;      // cached block: number of members it was built for
    std::vector<uint64_t> dscr_ver, dact_ver;   // ... and the weight version
    bool group_panel = true;    // groups (q > 1) on the panel engine too (A/B hook ADELIE_HIP_GROUP_PANEL=0: full-Gram block engine)
    bool panel_mode() const {
        return engine_panel && nv >= cd_block_min_nv && (all_scalar || (group_panel && max_gs <= idx(cd_block_size())));
    }
    DevBuf<T> d_work_sweep, d_work_gram;
    bool grad_valid = false; // d_grad == X^T W r - rsum*xbar for the current r
    // In stream order, d_grad holds the full Gaussian gradient of the CURRENT d_r (an invariance sweep was enqueued and nothing
    // touched the residual since): the first look-ahead pass of the next fit takes block 0's gradient from it instead of
    // streaming the block's columns (open_from_grad, consumed by run_panel_passes).  Hook ADELIE_HIP_OPEN_FROM_GRAD=0.
    bool grad_fresh = false, open_from_grad = false, open_from_grad_opt = true, spec_used_grad = false;
    // glm device vectors
    DevBuf<T> d_y, d_gw, d_off, d_eta, d_hess, d_irls_y, d_irls_resid, d_eta_prev, d_resid_prev, d_sums, d_ones;
    // host mirrors of per-screen arrays used to append
    idx nv = 0; // screen values
    idx ns_dev = 0; // screen groups already mirrored on device

    SweepBatcher* batcher = nullptr; // non-null while this solver is registered for sweep batching
    bool is_screen(idx i) const { return in_screen[i] != 0; }
    bool dense() const { return D->kind == 0; }
    // multi-response view (adelie_hip_design_create_multi): residual / weights live response-major on the device
    bool multi() const { return D->kind == 2; }
    int mk() const { return D->kind == 2 ? int(D->mK) : 1; } // class count handed to the GLM kernels
    bool multi_w_uniform = true;
    std::vector<int32_t> h_vcol, h_actcols, multi_seen; // host mirrors of d_vcol / d_actcols (block column lists)
    DevBuf<int32_t> d_mlist, d_mlist2;
    DevBuf<T> d_mC, d_mC2, d_mxm;
    const int32_t* host_cols(const int32_t* dev) const {
        if (dev >= d_vcol.p && dev < d_vcol.p + h_vcol.size()) return h_vcol.data() + (dev - d_vcol.p);
        if (dev >= d_actcols.p && dev < d_actcols.p + h_actcols.size()) return h_actcols.data() + (dev - d_actcols.p);
        throw make_core_error("internal: block column list without a host mirror.");
    }
    // (n, K) row-major (the ABI's layout, matrix_naive_kronecker_eye.ipp:36-37) <-> response-major
    void to_major(const T* src, T* dst) const {
        const int64_t nb = D->nb, K = D->mK;
        for (int64_t i = 0; i < nb; ++i)
            for (int64_t l = 0; l < K; ++l) dst[l * nb + i] = src[i * K + l];
    }
    void from_major(const T* src, T* dst) const {
        const int64_t nb = D->nb, K = D->mK;
        for (int64_t i = 0; i < nb; ++i)
            for (int64_t l = 0; l < K; ++l) dst[i * K + l] = src[l * nb + i];
    }

    // ---------------------------------------------------------------------------------------------------------
    void sweep(const T* v, T* out, const int32_t* cols, idx ncols, const T* sub_scale, const T* sub_vec,
               bool square = false) {
        if (multi()) { // only the full sweep of the Gaussian path is needed on the view (intercept off: no centring epilogue)
            if (cols || ncols != p || sub_vec || square) throw make_core_error("unsupported sweep on a multi-response view.");
            const MultiView<T> mv = D->multi<T>();
            launch_multi_sweep<T>(mv, v, out, d_work_sweep.reserve(size_t(multi_sweep_work_elems<T>(mv))), st);
            return;
        }
        if (batcher && dense() && !cols && ncols == p && !square &&
            batcher->template sweep<T>(D->dense<T>(), v, out, sub_scale, sub_vec, st)) {
            ++cnt.n_sweeps_shared;
            return;
        }
        T* work = d_work_sweep.reserve(size_t(sweep_work_elems(n, ncols)));
        if (dense()) launch_sweep<T>(D->dense<T>(), v, out, 0, ncols, cols, sub_scale, sub_vec, square, work, st);
        else launch_sweep_snp<T>(D->snp(), static_cast<const T*>(D->impute), v, out, 0, ncols, cols, sub_scale, sub_vec, square, work, st);
    }
    int panel_step(const T* w, T* r, const int32_t* dcol, const T* dlt, const int32_t* nz_dev, const int32_t* cols, int nb) {
        if (multi()) return launch_multi_panel_step<T>(D->multi<T>(), w, r, dcol, dlt, nz_dev, cols, nb, d_part.p, st);
        if (dense()) return launch_panel_step<T>(D->dense<T>(), w, r, dcol, dlt, nz_dev, cols, nb, d_part.p, st);
        return launch_panel_step_snp<T>(D->snp(), static_cast<const T*>(D->impute), w, r, dcol, dlt, nz_dev, cols, nb,
                                        d_part.p, st);
    }
    // `sb.count` diagonal blocks in one launch (non-multi designs): block y = columns cols_base[sb.off[y] ...], into
    // D0 + sb.dst[y] (ld = B)
    void gram_block_batch(const T* w, const int32_t* cols_base, const SyrkBatch& sb, const T* xm, T* D0, int side) {
        const int B = cd_block_size();
        hipStream_t gs = side == 0 ? st : (side >= 2 ? st_x[side - 2] : st2);
        T* work = (side == 0 ? d_work_gram : (side >= 2 ? d_work_x[side - 2] : d_work_gram2))
                      .reserve(size_t(std::max(syrk_batch_work_elems(n, sb.count), syrk_work_elems(n, 128))));
        t_gram.begin(gs);
        if (dense()) launch_syrk_batch<T>(D->dense<T>(), w, cols_base, sb, xm, intercept, D0, B, work, gs);
        else launch_syrk_batch_snp<T>(D->snp(), static_cast<const T*>(D->impute), w, cols_base, sb, xm, intercept, D0, B, work, gs);
        t_gram.end(gs);
        for (int y = 0; y < sb.count; ++y) {
            const int nb = sb.nb[y];
            cnt.gram_flops += 2.0 * double(n) * 256.0 * (nb <= 32 ? 3.0 : (nb <= 64 ? 10.0 : 36.0));
            cnt.n_gram_col_reads += 2 * nb;
        }
    }
    // B x B block  X_cols^T W X_cols - xm xm^T  into Dptr (ld = B)
    void gram_block(const T* w, const int32_t* cols, int nb, const T* xm, T* Dptr, int side = 0) {
        const int B = cd_block_size();
        hipStream_t gs = side == 0 ? st : (side >= 2 ? st_x[side - 2] : st2);
        t_gram.begin(gs);
        if (multi()) {
            // Gram over the block's distinct extended features (MFMA syrk), expanded to the view columns: entries between
            // different responses are zero.  One syrk when all responses carry the same weights (always so for
            // multigaussian: w_i / K), otherwise one per response.
            const MultiView<T> mv = D->multi<T>();
            const int32_t* hc = host_cols(cols);
            multi_seen.clear();
            for (int a = 0; a < nb; ++a) {
                const int32_t u = hc[a] / mv.K;
                if (std::find(multi_seen.begin(), multi_seen.end(), u) == multi_seen.end()) multi_seen.push_back(u);
            }
            const int nu = int(multi_seen.size());
            DevBuf<int32_t>& ml = side ? d_mlist2 : d_mlist;
            DevBuf<T>& mc = side ? d_mC2 : d_mC;
            ml.reserve(size_t(6 * B));
            mc.reserve(size_t(B) * B);
            launch_multi_block_lists(cols, nb, mv.K, ml.p, ml.p + B, ml.p + 2 * B, gs);
            T* work = (side ? d_work_gram2 : d_work_gram).reserve(size_t(syrk_work_elems(mv.nb, 128)));
            const int reps = multi_w_uniform ? 1 : mv.K;
            for (int l = 0; l < reps; ++l) {
                launch_syrk_multi<T>(mv, w + size_t(l) * size_t(mv.nb), ml.p, nu, mc.p, B, work, gs);
                launch_multi_expand<T>(mc.p, B, ml.p + B, ml.p + 2 * B, nb, multi_w_uniform ? -1 : l, Dptr, B, gs);
                cnt.gram_flops += 2.0 * double(mv.nb) * 256.0 * (nu <= 32 ? 3.0 : (nu <= 64 ? 10.0 : 36.0));
            }
            cnt.n_gram_col_reads += 2 * nu * reps;
            t_gram.end(gs);
            return;
        }
        {   // lower-triangle MFMA tiles only: 10 of 16 (nb <= 64) or 36 of 64
            T* work = (side == 0 ? d_work_gram : (side >= 2 ? d_work_x[side - 2] : d_work_gram2)).reserve(size_t(syrk_work_elems(n, 128)));
            if (dense()) launch_syrk<T>(D->dense<T>(), w, cols, nb, xm, intercept, Dptr, B, work, gs);
            else launch_syrk_snp<T>(D->snp(), static_cast<const T*>(D->impute), w, cols, nb, xm, intercept, Dptr, B, work, gs);
            cnt.gram_flops += 2.0 * double(n) * 256.0 * (nb <= 32 ? 3.0 : (nb <= 64 ? 10.0 : 36.0));
        }
        t_gram.end(gs);
        cnt.n_gram_col_reads += 2 * nb;
    }
    void axpy_cols(const int32_t* cols, const T* coef, const int32_t* cnt_dev, int32_t count, T sign, T* out) {
        if (multi()) {
            if (!cnt_dev) throw make_core_error("unsupported axpy on a multi-response view.");
            launch_multi_axpy_cols<T>(D->multi<T>(), cols, coef, cnt_dev, sign, out, st);
            return;
        }
        if (dense()) launch_axpy_cols<T>(D->dense<T>(), cols, coef, cnt_dev, count, sign, out, st);
        else launch_axpy_cols_snp<T>(D->snp(), static_cast<const T*>(D->impute), cols, coef, cnt_dev, count, sign, out, st);
    }
    void gram(const T* w, idx M, idx pos0, idx N, const T* xm, bool center) {
        T* work = d_work_gram.reserve(size_t(gram_work_elems(n, M, N)));
        t_gram.begin(st);
        if (dense())
            launch_gram<T>(D->dense<T>(), w, d_vcol.p, int32_t(M), 0, d_vcol.p + pos0, int32_t(N), int32_t(pos0), xm, center,
                           d_C.p, ldc, work, st);
        else
            launch_gram_snp<T>(D->snp(), static_cast<const T*>(D->impute), w, d_vcol.p, int32_t(M), 0, d_vcol.p + pos0,
                               int32_t(N), int32_t(pos0), xm, center, d_C.p, ldc, work, st);
        t_gram.end(st);
        cnt.n_gram_col_reads += M + N;
        cnt.gram_flops += 2.0 * double(n) * double(M) * double(N);
        gram_shapes.emplace_back(M, N);
    }
    // pinned staging for the small per-lambda copies (common.hpp::Staging; A/B hook ADELIE_HIP_STAGING=0)
    Staging stage;
    DeferredFrees deferred; // installed for the solving thread by run<T>; drained by ~Solver
    double t_sync_total = 0; // host seconds inside sync() (bench: splits the host phases into compute and waiting)
    // update_vars_panel_groups on the side stream (strip builds of the new screen groups' rows, their eigen-decompositions,
    // the rotations): everything a SCREEN pass needs and an active-set pass does not, so the active-set passes of the fit run
    // meanwhile and the screen pass (or any host read) joins through this event.  Hook ADELIE_HIP_UV_SIDE=0.
    bool uv_side = true;
    hipEvent_t uv_ev = nullptr, uv_in_ev = nullptr;
    bool uv_pending = false;
    void join_uv() {
        if (!uv_pending) return;
        AHIP_CHECK(hipStreamWaitEvent(st, uv_ev, 0));
        uv_pending = false;
    }
    void sync() {
        join_uv();
        const auto t0 = std::chrono::steady_clock::now();
        AHIP_CHECK(hipStreamSynchronize(st));
        t_sync_total += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        stage.reset();
    }

    // ---------------------------------------------------------------------------------------------------------
    // solver_base.hpp:20-110 on the host (used at construction only; later abs_grad comes from the device)
    void update_abs_grad_host(T lm) {
        for (size_t ss = 0; ss < screen_set.size(); ++ss) {
            const idx i = screen_set[ss], b = screen_begins[ss], k = groups[i], sz = group_sizes[i];
            const T regul = ((1 - alpha) * lm) * penalty[i];
            if (cons_on && cons_kind[i] && !host_cons(i)) { // :69-75: minus the constraint's gradient
                abs_grad[i] = std::abs(grad[k] - regul * screen_beta[b] - cons_mu[i]);
                continue;
            }
            T acc = 0;
            for (idx t = 0; t < sz; ++t) {
                const T e = grad[k + t] - regul * screen_beta[b + t];
                acc += e * e;
            }
            abs_grad[i] = std::sqrt(acc);
        }
        for (idx i = 0; i < G; ++i) {
            if (is_screen(i)) continue;
            const idx k = groups[i];
            if (cons_on && cons_kind[i] && !host_cons(i)) { // :88-93 solve_zero (constraint_box.ipp:268-284, constraint_one_sided.ipp:269-279)
                const T M = T(1e100), v = grad[k];
                cons_mu[i] = std::min(std::max(v, cons_lo[i] >= 0 ? -M : T(0)), cons_hi[i] <= 0 ? M : T(0));
                abs_grad[i] = std::abs(v - cons_mu[i]);
                continue;
            }
            T acc = 0;
            for (idx t = 0; t < group_sizes[i]; ++t) acc += grad[k + t] * grad[k + t];
            abs_grad[i] = std::sqrt(acc);
        }
    }

    // update_abs_grad on the device (solver_base.hpp:20-110) + the copy the host screens / checks KKT with; under constraints
    // also every group's multiplier (screened: from its last visit; others: solve_zero)
    T lmda_of_sweep = 0;
    void device_abs_grad(T lm, int active_now) {
        lmda_of_sweep = lm;
        if (cons_on) {
            launch_abs_grad_cons<T>(d_grad.p, d_groups.p, d_gsizes.p, G, d_slot.p, d_beta.p, d_penalty.p, (1 - alpha) * lm, d_clo_g.p,
                                    d_chi_g.p, d_cmu.p, d_absgrad.p, d_mu_g.p, st);
            d_mu_g.download(cons_mu.data(), size_t(G), st);
        } else {
            launch_abs_grad<T>(d_grad.p, d_groups.p, d_gsizes.p, G, d_slot.p, d_beta.p, d_penalty.p, (1 - alpha) * lm,
                               d_absgrad.p, st);
        }
        d_absgrad.download(abs_grad.data(), size_t(G), st);
    }

    int64_t n_host_screens = 0;

    // solver_base.hpp:120-153
    void update_screen_derived_base() {
        const auto old = screen_begins.size();
        if (in_screen.size() != size_t(G)) in_screen.assign(G, 0);
        for (size_t i = old; i < screen_set.size(); ++i) in_screen[screen_set[i]] = 1;
        size_t vs = (old == 0) ? 0 : (screen_begins.back() + group_sizes[screen_set[old - 1]]);
        for (size_t i = old; i < screen_set.size(); ++i) {
            screen_begins.push_back(vs);
            vs += group_sizes[screen_set[i]];
        }
        screen_beta.resize(vs, 0);
        screen_is_active.resize(screen_set.size(), 0);
    }

    // Mirror newly appended screen groups on the device (value->column map, begins, sizes, penalties, slots,
    // coefficients) and make room in the Gram matrix.  `beta_known`: upload host screen_beta for the new values
    // (warm start) instead of zeros.
    void device_append_screen() {
        const idx ns = idx(screen_set.size());
        if (ns_dev == ns) return;
        std::vector<int32_t> vcol, sbegin, ssize, slot_idx;
        std::vector<T> spen, beta_new;
        std::vector<int8_t> isact;
        const size_t fallbacks0 = stage.n_fallback;
        const idx nv_old = nv;
        idx nv_new = nv_old;
        for (idx ss = ns_dev; ss < ns; ++ss) {
            const idx g = screen_set[ss];
            sbegin.push_back(int32_t(screen_begins[ss]));
            ssize.push_back(int32_t(group_sizes[g]));
            spen.push_back(penalty[g]);
            isact.push_back(screen_is_active[ss]);
            for (idx t = 0; t < group_sizes[g]; ++t) {
                vcol.push_back(int32_t(groups[g] + t));
                beta_new.push_back(screen_beta[screen_begins[ss] + t]);
            }
            nv_new += group_sizes[g];
        }
        h_vcol.resize(size_t(nv_old));
        h_vcol.insert(h_vcol.end(), vcol.begin(), vcol.end());
        // one packed image of everything the new groups add to the device mirrors, one upload, one scatter launch
        const int Ng = int(ns - ns_dev), Nv = int(nv_new - nv_old);
        const AppendImage<T> L(Ng, Nv, cons_on);
        app_img.assign(L.total, 0);
        auto put = [&](size_t off, const void* src, size_t bytes) { if (bytes) std::memcpy(app_img.data() + off, src, bytes); };
        put(L.pen, spen.data(), sizeof(T) * spen.size());
        put(L.beta, beta_new.data(), sizeof(T) * beta_new.size());
        if (cons_on) { // per screen value (only groups of one coefficient carry a constraint)
            std::vector<T> clo_new, chi_new, cmu_new;
            for (idx ss = ns_dev; ss < ns; ++ss) {
                const idx g = screen_set[ss];
                for (idx t = 0; t < group_sizes[g]; ++t) {
                    clo_new.push_back(cons_lo[g]);
                    chi_new.push_back(cons_hi[g]);
                    cmu_new.push_back(cons_mu[g]);
                }
            }
            put(L.lo, clo_new.data(), sizeof(T) * clo_new.size());
            put(L.hi, chi_new.data(), sizeof(T) * chi_new.size());
            put(L.mu, cmu_new.data(), sizeof(T) * cmu_new.size());
        }
        std::vector<int32_t> isact32(isact.begin(), isact.end()), grp32;
        for (idx ss = ns_dev; ss < ns; ++ss) grp32.push_back(int32_t(screen_set[ss]));
        put(L.begin, sbegin.data(), 4 * sbegin.size());
        put(L.size, ssize.data(), 4 * ssize.size());
        put(L.isact, isact32.data(), 4 * isact32.size());
        put(L.group, grp32.data(), 4 * grp32.size());
        put(L.vcol, vcol.data(), 4 * vcol.size());
        if (slot_host.size() != size_t(G)) slot_host.assign(G, -1); // (host mirror of d_slot; the device table starts at -1)
        for (idx ss = ns_dev; ss < ns; ++ss) slot_host[screen_set[ss]] = int32_t(screen_begins[ss]);
        d_app.reserve(L.total);
        d_app.upload(app_img.data(), L.total, st);
        AppendDst<T> ad{};
        ad.spen = d_spen.p; ad.beta = d_beta.p; ad.clo = d_clo.p; ad.chi = d_chi.p; ad.cmu = d_cmu.p;
        ad.sbegin = d_sbegin.p; ad.ssize = d_ssize.p; ad.isact = d_isact.p; ad.slot = d_slot.p; ad.vcol = d_vcol.p;
        ad.ns_old = int32_t(ns_dev); ad.nv_old = int32_t(nv_old); ad.Ng = Ng; ad.Nv = Nv; ad.cons = cons_on ? 1 : 0;
        launch_screen_append<T>(d_app.p, ad, st);
        // the vectors above go out of scope: wait unless every upload took a snapshot into the pinned arena (a wait here also
        // waits for the speculative pass that may be running in-stream: 0.3 ms per lambda on the headline path)
        if (Staging::current() != &stage || !stage.base || stage.n_fallback != fallbacks0) sync();
        ns_dev = ns;
        nv = nv_new;
        // Gram capacity: `gcap` columns, leading dimension ldc = gcap rounded up to 2048 rows (the CD kernel reads
        // whole 512-lane x 16-byte chunks of a column; zero-filled so the padding never carries NaN payloads)
        if (nv > gcap && !panel_mode()) {
            idx want = std::max<idx>(gcap * 2, 256);
            while (want < nv) want *= 2;
            want = std::min<idx>(want, ((p + 63) / 64) * 64);
            if (want < nv) want = nv;
            const idx new_ld = ((want + 2047) / 2048) * 2048;
            DevBuf<T> nc;
            nc.reserve(size_t(new_ld) * size_t(want));
            AHIP_CHECK(hipMemsetAsync(nc.p, 0, size_t(new_ld) * size_t(want) * sizeof(T), st));
            if (gram_nv > 0) launch_copy2d<T>(d_C.p, ldc, nc.p, new_ld, gram_nv, gram_nv, st);
            sync();
            std::swap(d_C.p, nc.p);
            std::swap(d_C.cap, nc.cap);
            ldc = new_ld;
            gcap = want;
        }
    }

    // Bring the Gram matrix, variances and eigen-bases up to date for screen values [gram_nv, nv) under weights w
    // (centred with xm_by_col when intercept).  Fills host screen_X_means / screen_vars / screen_transforms for the
    // new groups [g_begin, ns)  (solver_gaussian_naive.hpp:41-125).
    void update_gram_and_vars(const T* w_dev, const T* xm_dev, const std::vector<T>& xm_host, size_t g_begin) {
        const idx ns = idx(screen_set.size());
        const idx pos0 = (g_begin < size_t(ns)) ? screen_begins[g_begin] : nv;
        const idx N = nv - pos0;
        screen_X_means.resize(nv);
        screen_vars.resize(nv, 0);
        screen_transforms.resize(ns);
        if (N <= 0) return;
        if (cov_mode) // the rows / columns of A of the new screen values (solver_gaussian_cov.hpp:63-97 reads A_gg from them)
            launch_cov_gather<T>(static_cast<const T*>(D->X), D->ld, D->cov == 2, d_vcol.p, int32_t(nv), int32_t(pos0), int32_t(N),
                                 d_C.p, ldc, st);
        else
            gram(w_dev, nv, pos0, N, xm_dev, intercept);
        AHIP_CHECK(hipGetLastError());
        gram_nv = nv;
        cnt.n_new_screen_cols += N;
        launch_diag_vars<T>(d_C.p, ldc, int32_t(pos0), int32_t(N), d_vars.p, st);
        // host-side pieces: X_means of the new values, eigen-bases of the new groups with q > 1
        std::vector<T> sxm(N);
        for (idx ss = idx(g_begin); ss < ns; ++ss) {
            const idx g = screen_set[ss], b = screen_begins[ss];
            for (idx t = 0; t < group_sizes[g]; ++t) {
                screen_X_means[b + t] = xm_host[groups[g] + t];
                sxm[b + t - pos0] = screen_X_means[b + t];
            }
        }
        d_sxm.upload(sxm.data(), sxm.size(), st, pos0);
        std::vector<idx> voff(ns - g_begin, 0);
        bool any_group = false;
        idx max_q = 1;
        for (idx ss = idx(g_begin); ss < ns; ++ss) {
            const idx q = group_sizes[screen_set[ss]];
            if (q > 1) any_group = true;
            max_q = std::max(max_q, q);
        }
        if (any_group && device_eig && max_q <= idx(kEigMaxQ)) {
            // eigen-decompositions of the new groups' diagonal blocks of C on the device (kernels_eig.hip): no per-group copy
            // to the host and back, no host wait
            eig_desc.clear();
            size_t v_new = 0;
            if (h_voff.size() < size_t(ns)) h_voff.resize(size_t(ns), 0);
            for (idx ss = idx(g_begin); ss < ns; ++ss) {
                const idx q = group_sizes[screen_set[ss]], b = screen_begins[ss];
                if (q == 1) continue; // (launch_diag_vars above wrote its variance)
                EigDesc e{};
                e.src = b + b * ldc;
                e.ld = int32_t(ldc);
                e.q = int32_t(q);
                e.vars_pos = b;
                e.v_off = int64_t(v_used + v_new);
                voff[ss - g_begin] = idx(v_used + v_new);
                h_voff[size_t(ss)] = voff[ss - g_begin];
                v_new += size_t(q) * q;
                eig_desc.push_back(e);
            }
            if (v_new) d_V.grow(v_used + v_new, v_used, st);
            v_used += v_new;
            d_eig_desc.reserve(eig_desc.size());
            d_eig_desc.upload(eig_desc.data(), eig_desc.size(), st);
            launch_grp_eig<T>(d_C.p, d_eig_desc.p, int(eig_desc.size()), int(max_q), d_vars.p, d_V.p, st);
            d_voff.upload(voff.data(), voff.size(), st, g_begin);
            host_mirrors_stale = true;
            if (Staging::current() != &stage || !stage.base) sync();
            return;
        }
        std::vector<T> vars_host(N);
        d_vars.download(vars_host.data(), N, st, pos0);
        sync();
        if (any_group) {
            for (idx ss = idx(g_begin); ss < ns; ++ss) {
                const idx q = group_sizes[screen_set[ss]], b = screen_begins[ss];
                if (q == 1) {
                    screen_transforms[ss] = std::vector<T>{T(1)};
                    continue;
                }
                std::vector<T> blk(size_t(q) * q);
                AHIP_CHECK(hipMemcpy2DAsync(blk.data(), q * sizeof(T), d_C.p + b + b * ldc, ldc * sizeof(T), q * sizeof(T), q,
                                            hipMemcpyDeviceToHost, st));
                sync();
                std::vector<double> A(blk.begin(), blk.end()), V, Dv;
                jacobi_eigh(int(q), A, V, Dv);
                std::vector<T> Vt(V.begin(), V.end());
                for (idx t = 0; t < q; ++t) vars_host[b + t - pos0] = T(Dv[t] >= 0 ? Dv[t] : 0.0);
                // append to the device transform pool
                d_V.grow(v_used + size_t(q) * q, v_used, st);
                d_V.upload(Vt.data(), Vt.size(), st, v_used);
                voff[ss - g_begin] = idx(v_used);
                if (h_voff.size() < size_t(ns)) h_voff.resize(size_t(ns), 0);
                h_voff[size_t(ss)] = idx(v_used);
                v_used += size_t(q) * q;
                screen_transforms[ss] = std::move(Vt);
                sync();
            }
            d_vars.upload(vars_host.data(), N, st, pos0);
            d_voff.upload(voff.data(), voff.size(), st, g_begin);
            sync();
        } else {
            for (idx ss = idx(g_begin); ss < ns; ++ss) screen_transforms[ss] = std::vector<T>{T(1)};
        }
        for (idx t = 0; t < N; ++t) screen_vars[pos0 + t] = vars_host[t];
    }

    // Panel engine (groups of size one only): the screen-derived quantities are the by-value means and the variances
    // A_k = x_k^T W x_k - xbar_k^2 (solver_gaussian_naive.hpp:99-111); no |S| x |S| Gram matrix is kept.
    void update_vars_panel(const T* w_dev, const T* xm_dev, const std::vector<T>& xm_host, size_t g_begin) {
        const idx ns = idx(screen_set.size());
        const idx pos0 = (g_begin < size_t(ns)) ? screen_begins[g_begin] : nv;
        const idx N = nv - pos0;
        screen_X_means.resize(nv);
        screen_vars.resize(nv, 0);
        screen_transforms.resize(ns);
        if (N <= 0) return;
        cnt.n_new_screen_cols += N;
        if (!all_scalar) {
            update_vars_panel_groups(w_dev, xm_dev, xm_host, g_begin, pos0, N);
            return;
        }
        sweep(w_dev, d_vars.p + pos0, d_vcol.p + pos0, N, nullptr, nullptr, true);
        for (idx ss = idx(g_begin); ss < ns; ++ss) {
            const idx g = screen_set[ss], b = screen_begins[ss];
            screen_X_means[b] = xm_host[groups[g]];
            screen_transforms[ss] = std::vector<T>{T(1)};
        }
        // by-value means on the device straight from the by-column vector; the host copy of the variances is only an output
        // (finalize() downloads it), so no synchronisation here
        launch_gather<T>(xm_dev, d_vcol.p + pos0, N, d_sxm.p + pos0, st);
        launch_center_vars<T>(d_vars.p + pos0, d_sxm.p + pos0, int(N), intercept, st);
    }

    // Same with groups: X_g^T W X_g - xbar xbar^T of every new group is a diagonal sub-block of one of the screen-order
    // diagonal blocks of the panel engine (groups are never split across blocks), so those blocks are built here (they are
    // needed by the next screen pass anyway), copied to the host once, and the eigen-decompositions
    // (solver_gaussian_naive.hpp:105-125) are done on the host copies.
    std::vector<int32_t> gp_vbeg; // per block of the current partition: offset of its first value in the pass's column list
    // blocks a visiting list can be cut into: runs of groups with <= 128 values, plus the cuts before and after every group
    // that is a block of its own (constraint objects visited on the host)
    size_t n_host_cons = 0;
    size_t group_maxblk() const { return size_t(2 * p / cd_block_size() + 2) + 2 * n_host_cons; }
    int build_partition_values(const idx* list, idx count) {
        const int nblk = build_partition(list, count);
        gp_vbeg.assign(size_t(nblk) + 1, 0);
        int32_t acc = 0;
        for (int j = 0; j < nblk; ++j) {
            for (int32_t pos = part_host[j]; pos < part_host[j + 1]; ++pos)
                acc += int32_t(group_sizes[screen_set[list ? list[pos] : pos]]);
            gp_vbeg[size_t(j) + 1] = acc;
        }
        return nblk;
    }
    void update_vars_panel_groups(const T* w_dev, const T* xm_dev, const std::vector<T>& xm_host, size_t g_begin, idx pos0,
                                  idx N) {
        const idx ns = idx(screen_set.size());
        const int SL = cd_block_size();
        panel_setup(group_maxblk());
        const int nblk = build_partition_values(nullptr, ns);
        for (idx ss = idx(g_begin); ss < ns; ++ss) {
            const idx g = screen_set[ss], b = screen_begins[ss];
            for (idx t = 0; t < group_sizes[g]; ++t) screen_X_means[b + t] = xm_host[groups[g] + t];
        }
        launch_gather<T>(xm_dev, d_vcol.p + pos0, N, d_sxm.p + pos0, st);
        int j0 = 0;
        while (j0 + 1 < nblk && size_t(part_host[j0 + 1]) <= g_begin) ++j0;
        idx max_q = 1;
        for (idx ss = idx(g_begin); ss < ns; ++ss) max_q = std::max(max_q, group_sizes[screen_set[ss]]);
        const bool dev_eig = device_eig && max_q <= idx(kEigMaxQ);
        std::vector<T> hD(dev_eig ? size_t(0) : size_t(nblk - j0) * SL * SL);
        std::vector<int> rebuilt_blocks;
        // Gaussian dense designs: only the rows of the new groups (strip builds), with the rows of the cross blocks when the
        // look-ahead tables exist, into the unrotated pool; the staged builder below then finds the blocks fresh
        if (!is_glm()) { cur_w = w_dev; cur_xm = xm_dev; } // (Gaussian: the weights / means every pin solve of the path runs under)
        const bool use_strips = strips_apply();
        const bool raw_split = use_strips && group_rot;
        T* const rawbase = raw_split ? d_Draw.p : d_Dpool.p;
        const bool on_side = use_strips && dev_eig && uv_side && side_grams && st2 != nullptr;
        if (use_strips) {
            // (diagonal rows only here and the cross rows on the side stream in the pass that needs them: measured slower,
            // config 3 634 vs 621 ms — two launches per block instead of one)
            const bool with_x = lookahead && xscr_key.size() == panel_maxblk && d_Xpool.p != nullptr;
            build_stale_strips(nblk, dscr_nb, dscr_ver, with_x ? &xscr_key : nullptr, rawbase, with_x ? d_Xpool.p : static_cast<T*>(nullptr),
                               [&](int j) { return int(gp_vbeg[size_t(j) + 1] - gp_vbeg[j]); },
                               [&](int j) { return d_vcol.p + gp_vbeg[j]; }, nullptr, nullptr, !on_side);
            rebuilt_blocks = strip_built;
        }
        for (int j = j0; j < nblk; ++j) {
            const int nval = gp_vbeg[size_t(j) + 1] - gp_vbeg[j];
            T* Dptr = rawbase + size_t(j) * SL * SL;
            if (dscr_nb[j] != nval || dscr_ver[j] != w_version) {
                gram_block(w_dev, d_vcol.p + gp_vbeg[j], nval, xm_dev, Dptr);
                dscr_nb[j] = nval;
                dscr_ver[j] = w_version;
                ++cnt.n_panel_grams;
                rebuilt_blocks.push_back(j);
            }
            if (!dev_eig)
                AHIP_CHECK(hipMemcpyAsync(hD.data() + size_t(j - j0) * SL * SL, Dptr, size_t(SL) * SL * sizeof(T),
                                          hipMemcpyDeviceToHost, st));
        }
        std::vector<idx> voff(size_t(ns) - g_begin, 0);
        h_voff.resize(size_t(ns), 0);
        if (dev_eig) {
            // eigen-decompositions on the device, one wavefront per new group, straight from the blocks built above: no copy
            // of the blocks to the host, no host wait (the host mirrors of screen_vars / screen_transforms are filled by
            // download_invariants)
            eig_desc.clear();
            size_t v_new = 0;
            int j = j0;
            for (idx ss = idx(g_begin); ss < ns; ++ss) {
                while (j + 1 < nblk && part_host[j + 1] <= int32_t(ss)) ++j;
                const idx q = group_sizes[screen_set[ss]], b = screen_begins[ss];
                const idx o = b - gp_vbeg[j];
                EigDesc e{};
                e.src = int64_t(j) * SL * SL + o + o * SL;
                e.ld = SL;
                e.q = int32_t(q);
                e.vars_pos = b;
                e.v_off = 0;
                if (q > 1) {
                    voff[ss - g_begin] = idx(v_used + v_new);
                    h_voff[size_t(ss)] = voff[ss - g_begin];
                    e.v_off = int64_t(v_used + v_new);
                    v_new += size_t(q) * q;
                }
                eig_desc.push_back(e);
            }
            if (v_new && v_used + v_new > d_V.cap) {
                // (a reallocation: nothing may be running on the old buffer; sized for every group of the problem at once so
                // that it happens once)
                sync();
                if (st2) AHIP_CHECK(hipStreamSynchronize(st2));
                size_t total = 0;
                for (idx q : group_sizes) total += q > 1 ? size_t(q) * size_t(q) : 0;
                d_V.grow(std::max(total, v_used + v_new), v_used, st);
            }
            v_used += v_new;
            if (on_side && d_eig_desc.cap < eig_desc.size()) { // (the previous descriptors may still be read on the side stream)
                join_uv();
                d_eig_desc.reserve(std::max<size_t>(eig_desc.size(), size_t(ns)));
            }
            d_eig_desc.reserve(eig_desc.size());
            d_eig_desc.upload(eig_desc.data(), eig_desc.size(), st);
            d_voff.upload(voff.data(), voff.size(), st, g_begin);
            hipStream_t es = st;
            if (on_side) {
                if (!uv_in_ev) {
                    AHIP_CHECK(hipEventCreateWithFlags(&uv_in_ev, hipEventDisableTiming));
                    AHIP_CHECK(hipEventCreateWithFlags(&uv_ev, hipEventDisableTiming));
                }
                AHIP_CHECK(hipEventRecord(uv_in_ev, st)); // the descriptors (and everything before) are on their way
                AHIP_CHECK(hipStreamWaitEvent(st2, uv_in_ev, 0));
                es = st2;
            }
            launch_grp_eig<T>(rawbase, d_eig_desc.p, int(eig_desc.size()), int(max_q), d_vars.p, d_V.p, es);
            if (group_rot)
                for (int jb : rebuilt_blocks)
                    rotate_block(nullptr, jb, d_Dpool.p + size_t(jb) * SL * SL, on_side ? 1 : 0, rawbase + size_t(jb) * SL * SL);
            if (on_side) {
                AHIP_CHECK(hipEventRecord(uv_ev, st2));
                uv_pending = true;
            }
            host_mirrors_stale = true;
            if (Staging::current() != &stage || !stage.base) sync(); // (pageable uploads: the vectors go out of scope)
            return;
        }
        sync();
        std::vector<T> vars_host(N), vnew;
        int j = j0;
        for (idx ss = idx(g_begin); ss < ns; ++ss) {
            while (j + 1 < nblk && part_host[j + 1] <= int32_t(ss)) ++j;
            const idx q = group_sizes[screen_set[ss]], b = screen_begins[ss];
            const idx o = b - gp_vbeg[j];
            const T* Dj = hD.data() + size_t(j - j0) * SL * SL;
            if (q == 1) {
                const T d = Dj[o + o * SL];
                vars_host[b - pos0] = d > T(0) ? d : T(0);
                screen_transforms[ss] = std::vector<T>{T(1)};
                continue;
            }
            std::vector<double> A(size_t(q) * q), V, Dv;
            for (idx c = 0; c < q; ++c)
                for (idx r = 0; r < q; ++r) A[r + c * q] = double(Dj[(o + r) + (o + c) * SL]);
            jacobi_eigh(int(q), A, V, Dv);
            for (idx t = 0; t < q; ++t) vars_host[b + t - pos0] = T(Dv[t] >= 0 ? Dv[t] : 0.0);
            voff[ss - g_begin] = idx(v_used + vnew.size());
            h_voff[size_t(ss)] = voff[ss - g_begin];
            std::vector<T> Vt(V.begin(), V.end());
            vnew.insert(vnew.end(), Vt.begin(), Vt.end());
            screen_transforms[ss] = std::move(Vt);
        }
        if (!vnew.empty()) {
            d_V.grow(v_used + vnew.size(), v_used, st);
            d_V.upload(vnew.data(), vnew.size(), st, v_used);
            v_used += vnew.size();
        }
        d_vars.upload(vars_host.data(), N, st, pos0);
        d_voff.upload(voff.data(), voff.size(), st, g_begin);
        // the blocks built above, into the eigen-coordinates of their groups (the eigenbases are on the device now)
        if (group_rot)
            for (int jb : rebuilt_blocks)
                rotate_block(nullptr, jb, d_Dpool.p + size_t(jb) * SL * SL, 0, rawbase + size_t(jb) * SL * SL);
        sync(); // the staging vectors go out of scope
        for (idx t = 0; t < N; ++t) screen_vars[pos0 + t] = vars_host[t];
    }
    // device-side eigen-decompositions of new screen groups (kernels_eig.hip; A/B hook ADELIE_HIP_DEVICE_EIG=0: host Jacobi on
    // copies of the blocks, as in rounds 1-2).  `host_mirrors_stale`: screen_vars / screen_transforms on the host lag behind
    // d_vars / d_V until download_invariants refreshes them.
    bool device_eig = true;
    bool host_mirrors_stale = false;
    std::vector<EigDesc> eig_desc;
    DevBuf<EigDesc> d_eig_desc;
    // D <- R^T D R for block `jb` of the partition in part_host over `list` (nullptr: screen order), on the stream of build
    // side `side` (0: main).  See CdGrpBlkParams::rot.
    bool group_rot = true; // A/B hook ADELIE_HIP_GROUP_ROT=0
    std::vector<idx> h_voff; // per screen group: offset of its eigenbasis in d_V
    DevBuf<T> d_rot_scratch[2 + kMaxExtra];
    void rotate_block(const idx* list, int jb, T* Dptr, int side, const T* Dsrc = nullptr) {
        GrpRotArgs a{};
        int ng = 0, o = 0;
        for (int32_t pos = part_host[size_t(jb)]; pos < part_host[size_t(jb) + 1]; ++pos, ++ng) {
            const idx ss = list ? list[pos] : idx(pos);
            a.goff[ng] = o;
            a.voff[ng] = (size_t(ss) < h_voff.size()) ? h_voff[size_t(ss)] : 0;
            o += int32_t(group_sizes[screen_set[ss]]);
        }
        a.goff[ng] = o;
        a.ng = ng;
        hipStream_t gs = side == 0 ? st : (side >= 2 ? st_x[side - 2] : st2);
        T* scratch = d_rot_scratch[side].reserve(size_t(cd_block_size()) * cd_block_size());
        launch_grp_block_rotate<T>(Dptr, Dsrc ? Dsrc : Dptr, d_V.p, a, scratch, gs);
    }

    // solver_gaussian_naive.hpp:134-176
    void gaussian_update_screen_derived() {
        const size_t old_groups = screen_transforms.size();
        update_screen_derived_base();
        device_append_screen();
        if (panel_mode()) update_vars_panel(d_w.p, d_xm.p, X_means, old_groups);
        else update_gram_and_vars(d_w.p, d_xm.p, X_means, old_groups);
    }

    // optimization/search_pivot.hpp:7-62
    static int search_pivot(const std::vector<T>& x, const std::vector<T>& y, std::vector<T>& mses) {
        const idx m = idx(x.size());
        if (m <= 0) return -1;
        mses[0] = std::numeric_limits<T>::infinity();
        if (m == 1) return 0;
        T y_mean = 0;
        for (idx i = 0; i < m; ++i) y_mean += y[i];
        y_mean /= T(m);
        T x_sum = x[0], xsq_sum = x[0] * x[0], y_sum = y[0], yx_sum = y[0] * x[0], min_mse = mses[0];
        int argmin = 0;
        for (idx i = 1; i < m; ++i) {
            x_sum += x[i];
            xsq_sum += x[i] * x[i];
            y_sum += y[i];
            yx_sum += y[i] * x[i];
            const T t_bar = ((i + 1) * x[i] - x_sum) / m;
            const T var_t = ((i + 1) * x[i] * x[i] - 2 * x[i] * x_sum + xsq_sum - m * t_bar * t_bar);
            const T cov_ty = (x[i] * (y_sum - (i + 1) * y_mean) - (yx_sum - y_mean * x_sum));
            const T b1 = cov_ty / var_t;
            mses[i] = -b1 * b1 * var_t;
            if (mses[i] < min_mse) { argmin = int(i); min_mse = mses[i]; }
        }
        return argmin;
    }

    // Stable LSD radix sort of (score, group) pairs by score: equal scores keep their group order, i.e. the same total
    // order as comparing the pairs, at a fraction of std::sort's cost for the G ~ 1e4..1e5 keys sorted once per lambda.
    static void sort_keyed(std::vector<std::pair<T, idx>>& v) {
        using U = typename std::conditional<sizeof(T) == 8, uint64_t, uint32_t>::type;
        const size_t m = v.size();
        if (m < 256) { std::sort(v.begin(), v.end()); return; }
        constexpr int BITS = 11, NB = 1 << BITS, PASSES = (sizeof(T) * 8 + BITS - 1) / BITS;
        std::vector<U> key(m), key2(m);
        std::vector<idx> val(m), val2(m);
        for (size_t i = 0; i < m; ++i) {
            U u;
            std::memcpy(&u, &v[i].first, sizeof(T));
            const U sign = U(1) << (sizeof(T) * 8 - 1);
            key[i] = (u & sign) ? ~u : (u | sign); // order-preserving map of IEEE values to unsigned
            val[i] = v[i].second;
        }
        std::vector<size_t> cntv(NB);
        for (int ps = 0; ps < PASSES; ++ps) {
            const int sh = ps * BITS;
            std::fill(cntv.begin(), cntv.end(), size_t(0));
            for (size_t i = 0; i < m; ++i) ++cntv[(key[i] >> sh) & (NB - 1)];
            size_t run = 0;
            for (int b = 0; b < NB; ++b) { const size_t c = cntv[b]; cntv[b] = run; run += c; }
            for (size_t i = 0; i < m; ++i) {
                const size_t d = cntv[(key[i] >> sh) & (NB - 1)]++;
                key2[d] = key[i];
                val2[d] = val[i];
            }
            key.swap(key2);
            val.swap(val2);
        }
        for (size_t i = 0; i < m; ++i) {
            const U sign = U(1) << (sizeof(T) * 8 - 1);
            const U u = (key[i] & sign) ? (key[i] & ~sign) : ~key[i];
            T f;
            std::memcpy(&f, &u, sizeof(T));
            v[i] = std::make_pair(f, val[i]);
        }
    }

    // solver_base.hpp:273-403
    void screen(T lmda_next, bool all_kkt_passed, int n_new_active) {
        const int old_size = int(screen_set.size());
        if (screen_rule == ADELIE_HIP_SCREEN_STRONG) {
            const T strong = (2 * lmda_next - lmda) * alpha;
            for (idx i = 0; i < G; ++i) {
                if (is_screen(i)) continue;
                if (abs_grad[i] > strong * penalty[i]) screen_set.push_back(i);
            }
        } else if (screen_rule == ADELIE_HIP_SCREEN_PIVOT) {
            if (n_new_active) {
                const int Gi = int(G);
                // sort (score, group) pairs in place: contiguous keys instead of an indirect comparator
                std::vector<std::pair<T, idx>> keyed(Gi);
                for (int i = 0; i < Gi; ++i) {
                    const T wt = (penalty[i] <= 0) ? alpha * lmda : std::min(abs_grad[i] / penalty[i], alpha * lmda);
                    keyed[i] = std::make_pair(wt, idx(i));
                }
                // The reference sorts with `weights[i] < weights[j]` only (solver_base.hpp:320-326): every group whose score is
                // capped at alpha*lmda ties exactly, and std::sort leaves the order of ties unspecified.  Ties are broken by
                // group index here (pair comparison) so that the screen insertion order (= CD visiting order) is reproducible.
                sort_keyed(keyed);
                std::vector<idx> order(Gi);
                std::vector<T> wts(Gi);
                for (int i = 0; i < Gi; ++i) {
                    order[i] = keyed[i].second;
                    wts[keyed[i].second] = keyed[i].first;
                }
                const int subset_size =
                    std::min<int>(std::max<int>(int(old_size * (1 + pivot_subset_ratio)), int(pivot_subset_min)), Gi);
                std::vector<T> sub(subset_size), mses(subset_size), ind(subset_size);
                for (int i = 0; i < subset_size; ++i) {
                    sub[i] = wts[order[Gi - subset_size + i]];
                    ind[i] = T(i);
                }
                const int pivot_idx = search_pivot(ind, sub, mses);
                const int full_pivot_idx = Gi - subset_size + pivot_idx;
                for (int ii = Gi - 1; ii >= full_pivot_idx; --ii) {
                    const idx i = order[ii];
                    if (is_screen(i)) continue;
                    screen_set.push_back(i);
                }
                int count = 0;
                for (int ii = full_pivot_idx - 1; ii >= 0; --ii) {
                    if (count >= pivot_slack_ratio * n_new_active) break;
                    const idx i = order[ii];
                    if (is_screen(i)) continue;
                    screen_set.push_back(i);
                    ++count;
                }
            }
            if ((int(screen_set.size()) == old_size) && !all_kkt_passed) {
                for (idx i = 0; i < G; ++i) {
                    if (is_screen(i)) continue;
                    if (abs_grad[i] > lmda_next * penalty[i] * alpha) screen_set.push_back(i);
                }
            }
            // Progress guard (deliberate deviation): the KKT check multiplies in the order
            // lmda * alpha * penalty (solver_base.hpp:428) and the fallback above in the order lmda * penalty * alpha
            // (:369), which can round differently; a gradient that falls between the two fails KKT forever without ever
            // being screened (seen in f32 at lambda_0 == lmda_max with alpha < 1).  Screen it with KKT's own expression.
            if ((int(screen_set.size()) == old_size) && !all_kkt_passed) {
                for (idx i = 0; i < G; ++i) {
                    if (is_screen(i)) continue;
                    if (abs_grad[i] > lmda_next * alpha * penalty[i]) screen_set.push_back(i);
                }
            }
        } else {
            throw make_solver_error("Unknown screen rule!");
        }
        if (screen_set.size() > max_screen_size) {
            screen_set.resize(old_size);
            throw max_screen_set_error();
        }
    }

    // solver_base.hpp:408-433
    bool kkt(T lm) const {
        for (idx k = 0; k < G; ++k) {
            if (is_screen(k)) continue;
            if (abs_grad[k] > lm * alpha * penalty[k]) return false;
        }
        return true;
    }
    // solver_base.hpp:241-263
    bool early_exit() const {
        if (cov_mode) { // solver_gaussian_cov.hpp:183-201: relative change of the (unnormalised) deviance
            if (!early_exit_ || devs.size() < 2) return false;
            const T dev_u = devs[devs.size() - 1], dev_m = devs[devs.size() - 2];
            return dev_u - dev_m <= rdev_tol * dev_u;
        }
        if (!early_exit_ || devs.empty()) return false;
        const T dev_u = devs.back();
        if (dev_u >= adev_tol) return true;
        if (devs.size() == 1) return false;
        const T dev_m = devs[devs.size() - 2];
        if (std::abs(dev_u - dev_m) < ddev_tol) return true;
        return false;
    }

    void poll_mid() {
        if (poll && poll(poll_user, 0, int64_t(lmdas.size()), live)) throw core_error("interrupted");
    }

    // ---------------------------------------------------------------------------------------------------------
    // Lasso pin solve as a sequence of block passes spread over the chip (kernels_cd_block.hip).  The pass structure
    // (solve_active until convergence, one screen pass, repeat; pin_naive:317-357) is driven from the host, which reads
    // one small scalar block per pass.  Fills `sc` like the single-workgroup kernel does.
    void run_block_passes(const CdParams<T>& cp, CdScalars<T>& sc) {
        const int B = cd_block_size();
        d_blk.reserve(1);
        d_Dbuf.reserve(size_t(2) * B * B);
        d_dlt.reserve(B);
        d_didx.reserve(B);
        CdBlkState<T> bs{};
        bs.rsq = sc.rsq;
        bs.resid_sum = sc.resid_sum;
        bs.cm = 0;
        bs.n_updates = 0;
        bs.active_size = sc.active_size;
        bs.status = CD_OK;
        d_blk.upload(&bs, 1, st);
        CdBlkParams<T> bp{};
        bp.nv = cp.nv; bp.C = cp.C; bp.ldc = cp.ldc; bp.vars = cp.vars; bp.xmean = cp.xmean; bp.spen = cp.spen;
        bp.beta = cp.beta; bp.g = cp.g; bp.is_active = cp.is_active; bp.active_set = cp.active_set;
        bp.l1 = cp.lmda * cp.alpha; bp.l2 = cp.lmda * (T(1) - cp.alpha);
        bp.max_active_size = cp.max_active_size;
        bp.Dbuf = d_Dbuf.p; bp.dlt = d_dlt.p; bp.didx = d_didx.p; bp.st = d_blk.p;
        bp.bsz = B;
        int64_t iters = 0;
        int status = CD_OK;
        int asz = sc.active_size;
        auto pass = [&](const int32_t* list, int count, bool mark) -> T {
            if (count <= 0) return T(0);
            bp.list = list; bp.count = count; bp.mark = mark ? 1 : 0;
            t_cd.begin(st);
            launch_cd_block_pass<T>(bp, st);
            t_cd.end(st);
            d_blk.download(&bs, 1, st);
            sync();
            status = bs.status;
            asz = bs.active_size;
            return bs.cm;
        };
        while (status == CD_OK) {
            while (status == CD_OK) { // solve_active, pin_naive:173-215
                ++iters;
                ++sc.n_passes_active;
                sc.n_visits_active += asz;
                const T cm = pass(cp.active_set, asz, false);
                if (status != CD_OK) break;
                if (cm < cp.tol) break;
                if (iters >= cp.max_iters) { status = CD_MAX_CDS; break; }
            }
            if (status != CD_OK) break;
            ++iters;
            ++sc.n_passes_screen;
            sc.n_visits_screen += cp.nv;
            const T cm = pass(nullptr, cp.nv, true);
            if (status != CD_OK) break;
            if (cm < cp.tol) break;
            if (iters >= cp.max_iters) { status = CD_MAX_CDS; break; }
        }
        sc.rsq = bs.rsq;
        sc.resid_sum = bs.resid_sum;
        sc.iters = iters;
        sc.n_updates = bs.n_updates;
        sc.active_size = asz;
        sc.status = status;
        // (column, delta) list of the residual update + the device copy of resid_sum for the sweep epilogue
        launch_cd_compact<T>(cp.beta, cp.beta0, cp.vcol, cp.nv, cp.dcols, cp.dvals, &cp.sc->n_delta, st);
        AHIP_CHECK(hipMemcpyAsync(&sc.n_delta, &cp.sc->n_delta, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        sync();
    }

    // Buffers of the panel engine: per-block vectors, slice partials, the two tables of cached diagonal blocks (screen order /
    // activation order, `maxblk` slots of 128 x 128 each) and the host-mapped end-of-pass report.
    CdBlkState<T>* rep_st_dev = nullptr;
    int32_t* rep_seq_dev = nullptr;
    size_t panel_maxblk = 0;
    void panel_setup(size_t maxblk) {
        const int SL = cd_block_size();
        d_blk.reserve(1);
        d_dlt.reserve(SL);
        d_dcolblk.reserve(SL);
        d_gblk.reserve(SL);
        d_actcols.reserve(size_t(p) + SL);
        d_part.reserve(size_t(panel_part_elems(n)));
        if (panel_maxblk != maxblk) {
            d_Dpool.reserve(size_t(2) * maxblk * SL * SL);
            AHIP_CHECK(hipMemsetAsync(d_Dpool.p, 0, size_t(2) * maxblk * SL * SL * sizeof(T), st));
            dscr_nb.assign(maxblk, 0); dact_nb.assign(maxblk, 0);
            dscr_ver.assign(maxblk, 0); dact_ver.assign(maxblk, 0);
            if (strips_apply() && !all_scalar) { // unrotated copies of the group engine's blocks (build_stale_strips)
                d_Draw.reserve(size_t(2) * maxblk * SL * SL);
                AHIP_CHECK(hipMemsetAsync(d_Draw.p, 0, size_t(2) * maxblk * SL * SL * sizeof(T), st));
            }
            panel_maxblk = maxblk;
        }
        if (side_grams && !st2) st2 = StreamPool::take();
        for (int k = 0; side_grams && k < std::min(n_side - 1, kMaxExtra); ++k)
            if (!st_x[k]) st_x[k] = StreamPool::take();
        if (use_report && !h_report) {
            void* hp = HostPool::take(sizeof(PassReport), hipHostMallocMapped);
            void* dp = nullptr;
            if (hp && hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
                h_report = static_cast<PassReport*>(hp);
                std::memset(h_report, 0, sizeof(PassReport));
                rep_st_dev = &static_cast<PassReport*>(dp)->st;
                rep_seq_dev = &static_cast<PassReport*>(dp)->seq;
            } else {
                (void)hipGetLastError();
                HostPool::give(hp, sizeof(PassReport), hipHostMallocMapped);
                use_report = false;
            }
        }
    }
    // state of the pass that was just enqueued: spin on the sequence number its last solve publishes in host-mapped memory
    void wait_pass_state(CdBlkState<T>& bs) {
        if (h_report) {
            const auto t_spin = std::chrono::steady_clock::now();
            int spins = 0;
            while (__atomic_load_n(&h_report->seq, __ATOMIC_ACQUIRE) != report_seq) {
                if ((++spins & 0xFFFF) == 0 &&
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t_spin).count() > 20.0)
                    break; // something is wrong on the device side: fall back to a real synchronisation
            }
            if (__atomic_load_n(&h_report->seq, __ATOMIC_ACQUIRE) == report_seq) {
                bs = h_report->st;
                return;
            }
        }
        d_blk.download(&bs, 1, st);
        sync();
    }

    // Residual-based block passes (kernels_cd_panel.hip).  Per block: panel step (apply the previous block's changes to the
    // residual, partial gradients of this block) -> reduce -> one-workgroup solve against the cached diagonal block.
    // The residual is current when this returns (no end-of-fit update), also on failure (changes are undone).
    void run_panel_passes(const CdParams<T>& cp, CdScalars<T>& sc, T* r_dev) {
        // Block size: 128 visits under fixed weights (Gaussian: a diagonal block is built once and re-used for the rest of
        // the path); 64 under IRLS, where every block is rebuilt per IRLS iteration and used about once, so the MFMA cost
        // per coordinate (block size x n MACs, lower triangle only below 64) matters more than the per-block latencies.
        // (32-visit blocks with a 3-tile kernel were measured too: the fixed cost per block build and per chain step wins back
        // nothing - 0.52 vs 0.44 s on a 500k x 8000 SNP path, 2.05 vs 1.58 s on the dense 100k x 10k binomial path.)
        const int B = panel_bsz > 0 ? panel_bsz : (is_glm() ? 64 : cd_block_size());
        const int SL = cd_block_size(); // D slot: SL x SL, leading dimension SL
        const size_t maxblk = size_t((p + B - 1) / B + 1);
        panel_setup(maxblk);
        CdBlkState<T> bs{};
        bs.rsq = sc.rsq;
        bs.resid_sum = sc.resid_sum;
        bs.active_size = sc.active_size;
        bs.status = CD_OK;
        bs.nz = 0;
        const int mode = spec_mode; // 1: enqueue one (speculative) active pass and return; 2: that pass is already in flight
        if (mode != 2) d_blk.upload(&bs, 1, st);
        bool first_open = open_from_grad && mode != 2 && !cons_on; // block 0 of the first pass: gradient from the sweep
        open_from_grad = false;
        CdBlkParams<T> bp{};
        bp.nv = cp.nv; bp.vars = cp.vars; bp.xmean = cp.xmean; bp.spen = cp.spen;
        bp.beta = cp.beta; bp.is_active = cp.is_active; bp.active_set = cp.active_set;
        bp.l1 = cp.lmda * cp.alpha; bp.l2 = cp.lmda * (T(1) - cp.alpha);
        bp.max_active_size = cp.max_active_size;
        bp.dlt = d_dlt.p; bp.st = d_blk.p;
        bp.gblk = d_gblk.p; bp.vcol = cp.vcol; bp.dcol = d_dcolblk.p;
        if (cons_on) { bp.clo = d_clo.p; bp.chi = d_chi.p; bp.cmu = d_cmu.p; } // -> blk_solve_cons_kernel
        bp.bsz = B;
        bp.host_st = rep_st_dev; bp.host_seq = rep_seq_dev; bp.report_j = -1; bp.report_seq = 0;
        const T* xm_c = intercept ? cur_xm : nullptr;
        const bool trace = hooks.trace >= 1;
        int64_t iters = 0;
        int status = CD_OK;
        int asz = sc.active_size;
        // blocks prebuilt by a fit that ended before its screen pass (error paths): let them finish before anything reuses
        // their slots
        for (hipEvent_t e : pre_ev)
            if (e) AHIP_CHECK(hipStreamWaitEvent(st, e, 0));
        pre_ev.clear();
        pre_used = 0;
        const bool prebuild_screen = is_glm() && prebuild_enabled;
        bool screen_prebuilt = false;
        // look-ahead only under fixed weights: the cross blocks are built once per block pair and re-used for the rest of the
        // path; under IRLS they would double the MFMA work of every iteration
        const bool la = lookahead && !is_glm() && B == SL;
        if (la) {
            if (xscr_key.size() != maxblk) {
                d_Xpool.reserve(size_t(2) * maxblk * SL * SL);
                xscr_key.assign(maxblk, XKey{});
                xact_key.assign(maxblk, XKey{});
            }
            d_la_dlt.reserve(size_t(2) * SL); d_la_g.reserve(size_t(2) * SL); d_la_rsum.reserve(2); d_la_dd.reserve(size_t(2) * SL);
            d_la_dcol.reserve(size_t(2) * SL); d_la_dpos.reserve(size_t(2) * SL); d_la_nz.reserve(2);
            if (!d_zero_i32.p) {
                d_zero_i32.reserve(1);
                AHIP_CHECK(hipMemsetAsync(d_zero_i32.p, 0, sizeof(int32_t), st));
            }
            d_part.reserve(size_t(2 * panel_part_elems(n) + 2048));
            part2_half = size_t(panel_part_elems(n));
            d_part2.reserve(2 * part2_half);
            if (mode != 2) pending_slot = -1; // (mode 2: the pass in flight leaves its last block's changes pending)
        }
        bool no_wait = false;
        auto pass_la = [&](bool screen_pass) -> T {
            const bool first_pass = first_open;
            first_open = false;
            const int count = screen_pass ? cp.nv : asz;
            if (count <= 0) return T(0);
            const int32_t* cols_all = d_vcol.p;
            if (!screen_pass) { // (the active list only grows by appending: the gathered columns of `count` entries stay valid)
                if (actcols_key != count) launch_gather_i32(d_vcol.p, cp.active_set, count, d_actcols.p, st);
                actcols_key = count;
                cols_all = d_actcols.p;
            }
            auto& tab_nb = screen_pass ? dscr_nb : dact_nb;
            auto& tab_ver = screen_pass ? dscr_ver : dact_ver;
            T* pool = d_Dpool.p + (screen_pass ? size_t(0) : maxblk * SL * SL);
            T* xpool = d_Xpool.p + (screen_pass ? size_t(0) : maxblk * SL * SL);
            bp.list = screen_pass ? nullptr : cp.active_set;
            bp.count = count;
            bp.mark = screen_pass ? 1 : 0;
            const int nblk = (count + B - 1) / B;
            auto nb_of = [&](int j) { return std::min(B, count - j * B); };
            auto cols_of = [&](int j) { return cols_all + size_t(j) * B; };
            Stopwatch sw_enq;
            sw_enq.start();
            record_pass_e0();
            t_cd.begin(st);
            // first step of the pass: applies the pending changes of the previous pass's last block and prepares blocks 0 AND 1
            // (block 1 from a residual without block 0's changes).  It goes out before the block builds are enqueued: it does
            // not depend on them, and enqueueing them takes the host about as long as the step runs.
            // Fused opening (fuse_reduce): the first launch is a fused launch WITHOUT a solve (j = -1) that prepares block 0
            // only and leaves slice partials; block 0 is then solved by a regular fused launch whose step applies nothing and
            // prepares block 1 — one launch, one boundary and 93 MB of the first step less per pass than step + reduce + solve.
            const bool fr_open = fuse_reduce && la_fused_open;
            int prev_ld = 0;         // partials of block j left behind by the previous fused launch (0: none, gblk is ready)
            const bool from_grad = fr_open && first_pass && pending_slot < 0;
            if (from_grad) {
                // first pass of a fit right behind the invariance sweep: nothing is pending and the sweep's gradient IS the
                // block-entry gradient of block 0 — no opening launch (93 MB of columns and a launch less per fit)
                launch_la_open_from_grad<T>(d_grad.p, cols_all, nb_of(0), d_la_g.p, xm_c ? &d_blk.p->resid_sum : nullptr,
                                            d_la_rsum.p, st);
            } else if (fr_open) {
                const int ps = pending_slot;
                CdBlkParams<T> op = bp;
                op.report_j = -1;
                op.rsum_out = d_la_rsum.p;                                 // both slots <- resid_sum at the start of the pass
                op.part_rsum = xm_c ? &d_blk.p->resid_sum : nullptr;
                const int32_t* dc = ps < 0 ? d_dcolblk.p : d_la_dcol.p + size_t(ps) * SL;
                const T* dl = ps < 0 ? d_dlt.p : d_la_dlt.p + size_t(ps) * SL;
                const int32_t* nzp = ps < 0 ? &d_blk.p->nz : d_la_nz.p + ps;
                T* part_out = d_part2.p + part2_half; // parity of "launch -1"
                if (time_panel) t_step.begin(st);
                if (dense())
                    prev_ld = launch_panel_fused<T>(op, -1, D->dense<T>(), cur_w, r_dev, dc, dl, nzp, cols_all, nb_of(0), part_out, true, st);
                else
                    prev_ld = launch_panel_fused_snp<T>(op, -1, D->snp(), static_cast<const T*>(D->impute), cur_w, r_dev, dc, dl, nzp,
                                                        cols_all, nb_of(0), part_out, true, st);
                if (time_panel) t_step.end(st);
                cnt.n_panel_cols += nb_of(0);
            } else {
                const int nb01 = nb_of(0) + (nblk > 1 ? nb_of(1) : 0);
                const int ps = pending_slot;
                if (time_panel) t_step.begin(st);
                const int nsl = panel_step(cur_w, r_dev, ps < 0 ? d_dcolblk.p : d_la_dcol.p + size_t(ps) * SL,
                                           ps < 0 ? d_dlt.p : d_la_dlt.p + size_t(ps) * SL,
                                           ps < 0 ? &d_blk.p->nz : d_la_nz.p + ps, cols_all, nb01);
                if (time_panel) t_step.end(st);
                launch_panel_reduce<T>(d_part.p, nsl, nb01, cols_all, &d_blk.p->resid_sum, xm_c, d_la_g.p, st);
                cnt.n_panel_cols += nb01;
            }
            build_stale_strips(nblk, tab_nb, tab_ver, screen_pass ? &xscr_key : &xact_key, pool, xpool, nb_of, cols_of);
            build_stale_blocks(nblk, tab_nb, tab_ver, pool, nb_of, cols_of);
            build_stale_cross(nblk, screen_pass ? xscr_key : xact_key, xpool, nb_of, cols_of);
            merge_strip_events(true);
            pass_e0_valid = false;
            for (int j = 0; j < nblk; ++j) {
                const int slot = j & 1, pslot = slot ^ 1;
                bp.gblk = d_la_g.p + size_t(slot) * B;
                // fuse_reduce: the solve of block j sums the slice partials that launch j-1 left in the buffer of parity
                // (j-1)&1 itself (no panel_reduce launch in between); resid_sum of the residual they were taken from = the
                // one after block j-2's solve, which sits in this block's own rsum slot until this solve overwrites it
                bp.part = (fuse_reduce && prev_ld > 0) ? d_part2.p + size_t((j - 1) & 1) * part2_half : nullptr;
                bp.part_ld = 0; // slice-major
                bp.part_n = prev_ld;
                bp.part_rsum = xm_c ? d_la_rsum.p + slot : nullptr;
                prev_ld = 0;
                bp.Dptr = pool + size_t(j) * SL * SL;
                bp.Cprev = j > 0 ? xpool + size_t(j) * SL * SL : nullptr;
                bp.pdlt = d_la_dlt.p + size_t(pslot) * SL;
                bp.ppos = d_la_dpos.p + size_t(pslot) * SL;
                bp.pnz = d_la_nz.p + pslot;
                bp.dlt = d_la_dlt.p + size_t(slot) * SL;
                bp.dcol = d_la_dcol.p + size_t(slot) * SL;
                bp.dpos = d_la_dpos.p + size_t(slot) * SL;
                bp.nz_out = d_la_nz.p + slot;
                bp.rsum_out = d_la_rsum.p + slot;
                bp.pdd = d_la_dd.p + size_t(pslot) * SL;
                bp.dd = d_la_dd.p + size_t(slot) * SL;
                if (h_report && j == nblk - 1) {
                    bp.report_j = j;
                    bp.report_seq = ++report_seq;
                } else {
                    bp.report_j = -1;
                }
                if (blk_ev[size_t(j)]) AHIP_CHECK(hipStreamWaitEvent(st, blk_ev[size_t(j)], 0));
                if (x_ev[size_t(j)]) AHIP_CHECK(hipStreamWaitEvent(st, x_ev[size_t(j)], 0));
                if (j == 0 && !fr_open) { // nothing to overlap with: the step above already prepared block 1
                    launch_cd_panel_solve<T>(bp, 0, st);
                    continue;
                }
                // solve of block j  ||  step: apply block j-1's changes, partial gradients of block j+1
                // (j = 0 of a fused opening: nothing to apply)
                const int nbn = (j + 1 < nblk) ? nb_of(j + 1) : 0;
                const int32_t* cols_n = cols_all + size_t(j + 1) * B;
                const int32_t* nz_apply = (j == 0) ? d_zero_i32.p : d_la_nz.p + pslot;
                int ld;
                T* part_out = fuse_reduce ? d_part2.p + size_t(j & 1) * part2_half : d_part.p;
                if (time_panel) t_step.begin(st);
                if (dense())
                    ld = launch_panel_fused<T>(bp, j, D->dense<T>(), cur_w, r_dev, d_la_dcol.p + size_t(pslot) * SL,
                                               d_la_dlt.p + size_t(pslot) * SL, nz_apply, cols_n, nbn, part_out, fuse_reduce, st);
                else
                    ld = launch_panel_fused_snp<T>(bp, j, D->snp(), static_cast<const T*>(D->impute), cur_w, r_dev,
                                                   d_la_dcol.p + size_t(pslot) * SL, d_la_dlt.p + size_t(pslot) * SL,
                                                   nz_apply, cols_n, nbn, part_out, fuse_reduce, st);
                if (time_panel) t_step.end(st);
                if (nbn > 0) {
                    if (fuse_reduce) {
                        prev_ld = ld; // summed by the next solve
                    } else {
                        // resid_sum as it was before block j's solve (the residual the partials were taken from)
                        launch_panel_reduce_ld<T>(d_part.p, ld, ld, nbn, cols_n, d_la_rsum.p + pslot, xm_c,
                                                  d_la_g.p + size_t(pslot) * B, st);
                    }
                    cnt.n_panel_cols += nbn;
                }
            }
            pending_slot = (nblk - 1) & 1;
            t_cd.end(st);
            AHIP_CHECK(hipGetLastError());
            cnt.n_panel_blocks += nblk;
            t_enq += sw_enq.elapsed();
            if (no_wait) { spec_blocks = nblk; return T(0); }
            sw_enq.start();
            wait_pass_state(bs);
            t_wait += sw_enq.elapsed();
            status = bs.status;
            asz = bs.active_size;
            return bs.cm;
        };
        auto pass_plain = [&](bool screen_pass) -> T {
            first_open = false; // (only the very first pass of a fit starts from the residual the sweep saw)
            const int count = screen_pass ? cp.nv : asz;
            if (count <= 0) return T(0);
            const int32_t* cols_all = d_vcol.p;
            if (!screen_pass) { // (the active list only grows by appending: the gathered columns of `count` entries stay valid)
                if (actcols_key != count) launch_gather_i32(d_vcol.p, cp.active_set, count, d_actcols.p, st);
                actcols_key = count;
                cols_all = d_actcols.p;
            }
            auto& tab_nb = screen_pass ? dscr_nb : dact_nb;
            auto& tab_ver = screen_pass ? dscr_ver : dact_ver;
            T* pool = d_Dpool.p + (screen_pass ? size_t(0) : maxblk * SL * SL);
            bp.list = screen_pass ? nullptr : cp.active_set;
            bp.count = count;
            bp.mark = screen_pass ? 1 : 0;
            const int nblk = (count + B - 1) / B;
            Stopwatch sw_enq;
            sw_enq.start();
            // the step of block 0 goes out before the builds are enqueued (it does not depend on them; see record_pass_e0)
            auto step_of = [&](int j) {
                const int nb = std::min(B, count - j * B);
                const int32_t* cols = cols_all + size_t(j) * B;
                const int ps = (j == 0) ? pending_slot : -1;
                if (time_panel) t_step.begin(st);
                const int nsl = panel_step(cur_w, r_dev, ps < 0 ? d_dcolblk.p : d_la_dcol.p + size_t(ps) * SL,
                                           ps < 0 ? d_dlt.p : d_la_dlt.p + size_t(ps) * SL,
                                           ps < 0 ? &d_blk.p->nz : d_la_nz.p + ps, cols, nb);
                if (time_panel) t_step.end(st);
                return nsl;
            };
            record_pass_e0();
            t_cd.begin(st);
            const int nsl0 = step_of(0);
            build_stale_strips(nblk, tab_nb, tab_ver, nullptr, pool, static_cast<T*>(nullptr),
                               [&](int j) { return std::min(B, count - j * B); }, [&](int j) { return cols_all + size_t(j) * B; });
            build_stale_blocks(nblk, tab_nb, tab_ver, pool, [&](int j) { return std::min(B, count - j * B); },
                               [&](int j) { return cols_all + size_t(j) * B; }, false, screen_pass);
            merge_strip_events(false);
            pass_e0_valid = false;
            if (!screen_pass && prebuild_screen && !screen_prebuilt && side_grams && st2) {
                // IRLS: every screen-order block is stale as well (new weights) and the screen pass follows the active-set
                // passes of this fit: enqueue those builds now, behind the ones this pass waits for, so that they run while
                // the active-set passes iterate
                screen_prebuilt = true;
                const int cnt_s = cp.nv, nblk_s = (cnt_s + B - 1) / B;
                build_stale_blocks(nblk_s, dscr_nb, dscr_ver, d_Dpool.p, [&](int j) { return std::min(B, cnt_s - j * B); },
                                   [&](int j) { return d_vcol.p + size_t(j) * B; }, true);
            }
            // (a look-ahead pass may have run before: plain buffers for the solves, its pending changes for the first step)
            bp.gblk = d_gblk.p; bp.dlt = d_dlt.p; bp.dcol = d_dcolblk.p;
            bp.Cprev = nullptr; bp.dpos = nullptr; bp.nz_out = nullptr; bp.rsum_out = nullptr;
            bp.part = nullptr; bp.pdd = nullptr; bp.dd = nullptr;
            for (int j = 0; j < nblk; ++j) {
                const int nb = std::min(B, count - j * B);
                const int32_t* cols = cols_all + size_t(j) * B;
                T* Dptr = pool + size_t(j) * SL * SL;
                const int nsl = (j == 0) ? nsl0 : step_of(j);
                pending_slot = -1;
                cnt.n_panel_cols += nb;
                launch_panel_reduce<T>(d_part.p, nsl, nb, cols, &d_blk.p->resid_sum, xm_c, d_gblk.p, st);
                bp.Dptr = Dptr;
                if (h_report && j == nblk - 1) {
                    bp.report_j = j;
                    bp.report_seq = ++report_seq;
                } else {
                    bp.report_j = -1;
                }
                if (blk_ev[size_t(j)]) AHIP_CHECK(hipStreamWaitEvent(st, blk_ev[size_t(j)], 0));
                launch_cd_panel_solve<T>(bp, j, st);
            }
            t_cd.end(st);
            AHIP_CHECK(hipGetLastError()); // a failed launch would otherwise only show up as a stalled pass report
            cnt.n_panel_blocks += nblk;
            t_enq += sw_enq.elapsed();
            if (no_wait) { spec_blocks = nblk; return T(0); }
            sw_enq.start();
            wait_pass_state(bs);
            t_wait += sw_enq.elapsed();
            status = bs.status;
            asz = bs.active_size;
            if (trace) std::fprintf(stderr, "[panel] %s count=%d nblk=%d cm=%g tol=%g status=%d asz=%d nz=%d rsq=%g rsum=%g nupd=%lld\n",
                                    screen_pass ? "screen" : "active", count, nblk, double(bs.cm), double(cp.tol), status, asz,
                                    bs.nz, double(bs.rsq), double(bs.resid_sum), (long long)bs.n_updates);
            return bs.cm;
        };
        // short passes gain nothing from the look-ahead (its first two blocks run as in the plain form) and would still pay
        // for the cross blocks
        bool resume_first = mode == 2;
        auto pass = [&](bool screen_pass) -> T {
            if (resume_first) { // the first active pass of this fit was enqueued behind the previous lambda's sweep
                resume_first = false;
                Stopwatch sw_w;
                sw_w.start();
                wait_pass_state(bs);
                t_wait += sw_w.elapsed();
                status = bs.status;
                // an active-set pass never marks (CdBlkParams::mark == 0): the active list it leaves is the one it was
                // speculated on, which is what makes spec_rollback's restore of beta and the residual complete
                if (bs.active_size != int32_t(spec_asz))
                    throw make_core_error("speculative pass changed the active set (internal error).");
                asz = bs.active_size;
                return bs.cm;
            }
            const int count = screen_pass ? cp.nv : asz;
            return (la && (count + B - 1) / B >= la_min_blocks) ? pass_la(screen_pass) : pass_plain(screen_pass);
        };
        if (mode == 1) {
            spec_enqueued = false;
            if (asz > 0 && !is_glm()) {
                const int64_t cols0 = cnt.n_panel_cols;
                no_wait = true;
                if (la && (asz + B - 1) / B >= la_min_blocks) pass_la(false);
                else pass_plain(false);
                spec_cols = cnt.n_panel_cols - cols0;
                spec_enqueued = true;
            }
            return;
        }
        while (status == CD_OK) {
            while (status == CD_OK) { // solve_active, pin_naive:173-215
                ++iters;
                ++sc.n_passes_active;
                sc.n_visits_active += asz;
                const T cm = pass(false);
                if (status != CD_OK) break;
                if (cm < cp.tol) break;
                if (iters >= cp.max_iters) { status = CD_MAX_CDS; break; }
            }
            if (status != CD_OK) break;
            ++iters;
            ++sc.n_passes_screen;
            sc.n_visits_screen += cp.nv;
            const T cm = pass(true);
            if (status != CD_OK) break;
            if (cm < cp.tol) break;
            if (iters >= cp.max_iters) { status = CD_MAX_CDS; break; }
        }
        // flush the last block's changes into the residual
        t_cd.begin(st);
        if (la && pending_slot >= 0) {
            panel_step(cur_w, r_dev, d_la_dcol.p + size_t(pending_slot) * SL, d_la_dlt.p + size_t(pending_slot) * SL,
                       d_la_nz.p + pending_slot, d_vcol.p, 0);
            pending_slot = -1;
        } else {
            panel_step(cur_w, r_dev, d_dcolblk.p, d_dlt.p, &d_blk.p->nz, d_vcol.p, 0);
        }
        t_cd.end(st);
        sc.rsq = bs.rsq;
        sc.resid_sum = bs.resid_sum;
        sc.iters = iters;
        sc.n_updates = bs.n_updates;
        sc.active_size = asz;
        sc.status = status;
        sc.n_delta = 0;
        if (status != CD_OK) {
            // undo: r += X_S (beta - beta0)   (solver_gaussian_naive.hpp:286-290,326-329 restore the saved residual)
            launch_cd_compact<T>(cp.beta, cp.beta0, cp.vcol, cp.nv, cp.dcols, cp.dvals, &cp.sc->n_delta, st);
            axpy_cols(cp.dcols, cp.dvals, &cp.sc->n_delta, 0, T(1), r_dev);
            sync();
        }
    }

    // Same for problems with groups: blocks of consecutive groups (<= 128 values), partition built on the host.
    DevBuf<int32_t> d_blk_g0;
    // Per-pass tables of the panel engines, kept across passes: both visiting lists only grow by appending, so the partition
    // of the first `count` entries, the design columns behind them and the layout descriptors of their blocks are those of the
    // previous pass over the same list unless the list grew.  One copy per list (the screen list uses d_blk_g0 / d_gdesc).
    int64_t actcols_key = -1;                 // lasso engine: entries of the active list gathered into d_actcols
    struct PassTables { int64_t count = -1; int nblk = 0; };
    PassTables ptab_scr, ptab_act;
    DevBuf<int32_t> d_blk_g0_act, d_gdesc_act;
    bool pass_tables_cached = true;           // A/B hook ADELIE_HIP_PASS_TABLES=0
    std::vector<int32_t> part_host;
    int build_partition(const idx* list, idx count) { // returns nblk; fills part_host with nblk+1 list positions
        const int B = cd_block_size();
        part_host.clear();
        part_host.push_back(0);
        idx acc = 0;
        for (idx pos = 0; pos < count; ++pos) {
            const idx ss = list ? list[pos] : pos;
            const idx q = group_sizes[screen_set[ss]];
            const bool alone = host_cons(screen_set[ss]); // visited on the host: a block of its own
            if (acc > 0 && (acc + q > B || alone)) {
                part_host.push_back(int32_t(pos));
                acc = 0;
            }
            acc += q;
            if (alone) acc = B; // nothing joins it
        }
        if (count > 0) part_host.push_back(int32_t(count));
        return int(part_host.size()) - 1;
    }
    void run_group_block_passes(const CdParams<T>& cp, CdScalars<T>& sc) {
        const int B = cd_block_size();
        d_blk.reserve(1);
        d_Dbuf.reserve(size_t(2) * B * B);
        d_dlt.reserve(B);
        d_didx.reserve(B);
        CdBlkState<T> bs{};
        bs.rsq = sc.rsq;
        bs.resid_sum = sc.resid_sum;
        bs.active_size = sc.active_size;
        bs.status = CD_OK;
        d_blk.upload(&bs, 1, st);
        CdGrpBlkParams<T> bp{};
        bp.nv = cp.nv; bp.C = cp.C; bp.ldc = cp.ldc; bp.vars = cp.vars; bp.xmean = cp.xmean; bp.beta = cp.beta; bp.g = cp.g;
        bp.is_active = cp.is_active; bp.active_set = cp.active_set;
        bp.l1 = cp.lmda * cp.alpha; bp.l2 = cp.lmda * (T(1) - cp.alpha);
        bp.newton_tol = cp.newton_tol; bp.dbeta_tol = cp.dbeta_tol; bp.newton_max_iters = cp.newton_max_iters;
        bp.max_active_size = cp.max_active_size;
        bp.V = cp.V; bp.voff = cp.voff; bp.spen = cp.spen; bp.sbegin = cp.sbegin; bp.ssize = cp.ssize;
        bp.Dbuf = d_Dbuf.p; bp.dlt = d_dlt.p; bp.didx = d_didx.p; bp.st = d_blk.p;
        int64_t iters = 0;
        int status = CD_OK;
        int asz = sc.active_size;
        std::vector<idx> act_host(active_set.begin(), active_set.begin() + asz); // host mirror of the active list
        auto pass = [&](bool screen_pass) -> T {
            const idx count = screen_pass ? idx(cp.ns) : idx(asz);
            if (count <= 0) return T(0);
            const int nblk = build_partition(screen_pass ? nullptr : act_host.data(), count);
            d_blk_g0.reserve(part_host.size());
            d_blk_g0.upload(part_host.data(), part_host.size(), st);
            ptab_scr.count = -1; // (this engine shares d_blk_g0 with the panel engine's screen-list tables)
            bp.blk_g0 = d_blk_g0.p;
            bp.list = screen_pass ? nullptr : cp.active_set;
            bp.nblk = nblk;
            bp.mark = screen_pass ? 1 : 0;
            t_cd.begin(st);
            launch_cd_group_block_pass<T>(bp, st);
            t_cd.end(st);
            d_blk.download(&bs, 1, st);
            sync();
            status = bs.status;
            if (bs.active_size > asz) { // pick up the groups activated by this screen pass
                std::vector<int32_t> fresh(bs.active_size - asz);
                d_actset.download(fresh.data(), fresh.size(), st, asz);
                sync();
                for (int32_t v : fresh) act_host.push_back(v);
            }
            asz = bs.active_size;
            return bs.cm;
        };
        while (status == CD_OK) {
            while (status == CD_OK) { // solve_active, pin_naive:173-215
                ++iters;
                ++sc.n_passes_active;
                sc.n_visits_active += asz;
                const T cm = pass(false);
                if (status != CD_OK) break;
                if (cm < cp.tol) break;
                if (iters >= cp.max_iters) { status = CD_MAX_CDS; break; }
            }
            if (status != CD_OK) break;
            ++iters;
            ++sc.n_passes_screen;
            sc.n_visits_screen += cp.ns;
            const T cm = pass(true);
            if (status != CD_OK) break;
            if (cm < cp.tol) break;
            if (iters >= cp.max_iters) { status = CD_MAX_CDS; break; }
        }
        sc.rsq = bs.rsq;
        sc.resid_sum = bs.resid_sum;
        sc.iters = iters;
        sc.n_updates = bs.n_updates;
        sc.active_size = asz;
        sc.status = status;
        launch_cd_compact<T>(cp.beta, cp.beta0, cp.vcol, cp.nv, cp.dcols, cp.dvals, &cp.sc->n_delta, st);
        AHIP_CHECK(hipMemcpyAsync(&sc.n_delta, &cp.sc->n_delta, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        sync();
    }

    // Panel engine with groups: blocks = consecutive groups of the visiting list with <= 128 values (partition built on the
    // host, prefix-stable because both lists are append-only); otherwise the same data flow as run_panel_passes.
    // One visit of a group whose constraint object lives on the caller's side (pin_naive:110-168 with update_coordinate_g1_f =
    // constraint->solve, :439-458).  The group is a block of its own: its gradient was just formed by a panel step + reduce
    // (d_gblk), its coefficients, variances and eigenbasis are read back, the object's solve runs through the callback, and the
    // changes go out the way a device solve leaves them (d_beta, the compacted (column, delta) list of the next step's
    // residual update, the pass state in d_blk).  Returns the pass state after the visit.
    CdBlkState<T> host_group_visit(const CdParams<T>& cp, idx ss, bool mark, bool first_of_pass) {
        const idx g = screen_set[ss], q = group_sizes[g], b = screen_begins[ss];
        const size_t uq = static_cast<size_t>(q);
        const bool trace_hv = hooks.trace >= 1;
        if (trace_hv) std::fprintf(stderr, "[host visit] ss=%lld g=%lld q=%lld b=%lld voff=%lld v_used=%zu nv=%lld\n", (long long)ss, (long long)g,
                                   (long long)q, (long long)b, (long long)(size_t(ss) < h_voff.size() ? h_voff[size_t(ss)] : -1), v_used, (long long)nv);
        std::vector<T> gk(uq), ak(uq), Ak(uq), Vk(uq * uq, T(1));
        CdBlkState<T> bs{};
        int8_t was_active = 0;
        d_gblk.download(gk.data(), size_t(q), st);
        d_beta.download(ak.data(), size_t(q), st, size_t(b));
        d_vars.download(Ak.data(), size_t(q), st, size_t(b));
        if (q > 1) d_V.download(Vk.data(), size_t(q) * q, st, size_t(h_voff[size_t(ss)]));
        d_blk.download(&bs, 1, st);
        d_isact.download(&was_active, 1, st, size_t(ss));
        sync();
        if (first_of_pass) bs.cm = T(0); // the convergence measure is per pass (the device solves reset it in block 0)
        const T pk = penalty[g];
        const double l1 = double(cp.lmda * cp.alpha) * double(pk), l2 = double(cp.lmda * (T(1) - cp.alpha)) * double(pk);
        std::vector<double> gt(uq), a_old_t(uq), x(uq), quad(uq), lin(uq), Qd(uq * uq);
        for (idx j = 0; j < q; ++j) { // into the eigenbasis: g V, beta V  (:123-135)
            double s1 = 0, s2 = 0;
            for (idx i = 0; i < q; ++i) {
                s1 += double(gk[size_t(i)]) * double(Vk[size_t(i + j * q)]);
                s2 += double(ak[size_t(i)]) * double(Vk[size_t(i + j * q)]);
            }
            gt[size_t(j)] = s1;
            a_old_t[size_t(j)] = s2;
            x[size_t(j)] = s2;
            quad[size_t(j)] = double(Ak[size_t(j)]);
            lin[size_t(j)] = s1 + double(Ak[size_t(j)]) * s2;
        }
        for (size_t e = 0; e < Qd.size(); ++e) Qd[e] = double(Vk[e]);
        if (cons_cb->solve(cons_cb->user, g, q, x.data(), quad.data(), lin.data(), l1, l2, Qd.data()))
            throw make_solver_error("constraint.solve() raised.");
        double dn = 0;
        for (idx j = 0; j < q; ++j) dn += (a_old_t[size_t(j)] - x[size_t(j)]) * (a_old_t[size_t(j)] - x[size_t(j)]);
        bs.nz = 0;
        if (!(std::sqrt(dn) <= g_dbeta_tol * std::sqrt(double(q)))) { // :144: the group changed
            double cmv = 0, rs = 0;
            for (idx j = 0; j < q; ++j) {
                const double dl = x[size_t(j)] - a_old_t[size_t(j)];
                cmv += quad[size_t(j)] * dl * dl;
                rs += dl * (2 * gt[size_t(j)] - dl * quad[size_t(j)]);
            }
            bs.cm = std::max(bs.cm, T(cmv / double(q))); // pin_base:100-110
            bs.rsq += T(rs);                              // pin_base:124-134
            std::vector<T> a_new(uq), dlt(uq);
            std::vector<int32_t> dcol(uq);
            double rsum = 0;
            for (idx i = 0; i < q; ++i) { // back: beta = x V^T  (:156-157)
                double acc = 0;
                for (idx j = 0; j < q; ++j) acc += x[size_t(j)] * double(Vk[size_t(i + j * q)]);
                a_new[size_t(i)] = T(acc);
                dlt[size_t(i)] = a_new[size_t(i)] - ak[size_t(i)];
                dcol[size_t(i)] = int32_t(groups[g] + i);
                rsum += double(screen_X_means[size_t(b + i)]) * double(ak[size_t(i)] - a_new[size_t(i)]);
            }
            bs.resid_sum += T(rsum);
            bs.n_updates += 1;
            bs.nz = int32_t(q);
            d_beta.upload(a_new.data(), size_t(q), st, size_t(b));
            d_dcolblk.upload(dcol.data(), size_t(q), st);
            d_dlt.upload(dlt.data(), size_t(q), st);
            if (mark && !was_active) { // add_active_set, pin_naive:294-304
                if (size_t(bs.active_size) >= max_active_size) {
                    bs.status = CD_MAX_ACTIVE;
                } else {
                    const int8_t one = 1;
                    const int32_t ssi = int32_t(ss);
                    d_isact.upload(&one, 1, st, size_t(ss));
                    d_actset.upload(&ssi, 1, st, size_t(bs.active_size));
                    bs.active_size += 1;
                }
            }
        }
        d_blk.upload(&bs, 1, st);
        sync();
        return bs;
    }
    // abs_grad of the groups with host constraint objects (solver_base.hpp:62-93): the constraint's gradient for screened groups,
    // its solve_zero for the others; overrides what the device kernel wrote for them (it knows no bounds for these groups)
    void host_cons_abs_grad(T lm) {
        if (!cons_host) return;
        d_grad.download(grad.data(), size_t(p), st);
        sync();
        std::vector<double> v, out;
        std::vector<idx> begin_of(static_cast<size_t>(G), idx(-1));
        for (size_t ss = 0; ss < screen_set.size() && ss < screen_begins.size(); ++ss) begin_of[size_t(screen_set[ss])] = screen_begins[ss];
        for (idx g = 0; g < G; ++g) {
            if (!host_cons(g)) continue;
            const idx q = group_sizes[g], k = groups[g];
            v.assign(size_t(q), 0);
            if (begin_of[size_t(g)] >= 0) {
                const idx b = begin_of[size_t(g)];
                const double regul = double((1 - alpha) * lm) * double(penalty[g]);
                for (idx t = 0; t < q; ++t) v[size_t(t)] = double(screen_beta[size_t(b + t)]);
                out.assign(size_t(q), 0);
                if (cons_cb->gradient(cons_cb->user, g, q, v.data(), out.data()))
                    throw make_solver_error("constraint.gradient() raised.");
                double acc = 0;
                for (idx t = 0; t < q; ++t) {
                    const double e = double(grad[size_t(k + t)]) - regul * v[size_t(t)] - out[size_t(t)];
                    acc += e * e;
                }
                abs_grad[size_t(g)] = T(std::sqrt(acc));
            } else {
                for (idx t = 0; t < q; ++t) v[size_t(t)] = double(grad[size_t(k + t)]);
                double nrm = 0;
                if (cons_cb->solve_zero(cons_cb->user, g, q, v.data(), &nrm))
                    throw make_solver_error("constraint.solve_zero() raised.");
                abs_grad[size_t(g)] = T(nrm);
            }
        }
    }

    void run_group_panel_passes(const CdParams<T>& cp, CdScalars<T>& sc, T* r_dev) {
        const int SL = cd_block_size();
        panel_setup(group_maxblk());
        const size_t maxblk = panel_maxblk;
        CdBlkState<T> bs{};
        bs.rsq = sc.rsq;
        bs.resid_sum = sc.resid_sum;
        bs.active_size = sc.active_size;
        bs.status = CD_OK;
        bs.nz = 0;
        const int mode = spec_mode; // see run_panel_passes
        if (mode != 2) d_blk.upload(&bs, 1, st);
        bool first_open = open_from_grad && mode != 2 && !cons_on && !multi(); // see run_panel_passes
        open_from_grad = false;
        CdGrpBlkParams<T> bp{};
        bp.nv = cp.nv; bp.vars = cp.vars; bp.xmean = cp.xmean; bp.beta = cp.beta;
        bp.is_active = cp.is_active; bp.active_set = cp.active_set;
        bp.l1 = cp.lmda * cp.alpha; bp.l2 = cp.lmda * (T(1) - cp.alpha);
        bp.newton_tol = cp.newton_tol; bp.dbeta_tol = cp.dbeta_tol; bp.newton_max_iters = cp.newton_max_iters;
        bp.max_active_size = cp.max_active_size;
        bp.V = cp.V; bp.voff = cp.voff; bp.spen = cp.spen; bp.sbegin = cp.sbegin; bp.ssize = cp.ssize;
        bp.dlt = d_dlt.p; bp.st = d_blk.p;
        bp.gblk = d_gblk.p; bp.vcol = cp.vcol; bp.dcol = d_dcolblk.p;
        bp.host_st = rep_st_dev; bp.host_seq = rep_seq_dev; bp.report_j = -1; bp.report_seq = 0;
        bp.rot = group_rot ? 1 : 0;
        if (cons_on) { bp.clo = d_clo.p; bp.chi = d_chi.p; bp.cmu = d_cmu.p; }
        struct RotGuard { // builds of this fit are rotated behind their launch (build_stale_blocks); off again on any exit
            Solver* s;
            ~RotGuard() { s->rot_on = false; s->rot_list = nullptr; }
        } rot_guard{this};
#ifdef AHIP_GRP_PROFILE // (profile build, scripts/grp_profile.py: cycle counters of the group solve)
        if (!d_grp_dbg.p) { d_grp_dbg.reserve(8); AHIP_CHECK(hipMemsetAsync(d_grp_dbg.p, 0, 8 * sizeof(int64_t), st)); }
        bp.dbg = d_grp_dbg.p;
#endif
        const T* xm_c = intercept ? cur_xm : nullptr;
        int64_t iters = 0;
        int status = CD_OK;
        int asz = sc.active_size;
        std::vector<idx> act_host(active_set.begin(), active_set.begin() + asz); // host mirror of the active list
        std::vector<int32_t>& acols = h_actcols;
        // look-ahead form (see run_panel_passes); not on the multi-response view, whose step is a different kernel
        const bool la = lookahead && !is_glm() && (!multi() || multi_w_uniform);
        if (la) {
            if (xscr_key.size() != maxblk) {
                d_Xpool.reserve(size_t(2) * maxblk * SL * SL);
                xscr_key.assign(maxblk, XKey{});
                xact_key.assign(maxblk, XKey{});
            }
            d_la_dlt.reserve(size_t(2) * SL); d_la_g.reserve(size_t(2) * SL); d_la_rsum.reserve(2); d_la_dd.reserve(size_t(2) * SL);
            d_la_dcol.reserve(size_t(2) * SL); d_la_dpos.reserve(size_t(2) * SL); d_la_nz.reserve(2);
            d_la_dd.reserve(size_t(2) * SL);
            d_part.reserve(size_t(2 * panel_part_elems(n) + 2048));
            part2_half = size_t(panel_part_elems(n));
            d_part2.reserve(2 * part2_half);
            if (mode != 2) pending_slot = -1;
        }
        // (The group solve summing the slice partials itself, as the lasso solve does, was measured slower — config 3: 722.7 ms
        // with, 654.1 ms without: a group launch is bound by its solve — and removed in round 4.)
        // The LAST STEP WORKGROUP of a fused launch sums them instead: it finishes ~9 us before the solve does, and summing
        // 196 x 128 partials takes one workgroup 3 us (CdGrpBlkParams::tail_counter).  No panel_reduce launch between two fused
        // launches (5.3 us + two boundaries per block).  One partial per column and workgroup is what the kernel sums: 16-byte
        // aligned dense designs in double precision / any SNP design.
        bool tail_ok = false;
        if (!multi()) {
            if (dense()) {
                constexpr int V = int(16 / sizeof(T));
                tail_ok = (64 * V >= 128) && (D->ld % V == 0) && ((reinterpret_cast<uintptr_t>(D->X) % 16) == 0);
            } else {
                tail_ok = true; // (SNP: 4 rows per lane and load, 256-row slices)
            }
        }
        if (tail_ok && !d_tail_counter.p) {
            d_tail_counter.reserve(1);
            AHIP_CHECK(hipMemsetAsync(d_tail_counter.p, 0, sizeof(int32_t), st));
        }
        if (!d_zero_i32.p) {
            d_zero_i32.reserve(1);
            AHIP_CHECK(hipMemsetAsync(d_zero_i32.p, 0, sizeof(int32_t), st));
        }
        if (tail_ok) d_part2.reserve(2 * size_t(panel_part_elems(n)));
        bool no_wait = false;
        d_gdesc.reserve(maxblk * size_t(GDESC_STRIDE));
        auto pass_la = [&](bool screen_pass) -> T {
            const bool first_pass = first_open;
            first_open = false;
            const idx count = screen_pass ? idx(cp.ns) : idx(asz);
            if (count <= 0) return T(0);
            const int nblk = build_partition_values(screen_pass ? nullptr : act_host.data(), count);
            PassTables& ptab = screen_pass ? ptab_scr : ptab_act;
            DevBuf<int32_t>& g0buf = screen_pass ? d_blk_g0 : d_blk_g0_act;
            DevBuf<int32_t>& descbuf = screen_pass ? d_gdesc : d_gdesc_act;
            const bool tables_hit = pass_tables_cached && ptab.count == int64_t(count) && ptab.nblk == nblk && g0buf.p && descbuf.p;
            if (!tables_hit) {
                g0buf.reserve(std::max<size_t>(part_host.size(), maxblk + 2));
                g0buf.upload(part_host.data(), part_host.size(), st);
            }
            const int32_t* cols_all = d_vcol.p;
            if (!screen_pass) {
                if (!tables_hit) {
                    acols.clear();
                    for (idx pos = 0; pos < count; ++pos) {
                        const idx g = screen_set[act_host[pos]];
                        for (idx t = 0; t < group_sizes[g]; ++t) acols.push_back(int32_t(groups[g] + t));
                    }
                    d_actcols.upload(acols.data(), acols.size(), st);
                }
                cols_all = d_actcols.p;
            }
            auto& tab_nb = screen_pass ? dscr_nb : dact_nb;
            auto& tab_ver = screen_pass ? dscr_ver : dact_ver;
            T* pool = d_Dpool.p + (screen_pass ? size_t(0) : maxblk * SL * SL);
            T* xpool = d_Xpool.p + (screen_pass ? size_t(0) : maxblk * SL * SL);
            bp.blk_g0 = g0buf.p;
            bp.list = screen_pass ? nullptr : cp.active_set;
            bp.nblk = nblk;
            bp.mark = screen_pass ? 1 : 0;
            descbuf.reserve(maxblk * size_t(GDESC_STRIDE));
            bp.desc = descbuf.p;
            if (bp.rot && !tables_hit) launch_grp_layout<T>(bp, nblk, descbuf.p, st);
            ptab.count = int64_t(count);
            ptab.nblk = nblk;
            auto nb_of = [&](int j) { return int(gp_vbeg[size_t(j) + 1] - gp_vbeg[j]); };
            auto cols_of = [&](int j) { return cols_all + gp_vbeg[j]; };
            record_pass_e0();
            t_cd.begin(st);
            // first step of the pass: pending changes of the previous pass's last block; blocks 0 and 1 prepared.  Enqueued
            // before the block builds (it does not depend on them, see record_pass_e0).
            // Fused opening (tail reduce available): a fused launch without a solve (j = -1) applies the pending changes and
            // prepares block 0 (its last step workgroup leaves the gradient); block 0 is then solved by a regular fused
            // launch whose step applies nothing and prepares block 1 - instead of step + two reduces + a stand-alone solve.
            const bool fr_open = tail_ok && la_fused_open && dense();
            if (fr_open && first_pass && pending_slot < 0) {
                // (as in run_panel_passes: block 0's gradient out of the sweep's result, no opening launch)
                launch_la_open_from_grad<T>(d_grad.p, cols_all, nb_of(0), d_la_g.p, xm_c ? &d_blk.p->resid_sum : nullptr,
                                            d_la_rsum.p, st);
            } else if (fr_open) {
                const int ps = pending_slot;
                CdGrpBlkParams<T> op = bp;
                op.report_j = -1;
                op.rsum_out = d_la_rsum.p;
                op.part_rsum = xm_c ? &d_blk.p->resid_sum : nullptr;
                op.tail_counter = d_tail_counter.p;
                op.tail_g = d_la_g.p;
                op.tail_rsum = &d_blk.p->resid_sum;
                op.tail_xm = xm_c;
                if (time_panel) t_step.begin(st);
                launch_panel_fused_grp<T>(op, -1, D->dense<T>(), cur_w, r_dev, ps < 0 ? d_dcolblk.p : d_la_dcol.p + size_t(ps) * SL,
                                          ps < 0 ? d_dlt.p : d_la_dlt.p + size_t(ps) * SL, ps < 0 ? &d_blk.p->nz : d_la_nz.p + ps,
                                          cols_all, nb_of(0), d_part2.p, true, st);
                if (time_panel) t_step.end(st);
                cnt.n_panel_cols += nb_of(0);
            } else {
                const int nv0 = nb_of(0), nv1 = nblk > 1 ? nb_of(1) : 0;
                const int ps = pending_slot;
                if (time_panel) t_step.begin(st);
                const int nsl = panel_step(cur_w, r_dev, ps < 0 ? d_dcolblk.p : d_la_dcol.p + size_t(ps) * SL,
                                           ps < 0 ? d_dlt.p : d_la_dlt.p + size_t(ps) * SL,
                                           ps < 0 ? &d_blk.p->nz : d_la_nz.p + ps, cols_all, nv0 + nv1);
                if (time_panel) t_step.end(st);
                launch_panel_reduce<T>(d_part.p, nsl, nv0, cols_all, &d_blk.p->resid_sum, xm_c, d_la_g.p, st);
                if (nv1 > 0)
                    launch_panel_reduce<T>(d_part.p + size_t(nv0) * size_t(nsl), nsl, nv1, cols_all + nv0, &d_blk.p->resid_sum,
                                           xm_c, d_la_g.p + SL, st);
                cnt.n_panel_cols += nv0 + nv1;
            }
            if (strips_apply() && !multi()) {
                T* raw = group_rot ? d_Draw.reserve(size_t(2) * maxblk * SL * SL) + (screen_pass ? size_t(0) : maxblk * SL * SL) : pool;
                build_stale_strips(nblk, tab_nb, tab_ver, screen_pass ? &xscr_key : &xact_key, raw, xpool, nb_of, cols_of,
                                   group_rot ? pool : nullptr, screen_pass ? nullptr : act_host.data());
            } else {
                strip_ev.clear();
            }
            rot_on = group_rot;
            rot_list = screen_pass ? nullptr : act_host.data();
            build_stale_blocks(nblk, tab_nb, tab_ver, pool, nb_of, cols_of);
            rot_on = false;
            build_stale_cross(nblk, screen_pass ? xscr_key : xact_key, xpool, nb_of, cols_of);
            merge_strip_events(true);
            pass_e0_valid = false;
            if (screen_pass) join_uv(); // the new screen groups' blocks / variances / eigenbases (update_vars_panel_groups)
            int prev_ld = 0; // partials of block j left behind by the previous fused launch (fr_grp), see run_panel_passes
            for (int j = 0; j < nblk; ++j) {
                const int slot = j & 1, pslot = slot ^ 1;
                bp.gblk = d_la_g.p + size_t(slot) * SL;
                bp.Dptr = pool + size_t(j) * SL * SL;
                bp.Cprev = j > 0 ? xpool + size_t(j) * SL * SL : nullptr;
                bp.pdlt = d_la_dlt.p + size_t(pslot) * SL;
                bp.ppos = d_la_dpos.p + size_t(pslot) * SL;
                bp.pnz = d_la_nz.p + pslot;
                bp.dlt = d_la_dlt.p + size_t(slot) * SL;
                bp.dcol = d_la_dcol.p + size_t(slot) * SL;
                bp.dpos = d_la_dpos.p + size_t(slot) * SL;
                bp.nz_out = d_la_nz.p + slot;
                bp.rsum_out = d_la_rsum.p + slot;
                bp.pdd = d_la_dd.p + size_t(pslot) * SL;
                bp.dd = d_la_dd.p + size_t(slot) * SL;
                bp.part = nullptr;
                bp.part_n = prev_ld;
                bp.part_rsum = xm_c ? d_la_rsum.p + slot : nullptr;
                prev_ld = 0;
                // tail reduce of this launch's partials (block j + 1): resid_sum as it was before block j's solve
                bp.tail_counter = tail_ok ? d_tail_counter.p : nullptr;
                bp.tail_g = d_la_g.p + size_t(pslot) * SL;
                bp.tail_rsum = d_la_rsum.p + pslot;
                bp.tail_xm = xm_c;
                if (h_report && j == nblk - 1) {
                    bp.report_j = j;
                    bp.report_seq = ++report_seq;
                } else {
                    bp.report_j = -1;
                }
                if (blk_ev[size_t(j)]) AHIP_CHECK(hipStreamWaitEvent(st, blk_ev[size_t(j)], 0));
                if (x_ev[size_t(j)]) AHIP_CHECK(hipStreamWaitEvent(st, x_ev[size_t(j)], 0));
                if (j == 0 && !fr_open) {
                    launch_cd_group_panel_solve<T>(bp, 0, st);
                    continue;
                }
                const int nbn = (j + 1 < nblk) ? nb_of(j + 1) : 0;
                const int32_t* cols_n = cols_all + gp_vbeg[size_t(j) + 1];
                const int32_t* nz_apply = (j == 0) ? d_zero_i32.p : d_la_nz.p + pslot; // (j = 0 of a fused opening: nothing to apply)
                int ld;
                if (time_panel) t_step.begin(st);
                if (multi())
                    ld = launch_multi_panel_fused<T>(bp, j, D->multi<T>(), cur_w, r_dev, d_la_dcol.p + size_t(pslot) * SL,
                                                     d_la_dlt.p + size_t(pslot) * SL, d_la_nz.p + pslot, cols_n, nbn, d_part.p, st);
                else if (dense())
                    ld = launch_panel_fused_grp<T>(bp, j, D->dense<T>(), cur_w, r_dev, d_la_dcol.p + size_t(pslot) * SL,
                                                   d_la_dlt.p + size_t(pslot) * SL, nz_apply, cols_n, nbn,
                                                   tail_ok ? d_part2.p : d_part.p,
                                                   tail_ok, st);
                else
                    ld = launch_panel_fused_grp_snp<T>(bp, j, D->snp(), static_cast<const T*>(D->impute), cur_w, r_dev,
                                                       d_la_dcol.p + size_t(pslot) * SL, d_la_dlt.p + size_t(pslot) * SL,
                                                       d_la_nz.p + pslot, cols_n, nbn,
                                                       tail_ok ? d_part2.p : d_part.p,
                                                       tail_ok, st);
                if (time_panel) t_step.end(st);
                if (nbn > 0) {
                    if (tail_ok) { /* summed by the launch's last step workgroup */ }
                    else
                        launch_panel_reduce_ld<T>(d_part.p, ld, ld, nbn, cols_n, d_la_rsum.p + pslot, xm_c,
                                                  d_la_g.p + size_t(pslot) * SL, st);
                    cnt.n_panel_cols += nbn;
                }
            }
            pending_slot = (nblk - 1) & 1;
            t_cd.end(st);
            AHIP_CHECK(hipGetLastError());
            cnt.n_panel_blocks += nblk;
            if (no_wait) { spec_blocks = nblk; return T(0); }
            wait_pass_state(bs);
            status = bs.status;
            if (bs.active_size > asz) {
                std::vector<int32_t> fresh(size_t(bs.active_size - asz));
                d_actset.download(fresh.data(), fresh.size(), st, asz);
                sync();
                for (int32_t v : fresh) act_host.push_back(v);
            }
            asz = bs.active_size;
            return bs.cm;
        };
        CdBlkState<T> host_bs{};
        bool last_on_host = false;
        auto pass_plain = [&](bool screen_pass) -> T {
            first_open = false;
            const idx count = screen_pass ? idx(cp.ns) : idx(asz);
            if (count <= 0) return T(0);
            last_on_host = false;
            const int nblk = build_partition_values(screen_pass ? nullptr : act_host.data(), count);
            PassTables& ptab = screen_pass ? ptab_scr : ptab_act;
            DevBuf<int32_t>& g0buf = screen_pass ? d_blk_g0 : d_blk_g0_act;
            DevBuf<int32_t>& descbuf = screen_pass ? d_gdesc : d_gdesc_act;
            const bool tables_hit = pass_tables_cached && ptab.count == int64_t(count) && ptab.nblk == nblk && g0buf.p && descbuf.p;
            if (!tables_hit) {
                g0buf.reserve(std::max<size_t>(part_host.size(), maxblk + 2));
                g0buf.upload(part_host.data(), part_host.size(), st);
            }
            const int32_t* cols_all = d_vcol.p;
            if (!screen_pass) { // design columns of the active values in visiting order
                if (!tables_hit) {
                    acols.clear();
                    for (idx pos = 0; pos < count; ++pos) {
                        const idx g = screen_set[act_host[pos]];
                        for (idx t = 0; t < group_sizes[g]; ++t) acols.push_back(int32_t(groups[g] + t));
                    }
                    d_actcols.upload(acols.data(), acols.size(), st);
                }
                cols_all = d_actcols.p;
            }
            auto& tab_nb = screen_pass ? dscr_nb : dact_nb;
            auto& tab_ver = screen_pass ? dscr_ver : dact_ver;
            T* pool = d_Dpool.p + (screen_pass ? size_t(0) : maxblk * SL * SL);
            bp.blk_g0 = g0buf.p;
            bp.list = screen_pass ? nullptr : cp.active_set;
            bp.nblk = nblk;
            bp.mark = screen_pass ? 1 : 0;
            descbuf.reserve(maxblk * size_t(GDESC_STRIDE));
            bp.desc = descbuf.p;
            bp.pdd = nullptr; bp.dd = nullptr;
            if (bp.rot && !tables_hit) launch_grp_layout<T>(bp, nblk, descbuf.p, st);
            ptab.count = int64_t(count);
            ptab.nblk = nblk;
            if (strips_apply()) {
                T* raw = group_rot ? d_Draw.reserve(size_t(2) * maxblk * SL * SL) + (screen_pass ? size_t(0) : maxblk * SL * SL) : pool;
                build_stale_strips(nblk, tab_nb, tab_ver, nullptr, raw, static_cast<T*>(nullptr),
                                   [&](int j) { return int(gp_vbeg[size_t(j) + 1] - gp_vbeg[j]); },
                                   [&](int j) { return cols_all + gp_vbeg[j]; }, group_rot ? pool : nullptr,
                                   screen_pass ? nullptr : act_host.data());
            } else {
                strip_ev.clear();
            }
            rot_on = group_rot;
            rot_list = screen_pass ? nullptr : act_host.data();
            build_stale_blocks(nblk, tab_nb, tab_ver, pool, [&](int j) { return int(gp_vbeg[size_t(j) + 1] - gp_vbeg[j]); },
                               [&](int j) { return cols_all + gp_vbeg[j]; });
            merge_strip_events(false);
            if (screen_pass) join_uv();
            t_cd.begin(st);
            bp.gblk = d_gblk.p; bp.dlt = d_dlt.p; bp.dcol = d_dcolblk.p;
            bp.Cprev = nullptr; bp.dpos = nullptr; bp.nz_out = nullptr; bp.rsum_out = nullptr;
            for (int j = 0; j < nblk; ++j) {
                const int nval = gp_vbeg[size_t(j) + 1] - gp_vbeg[j];
                const int32_t* cols = cols_all + gp_vbeg[j];
                T* Dptr = pool + size_t(j) * SL * SL;
                const int ps = (j == 0) ? pending_slot : -1;
                if (time_panel) t_step.begin(st);
                const int nsl = panel_step(cur_w, r_dev, ps < 0 ? d_dcolblk.p : d_la_dcol.p + size_t(ps) * SL,
                                           ps < 0 ? d_dlt.p : d_la_dlt.p + size_t(ps) * SL,
                                           ps < 0 ? &d_blk.p->nz : d_la_nz.p + ps, cols, nval);
                if (time_panel) t_step.end(st);
                pending_slot = -1;
                cnt.n_panel_cols += nval;
                launch_panel_reduce<T>(d_part.p, nsl, nval, cols, &d_blk.p->resid_sum, xm_c, d_gblk.p, st);
                if (cons_host) { // a block that is one group with a constraint object on the caller's side: visited on the host
                    const idx ss0 = screen_pass ? idx(part_host[size_t(j)]) : act_host[size_t(part_host[size_t(j)])];
                    if (part_host[size_t(j) + 1] - part_host[size_t(j)] == 1 && host_cons(screen_set[ss0])) {
                        if (blk_ev[size_t(j)]) AHIP_CHECK(hipStreamWaitEvent(st, blk_ev[size_t(j)], 0)); // (its eigenbasis)
                        host_bs = host_group_visit(cp, ss0, screen_pass, j == 0);
                        last_on_host = (j == nblk - 1);
                        continue;
                    }
                    last_on_host = false;
                }
                bp.Dptr = Dptr;
                if (h_report && j == nblk - 1) {
                    bp.report_j = j;
                    bp.report_seq = ++report_seq;
                } else {
                    bp.report_j = -1;
                }
                if (blk_ev[size_t(j)]) AHIP_CHECK(hipStreamWaitEvent(st, blk_ev[size_t(j)], 0));
                launch_cd_group_panel_solve<T>(bp, j, st);
            }
            t_cd.end(st);
            AHIP_CHECK(hipGetLastError()); // a failed launch would otherwise only show up as a stalled pass report
            cnt.n_panel_blocks += nblk;
            if (no_wait) { spec_blocks = nblk; return T(0); }
            if (last_on_host) bs = host_bs; // (no device solve published a report for this pass)
            else wait_pass_state(bs);
            status = bs.status;
            if (bs.active_size > asz) { // pick up the groups activated by this screen pass
                std::vector<int32_t> fresh(size_t(bs.active_size - asz));
                d_actset.download(fresh.data(), fresh.size(), st, asz);
                sync();
                for (int32_t v : fresh) act_host.push_back(v);
            }
            asz = bs.active_size;
            return bs.cm;
        };
        bool resume_first = mode == 2;
        auto pass = [&](bool screen_pass) -> T {
            if (resume_first) { // the first active pass of this fit was enqueued behind the previous lambda's sweep
                resume_first = false;
                wait_pass_state(bs);
                status = bs.status;
                asz = bs.active_size;
                return bs.cm;
            }
            if (!la) return pass_plain(screen_pass);
            const idx count = screen_pass ? idx(cp.ns) : idx(asz);
            const int nblk = count > 0 ? build_partition(screen_pass ? nullptr : act_host.data(), count) : 0;
            return nblk >= la_min_blocks ? pass_la(screen_pass) : pass_plain(screen_pass);
        };
        if (mode == 1) {
            spec_enqueued = false;
            if (asz > 0 && !is_glm()) {
                const int64_t cols0 = cnt.n_panel_cols;
                no_wait = true;
                pass(false);
                spec_cols = cnt.n_panel_cols - cols0;
                spec_enqueued = true;
            }
            return;
        }
        while (status == CD_OK) {
            while (status == CD_OK) { // solve_active, pin_naive:173-215
                ++iters;
                ++sc.n_passes_active;
                sc.n_visits_active += asz;
                const T cm = pass(false);
                if (status != CD_OK) break;
                if (cm < cp.tol) break;
                if (iters >= cp.max_iters) { status = CD_MAX_CDS; break; }
            }
            if (status != CD_OK) break;
            ++iters;
            ++sc.n_passes_screen;
            sc.n_visits_screen += cp.ns;
            const T cm = pass(true);
            if (status != CD_OK) break;
            if (cm < cp.tol) break;
            if (iters >= cp.max_iters) { status = CD_MAX_CDS; break; }
        }
        // flush the last block's changes into the residual
        t_cd.begin(st);
        if (la && pending_slot >= 0) {
            panel_step(cur_w, r_dev, d_la_dcol.p + size_t(pending_slot) * SL, d_la_dlt.p + size_t(pending_slot) * SL,
                       d_la_nz.p + pending_slot, d_vcol.p, 0);
            pending_slot = -1;
        } else {
            panel_step(cur_w, r_dev, d_dcolblk.p, d_dlt.p, &d_blk.p->nz, d_vcol.p, 0);
        }
        t_cd.end(st);
        sc.rsq = bs.rsq;
        sc.resid_sum = bs.resid_sum;
        sc.iters = iters;
        sc.n_updates = bs.n_updates;
        sc.active_size = asz;
        sc.status = status;
        sc.n_delta = 0;
        if (status != CD_OK) { // undo: r += X_S (beta - beta0)
            launch_cd_compact<T>(cp.beta, cp.beta0, cp.vcol, cp.nv, cp.dcols, cp.dvals, &cp.sc->n_delta, st);
            axpy_cols(cp.dcols, cp.dvals, &cp.sc->n_delta, 0, T(1), r_dev);
            sync();
        }
    }

    // ---------------------------------------------------------------------------------------------------------
    // One pin solve on the device (solver_gaussian_pin_naive.hpp:217-401 for a single lambda).
    // Preconditions: Gram/vars/sxm valid for [0,nv) under the weights in use; d_g holds the current gradient of the
    // screen values; d_beta the current coefficients.  On success the residual `r_dev` is updated.
    FitOut<T> pin_solve(T lm, T pin_tol, T rsq_in, T& rsum_io, T y_mean_pin, T* r_dev) {
        poll_mid();
        const idx ns = idx(screen_set.size());
        FitOut<T> o;
        bool resume = false;
        if (spec_active) {
            resume = lm == spec_lm && r_dev == d_r.p && !is_glm() && nv >= spec_nv && panel_mode() &&
                     active_set_size == spec_asz;
            if (!resume) spec_rollback();
        }
        if (!resume) {
            AHIP_CHECK(hipMemcpyAsync(d_beta0.p, d_beta.p, size_t(nv) * sizeof(T), hipMemcpyDeviceToDevice, st));
        } else if (nv > spec_nv) { // the screen values appended since: beta0 = beta at fit entry for them too
            AHIP_CHECK(hipMemcpyAsync(d_beta0.p + spec_nv, d_beta.p + spec_nv, size_t(nv - spec_nv) * sizeof(T),
                                      hipMemcpyDeviceToDevice, st));
        }
        CdScalars<T> sc{};
        sc.rsq = rsq_in;
        sc.resid_sum = rsum_io;
        sc.active_size = int32_t(active_set_size);
        d_sc.upload(&sc, 1, st);
        CdParams<T> cp{};
        cp.nv = int32_t(nv);
        cp.ns = int32_t(ns);
        cp.sbegin = d_sbegin.p;
        cp.ssize = d_ssize.p;
        cp.spen = d_spen.p;
        cp.C = d_C.p;
        cp.ldc = ldc;
        cp.vars = d_vars.p;
        cp.xmean = d_sxm.p;
        cp.V = d_V.p;
        cp.voff = d_voff.p;
        cp.beta = d_beta.p;
        cp.g = d_g.p;
        cp.is_active = d_isact.p;
        cp.active_set = d_actset.p;
        cp.lmda = lm;
        cp.alpha = alpha;
        cp.tol = pin_tol;
        cp.newton_tol = newton_tol;
        cp.dbeta_tol = T(g_dbeta_tol);
        cp.newton_max_iters = int32_t(std::min<size_t>(newton_max_iters, size_t(1) << 30));
        cp.max_active_size = int32_t(std::min<size_t>(max_active_size, size_t(1) << 30));
        cp.intercept = intercept;
        cp.all_scalar = all_scalar ? 1 : 0;
        cp.max_iters = int64_t(max_iters);
        cp.sc = d_sc.p;
        cp.beta0 = d_beta0.p;
        cp.vcol = d_vcol.p;
        cp.dcols = d_dcols.p;
        cp.dvals = d_dvals.p;
        cp.max_group_size = int32_t(max_gs);
        Stopwatch sw;
        sw.start();
        bool small_fit = false; // the whole pin solve ran in the single-workgroup kernel (its scalars are in d_sc)
        open_from_grad = grad_fresh && open_from_grad_opt && !is_glm() && !cov_mode && r_dev == d_r.p && !resume;
        grad_fresh = false; // (whatever engine runs, the residual moves)
        if (!(nv > 0 && panel_mode() && !all_scalar)) join_uv(); // (only the group panel passes know which of them need it)
        if (nv > 0 && panel_mode()) {
            spec_mode = resume ? 2 : 0;
            spec_active = false; // consumed (or never there)
            struct ModeGuard { int& m; ~ModeGuard() { m = 0; } } mode_guard{spec_mode};
            if (all_scalar) run_panel_passes(cp, sc, r_dev);
            else run_group_panel_passes(cp, sc, r_dev);
            // Gaussian: the residual is final and current on the device -> enqueue the invariance sweep of this lambda now,
            // so that it runs while the host does the post-fit bookkeeping below (otherwise the GPU idles ~0.2 ms per lambda)
            if (!is_glm() && sc.status == CD_OK && r_dev == d_r.p && prelaunch_sweep && inv_wanted) {
                launch_vmul<T>(d_w.p, d_r.p, d_v.p, n, st);
                t_sweep.begin(st);
                sweep(d_v.p, d_grad.p, nullptr, p, &d_blk.p->resid_sum, intercept ? d_xm.p : nullptr);
                t_sweep.end(st);
                device_abs_grad(lm, int(sc.active_size));
                grad_fresh = true;
                inv_prelaunched = true;
                inv_prelaunched_lm = lm;
            }
        } else if (nv > 0 && all_scalar && nv >= cd_block_min_nv) {
            run_block_passes(cp, sc);
        } else if (nv > 0 && !all_scalar && max_gs <= cd_block_size() && nv >= cd_block_min_nv) {
            run_group_block_passes(cp, sc);
        } else {
            if (nv > 0) {
                t_cd.begin(st);
                launch_cd<T>(cp, st);
                t_cd.end(st);
            }
            d_sc.download(&sc, 1, st);
            sync();
            small_fit = true;
        }
        const double t_cd = sw.elapsed();
        open_from_grad = false; // (only the panel engines take it)
        if (nv == 0) {
            // one (empty) active pass + one (empty) screen pass; their convergence measure is 0, so with a zero
            // tolerance (y_var == 0) the reference never leaves the loop and reports max_iters (pin_naive:317-357)
            if (!(T(0) < pin_tol)) throw max_cds_error(0);
            sc.status = CD_OK;
            sc.iters = 2;
        }
        cnt.n_cd_visits_screen += sc.n_visits_screen;
        cnt.n_cd_visits_active += sc.n_visits_active;
        cnt.n_updates += sc.n_updates;
        // in columns: exact for groups of one size (the mean size of the screened groups otherwise)
        cnt.n_update_cols += all_scalar ? sc.n_updates : int64_t(double(sc.n_updates) * double(nv) / double(std::max<idx>(ns, 1)) + 0.5);
        cnt.n_cd_passes_screen += sc.n_passes_screen;
        cnt.n_cd_passes_active += sc.n_passes_active;
        for (int i = 0; i < 8; ++i) cd_dbg[i] += sc.dbg[i];
        if (sc.status != CD_OK) {
            // restore the pre-fit invariants (solver_gaussian_naive.hpp:286-290,326-329)
            AHIP_CHECK(hipMemcpyAsync(d_beta.p, d_beta0.p, size_t(nv) * sizeof(T), hipMemcpyDeviceToDevice, st));
            d_isact.upload(screen_is_active.data(), screen_is_active.size(), st);
            actcols_key = -1; // (the failed fit may have appended to the device's active list: nothing cached about it survives)
            ptab_act.count = -1;
            sync();
            if (sc.status == CD_MAX_CDS) throw max_cds_error(0);
            if (sc.status == CD_MAX_ACTIVE) throw make_solver_error("Maximum number of active groups reached.");
            throw make_solver_error("Newton-ABS max iterations reached! Try increasing newton_max_iters.");
        }
        // residual update r -= X_S (beta - beta0), once per fit (the covariance method has no residual: its invariant, the
        // gradient, is recomputed from v and A by update_invariance)
        if (sc.n_delta > 0 && !cov_mode) {
            t_axpy.begin(st);
            axpy_cols(d_dcols.p, d_dvals.p, &d_sc.p->n_delta, 0, T(-1), r_dev);
            t_axpy.end(st);
            cnt.n_resid_col_reads += sc.n_delta;
        }
        // small screen sets (single-workgroup kernel): the invariance sweep of this lambda goes out right behind the residual
        // update as well, ahead of the downloads and the host bookkeeping below (the panel engines did this above)
        if (small_fit && !is_glm() && !cov_mode && prelaunch_sweep && inv_wanted && sc.status == CD_OK && r_dev == d_r.p &&
            !multi()) {
            launch_vmul<T>(d_w.p, d_r.p, d_v.p, n, st);
            t_sweep.begin(st);
            sweep(d_v.p, d_grad.p, nullptr, p, &d_sc.p->resid_sum, intercept ? d_xm.p : nullptr);
            t_sweep.end(st);
            device_abs_grad(lm, int(sc.active_size));
            grad_fresh = true;
            inv_prelaunched = true;
            inv_prelaunched_lm = lm;
        }
        grad_valid = false;
        // host mirrors
        const size_t old_active = active_set_size;
        active_set_size = size_t(sc.active_size);
        rsum_io = sc.resid_sum;
        o.rsq = sc.rsq;
        d_beta.download(screen_beta.data(), size_t(nv), st);
        std::vector<int32_t> act(active_set_size > old_active ? active_set_size - old_active : 0);
        if (!act.empty()) d_actset.download(act.data(), act.size(), st, old_active);
        // everything this fit hands back is enqueued; behind it, the first active pass of the next lambda (see spec_enabled)
        bool waited = false;
        if (spec_enabled && spec_next_lm > T(0) && inv_prelaunched && inv_prelaunched_lm == lm && !cons_on && nv > 0 &&
            panel_mode() && r_dev == d_r.p) {
            if (!spec_ev) AHIP_CHECK(hipEventCreateWithFlags(&spec_ev, hipEventDisableTiming));
            AHIP_CHECK(hipEventRecord(spec_ev, st));
            const size_t stage_mark = stage.mark();
            if (!all_scalar) { // the group engine partitions the active list on the host: it needs the newcomers first
                AHIP_CHECK(hipEventSynchronize(spec_ev));
                stage.flush();
                for (size_t i = 0; i < act.size(); ++i) {
                    active_set[old_active + i] = act[i];
                    screen_is_active[act[i]] = 1;
                }
                act.clear();
            }
            d_r_snap.reserve(size_t(n));
            AHIP_CHECK(hipMemcpyAsync(d_beta0.p, d_beta.p, size_t(nv) * sizeof(T), hipMemcpyDeviceToDevice, st));
            AHIP_CHECK(hipMemcpyAsync(d_r_snap.p, d_r.p, size_t(n) * sizeof(T), hipMemcpyDeviceToDevice, st));
            CdParams<T> cp2 = cp;
            cp2.lmda = spec_next_lm;
            CdScalars<T> sc2{};
            sc2.rsq = sc.rsq;
            sc2.resid_sum = sc.resid_sum;
            sc2.active_size = sc.active_size;
            spec_mode = 1;
            open_from_grad = grad_fresh && open_from_grad_opt; // (the sweep of this lambda went out just above, on the residual the pass starts from)
            spec_used_grad = grad_fresh;
            grad_fresh = false;
            {
                struct ModeGuard { int& m; ~ModeGuard() { m = 0; } } mode_guard{spec_mode};
                if (all_scalar) run_panel_passes(cp2, sc2, r_dev);
                else run_group_panel_passes(cp2, sc2, r_dev);
            }
            if (spec_enqueued) {
                spec_active = true;
                spec_lm = spec_next_lm;
                spec_nv = nv;
                spec_asz = active_set_size;
                ++n_spec;
            }
            AHIP_CHECK(hipEventSynchronize(spec_ev));
            stage.flush(); // every staged download of this fit was enqueued ahead of the event
            stage.release(stage_mark);
            waited = true;
        }
        if (!waited) sync();
        for (size_t i = 0; i < act.size(); ++i) {
            active_set[old_active + i] = act[i];
            screen_is_active[act[i]] = 1;
        }
        // pin_naive:359-394: active groups sorted by design column.  The active list only ever grows by appending, so the
        // sorted order is kept across fits and the newcomers are merged in (O(a + m log m) instead of a full sort per fit).
        {
            auto by_col = [&](idx i, idx j) { return groups[screen_set[active_set[i]]] < groups[screen_set[active_set[j]]]; };
            if (active_order.size() > active_set_size) active_order.clear();
            const size_t have = active_order.size();
            if (have < active_set_size) {
                std::vector<idx> fresh(active_set_size - have);
                std::iota(fresh.begin(), fresh.end(), idx(have));
                std::sort(fresh.begin(), fresh.end(), by_col);
                std::vector<idx> merged(active_set_size);
                std::merge(active_order.begin(), active_order.end(), fresh.begin(), fresh.end(), merged.begin(), by_col);
                active_order.swap(merged);
            }
        }
        const std::vector<idx>& order = active_order;
        o.beta_idx.reserve(size_t(nv));
        o.beta_val.reserve(size_t(nv));
        for (size_t i = 0; i < order.size(); ++i) {
            const idx ss = active_set[order[i]], g = screen_set[ss];
            for (idx t = 0; t < group_sizes[g]; ++t) {
                o.beta_idx.push_back(groups[g] + t);
                o.beta_val.push_back(screen_beta[screen_begins[ss] + t]);
            }
        }
        o.intercept = T(intercept) * (y_mean_pin + rsum_io);
        // the single kernel interleaves active and screen passes; split the wall time by visit counts
        const double va = double(sc.n_visits_active), vs = double(sc.n_visits_screen);
        o.t_active = (va + vs) > 0 ? t_cd * va / (va + vs) : 0;
        o.t_screen = t_cd - o.t_active;
        return o;
    }

    // gradient of the screen values into d_g
    void load_screen_gradient(const T* w_dev, const T* r_dev, const T* rsum_dev) {
        if (nv == 0) return;
        if (grad_valid) {
            launch_gather<T>(d_grad.p, d_vcol.p, nv, d_g.p, st);
        } else {
            launch_vmul<T>(w_dev, r_dev, d_v.p, n, st);
            sweep(d_v.p, d_g.p, d_vcol.p, nv, rsum_dev, intercept ? d_sxm_by_value() : nullptr);
        }
    }
    // the sweep epilogue indexes sub_vec by design column -> use the by-column means
    const T* d_sxm_by_value() const { return d_xm.p; }

    // gaussian::cov::fit, solver_gaussian_cov.hpp:232-357.  The reference's pin solver keeps `screen_grad` current with one
    // A.bmul per coordinate update; here the screen gradient of every fit is read from the full gradient of the last
    // invariance step (the same numbers in exact arithmetic) and the Gram kernels keep it current inside the fit.
    FitOut<T> cov_fit(T lm) {
        if (nv > 0) launch_gather<T>(d_grad.p, d_vcol.p, nv, d_g.p, st);
        T rsum = 0;
        FitOut<T> o = pin_solve(lm, tol, rsq, rsum, T(0), nullptr);
        rsq = o.rsq;
        return o;
    }

    // gaussian::naive::fit, solver_gaussian_naive.hpp:209-349
    FitOut<T> gaussian_fit(T lm) {
        // device scalar for the sweep epilogue
        CdScalars<T> sc{};
        sc.resid_sum = resid_sum;
        d_sc.upload(&sc, 1, st);
        cur_w = d_w.p;
        cur_xm = d_xm.p;
        if (!panel_mode()) load_screen_gradient(d_w.p, d_r.p, &d_sc.p->resid_sum);
        T rsum = resid_sum;
        FitOut<T> o = pin_solve(lm, tol * y_var, rsq, rsum, y_mean, d_r.p);
        resid_sum = rsum;
        rsq = o.rsq;
        return o;
    }

    // ---------------------------------------------------------------------------------------------------------
    // GLM: glm::naive::fit (IRLS), solver_glm_naive.hpp:234-459
    std::vector<T> irls_xm_host; // X_means under the IRLS weights (screen columns only), by design column
    DevBuf<T> d_irls_xm, d_irls_w;

    T device_scalar(const T* dptr) {
        T h;
        AHIP_CHECK(hipMemcpyAsync(&h, dptr, sizeof(T), hipMemcpyDeviceToHost, st));
        sync();
        return h;
    }

    // ---- GlmBase members: device kernels for the built-in families, host callbacks for a user-defined one ----
    bool glm_is_cb() const { return glm_kind == ADELIE_HIP_GLM_CALLBACK; }
    void cb_fetch(const T* dev, std::vector<T>& host) {
        host.resize(size_t(n));
        AHIP_CHECK(hipMemcpyAsync(host.data(), dev, size_t(n) * sizeof(T), hipMemcpyDeviceToHost, st));
    }
    void cb_store(const std::vector<T>& host, T* dev) {
        AHIP_CHECK(hipMemcpyAsync(dev, host.data(), size_t(n) * sizeof(T), hipMemcpyHostToDevice, st));
        sync(); // the host vector is reused by the next callback
    }
    // resid = glm.gradient(eta)
    void glm_gradient_dev(const T* eta_dev, T* r_dev) {
        if (!glm_is_cb()) {
            launch_glm_gradient<T>(glm_kind, d_y.p, d_gw.p, eta_dev, n, r_dev, st, mk());
            return;
        }
        cb_fetch(eta_dev, cb_eta);
        sync();
        cb_grad.resize(size_t(n));
        if (glm_cb.gradient(glm_cb.user, cb_eta.data(), cb_grad.data())) throw make_solver_error("glm.gradient() raised.");
        cb_store(cb_grad, r_dev);
    }
    // glm.loss(eta)
    T glm_loss_dev(const T* eta_dev) {
        if (!glm_is_cb()) {
            launch_glm_loss<T>(glm_kind, d_y.p, d_gw.p, eta_dev, n, d_sums.p, st, mk());
            return device_scalar(d_sums.p);
        }
        cb_fetch(eta_dev, cb_eta);
        sync();
        double l = 0;
        if (glm_cb.loss(glm_cb.user, cb_eta.data(), &l)) throw make_solver_error("glm.loss() raised.");
        return T(l);
    }
    // user-defined GLM: hess_dev = glm.hessian(eta, resid), z_dev = glm.inv_hessian_gradient(eta, resid, hess), which the
    // CALLBACK branch of the IRLS kernels reads instead of evaluating a built-in family
    void glm_hessian_cb(const T* eta_dev, const T* r_dev, T* hess_dev, T* z_dev) {
        cb_fetch(eta_dev, cb_eta);
        cb_fetch(r_dev, cb_grad);
        sync();
        cb_hess.resize(size_t(n));
        cb_z.resize(size_t(n));
        if (glm_cb.hessian(glm_cb.user, cb_eta.data(), cb_grad.data(), cb_hess.data(), cb_z.data()))
            throw make_solver_error("glm.hessian() raised.");
        cb_store(cb_hess, hess_dev);
        cb_store(cb_z, z_dev);
    }

    FitOut<T> glm_fit(T lm) {
        FitOut<T> o;
        size_t irls_it = 0;
        const T hmin = T(g_hessian_min);
        irls_xm_host.assign(p, 0);
        while (1) {
            if (irls_it >= irls_max_iters) throw make_solver_error("Maximum IRLS iterations reached.");
            ++cnt.n_irls_iters;
            cnt.n_irls_screen_cols += nv;
            Stopwatch sw_irls;
            sw_irls.start();
            // :336-348
            T sums[4];
            if (glm_is_cb()) glm_hessian_cb(d_eta.p, d_r.p, d_hess.p, d_irls_resid.p);
            launch_irls_prepare<T>(glm_kind, d_y.p, d_gw.p, d_eta.p, d_r.p, d_off.p, hmin, n, d_hess.p, d_irls_resid.p,
                                   d_irls_y.p, d_sums.p, st, mk());
            d_sums.download(sums, 1, st);
            sync();
            const T hess_sum = sums[0];
            launch_irls_weights<T>(d_hess.p, hess_sum, d_irls_y.p, T(0), n, d_irls_w.p, d_irls_resid.p, d_sums.p, st);
            d_sums.download(sums, 3, st);
            sync();
            const T ym = sums[0];
            T rsum;
            if (intercept) {
                const T shift = beta0 - ym;
                launch_irls_weights<T>(d_hess.p, hess_sum, d_irls_y.p, shift, n, d_irls_w.p, d_irls_resid.p, d_sums.p, st);
                d_sums.download(sums, 3, st);
                sync();
            }
            rsum = sums[2];
            T lmda_adj = lm / hess_sum;
            if (std::isinf(lmda_adj)) {
                if (lm == std::numeric_limits<T>::max()) lmda_adj = lm;
                else
                    throw make_solver_error(
                        "IRLS lambda is unexpectedly inf. This likely indicates a bug in the code. Please report this!");
            }
            // :361-385  X_means on the screen columns and all screen-derived quantities under the IRLS weights
            if (nv > 0) {
                if (multi()) { // the view's sweep covers all columns in one pass over X; pick the screen values out of it
                    d_mxm.reserve(size_t(p));
                    sweep(d_irls_w.p, d_mxm.p, nullptr, p, nullptr, nullptr);
                    launch_gather<T>(d_mxm.p, d_vcol.p, nv, d_g.p, st);
                } else {
                    sweep(d_irls_w.p, d_g.p, d_vcol.p, nv, nullptr, nullptr); // means by value
                }
                std::vector<T> m(nv);
                d_g.download(m.data(), size_t(nv), st);
                T drift = T(1e30);
                const bool track = irls_reuse > 0 && all_scalar && panel_mode();
                if (track) { // how far the weights moved since the previous iteration (and keep a copy for the next one)
                    d_irls_w_prev.reserve(size_t(n));
                    if (!irls_w_prev_valid) AHIP_CHECK(hipMemsetAsync(d_irls_w_prev.p, 0, size_t(n) * sizeof(T), st));
                    launch_rel_change<T>(d_irls_w.p, d_irls_w_prev.p, n, d_sums.p + 15, st);
                    AHIP_CHECK(hipMemcpyAsync(&drift, d_sums.p + 15, sizeof(T), hipMemcpyDeviceToHost, st));
                }
                sync();
                for (idx ss = 0; ss < idx(screen_set.size()); ++ss) {
                    const idx g = screen_set[ss];
                    for (idx t = 0; t < group_sizes[g]; ++t) irls_xm_host[groups[g] + t] = m[screen_begins[ss] + t];
                }
                d_irls_xm.upload(irls_xm_host.data(), size_t(p), st);
                gram_nv = 0;
                v_used = 0;
                screen_transforms.clear();
                ++w_version; // diagonal blocks built from here on belong to this iteration's weights
                if (track) {
                    note_weight_drift(irls_w_prev_valid ? double(drift) : 1e300);
                    irls_w_prev_valid = true;
                }
                if (panel_mode()) update_vars_panel(d_irls_w.p, d_irls_xm.p, irls_xm_host, 0);
                else update_gram_and_vars(d_irls_w.p, d_irls_xm.p, irls_xm_host, 0);
            }
            cur_w = d_irls_w.p;
            cur_xm = d_irls_xm.p;
            // gradient of the screen values for the working response
            CdScalars<T> sc{};
            sc.resid_sum = rsum;
            d_sc.upload(&sc, 1, st);
            grad_valid = false;
            if (nv > 0 && !panel_mode()) {
                launch_vmul<T>(d_irls_w.p, d_irls_resid.p, d_v.p, n, st);
                sweep(d_v.p, d_g.p, d_vcol.p, nv, &d_sc.p->resid_sum, intercept ? d_irls_xm.p : nullptr);
            }
            const T pin_tol = tol * (loss_null - loss_full) / hess_sum; // :407
            sync();
            t_host[6] += sw_irls.elapsed(); // IRLS set-up of the iteration (weights, means, screen-derived quantities)
            sw_irls.start();
            FitOut<T> po = pin_solve(lmda_adj, pin_tol, T(0), rsum, ym, d_irls_resid.p);
            t_host[7] += sw_irls.elapsed(); // the weighted least-squares pin solve
            o.t_screen += po.t_screen;
            o.t_active += po.t_active;
            beta0 = po.intercept;
            // :439-449
            std::swap(d_eta.p, d_eta_prev.p);
            std::swap(d_r.p, d_resid_prev.p);
            launch_irls_finish<T>(glm_kind, d_y.p, d_gw.p, d_irls_y.p, d_off.p, d_irls_resid.p,
                                  intercept ? (beta0 - ym) : T(0), n, d_eta.p, d_r.p, d_sums.p, st, mk());
            if (glm_is_cb()) glm_gradient_dev(d_eta.p, d_r.p);
            launch_dot_diff<T>(d_r.p, d_resid_prev.p, d_eta.p, d_eta_prev.p, n, d_sums.p, st);
            const T conv = device_scalar(d_sums.p);
            if (std::abs(conv) <= irls_tol) {
                o.beta_idx.swap(po.beta_idx);
                o.beta_val.swap(po.beta_val);
                o.intercept = po.intercept;
                o.rsq = po.rsq;
                return o;
            }
            ++irls_it;
        }
    }

    // update_loss_null, solver_glm_naive.hpp:160-232
    void update_loss_null() {
        if (multi() && D->micpt) { // solver_multiglm_naive.hpp:99-184: intercept-only model with one intercept per class
            const int64_t nb_ = D->nb;
            const int K_ = int(D->mK);
            DevBuf<T> e, r, e_prev, r_prev;
            e.reserve(n); r.reserve(n); e_prev.reserve(n); r_prev.reserve(n);
            AHIP_CHECK(hipMemcpyAsync(e.p, d_eta.p, n * sizeof(T), hipMemcpyDeviceToDevice, st));
            AHIP_CHECK(hipMemcpyAsync(r.p, d_r.p, n * sizeof(T), hipMemcpyDeviceToDevice, st));
            size_t it = 0;
            const T hmin = T(g_hessian_min);
            std::vector<T> b0(size_t(K_), T(0));
            while (1) {
                if (it >= irls_max_iters) throw make_solver_error("Maximum IRLS iterations reached.");
                // per class: sum of raised hessians and of hess * working response; the common 1 / hess_sum cancels
                for (int l = 0; l < K_; ++l) {
                    const int64_t o = int64_t(l) * nb_;
                    T sums[2];
                    launch_null_step<T>(glm_kind, d_y.p + o, d_gw.p + o, e.p + o, r.p + o, d_off.p + o, hmin, nb_, d_sums.p, st, K_);
                    d_sums.download(sums, 2, st);
                    sync();
                    b0[size_t(l)] = sums[1] / sums[0];
                }
                std::swap(e.p, e_prev.p);
                for (int l = 0; l < K_; ++l) {
                    const int64_t o = int64_t(l) * nb_;
                    launch_set_eta<T>(d_off.p + o, b0[size_t(l)], nb_, e.p + o, st);
                }
                std::swap(r.p, r_prev.p);
                launch_glm_gradient<T>(glm_kind, d_y.p, d_gw.p, e.p, n, r.p, st, K_);
                launch_dot_diff<T>(r.p, r_prev.p, e.p, e_prev.p, n, d_sums.p, st);
                const T conv = device_scalar(d_sums.p);
                if (std::abs(conv) <= irls_tol) {
                    launch_glm_loss<T>(glm_kind, d_y.p, d_gw.p, e.p, n, d_sums.p, st, K_);
                    loss_null = device_scalar(d_sums.p);
                    return;
                }
                ++it;
            }
        }
        if (!intercept) {
            loss_null = glm_loss_dev(d_off.p);
            return;
        }
        T b0 = beta0;
        DevBuf<T> e, r, e_prev, r_prev;
        e.reserve(n); r.reserve(n); e_prev.reserve(n); r_prev.reserve(n);
        AHIP_CHECK(hipMemcpyAsync(e.p, d_eta.p, n * sizeof(T), hipMemcpyDeviceToDevice, st));
        AHIP_CHECK(hipMemcpyAsync(r.p, d_r.p, n * sizeof(T), hipMemcpyDeviceToDevice, st));
        size_t it = 0;
        const T hmin = T(g_hessian_min);
        while (1) {
            if (it >= irls_max_iters) throw make_solver_error("Maximum IRLS iterations reached.");
            T sums[2];
            if (glm_is_cb()) glm_hessian_cb(e.p, r.p, d_hess.p, d_irls_y.p);
            launch_null_step<T>(glm_kind, d_y.p, d_gw.p, e.p, r.p, d_off.p, hmin, n, d_sums.p, st, mk(), d_hess.p, d_irls_y.p);
            d_sums.download(sums, 2, st);
            sync();
            b0 = sums[1] / sums[0];
            std::swap(e.p, e_prev.p);
            launch_set_eta<T>(d_off.p, b0, n, e.p, st);
            std::swap(r.p, r_prev.p);
            glm_gradient_dev(e.p, r.p);
            launch_dot_diff<T>(r.p, r_prev.p, e.p, e_prev.p, n, d_sums.p, st);
            const T conv = device_scalar(d_sums.p);
            if (std::abs(conv) <= irls_tol) {
                loss_null = glm_loss_dev(e.p);
                return;
            }
            ++it;
        }
    }

    // ---------------------------------------------------------------------------------------------------------
    static T compute_lmda_max(const Solver& s) { // solver/utils.hpp:7-23
        const T factor = (s.alpha <= 0) ? T(1e-3) : s.alpha;
        T mx = -std::numeric_limits<T>::infinity();
        for (idx i = 0; i < s.G; ++i) mx = std::max<T>(mx, (s.penalty[i] <= 0.0) ? T(0) : s.abs_grad[i] / s.penalty[i]);
        return mx / factor;
    }
    static void compute_lmda_path(std::vector<T>& path, T mr, T lmax) { // solver/utils.hpp:25-42
        const idx L = idx(path.size());
        if (L > 1) {
            const T log_factor = std::log(mr) / (L - 1);
            for (idx i = 0; i < L; ++i) path[i] = lmax * std::exp(log_factor * T(i));
        }
        path[0] = lmax;
    }

    bool is_glm() const { return glm_kind != ADELIE_HIP_GLM_GAUSSIAN; }

    // update_invariance_f: solver_gaussian_naive.hpp:377-393 / solver_glm_naive.hpp:495-503, + update_abs_grad
    bool inv_wanted = true; // set by solve(): the fit about to run is followed by update_invariance at the same lambda
    bool prelaunch_sweep = true, inv_prelaunched = false;
    T inv_prelaunched_lm = 0;
    // ---- speculative first active-set pass of the NEXT lambda (Gaussian lasso on the look-ahead panel engine) ----
    // Between the invariance sweep of lambda_k and the first kernel of the fit at lambda_{k+1} the host checks KKT, screens,
    // appends the new screen groups and computes their variances: 0.2-0.4 ms per lambda with the GPU idle.  The fit at
    // lambda_{k+1} always begins with a pass over the active set as lambda_k left it (pin_naive:173-215), which depends on
    // none of that host work, so it is enqueued right behind the sweep and the next fit picks its result up instead of
    // launching it.  Same operations in the same order: bit-identical paths.  If the next fit turns out to be something else
    // (KKT failed: refit at lambda_k; early exit; the caller reads the live state) the coefficients and the residual are put
    // back from the copies taken before the pass.
    bool spec_enabled = true;    // A/B hook ADELIE_HIP_SPECULATE=0
    T spec_next_lm = 0;          // set by solve() before a fit: the lambda that follows if KKT passes (0: none)
    bool spec_active = false;    // a speculative pass is in flight / done and not yet consumed
    T spec_lm = 0;
    idx spec_nv = 0;
    size_t spec_asz = 0;
    int spec_mode = 0;           // read by run_panel_passes: 1 = enqueue one active pass and return, 2 = its first pass is in flight
    bool spec_enqueued = false;
    int64_t spec_blocks = 0, spec_cols = 0, n_spec = 0, n_spec_rollback = 0;
    DevBuf<T> d_r_snap;
    hipEvent_t spec_ev = nullptr;
    void spec_rollback() {
        if (!spec_active) return;
        sync();
        AHIP_CHECK(hipMemcpyAsync(d_beta.p, d_beta0.p, size_t(spec_nv) * sizeof(T), hipMemcpyDeviceToDevice, st));
        AHIP_CHECK(hipMemcpyAsync(d_r.p, d_r_snap.p, size_t(n) * sizeof(T), hipMemcpyDeviceToDevice, st));
        sync();
        pending_slot = -1;
        cnt.n_panel_blocks -= spec_blocks;
        cnt.n_panel_cols -= spec_cols;
        spec_active = false;
        grad_fresh = spec_used_grad; // the residual is the one the sweep saw again: the refit opens as the pass taken back did
        ++n_spec_rollback;
    }
    void update_invariance(T lm) {
        lmda = lm;
        ++cnt.n_sweeps;
        if (inv_prelaunched) { // enqueued at the end of the fit (pin_solve); the fit's own synchronisation covered it
            inv_prelaunched = false;
            if (inv_prelaunched_lm == lm) {
                if (!spec_active) sync(); // (pin_solve waited for the downloads; a full sync would wait for the speculative pass)
                grad_valid = true;
                host_cons_abs_grad(lm);
                return;
            }
        }
        CdScalars<T> sc{};
        sc.resid_sum = resid_sum;
        d_sc.upload(&sc, 1, st);
        if (cov_mode) { // solver_gaussian_cov.hpp:392-418: grad = v - A beta over the non-zero coefficients
            if (nv > 0) {
                d_zero.reserve(size_t(nv));
                AHIP_CHECK(hipMemsetAsync(d_zero.p, 0, size_t(nv) * sizeof(T), st));
                launch_cd_compact<T>(d_beta.p, d_zero.p, d_vcol.p, int(nv), d_dcols.p, d_dvals.p, &d_sc.p->n_delta, st);
            }
            t_sweep.begin(st);
            launch_cov_grad<T>(static_cast<const T*>(D->X), D->ld, p, d_covv.p, d_dcols.p, d_dvals.p, &d_sc.p->n_delta, d_grad.p, st);
            t_sweep.end(st);
        } else if (is_glm()) {
            t_sweep.begin(st);
            sweep(d_r.p, d_grad.p, nullptr, p, nullptr, nullptr); // resid already carries the weights
            t_sweep.end(st);
        } else {
            launch_vmul<T>(d_w.p, d_r.p, d_v.p, n, st);
            t_sweep.begin(st);
            sweep(d_v.p, d_grad.p, nullptr, p, &d_sc.p->resid_sum, intercept ? d_xm.p : nullptr);
            t_sweep.end(st);
            grad_valid = true;
            grad_fresh = true;
        }
        device_abs_grad(lm, int(active_set_size));
        sync();
        host_cons_abs_grad(lm);
    }

    void update_solutions(FitOut<T>& fo, T lm) {
        betas_idx.emplace_back(std::move(fo.beta_idx));
        betas_val.emplace_back(std::move(fo.beta_val));
        intercepts.push_back(fo.intercept);
        lmdas.push_back(lm);
        if (cons_on) { // sparsify_dual, solver_base.hpp:158-222: the non-zero multipliers of every constraint, screened or not
            refresh_screen_multipliers();
            std::vector<idx> di;
            std::vector<T> dv;
            std::vector<double> mu_obj;
            for (idx g = 0; g < G; ++g) {
                if (!cons_kind[g]) continue;
                if (host_cons(g)) { // the object's own multipliers
                    mu_obj.assign(size_t(cons_m[g]), 0.0);
                    if (cons_m[g] > 0 && cons_cb->dual(cons_cb->user, g, cons_m[g], mu_obj.data()))
                        throw make_solver_error("constraint.dual() raised.");
                    for (idx t = 0; t < cons_m[g]; ++t)
                        if (mu_obj[size_t(t)] != 0) { di.push_back(dual_groups[g] + t); dv.push_back(T(mu_obj[size_t(t)])); }
                } else if (cons_mu[g] != 0) {
                    di.push_back(dual_groups[g]);
                    dv.push_back(cons_dual_of(g));
                }
            }
            duals_idx.emplace_back(std::move(di));
            duals_val.emplace_back(std::move(dv));
        } else {
            duals_idx.emplace_back();
            duals_val.emplace_back();
        }
        if (cov_mode) { // solver_gaussian_cov.hpp:203-229: the deviance is rsq itself (the saturated loss is unknown)
            devs.push_back(fo.rsq);
        } else if (is_glm()) { // solver_glm_naive.hpp:153-157
            const T loss = glm_loss_dev(d_eta.p);
            devs.push_back((loss_null - loss) / (loss_null - loss_full));
        } else {
            devs.push_back(fo.rsq / y_var);
        }
    }

    bool early_exit_f() {
        const bool ee = early_exit();
        const bool ec = poll && poll(poll_user, 1, int64_t(lmdas.size()), live);
        return ee || ec;
    }

    void screen_f(T lm, bool kkt_passed, int n_new_active) {
        Stopwatch sw;
        sw.start();
        screen(lm, kkt_passed, n_new_active);
        ++n_host_screens;
        t_host[0] += sw.elapsed();
        sw.start();
        if (is_glm()) {
            update_screen_derived_base();
            device_append_screen();
            t_host[1] += sw.elapsed();
        } else {
            const size_t old_groups = screen_transforms.size();
            update_screen_derived_base();
            device_append_screen();
            t_host[1] += sw.elapsed();
            sw.start();
            if (panel_mode()) update_vars_panel(d_w.p, d_xm.p, X_means, old_groups);
            else update_gram_and_vars(d_w.p, d_xm.p, X_means, old_groups);
            t_host[2] += sw.elapsed();
        }
    }

    FitOut<T> fit_f(T lm) {
        Stopwatch sw;
        sw.start();
        FitOut<T> o = cov_mode ? cov_fit(lm) : is_glm() ? glm_fit(lm) : gaussian_fit(lm);
        t_host[3] += sw.elapsed();
        return o;
    }

    // solve_core, solver_base.hpp:435-687
    void solve() {
        if (screen_set.size() > max_screen_size) throw max_screen_set_error();
        if (is_glm() && setup_loss_null) update_loss_null();

        if (setup_lmda_max) { // :500-515
            T pmax = -std::numeric_limits<T>::infinity();
            for (idx i = 0; i < G; ++i) pmax = std::max(pmax, penalty[i]);
            const T large_lmda = T(1e-3) * std::numeric_limits<T>::max() / std::max<T>(1, pmax);
            fit_f(large_lmda);
            update_invariance(large_lmda);
            lmda_max = compute_lmda_max(*this);
        }
        if (setup_lmda_path) { // :520-526
            if (lmda_path_size <= 0) return;
            lmda_path.resize(lmda_path_size);
            compute_lmda_path(lmda_path, min_ratio, lmda_max);
        }
        const size_t L = lmda_path.size();
        size_t pb_it = 0, large_sz = 0;
        while (large_sz < L && !(lmda_path[large_sz] <= lmda_max)) ++large_sz;
        if (large_sz || setup_lmda_max) { // :553-591
            std::vector<T> large(large_sz + 1);
            for (size_t i = 0; i < large_sz; ++i) large[i] = lmda_path[i];
            large[large_sz] = lmda_max;
            for (size_t i = 0; i < large.size(); ++i) {
                inv_wanted = i + 1 == large.size(); // the solutions above lambda_max are saved without an invariance step
                auto fo = fit_f(large[i]);
                inv_wanted = true;
                if (i < large.size() - 1) {
                    update_solutions(fo, large[i]);
                    ++pb_it;
                    if (early_exit_f()) return;
                } else {
                    update_invariance(large[i]);
                }
            }
        }
        size_t lmda_path_idx = large_sz;
        int current_active_size = int(active_set_size);
        bool kkt_passed = true;
        int n_new_active = 0;
        Stopwatch sw;
        for (; pb_it < L; ++pb_it) { // :605-686
            const T lmda_curr = lmda_path[lmda_path_idx];
            while (1) {
                ++cnt.n_basil_iters;
                sw.start();
                const double sync0 = t_sync_total;
                screen_f(lmda_curr, kkt_passed, n_new_active);
                benchmark_screen.push_back(sw.elapsed());
                t_host_screen += benchmark_screen.back();
                t_host_screen_wait += t_sync_total - sync0;
                spec_next_lm = (lmda_path_idx + 1 < L) ? lmda_path[lmda_path_idx + 1] : T(0);
                auto fo = fit_f(lmda_curr);
                spec_next_lm = T(0);
                benchmark_fit_screen.push_back(fo.t_screen);
                benchmark_fit_active.push_back(fo.t_active);
                sw.start();
                update_invariance(lmda_curr);
                benchmark_invariance.push_back(sw.elapsed());
                t_host[4] += benchmark_invariance.back();
                sw.start();
                kkt_passed = kkt(lmda_curr);
                n_valid_solutions.push_back(kkt_passed);
                lmda_path_idx += kkt_passed;
                if (kkt_passed) update_solutions(fo, lmda_curr);
                benchmark_kkt.push_back(sw.elapsed());
                t_host[5] += benchmark_kkt.back();
                if (kkt_passed) {
                    active_sizes.push_back(int(active_set_size));
                    screen_sizes.push_back(int(screen_set.size()));
                }
                n_new_active = kkt_passed ? (active_sizes.back() - current_active_size) : n_new_active;
                current_active_size = kkt_passed ? active_sizes.back() : current_active_size;
                if (kkt_passed) break;
            }
            if (early_exit_f()) break;
        }
    }

    // pull the device-resident invariants back into the host mirrors that the result accessors expose
    void finalize() {
        if (hooks.trace >= 2)
            std::fprintf(stderr, "[enq] panel passes: host enqueue %.1f ms, host wait %.1f ms, blocks %lld (built %lld + %lld cross + %lld strips, reused across IRLS iterations %lld), speculated %lld (rolled back %lld)\n",
                         t_enq * 1e3, t_wait * 1e3, (long long)cnt.n_panel_blocks, (long long)cnt.n_panel_grams, (long long)n_cross_blocks, (long long)n_strip_builds, (long long)n_blocks_reused, (long long)n_spec, (long long)n_spec_rollback);
        if (hooks.trace >= 2)
            std::fprintf(stderr, "[alloc] hipMalloc/hipFree so far in this process: %ld calls, %.1f ms\n", DevAllocStats::calls(),
                         DevAllocStats::seconds() * 1e3);
        t_sweep.collect(); t_gram.collect(); t_cd.collect(); t_axpy.collect(); t_step.collect();
        if (d_grp_dbg.p) {
            sync();
            d_grp_dbg.download(cd_dbg, 8, st);
            sync();
        }
        if (hooks.trace >= 2) {
            for (size_t i = 0; i < gram_shapes.size() && i < t_gram.each.size(); ++i) {
                const double fl = 2.0 * double(n) * double(gram_shapes[i].first) * double(gram_shapes[i].second);
                std::fprintf(stderr, "gram M=%lld N=%lld ms=%.3f TF=%.1f\n", (long long)gram_shapes[i].first,
                             (long long)gram_shapes[i].second, t_gram.each[i], fl / (t_gram.each[i] * 1e-3) / 1e12);
            }
        }
        download_invariants();
    }

    // multipliers of the screened coordinates as their last visits left them (device) -> host mirror
    std::vector<T> cmu_stage;
    void refresh_screen_multipliers() {
        if (!cons_on || nv <= 0) return;
        cmu_stage.resize(size_t(nv));
        d_cmu.download(cmu_stage.data(), size_t(nv), st);
        sync();
        for (size_t ss = 0; ss < screen_set.size(); ++ss) // a constrained group has one coefficient: its screen value
            if (cons_kind[screen_set[ss]] && !host_cons(screen_set[ss])) cons_mu[screen_set[ss]] = cmu_stage[size_t(screen_begins[ss])];
    }

    // host mirrors of the device-resident invariants (grad, resid, eta, screen_beta, screen_X_means, screen_vars); also what
    // adelie_hip_result_sync does for the live state inside a poll callback
    void download_invariants() {
        spec_rollback();
        d_grad.download(grad.data(), size_t(p), st);
        if (!cov_mode) d_r.download(resid.data(), size_t(n), st);
        if (is_glm()) d_eta.download(eta.data(), size_t(n), st);
        if (nv > 0) {
            d_beta.download(screen_beta.data(), size_t(nv), st);
            screen_X_means.resize(nv);
            screen_vars.resize(nv);
            d_sxm.download(screen_X_means.data(), size_t(nv), st);
            d_vars.download(screen_vars.data(), size_t(nv), st);
        }
        std::vector<T> v_host;
        if (host_mirrors_stale && v_used > 0) {
            v_host.resize(v_used);
            d_V.download(v_host.data(), v_used, st);
        }
        sync();
        if (host_mirrors_stale) { // eigenbases computed on the device: (1) for single coefficients, a slice of d_V otherwise
            screen_transforms.resize(screen_set.size());
            for (size_t ss = 0; ss < screen_set.size(); ++ss) {
                const size_t q = size_t(group_sizes[screen_set[ss]]);
                if (q == 1) screen_transforms[ss] = std::vector<T>{T(1)};
                else if (ss < h_voff.size() && size_t(h_voff[ss]) + q * q <= v_host.size())
                    screen_transforms[ss].assign(v_host.begin() + h_voff[ss], v_host.begin() + h_voff[ss] + q * q);
            }
            host_mirrors_stale = false;
        }
        if (multi()) { // back to the ABI's (n, K) row-major layout
            std::vector<T> tmp(resid);
            from_major(tmp.data(), resid.data());
            if (is_glm()) {
                tmp = eta;
                from_major(tmp.data(), eta.data());
            }
        }
    }

    // ---------------------------------------------------------------------------------------------------------
    void build(adelie_hip_design* X, const adelie_hip_grpnet_args* a) {
        D = X;
        st = X->stream;
        n = X->n; p = X->p; G = a->G;
        cov_mode = X->cov != 0;
        if (G <= 0) throw make_core_error("groups must be non-empty.");
        groups.assign(a->groups, a->groups + G);
        group_sizes.assign(a->group_sizes, a->group_sizes + G);
        penalty.assign((const T*)a->penalty, (const T*)a->penalty + G);
        alpha = T(a->alpha); min_ratio = T(a->min_ratio);
        lmda_path_size = size_t(a->lmda_path_size);
        max_screen_size = size_t(a->max_screen_size); max_active_size = size_t(a->max_active_size);
        pivot_subset_ratio = T(a->pivot_subset_ratio); pivot_subset_min = size_t(a->pivot_subset_min);
        pivot_slack_ratio = T(a->pivot_slack_ratio); screen_rule = a->screen_rule;
        max_iters = size_t(a->max_iters); tol = T(a->tol); adev_tol = T(a->adev_tol); ddev_tol = T(a->ddev_tol);
        newton_tol = T(a->newton_tol); newton_max_iters = size_t(a->newton_max_iters);
        early_exit_ = a->early_exit; setup_lmda_max = a->setup_lmda_max; setup_lmda_path = a->setup_lmda_path;
        intercept = a->intercept; glm_kind = a->glm_kind;
        poll = a->poll; poll_user = a->poll_user;
        if (glm_kind == ADELIE_HIP_GLM_CALLBACK) {
            if (!a->glm_cb || !a->glm_cb->gradient || !a->glm_cb->hessian || !a->glm_cb->loss)
                throw make_core_error("glm_cb with gradient, hessian and loss is required for a user-defined GLM.");
            glm_cb = *a->glm_cb;
        }
        lmda_max = T(a->lmda_max);
        if (a->lmda_path && a->n_lmda_path > 0) lmda_path.assign((const T*)a->lmda_path, (const T*)a->lmda_path + a->n_lmda_path);
        screen_set.assign(a->screen_set, a->screen_set + a->screen_set_size);
        screen_beta.assign((const T*)a->screen_beta, (const T*)a->screen_beta + a->screen_beta_size);
        screen_is_active.assign(a->screen_is_active, a->screen_is_active + a->screen_set_size);
        active_set_size = size_t(a->active_set_size);
        active_set.assign(a->active_set, a->active_set + G);
        lmda = T(a->lmda);
        grad.assign((const T*)a->grad, (const T*)a->grad + p);
        abs_grad.assign(G, 0);
        for (idx g = 0; g < G; ++g) {
            max_gs = std::max(max_gs, group_sizes[g]);
            if (group_sizes[g] != 1) all_scalar = false;
        }

        // state_base.ipp:9-116
        if (alpha < 0 || alpha > 1) throw make_core_error("alpha must be in [0,1].");
        if (tol < 0) throw make_core_error("tol must be >= 0.");
        if (adev_tol < 0 || adev_tol > 1) throw make_core_error("adev_tol must be in [0,1].");
        if (ddev_tol < 0 || ddev_tol > 1) throw make_core_error("ddev_tol must be in [0,1].");
        if (newton_tol < 0) throw make_core_error("newton_tol must be >= 0.");
        if (a->n_threads < 1) throw make_core_error("n_threads must be >= 1.");
        if (min_ratio < 0 || min_ratio > 1) throw make_core_error("min_ratio must be in [0,1].");
        if (pivot_subset_ratio <= 0 || pivot_subset_ratio > 1) throw make_core_error("pivot_subset_ratio must be in (0,1].");
        if (pivot_subset_min < 1) throw make_core_error("pivot_subset_min must be >= 1.");
        if (pivot_slack_ratio < 0) throw make_core_error("pivot_slack_ratio must be >= 0.");
        if (screen_beta.size() < screen_set.size())
            throw make_core_error(
                "screen_beta must be (bs,) where bs >= s and screen_set is (s,). "
                "It is likely screen_beta has been initialized incorrectly. ");
        if (active_set_size > size_t(G)) throw make_core_error("active_set_size must be <= G where groups is (G,).");
        if (p != groups[G - 1] + group_sizes[G - 1])
            throw make_core_error(
                "grad.size() != groups[G-1] + group_sizes[G-1]. "
                "It is likely either grad has the wrong shape, "
                "or groups/group_sizes have been initialized incorrectly.");
        for (idx i : screen_set)
            if (i < 0 || i >= G) throw make_core_error("screen_set contains an out-of-range group index.");

        AHIP_CHECK(hipSetDevice(X->device));
        hooks = Hooks::from_env(); // (common.hpp: the library's seven environment hooks)
        if (hooks.cd_block_min_nv >= 0) cd_block_min_nv = hooks.cd_block_min_nv;
        time_panel = hooks.time_panel;
        // two build streams under IRLS (config 4: 8.2 -> 7.2 s; three or four are no better), one under fixed weights (the
        // few builds of a Gaussian path only add contention for the look-ahead launches: 3.13 vs 3.08 paths/s)
        n_side = is_glm() ? 2 : 1;
        if (hooks.lookahead >= 0) lookahead = hooks.lookahead != 0;
        fuse_reduce = !multi() && fused_partials() <= 200;
        if (hooks.speculate >= 0) spec_enabled = hooks.speculate != 0;
        if (hooks.irls_reuse >= 0) irls_reuse = hooks.irls_reuse;
        panel_bsz = hooks.panel_bsz;
        if (cov_mode) { // base state of the covariance method: no intercept, adev_tol = ddev_tol = 0 (state_gaussian_cov.hpp:118)
            engine_panel = false; // the panel engines work on the residual; the Gram engines on C = A[S, S] and its gradient
            glm_kind = ADELIE_HIP_GLM_GAUSSIAN;
            intercept = false;
            adev_tol = 0; ddev_tol = 0;
            rdev_tol = T(a->rdev_tol);
        }
        if (multi()) {
            // StateMultiGaussianNaive (state.py:2300-2380): the Gaussian naive solver, global intercept off, on the view.
            // Everything runs on the group panel engine (its blocks are what lets a column slice serve K responses).
            if (is_glm() && glm_kind != ADELIE_HIP_GLM_MULTINOMIAL)
                throw make_core_error("a multi-response view supports the multigaussian and multinomial families only.");
            if (glm_kind == ADELIE_HIP_GLM_MULTINOMIAL && D->mK < 2)
                throw make_core_error("y must have at least 2 columns (classes).");
            if (intercept) throw make_core_error("a multi-response view is solved with intercept = false (the intercepts are its first K columns).");
            if (max_gs > idx(cd_block_size()))
                throw make_core_error("multi-response groups (group size x K) must not exceed " + std::to_string(cd_block_size()) + " columns.");
            all_scalar = false;
            engine_panel = true;
            group_panel = true;
            cd_block_min_nv = 0;
        } else if (glm_kind == ADELIE_HIP_GLM_MULTINOMIAL) {
            throw make_core_error("the multinomial family needs a multi-response view as its design.");
        }
        if (a->constraint_kind) {
            bool any = false;
            for (idx g = 0; g < G; ++g) any = any || a->constraint_kind[g] != 0;
            if (any) {
                if (cov_mode) throw make_core_error("constraints are not implemented for the covariance method.");
                if (!all_scalar && max_gs > idx(cd_block_size()))
                    throw make_core_error("constraints are not implemented for problems with groups of more than " +
                                          std::to_string(cd_block_size()) + " coefficients.");
                if (!a->constraint_a || !a->constraint_b) throw make_core_error("constraint_a and constraint_b are required.");
                cons_m.assign(G, 0);
                for (idx g = 0; g < G; ++g)
                    if (a->constraint_kind[g] == ADELIE_HIP_CONSTRAINT_HOST) { cons_host = true; ++n_host_cons; }
                if (cons_host) {
                    if (!a->constraint_cb || !a->constraint_cb->solve || !a->constraint_cb->gradient ||
                        !a->constraint_cb->solve_zero || !a->constraint_cb->dual)
                        throw make_core_error("constraint_cb is required for host constraint objects.");
                    cons_cb = a->constraint_cb;
                    if (max_gs > idx(cd_block_size()))
                        throw make_core_error("constraints are not implemented for problems with groups of more than " +
                                              std::to_string(cd_block_size()) + " coefficients.");
                    all_scalar = false; // the group engine carries the host visits (it handles groups of one coefficient too)
                }
                const T* ca = static_cast<const T*>(a->constraint_a);
                const T* cb = static_cast<const T*>(a->constraint_b);
                const T* cm = static_cast<const T*>(a->constraint_mu);
                const T INF = std::numeric_limits<T>::infinity();
                cons_on = true;
                cons_kind.assign(a->constraint_kind, a->constraint_kind + G);
                cons_a.assign(ca, ca + G);
                cons_lo.assign(G, -INF);
                cons_hi.assign(G, INF);
                cons_mu.assign(G, 0);
                dual_groups.assign(G, 0);
                idx nd = 0;
                for (idx g = 0; g < G; ++g) {
                    dual_groups[g] = nd;
                    const int32_t kd = cons_kind[g];
                    if (!kd) continue;
                    if (kd == ADELIE_HIP_CONSTRAINT_HOST) {
                        cons_m[g] = a->constraint_duals ? a->constraint_duals[g] : group_sizes[g];
                        if (cons_m[g] < 0) throw make_core_error("constraint_duals must be >= 0.");
                        nd += cons_m[g];
                        continue;
                    }
                    if (group_sizes[g] != 1)
                        throw make_core_error("box / one-sided closed forms are for groups of one coefficient (pass the object as a host constraint).");
                    cons_m[g] = 1;
                    if (kd == 1) { // constraint_box.ipp:30-37
                        if (cb[g] < 0) throw make_core_error("upper must be >= 0.");
                        if (ca[g] > 0) throw make_core_error("lower must be <= 0.");
                        // the Python classes clamp absent sides to +-max_solver_value (1e100, configs.hpp:13), which is +-inf
                        // in f32 only: an absent side is +-INF here in either precision
                        cons_lo[g] = (ca[g] <= -T(1e100)) ? -INF : ca[g];
                        cons_hi[g] = (cb[g] >= T(1e100)) ? INF : cb[g];
                        if (cm) cons_mu[g] = cm[g];
                    } else if (kd == 2) { // constraint_one_sided.ipp:74-79: sgn * x <= b
                        if (std::abs(ca[g]) != 1) throw make_core_error("sgn must be a vector of +/-1.");
                        if (cb[g] < 0) throw make_core_error("b must be >= 0.");
                        if (cb[g] >= T(1e100)) { /* no bound on this side */ }
                        else if (ca[g] > 0) cons_hi[g] = cb[g];
                        else cons_lo[g] = -cb[g];
                        if (cm) cons_mu[g] = ca[g] * cm[g];
                    } else {
                        throw make_core_error("unknown constraint kind.");
                    }
                    ++nd;
                }
                // the clipped coordinate update lives in the panel solve (blk_solve_body<.., CONS>): that engine from the first
                // screened coefficient on, in its sequential form
                engine_panel = true;
                group_panel = true;
                cd_block_min_nv = 1;
                lookahead = false;
                d_clo_g.reserve(G); d_chi_g.reserve(G); d_mu_g.reserve(G);
                d_clo_g.upload(cons_lo.data(), size_t(G), st);
                d_chi_g.upload(cons_hi.data(), size_t(G), st);
                d_mu_g.upload(cons_mu.data(), size_t(G), st);
                d_clo.reserve(p); d_chi.reserve(p); d_cmu.reserve(p);
            }
        }
        // device allocations
        d_r.reserve(n); d_v.reserve(n); d_grad.reserve(p); d_absgrad.reserve(G); d_penalty.reserve(G);
        d_groups.reserve(G); d_gsizes.reserve(G); d_slot.reserve(G);
        d_vcol.reserve(p); d_sbegin.reserve(G); d_ssize.reserve(G); d_actset.reserve(G); d_dcols.reserve(p);
        d_spen.reserve(G); d_beta.reserve(p); d_beta0.reserve(p); d_g.reserve(p); d_vars.reserve(p); d_sxm.reserve(p);
        d_dvals.reserve(p); d_isact.reserve(G); d_voff.reserve(G); d_V.reserve(16); d_sc.reserve(1); d_sums.reserve(16 + 4 * 256);
        d_penalty.upload(penalty.data(), G, st);
        d_groups.upload(groups.data(), G, st);
        d_gsizes.upload(group_sizes.data(), G, st);
        AHIP_CHECK(hipMemsetAsync(d_slot.p, 0xFF, size_t(G) * sizeof(int32_t), st)); // -1
        AHIP_CHECK(hipMemsetAsync(d_voff.p, 0, size_t(G) * sizeof(idx), st));
        d_grad.upload(grad.data(), p, st);
        std::vector<int32_t> act32(G, 0);
        for (size_t i = 0; i < active_set_size; ++i) act32[i] = int32_t(active_set[i]);
        d_actset.upload(act32.data(), G, st);
        sync();

        update_screen_derived_base();
        update_abs_grad_host(lmda);
        host_cons_abs_grad(lmda);

        if (cov_mode) {
            if (!a->cov_v) throw make_core_error("v must be (p,) where A is (p, p).");
            d_covv.reserve(p); d_xm.reserve(p);
            d_covv.upload((const T*)a->cov_v, p, st);
            X_means.assign(size_t(p), T(0)); // no centring in the covariance method
            d_xm.upload(X_means.data(), p, st);
            rsq = T(a->rsq);
            sync();
            gaussian_update_screen_derived();
        } else if (!is_glm()) {
            // state_gaussian_naive.hpp:40-160
            const T* w = (const T*)a->weights;
            if (!w || !a->X_means || !a->resid) throw make_core_error("weights, X_means and resid are required.");
            d_w.reserve(n); d_xm.reserve(p);
            std::vector<T> w_major;
            if (multi()) {
                w_major.resize(size_t(n));
                to_major(w, w_major.data());
                const int64_t nb_ = D->nb;
                multi_w_uniform = true;
                for (int64_t l = 1; l < D->mK && multi_w_uniform; ++l)
                    multi_w_uniform = std::equal(w_major.begin(), w_major.begin() + nb_, w_major.begin() + l * nb_);
                w = w_major.data();
            }
            d_w.upload(w, n, st);
            X_means.assign((const T*)a->X_means, (const T*)a->X_means + p);
            d_xm.upload(X_means.data(), p, st);
            y_mean = T(a->y_mean); y_var = T(a->y_var);
            loss_null = -T(0.5) * y_mean * y_mean;
            loss_full = -T(0.5) * y_var + loss_null;
            rsq = T(a->rsq); resid_sum = T(a->resid_sum);
            resid.assign((const T*)a->resid, (const T*)a->resid + n);
            std::vector<T> r_major;
            if (multi()) {
                r_major.resize(size_t(n));
                to_major(resid.data(), r_major.data());
                d_r.upload(r_major.data(), n, st);
            } else {
                d_r.upload(resid.data(), n, st);
            }
            sync();
            grad_valid = true; // the caller's grad is X^T W r (and resid_sum*X_means is already folded in or zero)
            // (solver.py:891-904 passes the un-corrected gradient with resid_sum == 0 when intercept; a warm start
            //  passes the corrected invariant; in both cases grad equals the invariant the CD kernel needs.)
            gaussian_update_screen_derived();
        } else {
            // state_glm_naive.hpp:60-164
            if (a->irls_tol <= 0) throw make_core_error("irls_tol must be > 0.");
            if (!a->glm_y || !a->glm_weights || !a->offsets || !a->eta || !a->resid)
                throw make_core_error("glm_y, glm_weights, offsets, eta and resid are required.");
            d_y.reserve(n); d_gw.reserve(n); d_off.reserve(n); d_eta.reserve(n); d_hess.reserve(n); d_irls_y.reserve(n);
            d_irls_resid.reserve(n); d_eta_prev.reserve(n); d_resid_prev.reserve(n); d_irls_w.reserve(n); d_irls_xm.reserve(p);
            d_xm.reserve(p);
            eta.assign((const T*)a->eta, (const T*)a->eta + n);
            resid.assign((const T*)a->resid, (const T*)a->resid + n);
            std::vector<T> stage;
            if (multi()) {
                // response-major device layout; glm_weights is (n,): repeated per class so that the elementwise kernels index it
                // like every other vector (the multinomial kernels read its first segment)
                const size_t nb_ = size_t(D->nb), K_ = size_t(D->mK);
                stage.resize(5 * size_t(n));
                to_major((const T*)a->glm_y, stage.data());
                for (size_t l = 0; l < K_; ++l) std::copy((const T*)a->glm_weights, (const T*)a->glm_weights + nb_, stage.data() + size_t(n) + l * nb_);
                to_major((const T*)a->offsets, stage.data() + 2 * size_t(n));
                to_major(eta.data(), stage.data() + 3 * size_t(n));
                to_major(resid.data(), stage.data() + 4 * size_t(n));
                d_y.upload(stage.data(), n, st);
                d_gw.upload(stage.data() + size_t(n), n, st);
                d_off.upload(stage.data() + 2 * size_t(n), n, st);
                d_eta.upload(stage.data() + 3 * size_t(n), n, st);
                d_r.upload(stage.data() + 4 * size_t(n), n, st);
                multi_w_uniform = false; // IRLS weights differ between classes
            } else {
                d_y.upload((const T*)a->glm_y, n, st);
                d_gw.upload((const T*)a->glm_weights, n, st);
                d_off.upload((const T*)a->offsets, n, st);
                d_eta.upload(eta.data(), n, st);
                d_r.upload(resid.data(), n, st);
            }
            beta0 = T(a->beta0); loss_null = T(a->loss_null); loss_full = T(a->loss_full);
            irls_max_iters = size_t(a->irls_max_iters); irls_tol = T(a->irls_tol);
            setup_loss_null = a->setup_loss_null;
            sync();
            device_append_screen();
        }
    }
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
            work.reserve(size_t(sweep_work_elems(d->n, d->p)));
            launch_fill<T>(v.p, T(1) / T(d->n), d->n, s);
            launch_fill<T>(xm.p, T(0.5), d->p, s);
            launch_fill<T>(sc.p, T(0.25), 1, s);
            auto once = [&]() {
                if (d->kind == 0) launch_sweep<T>(d->dense<T>(), v.p, out.p, 0, d->p, nullptr, sc.p, xm.p, false, work.p, s);
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
