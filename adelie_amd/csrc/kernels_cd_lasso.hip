// kernels_cd_lasso.hip — coordinate-descent pin solver, lasso specialisation (every group has size 1, so the value
// index equals the screen index).  Same iteration as cd_kernel in kernels_cd.hip (which documents the algorithm and
// cites the reference lines); the differences are all about keeping one CU busy:
//   * 512 threads; thread t owns the chunks {(k*NT + t)*VEC .. +VEC}, k < kmax, of g in LDS (16-byte ds_read/ds_write);
//     g is padded to whole chunks and the Gram matrix has ldc >= the padded length, so no lane-divergent control flow
//     surrounds any load;
//   * the Gram column of a coordinate is fetched into registers two visits before it is needed (three rotating
//     slots), so the HBM/L2 latency of one column overlaps the two preceding visits.  In a screen pass only
//     currently-active coordinates are prefetched for real (an inactive one almost never moves; if it does, the
//     column is loaded on demand) — the others load a dummy column that stays in cache.  Every prefetch load is issued
//     unconditionally so the number of loads in flight is static and the compiler can wait with an exact vmcnt(N)
//     for the oldest slot only;
//   * the two barriers of an update wait for LDS only (s_waitcnt lgkmcnt(0); s_barrier) so the prefetches stay in
//     flight across them.  Everything the waves exchange inside a pass goes through LDS (g, the active flags);
//     beta / active_set go to global memory from lane 0 and are only re-read in a LATER pass, after the full
//     __syncthreads() that opens every pass.
#include "kernels.hpp"
#include <stdexcept>
#include <string>

namespace ahip {

namespace {

typedef double cd_d2 __attribute__((ext_vector_type(2)));
typedef float cd_f4 __attribute__((ext_vector_type(4)));
template <class T> struct CdVec;
template <> struct CdVec<double> { using type = cd_d2; static constexpr int N = 2; };
template <> struct CdVec<float> { using type = cd_f4; static constexpr int N = 4; };

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#ifdef AHIP_CD_PROFILE
#define CDP_DECL int64_t cdp[8] = {0, 0, 0, 0, 0, 0, 0, 0}; int64_t cdp_t = __builtin_readcyclecounter();
#define CDP_MARK(i) { const int64_t t_ = __builtin_readcyclecounter(); cdp[i] += t_ - cdp_t; cdp_t = t_; }
#define CDP_STORE if (tid == 0) { for (int i_ = 0; i_ < 8; ++i_) p.sc->dbg[i_] = cdp[i_]; }
#else
#define CDP_DECL
#define CDP_MARK(i)
#define CDP_STORE
#endif

template <class T, int NT, int K, bool TAIL>
__global__ __launch_bounds__(NT) void cd_lasso_kernel(CdParams<T> p) {
    using V = typename CdVec<T>::type;
    constexpr int VEC = CdVec<T>::N;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x;
    const int nv = p.nv;
    const int nvp = ((nv + NT * VEC - 1) / (NT * VEC)) * (NT * VEC);
    const int kmax = nvp / (NT * VEC); // chunks in use (uniform)
    T* gl = reinterpret_cast<T*>(smem_raw);
    int8_t* act = reinterpret_cast<int8_t*>(smem_raw + size_t(nvp) * sizeof(T));

    for (int a = tid; a < nvp; a += NT) gl[a] = a < nv ? p.g[a] : T(0);
    for (int a = tid; a < nv; a += NT) act[a] = p.is_active[a];
    __syncthreads();

    T rsq = p.sc->rsq, rsum = p.sc->resid_sum;
    int asz = p.sc->active_size;
    int64_t iters = 0, n_upd = 0, n_vis_s = 0, n_vis_a = 0, n_pass_s = 0, n_pass_a = 0;
    int status = CD_OK;
    const T l1 = p.lmda * p.alpha;
    const T l2 = p.lmda * (T(1) - p.alpha);
    const int64_t ldc = p.ldc;
    CDP_DECL

    struct Slot {
        int ss;       // coordinate (-1: past the end of the list)
        T beta, A, pk, pk2, xm;
        bool have;    // col[] holds the coordinate's Gram column (else a dummy column was loaded)
        V col[K];
    };

    auto load_col = [&](int b, V (&col)[K]) {
        const T* __restrict__ Cc = p.C + int64_t(b) * ldc;
#pragma unroll
        for (int k = 0; k < K; ++k) col[k] = *reinterpret_cast<const V*>(Cc + (k * NT + tid) * VEC);
    };
    auto apply_col = [&](int b, T del, const V (&col)[K]) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int a = (k * NT + tid) * VEC;
            V g = *reinterpret_cast<V*>(gl + a);
#pragma unroll
            for (int e = 0; e < VEC; ++e) g[e] = fma(-del, col[k][e], g[e]);
            *reinterpret_cast<V*>(gl + a) = g;
        }
        if (TAIL) { // chunks beyond the register-resident ones (nv > K*NT*VEC): on demand, 4 loads in flight
            const T* __restrict__ Cc = p.C + int64_t(b) * ldc;
            for (int k0 = K; k0 < kmax; k0 += 4) {
                V c[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int kk = k0 + u < kmax ? k0 + u : k0;
                    c[u] = *reinterpret_cast<const V*>(Cc + (kk * NT + tid) * VEC);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (k0 + u < kmax) {
                        const int a = ((k0 + u) * NT + tid) * VEC;
                        V g = *reinterpret_cast<V*>(gl + a);
#pragma unroll
                        for (int e = 0; e < VEC; ++e) g[e] = fma(-del, c[u][e], g[e]);
                        *reinterpret_cast<V*>(gl + a) = g;
                    }
                }
            }
        }
    };

    // one coordinate-descent pass (solver_gaussian_pin_naive.hpp:16-168); returns the convergence measure
    auto pass = [&](const int32_t* list, int count, bool mark) -> T {
        T cm = 0;
        if (count <= 0) return cm;
        __syncthreads(); // full barrier: global stores of the previous pass (beta, active_set) are visible
        Slot s0, s1, s2;
        auto fetch = [&](Slot& s, int ss) {
            s.ss = ss;
            const int sc = ss < 0 ? 0 : ss;
            s.pk = p.spen[sc];
            s.pk2 = p.spen2 ? p.spen2[sc] : s.pk; // (penalty_l2, ABI 8)
            s.beta = p.beta[sc];
            s.A = p.vars[sc];
            s.xm = p.xmean[sc];
            s.have = (ss >= 0) && (mark ? (act[sc] != 0) : true);
            load_col(s.have ? sc : 0, s.col);
        };
        // A visit is straight-line code apart from the (uniform) "changed" branch, which contains no global loads
        // except on the rare misprediction path: the compiler's vmcnt bookkeeping for the prefetched slots stays exact.
        auto visit = [&](Slot& s) {
            CDP_MARK(0) // fetch issue + loop overhead
            const bool valid = s.ss >= 0;
            const int b = valid ? s.ss : 0;
            const T gcur = gl[b];
            const T denom = s.A + l2 * s.pk2;                // pin_base:181-195
            const T rden = T(1) / denom;                     // independent of g: overlaps the LDS read
            const T gk = fma(s.beta, s.A, gcur);             // pin_naive:85-89
            const T v = fabs(gk) - l1 * s.pk;
            T ak = T(0);
            if (v > T(0)) {
                const T x = copysign(v, gk);
                const T q0 = x * rden;                       // x / denom via the reciprocal,
                const T r = fma(-q0, denom, x);              // with one residual correction step
                ak = fma(r, rden, q0);
            }
            if (!valid) ak = s.beta;                         // padding visit past the end of the list: no-op
            CDP_MARK(1) // g read + coefficient update (waits for the slot's scalar loads)
            if (ak != s.beta) {                              // pin_naive:97
                const T del = ak - s.beta;
                const T c1 = s.A * del * del;
                cm = c1 > cm ? c1 : cm;                      // pin_base:112-122
                rsq += del * (T(2) * gcur - del * s.A);      // pin_base:136-146
                rsum -= s.xm * del;                          // pin_naive:107
                const bool add = mark && (act[b] == 0);      // add_active_set, pin_naive:294-304
                const bool full = add && asz >= p.max_active_size;
                if (full) status = CD_MAX_ACTIVE;            // the host restores the pre-fit state on any error
                lds_barrier(); // every wave has read gl[b] and act[b]
                CDP_MARK(2) // barrier 1
                if (tid == 0) {
                    p.beta[b] = ak;
                    if (add && !full) { act[b] = 1; p.is_active[b] = 1; p.active_set[asz] = b; }
                }
                if (add && !full) ++asz;
                if (s.have) {
                    apply_col(b, del, s.col);
                } else { // mispredicted (an inactive coordinate moved): fetch its column now
                    V tmp[K];
                    load_col(b, tmp);
                    apply_col(b, del, tmp);
                }
                CDP_MARK(3) // wait for the column + LDS update
                lds_barrier();
                CDP_MARK(4) // barrier 2
                ++n_upd;
            }
        };
        auto idx_of = [&](int itx) -> int { return itx < count ? (list ? list[itx] : itx) : -1; };
        int ss_ahead = idx_of(2);
        fetch(s0, idx_of(0));
        fetch(s1, idx_of(1));
        for (int it = 0; it < count && status == CD_OK; it += 3) {
            int nx = idx_of(it + 3);
            fetch(s2, ss_ahead);
            ss_ahead = nx;
            visit(s0);
            nx = idx_of(it + 4);
            fetch(s0, ss_ahead);
            ss_ahead = nx;
            visit(s1);
            nx = idx_of(it + 5);
            fetch(s1, ss_ahead);
            ss_ahead = nx;
            visit(s2);
        }
        return cm;
    };

    // solver_gaussian_pin_naive.hpp:317-357 for a single lambda
    while (status == CD_OK) {
        while (status == CD_OK) { // solve_active, pin_naive:173-215
            ++iters;
            ++n_pass_a;
            n_vis_a += asz;
            const T cm = pass(p.active_set, asz, false);
            if (status != CD_OK) break;
            if (cm < p.tol) break;
            if (iters >= p.max_iters) { status = CD_MAX_CDS; break; }
        }
        if (status != CD_OK) break;
        ++iters;
        ++n_pass_s;
        n_vis_s += nv;
        const T cm = pass(nullptr, nv, true);
        if (status != CD_OK) break;
        if (cm < p.tol) break;
        if (iters >= p.max_iters) { status = CD_MAX_CDS; break; }
    }

    __syncthreads();
    for (int a = tid; a < nv; a += NT) p.g[a] = gl[a];

    // ordered compaction of the net coefficient changes -> (design column, delta) list for the residual update
    __shared__ int cnt[NT + 1];
    const int chunk = (nv + NT - 1) / NT;
    const int a0 = tid * chunk, a1 = min(nv, a0 + chunk);
    int c = 0;
    for (int a = a0; a < a1; ++a) c += (p.beta[a] != p.beta0[a]) ? 1 : 0;
    cnt[tid + 1] = c;
    if (tid == 0) cnt[0] = 0;
    __syncthreads();
    if (tid == 0)
        for (int t = 1; t <= NT; ++t) cnt[t] += cnt[t - 1];
    __syncthreads();
    int o = cnt[tid];
    for (int a = a0; a < a1; ++a) {
        const T b1 = p.beta[a], b0 = p.beta0[a];
        if (b1 != b0) { p.dcols[o] = p.vcol[a]; p.dvals[o] = b1 - b0; ++o; }
    }
    if (tid == 0) {
        p.sc->rsq = rsq;
        p.sc->resid_sum = rsum;
        p.sc->iters = iters;
        p.sc->n_visits_screen = n_vis_s;
        p.sc->n_visits_active = n_vis_a;
        p.sc->n_updates = n_upd;
        p.sc->n_passes_screen = n_pass_s;
        p.sc->n_passes_active = n_pass_a;
        p.sc->active_size = asz;
        p.sc->status = status;
        p.sc->n_delta = cnt[NT];
    }
    CDP_STORE
}

} // namespace

// Returns false when the problem does not fit this specialisation (the caller then uses the generic kernel).
// The kernel is instantiated per number of 512-lane x 16-byte chunks of g (K = 1..8, exactly the chunks in use: every
// load instruction moves 8 KB through the CU's vector cache whether it hits or not, so none is issued in vain);
// longer screen sets use K = 8 plus an on-demand tail.
template <class T, int NT, int K>
static void launch_k(const CdParams<T>& p, size_t bytes, bool tail, hipStream_t s) {
    auto k = tail ? cd_lasso_kernel<T, NT, K, true> : cd_lasso_kernel<T, NT, K, false>;
    // The limit is raised ONCE per instantiation, to the most any launch asks for (thread-safe static initialisation).  It used
    // to be set to this launch's size on every call: with several solves in flight (the folds of cv_grpnet run on threads) one
    // thread could lower it between another thread's call and its launch, and a launch above the limit does not run.
    static const bool raised = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cd_lasso_kernel<T, NT, K, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cd_lasso_kernel<T, NT, K, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        return true;
    }();
    (void)raised;
    (void)hipGetLastError(); // (whatever an earlier call of this thread left behind, e.g. an allocation the pool retried)
    hipLaunchKernelGGL(k, dim3(1), dim3(NT), bytes, s, p);
    if (const hipError_t e = hipGetLastError(); e != hipSuccess) // a refused launch must not pass for a solve
        throw std::runtime_error(std::string("adelie_hip: HIP error '") + hipGetErrorString(e) + "' launching cd_lasso_kernel");
}

template <class T>
bool launch_cd_lasso(const CdParams<T>& p, hipStream_t s) {
    if (!(p.all_scalar && p.nv == p.ns)) return false;
    constexpr int NT = 512, KMAX = 8, VEC = 16 / sizeof(T);
    const size_t lds_cap = 150 * 1024; // of the 160 KiB per CU (static arrays take a few KiB)
    const size_t nvp = size_t((p.nv + NT * VEC - 1) / (NT * VEC)) * (NT * VEC);
    const size_t bytes = nvp * sizeof(T) + size_t(p.ns) + 16;
    if (bytes > lds_cap || int64_t(nvp) > p.ldc) return false;
    const int kmax = int(nvp / (NT * VEC));
    switch (kmax < KMAX ? kmax : KMAX) {
        case 1: launch_k<T, NT, 1>(p, bytes, false, s); break;
        case 2: launch_k<T, NT, 2>(p, bytes, false, s); break;
        case 3: launch_k<T, NT, 3>(p, bytes, false, s); break;
        case 4: launch_k<T, NT, 4>(p, bytes, false, s); break;
        case 5: launch_k<T, NT, 5>(p, bytes, false, s); break;
        case 6: launch_k<T, NT, 6>(p, bytes, false, s); break;
        case 7: launch_k<T, NT, 7>(p, bytes, false, s); break;
        default: launch_k<T, NT, 8>(p, bytes, kmax > KMAX, s); break;
    }
    return true;
}

template bool launch_cd_lasso<double>(const CdParams<double>&, hipStream_t);
template bool launch_cd_lasso<float>(const CdParams<float>&, hipStream_t);

} // namespace ahip
