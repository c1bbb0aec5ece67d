// kernels_cd.hip — the block-coordinate-descent pin solver for one lambda, resident on one CU (gfx950).
//
// Restates solver_gaussian_pin_naive.hpp:16-401 (coordinate_descent / solve_active / solve) and
// solver_gaussian_pin_base.hpp:100-195 (convergence measure, rsq update, scalar and group coefficient updates) with
// bcd/unconstrained/newton.hpp:35-142 + optimization/newton.hpp:28-65 for the group prox — same visiting order
// (screen insertion order / activation order), same tolerances, same "changed" predicates, same active-set marking.
//
// What differs is how the per-visit gradient is obtained.  The reference recomputes  g_k = X_k^T W r - xbar_k * rsum
// from the n-vector residual at every visit and pushes  r -= X_k * del  after every change: two n-length passes per
// visit, strictly sequential.  On a GPU that is one grid-wide reduction per visit (>= several microseconds each,
// hundreds of thousands of visits per path).  Here the gradient of every screen value is kept *current* instead:
//      g_a  <-  g_a - C[a,k] * del      for all screen values a,     C = X_S^T W X_S - xbar xbar^T  (kernels_gram.hip)
// which is the same number in exact arithmetic (r' = r - x_k del, rsum' = rsum - xbar_k del), costs |S| instead of n
// per change, nothing per unchanged visit, and lets one workgroup run the whole active-set / screen-set alternation
// of a fit without leaving the CU: g lives in LDS, C streams from L2/HBM one column per change, the coefficient
// update itself is evaluated redundantly by every lane so no broadcast is needed.  The residual is brought up to
// date once per fit from the list of net coefficient changes this kernel emits (axpy_cols kernel).
#include "kernels.hpp"
#include <stdexcept>
#include <string>

namespace ahip {

namespace {

template <class T>
__device__ __forceinline__ T wsum(T x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

// Group coordinate update executed by ONE wave (lane-parallel over the q coefficients), results left in LDS.
//   scratch layout (T each, stride mq = max group size):
//     [0] gk  [1] gk_t  [2] ak_old  [3] ak_old_t  [4] ak_t  [5] buf1  [6] buf2  [7] del(raw) / x
//   res[0] = changed(0/1)  res[1] = convergence term  res[2] = rsq increment  res[3] = resid_sum increment
//   res[4] = status (0 ok, 1 newton failure)
template <class T>
__device__ __forceinline__ void group_update_wave0(const CdParams<T>& p, const T* gl, int b, int q, T pk, const T* __restrict__ V,
                                   T l1, T l2, T* scr, T* res, int lane) {
    const int mq = p.max_group_size;
    T* gk = scr;
    T* gk_t = scr + mq;
    T* ak_old = scr + 2 * mq;
    T* ak_old_t = scr + 3 * mq;
    T* ak_t = scr + 4 * mq;
    T* buf1 = scr + 5 * mq;
    T* buf2 = scr + 6 * mq;
    T* del = scr + 7 * mq;
    const T* A = p.vars + b;
    const T* xm = p.xmean + b;
    const T l1p = l1 * pk, l2p = l2 * pk;

    for (int i = lane; i < q; i += 64) {
        gk[i] = gl[b + i]; // already includes the -rsum*xbar correction (pin_naive:118-121)
        ak_old[i] = p.beta[b + i];
    }
    __builtin_amdgcn_wave_barrier();
    // gk_t = gk V ; ak_old_t = ak_old V ; gk_t += A * ak_old_t   (pin_naive:123-140)
    for (int j = lane; j < q; j += 64) {
        T s1 = 0, s2 = 0;
        const T* Vj = V + int64_t(j) * q;
        for (int i = 0; i < q; ++i) {
            s1 = fma(gk[i], Vj[i], s1);
            s2 = fma(ak_old[i], Vj[i], s2);
        }
        ak_old_t[j] = s2;
        gk_t[j] = s1 + A[j] * s2;
    }
    __builtin_amdgcn_wave_barrier();
    // ---- newton_solver (bcd/unconstrained/newton.hpp:35-142): v = gk_t, L = A -------------------------------
    T nrm2 = 0;
    for (int i = lane; i < q; i += 64) nrm2 = fma(gk_t[i], gk_t[i], nrm2);
    nrm2 = wsum(nrm2);
    const T v_l2 = sqrt(nrm2);
    int status = 0;
    if (v_l2 <= l1p) {
        for (int i = lane; i < q; i += 64) ak_t[i] = 0;
    } else if (l1p <= T(0)) {
        for (int i = lane; i < q; i += 64) ak_t[i] = gk_t[i] / (A[i] + l2p);
    } else {
        for (int i = lane; i < q; i += 64) buf1[i] = A[i] + l2p;
        T h = 0, fh, dfh;
        auto step = [&](T hh) {
            T t = 0, s = 0;
            for (int i = lane; i < q; i += 64) {
                const T b2 = T(1) / (buf1[i] * hh + l1p);
                const T z = gk_t[i] * b2;
                const T x = z * z;
                buf2[i] = b2;
                t += x;
                s += x * buf1[i] * b2;
            }
            t = wsum(t);
            s = wsum(s);
            fh = t - T(1);
            dfh = -s * (T(1) + sqrt(t)) / t;
        };
        step(h);
        int iters = 0;
        while ((fabs(fh) > p.newton_tol) && (iters < p.newton_max_iters)) {
            h -= fh / dfh;
            h = h > T(0) ? h : T(0);
            step(h);
            ++iters;
        }
        for (int i = lane; i < q; i += 64) ak_t[i] = h * gk_t[i] * buf2[i];
        if (iters >= p.newton_max_iters) status = 1;
    }
    __builtin_amdgcn_wave_barrier();
    // gk_t -= A*ak_old_t ; changed? ; convergence / rsq in rotated coordinates (pin_naive:144-154)
    T dn = 0, cm = 0, rs = 0;
    for (int i = lane; i < q; i += 64) {
        const T g = gk_t[i] - A[i] * ak_old_t[i];
        const T d = ak_t[i] - ak_old_t[i];
        dn = fma(d, d, dn);
        cm = fma(A[i] * d, d, cm);
        rs += d * (T(2) * g - d * A[i]);
    }
    dn = wsum(dn);
    cm = wsum(cm);
    rs = wsum(rs);
    const bool changed = !(sqrt(dn) <= p.dbeta_tol * sqrt(T(q)));
    T rsum = 0;
    if (changed) {
        // ak = ak_t V^T ; del = ak_old - ak ; resid_sum += xbar . del   (pin_naive:156-163)
        for (int i = lane; i < q; i += 64) {
            T s = 0;
            for (int j = 0; j < q; ++j) s = fma(ak_t[j], V[i + int64_t(j) * q], s);
            const T d = ak_old[i] - s;
            del[i] = d;
            p.beta[b + i] = s;
            rsum = fma(xm[i], d, rsum);
        }
        rsum = wsum(rsum);
    }
    if (lane == 0) {
        res[0] = changed ? T(1) : T(0);
        res[1] = cm / T(q);
        res[2] = rs;
        res[3] = rsum;
        res[4] = T(status);
    }
}

template <class T, int NT, bool GLDS>
__global__ __launch_bounds__(NT) void cd_kernel(CdParams<T> p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* smem = reinterpret_cast<T*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nv = p.nv, ns = p.ns;
    const int mq = p.max_group_size;
    T* scr = smem;                       // 8*mq group scratch
    T* res = smem + 8 * mq;              // 8 result words
    T* gl = GLDS ? (smem + 8 * mq + 8) : p.g;

    if (GLDS) {
        for (int a = tid; a < nv; a += NT) gl[a] = p.g[a];
    }
    __syncthreads();

    T rsq = p.sc->rsq, rsum = p.sc->resid_sum;
    int asz = p.sc->active_size;
    int64_t iters = 0, n_upd = 0, n_vis_s = 0, n_vis_a = 0, n_pass_s = 0, n_pass_a = 0;
    int status = CD_OK;
    const T l1 = p.lmda * p.alpha;
    const T l2 = p.lmda * (T(1) - p.alpha);
    const bool all_scalar = p.all_scalar != 0;

    // one coordinate-descent pass (pin_naive:16-168); returns the convergence measure
    auto pass = [&](const int32_t* list, int count, bool mark) -> T {
        T cm = 0;
        for (int it = 0; it < count && status == CD_OK; ++it) {
            const int ss = list ? list[it] : it;
            int b, q;
            if (all_scalar) { b = ss; q = 1; }
            else { b = p.sbegin[ss]; q = p.ssize[ss]; }
            const T pk = p.spen[ss];
            bool changed = false;
            if (q == 1) {
                const T ak_old = p.beta[b];
                const T A = p.vars[b];
                const T gcur = gl[b];
                const T gk = fma(ak_old, A, gcur);                 // pin_naive:85-89
                const T denom = A + l2 * (p.spen2 ? p.spen2[ss] : pk);                    // pin_base:181-195
                const T v = fabs(gk) - l1 * pk;
                const T ak = (v > T(0)) ? copysign(v, gk) / denom : T(0);
                if (ak != ak_old) {                                // pin_naive:97
                    changed = true;
                    const T del = ak - ak_old;
                    const T c1 = A * del * del;
                    cm = c1 > cm ? c1 : cm;                        // pin_base:112-122
                    rsq += del * (T(2) * gcur - del * A);          // pin_base:136-146 (gk - ak_old*A == gcur)
                    rsum -= p.xmean[b] * del;                      // pin_naive:107
                    const bool add = mark && !p.is_active[ss];     // add_active_set, pin_naive:294-304
                    if (add && asz >= p.max_active_size) status = CD_MAX_ACTIVE;
                    __syncthreads(); // every wave has read gl[b], beta[b], is_active[ss]
                    if (status != CD_OK) break;
                    if (tid == 0) {
                        p.beta[b] = ak;
                        if (add) { p.is_active[ss] = 1; p.active_set[asz] = ss; }
                    }
                    if (add) ++asz;
                    const T* __restrict__ Cc = p.C + int64_t(b) * p.ldc;
                    for (int a0 = tid; a0 < nv; a0 += 8 * NT) { // 8 independent loads in flight per lane
                        T c8[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) c8[u] = (a0 + u * NT < nv) ? Cc[a0 + u * NT] : T(0);
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            if (a0 + u * NT < nv) gl[a0 + u * NT] = fma(-del, c8[u], gl[a0 + u * NT]);
                    }
                    __syncthreads();
                    ++n_upd;
                }
            } else {
                const bool add = mark && !p.is_active[ss];
                __syncthreads(); // scratch free, gl stable, is_active[ss] read by everyone
                if (wv == 0) group_update_wave0<T>(p, gl, b, q, pk, p.V + p.voff[ss], l1, l2, scr, res, lane);
                __syncthreads();
                changed = res[0] != T(0);
                if (res[4] != T(0)) status = CD_NEWTON;
                if (changed && status == CD_OK) {
                    cm = res[1] > cm ? res[1] : cm;
                    rsq += res[2];
                    rsum += res[3];
                    if (add) {
                        if (asz >= p.max_active_size) status = CD_MAX_ACTIVE;
                        else {
                            if (tid == 0) { p.is_active[ss] = 1; p.active_set[asz] = ss; }
                            ++asz;
                        }
                    }
                    const T* del = scr + 7 * mq;
                    for (int a = tid; a < nv; a += NT) {
                        T acc = gl[a];
                        for (int t = 0; t < q; ++t) acc = fma(p.C[a + int64_t(b + t) * p.ldc], del[t], acc);
                        gl[a] = acc;
                    }
                    __syncthreads();
                    ++n_upd;
                }
            }
            (void)changed;
        }
        return cm;
    };

    // pin_naive:317-357 for a single lambda
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
        n_vis_s += ns;
        const T cm = pass(nullptr, ns, true);
        if (status != CD_OK) break;
        if (cm < p.tol) break;
        if (iters >= p.max_iters) { status = CD_MAX_CDS; break; }
    }

    __syncthreads();
    if (GLDS) {
        for (int a = tid; a < nv; a += NT) p.g[a] = gl[a];
    }

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
        const T d = p.beta[a] - p.beta0[a];
        if (p.beta[a] != p.beta0[a]) { p.dcols[o] = p.vcol[a]; p.dvals[o] = d; ++o; }
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
}

} // namespace

template <class T> bool launch_cd_lasso(const CdParams<T>& p, hipStream_t s); // kernels_cd_lasso.hip

template <class T>
void launch_cd(const CdParams<T>& p, hipStream_t s) {
    if (launch_cd_lasso<T>(p, s)) return;
    constexpr int NT = 256;
    const size_t base = size_t(8 * p.max_group_size + 8) * sizeof(T);
    const size_t with_g = base + size_t(p.nv) * sizeof(T);
    const size_t lds_cap = 150 * 1024; // of the 160 KiB per CU; the static cnt[] array takes ~1 KiB
    // the dynamic-LDS limit of both instantiations is raised once, to the cap (see launch_k in kernels_cd_lasso.hip: a
    // per-call value races between the threads of concurrent solves)
    static const bool raised = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cd_kernel<T, NT, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  150 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cd_kernel<T, NT, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  150 * 1024);
        return true;
    }();
    (void)raised;
    (void)hipGetLastError(); // (whatever an earlier call of this thread left behind, e.g. an allocation the pool retried)
    if (with_g <= lds_cap) {
        hipLaunchKernelGGL((cd_kernel<T, NT, true>), dim3(1), dim3(NT), with_g, s, p);
    } else {
        if (base > lds_cap) throw std::runtime_error("adelie_core: a group is too large for the single-workgroup coordinate descent.");
        hipLaunchKernelGGL((cd_kernel<T, NT, false>), dim3(1), dim3(NT), base, s, p);
    }
    if (const hipError_t e = hipGetLastError(); e != hipSuccess) // a refused launch must not pass for a solve
        throw std::runtime_error(std::string("adelie_hip: HIP error '") + hipGetErrorString(e) + "' launching cd_kernel");
}

template void launch_cd<double>(const CdParams<double>&, hipStream_t);
template void launch_cd<float>(const CdParams<float>&, hipStream_t);

} // namespace ahip
