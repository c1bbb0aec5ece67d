// kernels_cons.hip — box / one-sided constraints on groups of SEVERAL coefficients, solved on the device.
//
// The reference visits such a group through ConstraintBox::solve / ConstraintOneSided::solve (constraint_box.ipp:98-250,
// constraint_one_sided.ipp:107-262), both of which run the dual proximal-Newton driver of constraint/utils.hpp:24-243: ascent
// on the multipliers mu with the primal x*(mu) from the unconstrained group update (bcd/unconstrained/newton.hpp:35-142), a
// quadratic model of the dual whose Hessian is |x| Q D Q' + l1 kappa |x| a a', a coordinate-descent sub-solver that keeps mu
// feasible (optimization/pinball_full.hpp:84-118 for a box, nnqp_full.hpp:150-178 for one-sided bounds), and backtracking
// towards the ellipse |v - Q'A'mu| = l1 when a step lands where the primal is zero.
//
// Here ONE WAVEFRONT runs the whole visit of such a group (pin_naive:110-168 with update_coordinate_g1_f = constraint->solve,
// :439-458): lane i carries component i of every d-vector (d <= 64), the eigenbasis Q and the dual Hessian live in LDS, every
// decision of the driver is wave-uniform (it is taken on wave reductions), and the coordinate descent on mu is the same strictly
// sequential loop with the scalars of a coordinate read out of the lanes.  Arithmetic is double precision for either value type
// (the host visits it replaces handed the objects doubles as well).  The group is a block of its own in every pass
// (Solver::build_partition): its gradient comes from the panel step + reduce in front of it, its changes leave the way a device
// solve leaves them — coefficients, the (column, delta) list of the next step's residual update, the pass state, the end-of-pass
// report — so that a constrained group costs launches, not host round trips.
//
// abs_grad of such groups (solver_base.hpp:62-93): cons_abs_grad_kernel — the constraint's gradient for screened groups, its
// solve_zero (constraint_box.ipp:268-284, constraint_one_sided.ipp:269-279) for the others, which also leaves the multipliers
// the object would hold.
#include <stdexcept>
#include <string>
#include "kernels.hpp"
#include "wavered.hpp"

namespace ahip {

namespace {

constexpr int CW = 64;          // lanes = largest group
constexpr double kMaxSolver = 1e100; // configs.hpp:13 max_solver_value

__device__ __forceinline__ double wsum(double x) { return wave_sum64(x); }
__device__ __forceinline__ double rl(double x, int l) { return prdl(x, l); }

struct ConsCtx {
    int d, LD, lane;
    bool box;
    double* Q;   // [d][LD] column-major with odd leading dimension (conflict-free both ways)
    double* H;   // dual Hessian, same layout
    double* vs;  // [CW] broadcast slot
    double* vs2; // [CW] second broadcast slot
};

// y_i = sum_j M[i + j LD] x_j   (x handed over in registers, lane j holds x_j)
__device__ __forceinline__ double matvec(const ConsCtx& c, const double* M, double x) {
    __syncthreads();
    c.vs[c.lane] = x;
    __syncthreads();
    double acc = 0;
    if (c.lane < c.d)
        for (int j = 0; j < c.d; ++j) acc += M[c.lane + j * c.LD] * c.vs[j];
    return acc;
}
// y_j = sum_i M[i + j LD] x_i
__device__ __forceinline__ double matvec_t(const ConsCtx& c, const double* M, double x) {
    __syncthreads();
    c.vs[c.lane] = x;
    __syncthreads();
    double acc = 0;
    if (c.lane < c.d)
        for (int i = 0; i < c.d; ++i) acc += M[i + c.lane * c.LD] * c.vs[i];
    return acc;
}

// bcd/unconstrained/newton.hpp:35-142 + optimization/newton.hpp:28-65 (the unconstrained group update, started at h = 0 as the
// reference's): x, and the buffers b1 = L + l2, b2 = 1 / (b1 h + l1) the dual Hessian reads
__device__ __forceinline__ void group_prox(const ConsCtx& c, double L, double v, double l1, double l2, double& x, double& b1, double& b2) {
    const bool in = c.lane < c.d;
    const double vn = sqrt(wsum(in ? v * v : 0.0));
    if (vn <= l1) { x = 0; return; }
    if (l1 <= 0.0) { x = in ? v / (L + l2) : 0.0; return; }
    b1 = L + l2;
    double h = 0, fh, dfh;
    auto step_f = [&](double hh) {
        b2 = in ? 1.0 / (b1 * hh + l1) : 0.0;
        const double z = v * b2, zz = in ? z * z : 0.0;
        const double t = wsum(zz);
        fh = t - 1.0;
        const double s = wsum(zz * b1 * b2);
        dfh = -s * (1.0 + sqrt(t)) / t;
    };
    step_f(h);
    for (int it = 0; fabs(fh) > 1e-12 && it < 100000; ++it) {
        h -= fh / dfh;
        h = fmax(h, 0.0);
        step_f(h);
    }
    x = in ? h * v * b2 : 0.0;
}

enum ConsRc { CRC_OK = 0, CRC_PN = 1, CRC_QP = 2, CRC_UNEXPECTED = 3 };

// The quadratic sub-problem in mu by coordinate descent (ProxNewton's newton_step): g = gradient of the model at mu (kept
// current), var * qp_tol the stopping threshold.  Box: pinball_full.hpp:92-104 with penalty_neg = lo, penalty_pos = up;
// one-sided: nnqp_full.hpp:158-164 in the coordinates mu * sgn.
__device__ __forceinline__ int cons_qp(const ConsCtx& c, double& mu, double g, double va, double vb, double var, double qp_tol, int64_t qp_max_iters) {
    const int d = c.d;
    if (!c.box) { g *= va; mu *= va; } // (va = sgn)
    for (int64_t it = 0; it < qp_max_iters; ++it) {
        double cm = 0;
        for (int i = 0; i < d; ++i) {
            const double qii = c.H[i + i * c.LD], gi = rl(g, i), old = rl(mu, i);
            double nw;
            if (c.box) {
                const double lo_i = rl(va, i), up_i = rl(vb, i); // lo = -lower >= 0, up = upper >= 0
                const double gi0 = gi + qii * old;
                nw = copysign(fmax(fmax(-lo_i - gi0, gi0 - up_i), 0.0), gi0 + lo_i) / qii;
            } else {
                const double sg = rl(va, i);
                const double step = (qii <= 0) ? 0.0 : gi / qii;
                nw = (sg > 0) ? fmax(old + step, 0.0) : fmin(old + step, 0.0);
            }
            const double del = nw - old;
            if (del == 0) continue;
            cm = fmax(cm, qii * del * del);
            if (c.lane == i) mu = nw;
            if (c.lane < d) g -= del * c.H[c.lane + i * c.LD];
        }
        if (cm < var * qp_tol) {
            if (!c.box) mu *= va;
            return CRC_OK;
        }
    }
    return CRC_QP;
}

// constraint/utils.hpp:24-243.  Lane i: x (in/out, eigen-coordinates), quad, linear, the bounds (box: lo = -lower, up; one-sided:
// sgn, b) and the multiplier mu (in/out).  Returns a ConsRc.
__device__ __forceinline__ int cons_solve(const ConsCtx& c, double& x, double quad, double linear, double l1, double l2,
                                          double va, double vb, double& mu, const double* cfg) {
    const int d = c.d;
    const bool in = c.lane < d;
    const int max_iters = int(cfg[0]);
    const double tol = cfg[1];
    const int64_t qp_max_iters = int64_t(cfg[2]);
    const double qp_tol = cfg[3], slack = cfg[4];
    auto At = [&](double m) { return c.box ? m : va * m; };      // A' mu (A = I / diag(sgn))
    if (sqrt(wsum(in ? linear * linear : 0.0)) <= l1) { x = 0; mu = 0; return CRC_OK; } // box :119-123, one-sided :161-165
    const double Qv = matvec(c, c.Q, linear);
    double grad_prev = 0, grad = 0, mu_prev = 0, b1 = 0, b2 = 0;
    // multipliers that best explain v while the primal stays zero (box :148-166, one-sided :190-208)
    auto min_mu_resid = [&](bool prev_valid_old, bool init) -> double {
        const double keep = mu;
        if (in) {
            if (c.box) {
                const double lb = (va <= 0) ? -kMaxSolver : 0.0, ub = (vb <= 0) ? kMaxSolver : 0.0;
                mu = fmin(fmax(Qv, lb), ub);
            } else {
                const double ub = (vb <= 0) ? kMaxSolver : 0.0;
                mu = fmin(fmax(va * Qv, 0.0), ub);
            }
        }
        const double e = in ? Qv - At(mu) : 0.0;
        const double nsq = wsum(e * e);
        if ((init || prev_valid_old) && nsq > l1 * l1) mu = keep;
        return nsq;
    };
    const bool x_init_zero = __ballot(in && x != 0) == 0ull;
    bool prev_valid = false, zero_checked = false;
    double resid_norm_prev = -1;
    if (x_init_zero) { // :78-83
        zero_checked = true;
        if (min_mu_resid(false, true) <= l1 * l1) return CRC_OK;
    }
    for (int iters = 1; iters <= max_iters; ++iters) {
        const double qtam = matvec_t(c, c.Q, At(mu));
        const double mu_resid = in ? linear - qtam : 0.0;
        const double resid_norm = sqrt(wsum(mu_resid * mu_resid)), resid_norm_sq = resid_norm * resid_norm;
        double x_norm = -1;
        bool in_ellipse = resid_norm <= l1;
        if (!in_ellipse) { // compute_primal :85-93
            group_prox(c, quad, mu_resid, l1, l2, x, b1, b2);
            x_norm = sqrt(wsum(in ? x * x : 0.0));
            in_ellipse = x_norm <= 0;
            if (l1 <= 0) { b1 = quad + l2; b2 = in ? 1.0 / (b1 * x_norm + l1) : 0.0; } // (newton.hpp:72-75 leaves its buffers unset)
        }
        if (in_ellipse) {
            if (iters == 1 && x_init_zero) { x = 0; return CRC_OK; } // :117-120
            if (prev_valid) { // :124-129
                const double gp = c.box ? grad_prev : grad_prev + vb;
                if (fabs(wsum(in ? (mu - mu_prev) * gp : 0.0) / double(d)) <= tol) { x = 0; return CRC_OK; }
            }
            if (!zero_checked) { // :136-164
                zero_checked = true;
                const bool prev_valid_old = prev_valid;
                if (!prev_valid_old) {
                    resid_norm_prev = resid_norm;
                    prev_valid = true;
                    mu_prev = mu;
                    grad_prev = c.box ? 0.0 : -vb;
                }
                if (min_mu_resid(prev_valid_old, false) <= l1 * l1) { x = 0; return CRC_OK; }
                if (!prev_valid_old) continue;
            }
            if (!prev_valid || (resid_norm_prev <= l1 * 0.9999) || (resid_norm > l1 * 1.0001)) return CRC_UNEXPECTED;
            const double target = (1 - slack) * l1 + slack * resid_norm_prev; // :177-185
            const double dm = in ? mu - mu_prev : 0.0;
            const double a = wsum(dm * dm);
            const double bq = wsum(in ? ((c.box ? Qv : va * Qv) - mu) * dm : 0.0);
            const double cc = resid_norm_sq - target * target;
            const double t_star = (-bq + sqrt(fmax(bq * bq - a * cc, 0.0))) / a;
            const double step = fmin(fmax(1 - t_star, 0.0), 1.0);
            mu = mu_prev + step * (mu - mu_prev);
            continue;
        }
        const double z = matvec(c, c.Q, x);         // primal_gradient: box x Q^T; one-sided sgn * (x Q^T) - b
        grad = in ? (c.box ? z : va * z - vb) : 0.0;
        bool opt; // hard_optimal
        if (c.box) opt = (grad <= vb && grad >= -va) && (fmax(mu, 0.0) * (grad - vb) == 0) && (fmin(mu, 0.0) * (grad + va) == 0);
        else opt = (grad <= 0) && (mu * grad == 0);
        if (__ballot(in && !opt) == 0ull) return CRC_OK;
        if (prev_valid && fabs(wsum(in ? (mu - mu_prev) * (grad_prev - grad) : 0.0) / double(d)) <= tol) return CRC_OK;
        resid_norm_prev = resid_norm;
        prev_valid = true;
        mu_prev = mu;
        grad_prev = grad;
        // Hessian of the dual objective, :208-237: x_norm Q diag(b2) Q' + l1 kappa x_norm alpha alpha'
        const double alpha_tmp = in ? x * b2 / x_norm : 0.0;
        const double kappa = 1.0 / wsum(in ? x * b1 * alpha_tmp : 0.0);
        const double alpha = matvec(c, c.Q, alpha_tmp);
        const double l1kn = l1 * kappa * x_norm;
        __syncthreads();
        c.vs[c.lane] = b2;
        c.vs2[c.lane] = alpha;
        __syncthreads();
        if (in)
            for (int c2 = 0; c2 < d; ++c2) { // row `lane`, column c2; (Q_r Q_c) b2 and alpha_r alpha_c: exactly symmetric
                double acc = 0;
                for (int j = 0; j < d; ++j) acc += (c.Q[c.lane + j * c.LD] * c.Q[c2 + j * c.LD]) * c.vs[j];
                c.H[c.lane + c2 * c.LD] = x_norm * acc + l1kn * (alpha * c.vs2[c2]);
            }
        __syncthreads();
        const double xq = matvec_t(c, c.Q, x);      // x' S^-1 x by the Woodbury identity, :229-236
        const double xy = wsum(in ? x * xq : 0.0), s1 = wsum(in ? xq * xq / b2 : 0.0), s2 = wsum(in ? x * x * b2 : 0.0);
        double var = (s1 - (xy * xy) / ((x_norm * x_norm) / (l1 * kappa) + s2)) / x_norm;
        var = fmax(var, 0.0);
        const int rc = cons_qp(c, mu, grad, va, vb, var, qp_tol, qp_max_iters);
        if (rc != CRC_OK) return rc;
    }
    return CRC_PN;
}

template <class T>
__global__ __launch_bounds__(CW) void grp_cons_visit_kernel(ConsVisitParams<T> p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int lane = threadIdx.x, d = p.q;
    const int LD = d | 1;
    ConsCtx c;
    c.d = d; c.LD = LD; c.lane = lane; c.box = p.native == ADELIE_HIP_NATIVE_BOX;
    c.Q = reinterpret_cast<double*>(smem_raw);
    c.H = c.Q + size_t(d) * LD;
    c.vs = c.H + size_t(d) * LD;
    c.vs2 = c.vs + CW;
    const bool in = lane < d;
    for (int e = lane; e < d * d; e += CW) c.Q[(e % d) + (e / d) * LD] = d > 1 ? double(p.V[e]) : 1.0;
    const double gk = in ? double(p.gsrc[lane]) : 0.0, ak = in ? double(p.beta[lane]) : 0.0, Ak = in ? double(p.vars[lane]) : 0.0;
    double va = in ? double(p.va[lane]) : 0.0, vb = in ? double(p.vb[lane]) : 0.0;
    if (c.box) { // the class keeps lo = -lower and the bounds clamped to +-max_solver_value (adelie/constraint.py:262-263)
        const double lower = fmax(va, -kMaxSolver), upper = fmin(vb, kMaxSolver);
        va = -lower; vb = upper;
    } else {
        vb = fmin(vb, kMaxSolver);
    }
    double mu = in ? double(p.mu[lane]) : 0.0;
    CdBlkState<T> bs = *p.st;
    if (p.first_of_pass) bs.cm = T(0);
    // into the eigenbasis: g V, beta V  (pin_naive:123-135)
    const double gt = matvec_t(c, c.Q, gk), a_old_t = matvec_t(c, c.Q, ak);
    double x = in ? a_old_t : 0.0;
    const double quad = Ak, lin = gt + Ak * a_old_t;
    int rc = cons_solve(c, x, quad, lin, p.l1, p.l2, va, vb, mu, p.cfg);
    if (!in) x = 0;
    if (in) p.mu[lane] = T(mu);
    int status = bs.status;
    if (rc == CRC_PN) status = CD_CONS_PN;
    else if (rc == CRC_QP) status = c.box ? CD_CONS_QP_BOX : CD_CONS_QP_NNQP;
    else if (rc == CRC_UNEXPECTED) status = CD_CONS_UNEXPECTED;
    int nz = 0;
    if (rc == CRC_OK) {
        const double dl = in ? x - a_old_t : 0.0;
        const double dn = wsum(dl * dl);
        if (!(sqrt(dn) <= p.dbeta_tol * sqrt(double(d)))) { // :144: the group changed
            const double cmv = wsum(quad * dl * dl), rs = wsum(dl * (2 * gt - dl * quad));
            const T cmn = T(cmv / double(d));
            bs.cm = cmn > bs.cm ? cmn : bs.cm;     // pin_base:100-110
            bs.rsq += T(rs);                        // pin_base:124-134
            const double a_new_d = matvec(c, c.Q, x); // back: beta = x V^T  (:156-157)
            const T a_new = T(a_new_d), ak_t = in ? p.beta[lane] : T(0);
            const T dlt = in ? a_new - ak_t : T(0);
            double rsum_l = 0;
            if (in) {
                p.beta[lane] = a_new;
                p.dcol[lane] = p.gram ? int32_t(p.b + lane) : int32_t(p.col0 + lane);
                p.dlt[lane] = dlt;
                if (!p.gram && p.sxm) rsum_l = double(p.sxm[lane]) * double(ak_t - a_new);
            }
            bs.resid_sum += T(wsum(rsum_l));
            bs.n_updates += 1;
            nz = d;
            if (p.mark && !p.is_active[p.ss]) { // add_active_set, pin_naive:294-304
                if (bs.active_size >= p.max_active_size) {
                    status = CD_MAX_ACTIVE;
                } else if (lane == 0) {
                    p.is_active[p.ss] = 1;
                    p.active_set[bs.active_size] = p.ss;
                }
                if (status != CD_MAX_ACTIVE) bs.active_size += 1;
            }
        }
    }
    bs.nz = nz;
    bs.status = status;
    if (lane == 0) {
        *p.st = bs;
        if (p.n_visits) p.n_visits[0] += 1;
        if (p.host_st && p.report_seq) {
            *p.host_st = bs;
            __threadfence_system();
            __hip_atomic_store(p.host_seq, p.report_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// abs_grad of the groups with device-side constraint objects, one thread per such group (solver_base.hpp:62-93)
template <class T>
__global__ void cons_abs_grad_kernel(const int32_t* __restrict__ list, int count, const int32_t* __restrict__ native,
                                     const int64_t* __restrict__ groups, const int64_t* __restrict__ gsizes,
                                     const int32_t* __restrict__ slot, const T* __restrict__ grad, const T* __restrict__ beta, const T* __restrict__ penalty, T regul_scale,
                                     const T* __restrict__ va, const T* __restrict__ vb, T* __restrict__ mu, T* __restrict__ abs_grad) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const int g = list[t];
    const int64_t k = groups[g], q = gsizes[g];
    const bool box = native[g] == ADELIE_HIP_NATIVE_BOX;
    double acc = 0;
    const int b = slot[g]; // the group's first screen value, -1 outside the screen set
    if (b >= 0) { // :69-75: the group's gradient minus the penalty's and the constraint's terms
        const double regul = double(regul_scale) * double(penalty[g]);
        for (int64_t i = 0; i < q; ++i) {
            const double cg = box ? double(mu[k + i]) : double(va[k + i]) * double(mu[k + i]);
            const double e = double(grad[k + i]) - regul * double(beta[b + i]) - cg;
            acc += e * e;
        }
    } else { // :88-93 solve_zero
        for (int64_t i = 0; i < q; ++i) {
            const double v = double(grad[k + i]);
            double m, e;
            if (box) {
                const double lo = -fmax(double(va[k + i]), -kMaxSolver), up = fmin(double(vb[k + i]), kMaxSolver);
                m = fmin(fmax(v, (lo <= 0) ? -kMaxSolver : 0.0), (up <= 0) ? kMaxSolver : 0.0);
                e = v - m;
            } else {
                const double sg = double(va[k + i]), bb = fmin(double(vb[k + i]), kMaxSolver);
                m = fmin(fmax(sg * v, 0.0), (bb <= 0) ? kMaxSolver : 0.0);
                e = v - sg * m;
            }
            mu[k + i] = T(m);
            acc += e * e;
        }
    }
    abs_grad[g] = T(sqrt(acc));
}

} // namespace

size_t cons_visit_lds(int q) { return (size_t(2) * size_t(q) * size_t(q | 1) + 2 * CW) * sizeof(double); }

template <class T>
void launch_grp_cons_visit(const ConsVisitParams<T>& p, hipStream_t s) {
    if (p.q < 1 || p.q > CW) throw std::runtime_error("adelie_hip: device constraint visit of a group beyond 64 coefficients.");
    static const bool raised = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(grp_cons_visit_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  int(cons_visit_lds(CW)));
        return true;
    }();
    (void)raised;
    (void)hipGetLastError();
    hipLaunchKernelGGL((grp_cons_visit_kernel<T>), dim3(1), dim3(CW), cons_visit_lds(p.q), s, p);
    if (const hipError_t e = hipGetLastError(); e != hipSuccess)
        throw std::runtime_error(std::string("adelie_hip: grp_cons_visit_kernel launch refused: ") + hipGetErrorString(e));
}

template <class T>
void launch_cons_abs_grad(const int32_t* list, int count, const int32_t* native, const int64_t* groups, const int64_t* gsizes,
                          const int32_t* slot, const T* grad, const T* beta, const T* penalty, T regul_scale,
                          const T* va, const T* vb, T* mu, T* abs_grad, hipStream_t s) {
    if (count <= 0) return;
    hipLaunchKernelGGL((cons_abs_grad_kernel<T>), dim3((unsigned)((count + 63) / 64)), dim3(64), 0, s, list, count, native, groups,
                       gsizes, slot, grad, beta, penalty, regul_scale, va, vb, mu, abs_grad);
}

#define INST(T)                                                                                                              \
    template void launch_grp_cons_visit<T>(const ConsVisitParams<T>&, hipStream_t);                                         \
    template void launch_cons_abs_grad<T>(const int32_t*, int, const int32_t*, const int64_t*, const int64_t*, const int32_t*, \
                                          const T*, const T*, const T*, T, const T*, const T*, T*, T*, hipStream_t);
INST(double)
INST(float)
#undef INST

} // namespace ahip
