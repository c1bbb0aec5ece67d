/* linear_constraint.hpp — CPU restatement of the reference's ConstraintLinear (TEST INFRASTRUCTURE ONLY, part of oracle/).
 *
 * Included by grpnet_oracle.cpp after `newton_solver`, `ConsObject`, `make_core_error` / `make_solver_error` and `idx` exist.
 * Follows, function by function,
 *   adelie_core/constraint/constraint_linear.ipp      (solve :232-497, solve_zero :520-603, sparse-mu helpers :70-140)
 *   adelie_core/constraint/utils.hpp:24-243            (solve_proximal_newton: the driver shared with box / one-sided)
 *   adelie_core/solver/solver_bvls.hpp                 (bounded-variable least squares by coordinate descent: the "NNLS" of
 *                                                       compute_min_mu_resid and solve_zero, optimization/nnls.hpp:144-153)
 *   adelie_core/solver/solver_pinball.hpp              (pinball least squares with screening: the m >= d Newton step)
 *   adelie_core/optimization/pinball_full.hpp:84-118   (dense pinball coordinate descent: the m < d Newton step)
 * The constraint is  lower <= A z <= upper  on z = Q x (the group's coefficients in the design's coordinates), A (m, d)
 * row-major here; the class stores l = -lower >= 0 and u = upper >= 0 like the reference's Python wrapper hands them over
 * (adelie/constraint.py:262-263).  The multipliers are kept sparse in insertion order (`mu_active`, `mu_value`): that order IS
 * the visiting order of the two coordinate-descent sub-solvers, so it is restated, not replaced by a dense vector.
 * Written independently of adelie_amd/constraint.py (numpy: one dual driver in A' mu terms for three classes, dense multipliers
 * with an order list, one dense pinball branch): tests/test_constraint.py compares the two and replays the reference's own
 * recipe (tests/test_constraint.py:73-135) on this one. */
#pragma once

template <class T>
struct LinearCons : ConsObject<T> {
    /* configuration (constraint_linear.hpp:44-56; defaults adelie/constraint.py:272-283) */
    std::vector<T> A;           /* (m, d) row-major */
    std::vector<T> l, u, A_vars; /* l = -lower */
    size_t max_iters = 100, nnls_max_iters = 100000, pinball_max_iters = 100000;
    T tol = T(1e-9), nnls_tol = T(1e-7), pinball_tol = T(1e-7), slack = T(1e-4);
    static constexpr double kEps = 1e-16;  /* constraint_linear.hpp:40 */
    static constexpr double kMax = 1e100;  /* configs.hpp:13 */

    /* state (constraint_linear.hpp:57-63) */
    std::vector<idx> mu_active, mu_active_prev;
    std::vector<T> mu_value, mu_value_prev;
    std::vector<char> in_set, in_set_prev; /* _mu_active_set(_prev) as membership flags */
    std::vector<T> ATmu;

    void init(idx m_, idx d_) {
        this->m = m_; this->d = d_;
        in_set.assign(size_t(m_), 0); in_set_prev.assign(size_t(m_), 0);
        ATmu.assign(size_t(d_), 0);
        if (idx(l.size()) != m_) throw make_core_error("lower must be (m,) where A is (m, d).");
        if (idx(u.size()) != m_) throw make_core_error("upper must be (m,) where A is (m, d).");
        for (T v : u) if (v < 0) throw make_core_error("upper must be >= 0.");
        for (T v : l) if (v < 0) throw make_core_error("lower must be <= 0.");
        if (idx(A_vars.size()) != m_) throw make_core_error("A_vars must be (m,) where A is (m, d).");
        if (tol < 0) throw make_core_error("tol must be >= 0.");
        if (nnls_tol < 0) throw make_core_error("nnls_tol must be >= 0.");
        if (pinball_tol < 0) throw make_core_error("pinball_tol must be >= 0.");
        if (slack <= 0 || slack >= 1) throw make_core_error("slack must be in (0,1).");
    }

    /* ---- MatrixConstraintDense members (matrix_constraint_dense.ipp) on the row-major A ---- */
    const T* row(idx j) const { return A.data() + size_t(j) * size_t(this->d); }
    T rvmul(idx j, const T* v) const { /* A[j] . v */
        T acc = 0;
        const T* a = row(j);
        for (idx i = 0; i < this->d; ++i) acc += a[i] * v[i];
        return acc;
    }
    void rvtmul(idx j, T c, T* out) const { /* out += c A[j] */
        const T* a = row(j);
        for (idx i = 0; i < this->d; ++i) out[i] += c * a[i];
    }
    void tmul(const T* v, T* out) const { /* out = A v   (m,) */
        for (idx j = 0; j < this->m; ++j) out[j] = rvmul(j, v);
    }
    void rmmul(idx j, const T* S, T* out) const { /* out = A[j] S, S (d, d) column-major */
        const idx d = this->d;
        const T* a = row(j);
        for (idx c = 0; c < d; ++c) {
            T acc = 0;
            for (idx r = 0; r < d; ++r) acc += a[r] * S[r + c * d];
            out[c] = acc;
        }
    }

    /* ---- sparse multipliers, constraint_linear.ipp:70-140 ---- */
    void compute_ATmu() {
        std::fill(ATmu.begin(), ATmu.end(), T(0));
        for (size_t i = 0; i < mu_active.size(); ++i) rvtmul(mu_active[i], mu_value[i], ATmu.data());
    }
    void mu_to_dense(std::vector<T>& mu) const {
        std::fill(mu.begin(), mu.end(), T(0));
        for (size_t i = 0; i < mu_active.size(); ++i) mu[size_t(mu_active[i])] = mu_value[i];
    }
    void mu_prune() {
        size_t n = 0;
        for (size_t i = 0; i < mu_active.size(); ++i) {
            const idx k = mu_active[i];
            const T mi = mu_value[i];
            if (std::abs(mi) <= T(kEps)) { in_set[size_t(k)] = 0; continue; }
            mu_active[n] = k; mu_value[n] = mi; ++n;
        }
        mu_active.resize(n); mu_value.resize(n);
    }
    void mu_to_sparse(const std::vector<T>& mu) {
        for (size_t i = 0; i < mu_active.size(); ++i) mu_value[i] = mu[size_t(mu_active[i])];
        for (idx i = 0; i < this->m; ++i) {
            const T mi = mu[size_t(i)];
            if (mi == 0 || in_set[size_t(i)]) continue;
            in_set[size_t(i)] = 1;
            mu_active.push_back(i);
            mu_value.push_back(mi);
        }
        mu_prune();
    }
    void take_active(const std::vector<idx>& active_set, size_t active_size, const std::vector<T>& mu) {
        mu_active.clear(); mu_value.clear();
        std::fill(in_set.begin(), in_set.end(), char(0));
        for (size_t i = 0; i < active_size; ++i) {
            const idx k = active_set[i];
            mu_active.push_back(k); mu_value.push_back(mu[size_t(k)]); in_set[size_t(k)] = 1;
        }
        mu_prune();
    }
    void clear_all() {
        mu_active.clear(); mu_value.clear();
        std::fill(in_set.begin(), in_set.end(), char(0));
        std::fill(ATmu.begin(), ATmu.end(), T(0));
    }

    /* ---- working sets of the two coordinate-descent sub-solvers ---- */
    std::vector<idx> screen_set, active_set;
    std::vector<char> is_screen, is_active;
    size_t screen_size = 0, active_size = 0;
    void sets_from_mu(std::vector<T>& mu) { /* constraint_linear.ipp:301-315,425-438,546-558 */
        const size_t m = size_t(this->m);
        screen_set.assign(m, 0); active_set.assign(m, 0);
        is_screen.assign(m, 0); is_active.assign(m, 0);
        std::fill(mu.begin(), mu.end(), T(0));
        for (size_t i = 0; i < mu_active.size(); ++i) {
            const idx k = mu_active[i];
            screen_set[i] = k; active_set[i] = k;
            is_screen[size_t(k)] = 1; is_active[size_t(k)] = 1;
            mu[size_t(k)] = mu_value[i];
        }
        screen_size = active_size = mu_active.size();
    }

    /* ---- solver_bvls.hpp on X = A^T (d rows, m columns): min 1/2 |resid|^2, lower <= beta <= upper ----
     * coordinate_descent :23-70, solve_active :79-118, fit :127-222, kkt_screen :231-301, solve :310-341.
     * `early` is the early-exit predicate on the running loss; returns the final loss. */
    template <class Early>
    T bvls(std::vector<T>& beta, std::vector<T>& resid, std::vector<T>& grad, T loss, T y_var, const std::vector<T>& lower,
           const std::vector<T>& upper, Early early) {
        const idx m = this->m, d = this->d;
        const size_t kappa = size_t(std::min(m, d));
        size_t iters = 0, n_kkt = 0;
        auto cd = [&](const idx* begin, const idx* end, T& cm, bool add_active) {
            for (const idx* it = begin; it != end; ++it) {
                if (early(loss)) return;
                const idx k = *it;
                const T vk = A_vars[size_t(k)], lk = lower[size_t(k)], uk = upper[size_t(k)];
                const T gk = rvmul(k, resid.data());
                T& bk = beta[size_t(k)];
                const T bk_old = bk;
                const T step = (vk <= 0) ? T(0) : (gk / vk);
                bk = std::min<T>(std::max<T>(bk + step, lk), uk);
                if (bk == bk_old) continue;
                const T del = bk - bk_old;
                const T sds = vk * del * del;
                cm = std::max<T>(cm, sds);
                loss -= del * gk - T(0.5) * sds;
                rvtmul(k, -del, resid.data());
                if (add_active && !is_active[size_t(k)]) {
                    active_set[active_size++] = k;
                    is_active[size_t(k)] = 1;
                }
            }
        };
        auto prune = [&]() {
            size_t n = 0;
            for (size_t i = 0; i < active_size; ++i) {
                const idx k = active_set[i];
                const T bk = beta[size_t(k)];
                if (bk <= lower[size_t(k)] || bk >= upper[size_t(k)]) { is_active[size_t(k)] = 0; continue; }
                active_set[n++] = k;
            }
            active_size = n;
        };
        auto fit = [&]() {
            while (true) {
                ++iters;
                T cm = 0;
                cd(screen_set.data(), screen_set.data() + screen_size, cm, true);
                if (iters >= nnls_max_iters) throw make_solver_error("bvls: max iterations reached!");
                if (cm <= nnls_tol * y_var || early(loss)) { prune(); break; }
                while (true) { /* solve_active */
                    ++iters;
                    T cma = 0;
                    cd(active_set.data(), active_set.data() + active_size, cma, false);
                    if (iters >= nnls_max_iters) throw make_solver_error("bvls: max iterations reached!");
                    if (cma <= nnls_tol * y_var) break;
                }
                prune();
            }
        };
        std::vector<idx> order(size_t(m), 0);
        for (idx i = 0; i < m; ++i) order[size_t(i)] = i;
        while (true) {
            const T loss_prev = loss;
            fit();
            if (early(loss)) return loss;
            if (n_kkt > 0 && std::abs(loss - loss_prev) < T(1e-6) * std::abs(y_var)) return loss;
            /* kkt_screen */
            ++n_kkt;
            tmul(resid.data(), grad.data());
            for (idx k = 0; k < m; ++k) {
                const T g = grad[size_t(k)];
                grad[size_t(k)] = std::max<T>(g, 0) * T(beta[size_t(k)] < upper[size_t(k)]) -
                                  std::min<T>(g, 0) * T(beta[size_t(k)] > lower[size_t(k)]);
            }
            std::sort(order.begin(), order.end(), [&](idx i, idx j) { return grad[size_t(i)] > grad[size_t(j)]; });
            const size_t old = screen_size;
            bool passed = true;
            for (idx j = 0; j < m; ++j) {
                const idx k = order[size_t(j)];
                if (is_screen[size_t(k)] || grad[size_t(k)] <= 0) continue;
                passed = false;
                if (screen_size >= old + kappa) break;
                screen_set[screen_size++] = k;
                is_screen[size_t(k)] = 1;
            }
            if (passed) return loss;
        }
    }
    void sign_bounds(std::vector<T>& lower, std::vector<T>& upper) const { /* constraint_linear.ipp:286-293 */
        lower.resize(size_t(this->m)); upper.resize(size_t(this->m));
        for (idx i = 0; i < this->m; ++i) {
            lower[size_t(i)] = (l[size_t(i)] <= 0) ? T(-kMax) : T(0);
            upper[size_t(i)] = (u[size_t(i)] <= 0) ? T(kMax) : T(0);
        }
    }

    /* ---- solver_pinball.hpp: min over mu of the quadratic model with S = hess through A, penalties l (neg) / u (pos) ----
     * coordinate_descent :16-62, solve_active :68-100, fit :106-176, kkt_screen :182-251, solve :257-290 */
    void pinball(std::vector<T>& beta, std::vector<T>& resid, std::vector<T>& grad, const T* S, T y_var,
                 std::vector<T>& AS, std::vector<T>& ASAT_diag) {
        const idx m = this->m, d = this->d;
        const size_t kappa = size_t(std::min(m, d));
        size_t iters = 0, n_kkt = 0;
        T loss = 0;
        auto cd = [&](const idx* begin, const idx* end, T& cm, bool add_active) {
            for (const idx* it = begin; it != end; ++it) {
                const idx k = *it;
                const T vk = ASAT_diag[size_t(k)], lk = l[size_t(k)], uk = u[size_t(k)];
                const T* AS_k = AS.data() + size_t(k) * size_t(d);
                const T gk = rvmul(k, resid.data());
                T& bk = beta[size_t(k)];
                const T bk_old = bk;
                const T gk0 = gk + vk * bk_old, gk0_lk = gk0 + lk;
                bk = (vk <= 0) ? bk_old : std::copysign(std::max<T>(std::max<T>(-gk0_lk, gk0 - uk), 0), gk0_lk) / vk;
                if (bk == bk_old) continue;
                const T del = bk - bk_old;
                const T sds = vk * del * del;
                cm = std::max<T>(cm, sds);
                loss -= del * gk - T(0.5) * sds;
                for (idx i = 0; i < d; ++i) resid[size_t(i)] -= del * AS_k[i];
                if (add_active && !is_active[size_t(k)]) {
                    active_set[active_size++] = k;
                    is_active[size_t(k)] = 1;
                }
            }
        };
        auto prune = [&]() {
            size_t n = 0;
            for (size_t i = 0; i < active_size; ++i) {
                const idx k = active_set[i];
                if (beta[size_t(k)] == 0) { is_active[size_t(k)] = 0; continue; }
                active_set[n++] = k;
            }
            active_size = n;
        };
        auto fit = [&]() {
            while (true) {
                ++iters;
                T cm = 0;
                cd(screen_set.data(), screen_set.data() + screen_size, cm, true);
                if (iters >= pinball_max_iters) throw make_solver_error("pinball: max iterations reached!");
                if (cm <= pinball_tol * y_var) { prune(); break; }
                while (true) {
                    ++iters;
                    T cma = 0;
                    cd(active_set.data(), active_set.data() + active_size, cma, false);
                    if (iters >= pinball_max_iters) throw make_solver_error("pinball: max iterations reached!");
                    if (cma <= pinball_tol * y_var) break;
                }
                prune();
            }
        };
        std::vector<idx> order(size_t(m), 0);
        for (idx i = 0; i < m; ++i) order[size_t(i)] = i;
        while (true) {
            const T loss_prev = loss;
            fit();
            if (n_kkt > 0 && std::abs(loss - loss_prev) < T(1e-6) * std::abs(y_var)) return;
            ++n_kkt;
            tmul(resid.data(), grad.data());
            for (idx k = 0; k < m; ++k)
                grad[size_t(k)] = std::max<T>(grad[size_t(k)] - u[size_t(k)], -l[size_t(k)] - grad[size_t(k)]);
            std::sort(order.begin(), order.end(), [&](idx i, idx j) { return grad[size_t(i)] > grad[size_t(j)]; });
            const size_t old = screen_size;
            bool passed = true;
            for (idx j = 0; j < m; ++j) {
                const idx k = order[size_t(j)];
                if (is_screen[size_t(k)] || grad[size_t(k)] <= 0) continue;
                passed = false;
                if (screen_size >= old + kappa) break;
                screen_set[screen_size++] = k;
                is_screen[size_t(k)] = 1;
                T* AS_k = AS.data() + size_t(k) * size_t(d);
                rmmul(k, S, AS_k);
                ASAT_diag[size_t(k)] = std::max<T>(rvmul(k, AS_k), 0);
            }
            if (passed) return;
        }
    }
    /* optimization/pinball_full.hpp:84-118 on the dense (m, m) `quad` (symmetric) */
    void pinball_full(const std::vector<T>& quad, T y_var, std::vector<T>& x, std::vector<T>& grad) const {
        const idx m = this->m;
        size_t iters = 0;
        while (iters < pinball_max_iters) {
            T cm = 0;
            ++iters;
            for (idx i = 0; i < m; ++i) {
                const T qii = quad[size_t(i) + size_t(i) * size_t(m)];
                const T xo = x[size_t(i)], gi0 = grad[size_t(i)] + qii * xo;
                x[size_t(i)] = std::copysign(std::max<T>(std::max<T>(-l[size_t(i)] - gi0, gi0 - u[size_t(i)]), 0), gi0 + l[size_t(i)]) / qii;
                const T del = x[size_t(i)] - xo;
                if (del == 0) continue;
                cm = std::max<T>(cm, qii * del * del);
                for (idx r = 0; r < m; ++r) grad[size_t(r)] -= del * quad[size_t(r) + size_t(i) * size_t(m)];
            }
            if (cm < y_var * pinball_tol) return;
        }
        throw make_solver_error("StatePinballFull: max iterations reached!");
    }

    /* ---- ConstraintBase members ---- */
    void gradient(const T*, T* out) override { /* constraint_linear.ipp:499-506 */
        for (idx i = 0; i < this->d; ++i) out[i] = ATmu[size_t(i)];
    }
    void dual_dense(T* mu_out) override {
        for (idx i = 0; i < this->m; ++i) mu_out[i] = 0;
        for (size_t i = 0; i < mu_active.size(); ++i) mu_out[mu_active[i]] = mu_value[i];
    }
    T solve_zero(const T* v) override { /* constraint_linear.ipp:520-603 */
        const idx m = this->m, d = this->d;
        std::vector<T> mu(static_cast<size_t>(m), T(0)), resid(static_cast<size_t>(d), T(0)), grad(static_cast<size_t>(m), T(0)), lower, upper;
        sets_from_mu(mu);
        T vsq = 0, loss = 0;
        for (idx i = 0; i < d; ++i) {
            resid[size_t(i)] = v[i] - ATmu[size_t(i)];
            loss += resid[size_t(i)] * resid[size_t(i)];
            vsq += v[i] * v[i];
        }
        loss *= T(0.5);
        sign_bounds(lower, upper);
        loss = bvls(mu, resid, grad, loss, vsq, lower, upper, [](T) { return false; });
        take_active(active_set, active_size, mu);
        for (idx i = 0; i < d; ++i) ATmu[size_t(i)] = v[i] - resid[size_t(i)];
        return std::sqrt(std::max<T>(2 * loss, 0));
    }

    /* constraint_linear.ipp:232-497 with the driver constraint/utils.hpp:24-243 written out */
    void solve(T* x, const T* quad, const T* linear, T l1, T l2, const T* Q) override {
        const idx m = this->m, d = this->d;
        T vsq = 0;
        for (idx i = 0; i < d; ++i) vsq += linear[i] * linear[i];
        const T v_norm = std::sqrt(vsq);
        if (v_norm <= l1) { /* :252-257 */
            for (idx i = 0; i < d; ++i) x[i] = 0;
            clear_all();
            return;
        }
        std::vector<T> grad_prev(size_t(d), 0), grad(size_t(d), 0), ATmu_prev(size_t(d), 0), mu(size_t(m), 0),
            pinball_grad(size_t(m), 0), nnls_grad(size_t(m), 0), xb1(size_t(d), 0), xb2(size_t(d), 0), mu_resid(size_t(d), 0),
            hess(size_t(d) * size_t(d), 0), alpha_tmp(size_t(d), 0), alpha(size_t(d), 0), Qv(size_t(d), 0), lower, upper;
        for (idx i = 0; i < d; ++i) { /* Qv = v Q^T, utils.hpp:70 */
            T acc = 0;
            for (idx j = 0; j < d; ++j) acc += linear[j] * Q[i + j * d];
            Qv[size_t(i)] = acc;
        }
        auto compute_min_mu_resid = [&](bool prev_valid_old, bool is_init) -> T { /* :275-351 */
            T dist = 0;
            for (idx i = 0; i < d; ++i) dist += (Qv[size_t(i)] - ATmu[size_t(i)]) * (Qv[size_t(i)] - ATmu[size_t(i)]);
            if (dist <= l1 * l1) return T(0);
            sign_bounds(lower, upper);
            sets_from_mu(mu);
            std::vector<T>& resid = grad; /* (the reference reuses `grad` as Qmu_resid) */
            T loss = 0;
            for (idx i = 0; i < d; ++i) {
                resid[size_t(i)] = Qv[size_t(i)] - ATmu[size_t(i)];
                loss += resid[size_t(i)] * resid[size_t(i)];
            }
            loss *= T(0.5);
            loss = bvls(mu, resid, nnls_grad, loss, v_norm * v_norm, lower, upper, [&](T ls) { return 2 * ls <= l1 * l1; });
            const T nsq = 2 * loss;
            if ((!is_init && !prev_valid_old) || nsq <= l1 * l1) {
                take_active(active_set, active_size, mu);
                for (idx i = 0; i < d; ++i) ATmu[size_t(i)] = Qv[size_t(i)] - resid[size_t(i)];
            }
            return nsq;
        };
        auto save_prev = [&](bool in_ellipse) { /* :470-477 */
            in_set_prev = in_set; mu_active_prev = mu_active; mu_value_prev = mu_value;
            ATmu_prev = ATmu;
            if (in_ellipse) std::fill(grad_prev.begin(), grad_prev.end(), T(0));
            else grad_prev = grad;
        };
        auto convergence = [&](bool in_ellipse) -> T { /* :390-400 */
            T acc = 0;
            for (idx i = 0; i < d; ++i) {
                const T dm = ATmu[size_t(i)] - ATmu_prev[size_t(i)];
                acc += dm * (in_ellipse ? grad_prev[size_t(i)] : (grad_prev[size_t(i)] - grad[size_t(i)]));
            }
            return std::abs(acc / T(d));
        };
        auto backtrack = [&](T step) { /* :364-382 */
            for (size_t i = 0; i < mu_active_prev.size(); ++i) mu[size_t(mu_active_prev[i])] = (1 - step) * mu_value_prev[i];
            for (size_t i = 0; i < mu_active.size(); ++i) {
                const idx k = mu_active[i];
                mu_value[i] = step * mu_value[i] + (in_set_prev[size_t(k)] ? mu[size_t(k)] : T(0));
            }
            for (size_t i = 0; i < mu_active_prev.size(); ++i) {
                const idx k = mu_active_prev[i];
                if (in_set[size_t(k)]) continue;
                in_set[size_t(k)] = 1;
                mu_active.push_back(k);
                mu_value.push_back(mu[size_t(k)]);
            }
            compute_ATmu();
        };
        auto newton_step = [&](T var) { /* :402-469 */
            if (m < d) {
                mu_to_dense(mu);
                std::vector<T> AH(size_t(m) * size_t(d), T(0)), hs(size_t(m) * size_t(m), T(0));
                for (idx j = 0; j < m; ++j) rmmul(j, hess.data(), AH.data() + size_t(j) * size_t(d));
                for (idx r = 0; r < m; ++r)
                    for (idx c = 0; c < m; ++c) hs[size_t(r) + size_t(c) * size_t(m)] = rvmul(c, AH.data() + size_t(r) * size_t(d));
                tmul(grad.data(), pinball_grad.data());
                pinball_full(hs, var, mu, pinball_grad);
                mu_to_sparse(mu);
            } else {
                sets_from_mu(mu);
                std::vector<T> AS(size_t(m) * size_t(d), 0), diag(size_t(m), 0);
                for (size_t i = 0; i < mu_active.size(); ++i) {
                    const idx k = mu_active[i];
                    T* AS_k = AS.data() + size_t(k) * size_t(d);
                    rmmul(k, hess.data(), AS_k);
                    diag[size_t(k)] = std::max<T>(rvmul(k, AS_k), 0);
                }
                pinball(mu, grad, pinball_grad, hess.data(), var, AS, diag); /* (resid aliases grad, :423) */
                take_active(active_set, active_size, mu);
            }
            compute_ATmu();
        };

        bool x_init_zero = true;
        for (idx i = 0; i < d; ++i) x_init_zero = x_init_zero && x[i] == 0;
        bool prev_valid = false, zero_checked = false;
        T resid_norm_prev = -1;
        if (x_init_zero) { /* utils.hpp:78-83 */
            zero_checked = true;
            if (compute_min_mu_resid(false, true) <= l1 * l1) return;
        }
        size_t iters = 0;
        while (iters < max_iters) {
            ++iters;
            for (idx j = 0; j < d; ++j) { /* mu_resid = linear - ATmu Q, :270-273 */
                T acc = 0;
                for (idx i = 0; i < d; ++i) acc += ATmu[size_t(i)] * Q[i + j * d];
                mu_resid[size_t(j)] = linear[j] - acc;
            }
            T rn2 = 0;
            for (idx i = 0; i < d; ++i) rn2 += mu_resid[size_t(i)] * mu_resid[size_t(i)];
            const T resid_norm = std::sqrt(rn2), resid_norm_sq = resid_norm * resid_norm;
            T x_norm = -1;
            bool in_ellipse = resid_norm <= l1;
            if (!in_ellipse) { /* compute_primal, utils.hpp:85-93 */
                size_t nit;
                newton_solver<T>(d, quad, mu_resid.data(), l1, l2, T(1e-12), size_t(100000), x, nit, xb1.data(), xb2.data());
                T xn = 0;
                for (idx i = 0; i < d; ++i) xn += x[i] * x[i];
                x_norm = std::sqrt(xn);
                in_ellipse = x_norm <= 0;
                if (l1 <= 0) /* (newton.hpp:72-75 returns before it fills its buffers; what the hessian below reads from them) */
                    for (idx i = 0; i < d; ++i) { xb1[size_t(i)] = quad[i] + l2; xb2[size_t(i)] = 1 / (xb1[size_t(i)] * x_norm + l1); }
            }
            if (in_ellipse) {
                if (iters == 1 && x_init_zero) { /* utils.hpp:117-120 */
                    for (idx i = 0; i < d; ++i) x[i] = 0;
                    return;
                }
                if (prev_valid && convergence(true) <= tol) { /* :124-129 */
                    for (idx i = 0; i < d; ++i) x[i] = 0;
                    return;
                }
                if (!zero_checked) { /* :136-164 */
                    zero_checked = true;
                    const bool prev_valid_old = prev_valid;
                    if (!prev_valid_old) {
                        resid_norm_prev = resid_norm;
                        prev_valid = true;
                        save_prev(true);
                    }
                    if (compute_min_mu_resid(prev_valid_old, false) <= l1 * l1) {
                        for (idx i = 0; i < d; ++i) x[i] = 0;
                        return;
                    }
                    if (!prev_valid_old) continue;
                }
                if (!prev_valid || (resid_norm_prev <= l1 * T(0.9999)) || (resid_norm > l1 * T(1.0001)))
                    throw make_core_error("Possibly an unexpected error! Previous iterate should have been properly initialized. ");
                const T target = (1 - slack) * l1 + slack * resid_norm_prev; /* :177-185 */
                T a = 0, bq = 0;
                for (idx i = 0; i < d; ++i) {
                    const T dm = ATmu[size_t(i)] - ATmu_prev[size_t(i)];
                    a += dm * dm;
                    T rq = 0; /* (mu_resid Q^T)_i, constraint_linear.ipp:357-363 */
                    for (idx j = 0; j < d; ++j) rq += mu_resid[size_t(j)] * Q[i + j * d];
                    bq += rq * dm;
                }
                const T c = resid_norm_sq - target * target;
                const T t_star = (-bq + std::sqrt(std::max<T>(bq * bq - a * c, 0))) / a;
                backtrack(std::min<T>(std::max<T>(1 - t_star, 0), 1));
                continue;
            }
            for (idx i = 0; i < d; ++i) { /* compute_gradient: grad = x Q^T, :383-385 */
                T acc = 0;
                for (idx j = 0; j < d; ++j) acc += x[j] * Q[i + j * d];
                grad[size_t(i)] = acc;
            }
            /* compute_hard_optimality is `false` for this class (:386-391) */
            if (prev_valid && convergence(false) <= tol) return;
            resid_norm_prev = resid_norm;
            prev_valid = true;
            save_prev(false);
            /* dual Hessian, utils.hpp:208-237 */
            for (idx i = 0; i < d; ++i) alpha_tmp[size_t(i)] = x[i] * xb2[size_t(i)] / x_norm;
            T ks = 0;
            for (idx i = 0; i < d; ++i) ks += x[i] * xb1[size_t(i)] * alpha_tmp[size_t(i)];
            const T kappa = 1 / ks;
            for (idx i = 0; i < d; ++i) {
                T acc = 0;
                for (idx j = 0; j < d; ++j) acc += alpha_tmp[size_t(j)] * Q[i + j * d];
                alpha[size_t(i)] = acc;
            }
            const T l1kn = l1 * kappa * x_norm;
            for (idx c2 = 0; c2 < d; ++c2)
                for (idx r = c2; r < d; ++r) {
                    T acc = 0;
                    for (idx j = 0; j < d; ++j) acc += Q[r + j * d] * xb2[size_t(j)] * Q[c2 + j * d];
                    const T h = x_norm * acc + l1kn * alpha[size_t(r)] * alpha[size_t(c2)];
                    hess[size_t(r) + size_t(c2) * size_t(d)] = h;
                    hess[size_t(c2) + size_t(r) * size_t(d)] = h;
                }
            T xy = 0, s1 = 0, s2 = 0;
            for (idx j = 0; j < d; ++j) { /* alpha_tmp = x Q */
                T xq = 0;
                for (idx i = 0; i < d; ++i) xq += x[i] * Q[i + j * d];
                alpha_tmp[size_t(j)] = xq;
            }
            for (idx j = 0; j < d; ++j) {
                xy += x[j] * alpha_tmp[size_t(j)];
                s1 += alpha_tmp[size_t(j)] * alpha_tmp[size_t(j)] / xb2[size_t(j)];
                s2 += x[j] * x[j] * xb2[size_t(j)];
            }
            T var = (s1 - (xy * xy) / ((x_norm * x_norm) / (l1 * kappa) + s2)) / x_norm;
            var = std::max<T>(var, 0);
            newton_step(var);
        }
        throw make_solver_error("ConstraintBase: proximal newton max iterations reached!");
    }
};
