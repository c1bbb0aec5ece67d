// solver_path.hpp — part of `template <class T> struct Solver` (solver.hip includes this file INSIDE the struct body, in this order:
// solver_builds, solver_screen, solver_panel, solver_fit, solver_path; one translation unit, several readable files).
// Contents: the path driver: lambda_max / path generation (solver/utils.hpp:7-42), the speculative first pass, invariance, solutions,
// the BASIL loop (solver_base.hpp:435-687), finalize, and the construction of the state from the ABI arguments.
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
                if (on_device(g)) { // multipliers per coefficient, in the object's order (the gradient convention: box mu, one-sided mu >= 0)
                    for (idx t = 0; t < group_sizes[g]; ++t) {
                        const T m = cons_vmu[size_t(groups[g] + t)];
                        if (m != 0) { di.push_back(dual_groups[g] + t); dv.push_back(m); }
                    }
                } else if (host_cons(g)) { // the object's own multipliers
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
        } else if (!lmda_aug.empty()) {
            // adelie/cv.py:255-264 inside the solve (adelie_hip_grpnet_args::lmda_aug_ratios): the fold's own grid above the head
            // of the full-data grid joins the given path; every value kept, descending
            std::vector<double> all(lmda_path.begin(), lmda_path.end());
            for (double r : lmda_aug) {
                const double v = double(lmda_max) * r;
                if (v > lmda_aug_min) all.push_back(v);
            }
            std::sort(all.begin(), all.end(), std::greater<double>());
            lmda_path.resize(all.size());
            for (size_t i = 0; i < all.size(); ++i) lmda_path[i] = T(all[i]);
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
        if (hooks.trace >= 2 && is_glm())
            std::fprintf(stderr, "[irls] %lld iterations: set-up (weights, means, screen-derived) %.1f ms, pin solves %.1f ms (host wall)\n",
                         (long long)cnt.n_irls_iters, t_host[6] * 1e3, t_host[7] * 1e3);
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
        if (cons_on && cons_dev) { // the device objects' multipliers, screened or not (cons_abs_grad_kernel keeps the others)
            d_cons_mu.download(cons_vmu.data(), size_t(p), st);
            if (nv <= 0) sync();
        }
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
        if (cons_dev) { // what the device-side constraint objects hold, and how often they were visited
            d_cons_mu.download(cons_vmu.data(), size_t(p), st);
            d_cons_nvis.download(&n_dev_cons_visits_final, 1, st);
        }
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
        has_pen2 = a->penalty_l2 != nullptr;
        if (has_pen2) penalty2.assign((const T*)a->penalty_l2, (const T*)a->penalty_l2 + G);
        else penalty2 = penalty;
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
        lmda_aug.clear();
        if (a->lmda_aug_ratios && a->n_lmda_aug > 0) {
            if (setup_lmda_path || !setup_lmda_max)
                throw make_core_error("lmda_aug_ratios needs a given lmda_path (setup_lmda_path = false) and setup_lmda_max = true.");
            lmda_aug.assign(a->lmda_aug_ratios, a->lmda_aug_ratios + a->n_lmda_aug);
            lmda_aug_min = a->lmda_aug_min;
        }
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
        if (has_pen2) { // (ABI 8) the separate quadratic factors live in the one-coefficient solves only
            if (!all_scalar) throw make_core_error("penalty_l2 needs groups of one coefficient.");
            if (cov_mode || X->kind == 2) throw make_core_error("penalty_l2 is not offered by the covariance method / the multi-response view.");
            if (a->constraint_kind)
                for (idx g = 0; g < G; ++g)
                    if (a->constraint_kind[g] != 0) throw make_core_error("penalty_l2 is not offered with constraints.");
            for (idx g = 0; g < G; ++g)
                if (!(penalty2[g] >= 0)) throw make_core_error("penalty_l2 must be >= 0.");
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
        if (hooks.group_next_corr >= 0) group_next_corr = hooks.group_next_corr != 0;
        fuse_reduce = !multi() && fused_partials() <= 200;
        if (hooks.speculate >= 0) spec_enabled = hooks.speculate != 0;
        if (hooks.irls_reuse >= 0) irls_reuse = hooks.irls_reuse;
        if (hooks.solve_sums >= 0) plain_solve_sums = hooks.solve_sums != 0;
        if (hooks.step_tail >= 0) step_tail = hooks.step_tail != 0;
        if (hooks.step_means >= 0) step_means_opt = hooks.step_means != 0;
        panel_bsz = hooks.panel_bsz;
        if (cov_mode) { // base state of the covariance method: no intercept, adev_tol = ddev_tol = 0 (state_gaussian_cov.hpp:118)
            engine_panel = false; // the panel engines work on the residual; the Gram engines on C = A[S, S] and its gradient
            glm_kind = ADELIE_HIP_GLM_GAUSSIAN;
            intercept = false;
            adev_tol = 0; ddev_tol = 0;
            rdev_tol = T(a->rdev_tol);
        }
        if (sparse() || std_generic()) engine_panel = false; // Gram engines: C = X_S^T W X_S from launch_gram_csc, residual updated once per fit
        // ... except IRLS on the plain compressed columns (groups of one, or groups of at most 128): the panel engine's sequential form (step over
        // the stored entries, 64-visit diagonal blocks by row-list merges; kernels_sparse.hip) -- a Gram of the whole screen set
        // per IRLS iteration is what such a path spends its time on otherwise (hook ADELIE_HIP_SPARSE_PANEL=0)
        if (sparse() && is_glm() && !cov_mode && hooks.sparse_panel != 0 && (D->std_center == nullptr || hooks.std_panel != 0))
            engine_panel = true; // (its standardized view: with the corrections of the next paragraph)
        // ... and a standardized view of a dense / 2-bit design: the panel engines' sequential form on the BASE design's columns
        // with the view's corrections around every step (changes over the scales + kappa off every row before it, the block's
        // gradient and the diagonal blocks corrected from the raw sums behind it; kernels_sparse.hip), no look-ahead, no
        // speculation (hook ADELIE_HIP_STD_PANEL=0: the view's full-Gram engines)
        if (std_generic() && !cov_mode && hooks.std_panel != 0) {
            engine_panel = true;
            spec_enabled = false;
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
                // the constrained solves live in the panel engines: a design kept sparse has them under IRLS only (compressed-column
                // panel form), a standardized view of a dense / 2-bit design whenever its panel form is on (ADELIE_HIP_STD_PANEL)
                if (sparse() && !engine_panel)
                    throw make_core_error("constraints are not implemented on a design kept sparse; use matrix.sparse(..., resident=\"dense\").");
                if (std_generic() && !engine_panel)
                    throw make_core_error("constraints are not implemented on a lazily standardized design; use matrix.standardize(..., lazy=False).");
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
                engine_panel = !cov_mode; // (covariance method: the Gram group engine carries clips and host visits,
                group_panel = true;       //  solver_gaussian_pin_cov.hpp:287-355,723-763)
                if (cov_mode) {
                    if (max_gs > idx(cd_block_size()))
                        throw make_core_error("constraints are not implemented for problems with groups of more than " +
                                              std::to_string(cd_block_size()) + " coefficients.");
                    all_scalar = false;
                }
                cd_block_min_nv = 1;
                lookahead = false;
                d_clo_g.reserve(G); d_chi_g.reserve(G); d_mu_g.reserve(G);
                d_clo_g.upload(cons_lo.data(), size_t(G), st);
                d_chi_g.upload(cons_hi.data(), size_t(G), st);
                d_mu_g.upload(cons_mu.data(), size_t(G), st);
                d_clo.reserve(p); d_chi.reserve(p); d_cmu.reserve(p);
                // box / one-sided objects on several coefficients: on the device when their description came along (ABI 7)
                if (cons_host && a->constraint_native && a->constraint_va && a->constraint_vb && a->constraint_cfg && !hooks.cons_host) {
                    cons_native.assign(a->constraint_native, a->constraint_native + G);
                    cons_cfg.assign(a->constraint_cfg, a->constraint_cfg + size_t(G) * 5);
                    cons_dev = true; // (on_device() reads cons_native)
                    devcons_list.clear();
                    for (idx g = 0; g < G; ++g)
                        if (on_device(g)) devcons_list.push_back(int32_t(g));
                    cons_dev = !devcons_list.empty();
                    if (cons_dev) {
                        const T* va = static_cast<const T*>(a->constraint_va);
                        const T* vb = static_cast<const T*>(a->constraint_vb);
                        for (int32_t g : devcons_list)
                            for (idx t = 0; t < group_sizes[g]; ++t) {
                                const idx k = groups[g] + t;
                                if (cons_native[g] == ADELIE_HIP_NATIVE_BOX) { // constraint_box.ipp:30-37
                                    if (vb[k] < 0) throw make_core_error("upper must be >= 0.");
                                    if (va[k] > 0) throw make_core_error("lower must be <= 0.");
                                } else {                                       // constraint_one_sided.ipp:74-79
                                    if (std::abs(va[k]) != 1) throw make_core_error("sgn must be a vector of +/-1.");
                                    if (vb[k] < 0) throw make_core_error("b must be >= 0.");
                                }
                            }
                        cons_vmu.assign(size_t(p), T(0));
                        if (a->constraint_vmu) {
                            const T* vm = static_cast<const T*>(a->constraint_vmu);
                            for (int32_t g : devcons_list)
                                for (idx t = 0; t < group_sizes[g]; ++t) cons_vmu[size_t(groups[g] + t)] = vm[groups[g] + t];
                        }
                        d_cons_va.reserve(p); d_cons_vb.reserve(p); d_cons_mu.reserve(p); d_cons_native.reserve(G);
                        d_devcons_list.reserve(devcons_list.size()); d_cons_nvis.reserve(1);
                        d_cons_va.upload(va, size_t(p), st);
                        d_cons_vb.upload(vb, size_t(p), st);
                        d_cons_mu.upload(cons_vmu.data(), size_t(p), st);
                        d_cons_native.upload(cons_native.data(), size_t(G), st);
                        d_devcons_list.upload(devcons_list.data(), devcons_list.size(), st);
                        AHIP_CHECK(hipMemsetAsync(d_cons_nvis.p, 0, sizeof(int64_t), st));
                    }
                }
            }
        }
        // device allocations
        d_r.reserve(n); d_v.reserve(n); d_grad.reserve(p); d_absgrad.reserve(G); d_penalty.reserve(G);
        d_groups.reserve(G); d_gsizes.reserve(G); d_slot.reserve(G);
        d_vcol.reserve(p); d_sbegin.reserve(G); d_ssize.reserve(G); d_actset.reserve(G); d_dcols.reserve(p);
        d_spen.reserve(G); d_beta.reserve(p); d_beta0.reserve(p); d_g.reserve(p); d_vars.reserve(p); d_sxm.reserve(p);
        d_dvals.reserve(p); d_isact.reserve(G); d_voff.reserve(G); d_V.reserve(16); d_sc.reserve(1); d_sums.reserve(16 + 4 * 256);
        d_penalty.upload(penalty.data(), G, st);
        if (has_pen2) {
            d_penalty2.reserve(G); d_spen2.reserve(G);
            d_penalty2.upload(penalty2.data(), G, st);
        }
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
        dev_cons_abs_grad_host(lmda);

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
