// solver_fit.hpp — part of `template <class T> struct Solver` (solver.hip includes this file INSIDE the struct body, in this order:
// solver_builds, solver_screen, solver_panel, solver_fit, solver_path; one translation unit, several readable files).
// Contents: one fit at one lambda: pin_solve (solver_gaussian_pin_naive.hpp:217-401), the Gaussian / covariance fits, the GLM members
// and the IRLS loop (solver_glm_naive.hpp:234-459), the null model.
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
        cp.spen2 = has_pen2 ? d_spen2.p : nullptr;
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
        open_from_grad = grad_fresh && open_from_grad_opt && !is_glm() && !cov_mode && r_dev == d_r.p && !resume && !std_generic();
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
            run_block_passes(cp, sc, (is_glm() && gram_stale) ? r_dev : nullptr);
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
            // constraint objects solved on the device (kernels_cons.hip): the errors of constraint/utils.hpp and its sub-solvers
            if (sc.status == CD_CONS_PN) throw make_solver_error("ConstraintBase: proximal newton max iterations reached!");
            if (sc.status == CD_CONS_QP_BOX) throw make_solver_error("StatePinballFull: max iterations reached!");
            if (sc.status == CD_CONS_QP_NNQP) throw make_solver_error("StateNNQPFull: max iterations reached!");
            if (sc.status == CD_CONS_UNEXPECTED)
                throw make_core_error("Possibly an unexpected error! Previous iterate should have been properly initialized. ");
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
        if (!defer_fit_output) d_beta.download(screen_beta.data(), size_t(nv), st);
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
        o.intercept = T(intercept) * (y_mean_pin + rsum_io);
        // the single kernel interleaves active and screen passes; split the wall time by visit counts
        const double va = double(sc.n_visits_active), vs = double(sc.n_visits_screen);
        o.t_active = (va + vs) > 0 ? t_cd * va / (va + vs) : 0;
        o.t_screen = t_cd - o.t_active;
        if (!defer_fit_output) assemble_fit_output(o, true);
        return o;
    }

    // The coefficient rows a fit hands back (pin_naive:359-394).  Under IRLS only the LAST pin solve of a lambda needs them: the
    // iterations before it read the intercept and nothing else, and 32 k coefficients downloaded, merged and pushed one by one
    // per IRLS iteration left the device idle for 0.17 ms each (profiles/r06_cfg4_timeline.txt: 36 ms per config-4 path).
    // glm_fit sets defer_fit_output and calls this once per lambda (`have_beta` false: the coefficients are still on the device).
    bool defer_fit_output = false;
    void assemble_fit_output(FitOut<T>& o, bool have_beta) {
        if (!have_beta && nv > 0) {
            d_beta.download(screen_beta.data(), size_t(nv), st);
            sync();
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
        struct DeferGuard { bool& f; ~DeferGuard() { f = false; } } defer_guard{defer_fit_output};
        defer_fit_output = true; // (the coefficient rows are assembled once, below, from the last pin solve of this lambda)
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
            step_means_now = false;
            if (nv > 0) {
                // 2-bit designs on the panel engine: the steps produce the means of the columns they read under the current weights
                // and the block builds take those of their own columns (solver_screen.hpp::step_means_now) -- no sweep here
                step_means_now = panel_mode() && step_means_possible();
                if (multi()) { // the view's sweep covers all columns in one pass over X; pick the screen values out of it
                    d_mxm.reserve(size_t(p));
                    sweep(d_irls_w.p, d_mxm.p, nullptr, p, nullptr, nullptr);
                    launch_gather<T>(d_mxm.p, d_vcol.p, nv, d_g.p, st);
                } else if (!step_means_now) {
                    sweep(d_irls_w.p, d_g.p, d_vcol.p, nv, nullptr, nullptr); // means by value
                }
                // groups of one on the panel engines: the by-column means stay on the device (scatter through the value -> column map);
                // the host mirrors of the screen means are not read under IRLS (a GLM state hands none back)
                const bool means_on_device = panel_mode() && all_scalar && !multi();
                std::vector<T> m(means_on_device ? 0 : nv);
                if (!means_on_device) d_g.download(m.data(), size_t(nv), st);
                T drift = T(1e30);
                const bool gram_keep_mode = irls_reuse > 0 && all_scalar && !panel_mode() && !cov_mode && !multi() &&
                                            (sparse() || std_generic()) && nv >= cd_block_min_nv;
                const bool track = irls_reuse > 0 && all_scalar && (panel_mode() || gram_keep_mode);
                if (track) { // how far the weights moved since the previous iteration (and keep a copy for the next one)
                    d_irls_w_prev.reserve(size_t(n));
                    if (!irls_w_prev_valid) AHIP_CHECK(hipMemsetAsync(d_irls_w_prev.p, 0, size_t(n) * sizeof(T), st));
                    launch_rel_change<T>(d_irls_w.p, d_irls_w_prev.p, n, d_sums.p + 15, st);
                    AHIP_CHECK(hipMemcpyAsync(&drift, d_sums.p + 15, sizeof(T), hipMemcpyDeviceToHost, st));
                }
                if (means_on_device) {
                    if (!step_means_now) launch_scatter<T>(d_g.p, d_vcol.p, nv, d_irls_xm.p, st);
                    if (track) sync(); // (the drift decides which blocks are kept)
                } else {
                    sync();
                    for (idx ss = 0; ss < idx(screen_set.size()); ++ss) {
                        const idx g = screen_set[ss];
                        for (idx t = 0; t < group_sizes[g]; ++t) irls_xm_host[groups[g] + t] = m[screen_begins[ss] + t];
                    }
                    d_irls_xm.upload(irls_xm_host.data(), size_t(p), st);
                }
                ++w_version; // diagonal blocks built from here on belong to this iteration's weights
                if (track) {
                    note_weight_drift(irls_w_prev_valid ? double(drift) : 1e300);
                    irls_w_prev_valid = true;
                }
                const bool keep = gram_keep_mode && gram_version != 0 && gram_version >= min_usable_version && gram_nv > 0 &&
                                  gram_nv <= nv;
                // (groups of one on the panel engines: no Gram, no eigenbases; the (1, 1) transforms of the host mirror stay as they
                // are -- 39 k one-element vectors freed and re-allocated per IRLS iteration were a third of the set-up's host time)
                if (!keep && !(panel_mode() && all_scalar)) {
                    gram_nv = 0;
                    v_used = 0;
                    screen_transforms.clear();
                }
                if (panel_mode()) {
                    update_vars_panel(d_irls_w.p, d_irls_xm.p, irls_xm_host, 0);
                } else if (keep) { // rows of the members that joined since, under the current weights; the means of ALL values
                    // are the current ones (the gradient's centring and the residual-sum tracking read them)
                    update_gram_and_vars(d_irls_w.p, d_irls_xm.p, irls_xm_host, size_t(gram_nv));
                    d_sxm.upload(m.data(), size_t(nv), st);
                    for (idx a = 0; a < nv; ++a) screen_X_means[a] = m[a];
                    gram_stale = true;
                    sync(); // (`m` goes out of scope below)
                } else {
                    update_gram_and_vars(d_irls_w.p, d_irls_xm.p, irls_xm_host, 0);
                    gram_version = gram_keep_mode ? w_version : 0;
                    gram_stale = false;
                }
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
                assemble_fit_output(po, false);
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

