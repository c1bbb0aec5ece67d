// solver_builds.hpp — part of `template <class T> struct Solver` (solver.hip includes this file INSIDE the struct body, in this order:
// solver_builds, solver_screen, solver_panel, solver_fit, solver_path; one translation unit, several readable files).
// Contents: state of a path solve (host mirrors, device working set) and the builders of the panel engine's cached blocks:
// diagonal blocks (build_stale_blocks), cross blocks of the look-ahead (build_stale_cross), strips of new members (build_stale_strips).
    adelie_hip_design* D = nullptr;
    hipStream_t st = nullptr;
    idx n, p, G;
    // ---- static inputs (host copies) ----
    std::vector<idx> groups, group_sizes;
    std::vector<T> penalty;
    std::vector<T> penalty2; // adelie_hip_grpnet_args::penalty_l2 (factors of the quadratic part), == penalty when absent
    bool has_pen2 = false;
    std::vector<T> h_spen2;
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
    // ... of which the built-in box / one-sided classes (constraint_native 4 / 5) on groups of <= 64 coefficients run on the
    // device (kernels_cons.hip): same block structure, the visit is a one-wavefront launch instead of a host round trip, the
    // multipliers live in d_cons_mu (per design coefficient) and abs_grad / the duals read them there
    bool cons_dev = false;
    std::vector<int32_t> cons_native;          // (G,) or empty
    std::vector<double> cons_cfg;              // (G, 5)
    std::vector<int32_t> devcons_list;         // the groups with on_device()
    std::vector<T> cons_vmu;                   // (p,) host mirror of d_cons_mu
    DevBuf<T> d_cons_va, d_cons_vb, d_cons_mu; // (p,)
    DevBuf<int32_t> d_cons_native, d_devcons_list;
    DevBuf<int64_t> d_cons_nvis;
    int64_t n_host_cons_visits = 0, n_dev_cons_visits_final = 0;
    bool on_device(idx g) const {
        return cons_dev && host_cons(g) && (cons_native[g] == ADELIE_HIP_NATIVE_BOX || cons_native[g] == ADELIE_HIP_NATIVE_ONE_SIDED) &&
               group_sizes[g] <= 64;
    }
    adelie_hip_glm_callbacks glm_cb{};       // glm_kind == CALLBACK: the user's GlmBase subclass, evaluated on the host
    std::vector<T> cb_eta, cb_grad, cb_hess, cb_z;
    idx max_gs = 1;
    bool all_scalar = true;
    // ---- dynamic host state ----
    T lmda_max;
    std::vector<T> lmda_path;
    std::vector<double> lmda_aug; // adelie_hip_grpnet_args::lmda_aug_ratios (a CV fold's own grid, joined to lmda_path once lmda_max is known)
    double lmda_aug_min = 0;
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
    DevBuf<T> d_spen2, d_penalty2; // (penalty_l2: per screen group / per group)
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
    // IRLS on the full-Gram engines (designs kept sparse, standardized views; groups of one): the screen set's Gram is kept
    // while the weights have drifted by at most `irls_reuse` since it was built (the rule of the panel engines' blocks,
    // note_weight_drift) instead of being rebuilt per IRLS iteration.  A kept Gram only carries the coupling INSIDE a pass:
    // every pass then starts from the exact gradient of the current residual (run_block_passes: residual update with the
    // pass's changes, X_S' W r), so the fixed point is the exact one and the iterates inside a pass differ by O(theta |delta|).
    uint64_t gram_version = 0; // w_version the bulk of C was built under (0: none)
    bool gram_stale = false;   // C belongs to older weights than the fit's: refresh the gradient per pass
    DevBuf<T> d_beta_ref;      // coefficients the residual currently reflects (stale mode)
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
    // IRLS with groups of one on the panel engines: A_jj = the diagonal of the block that carries coordinate j, written by the
    // block's build (build_stale_blocks) instead of a square sweep over the screen columns per IRLS iteration (config 4: 107
    // sweeps of 1.7 ms per path).  A block kept under the weight-drift rule keeps its variances with it; the fixed point of the
    // passes does not depend on A (the update beta <- S(g + A beta, l1) / (A + l2) has the KKT point as its fixed point for any
    // A > 0).  One vector per table (screen order / activation order): the builds of one table run while the passes of the
    // other read theirs.  Hook ADELIE_HIP_IRLS_REUSE=0 makes every block, and so every variance, current.
    bool vars_from_blocks() const { return is_glm() && all_scalar && !multi() && !cov_mode; }
    DevBuf<T> d_vars_act;
    T* vb_vars = nullptr;               // (set around a build_stale_blocks call: destination, the pass's column list, its visiting list)
    const int32_t* vb_cols_all = nullptr;
    const int32_t* vb_list = nullptr;
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
            else if (!step_means_now) gram_block_batch(cur_w, cols_base, sb, cur_xm, pool + size_t(j0) * SL * SL, sidx);
            else if (!dense() && syrk_batch_snp_brings_means(sb)) // the MFMA build computes its columns' means on the side
                gram_block_batch(cur_w, cols_base, sb, nullptr, pool + size_t(j0) * SL * SL, sidx,
                                 d_xmb[pool == d_Dpool.p ? 0 : 1].reserve(size_t(p) + 8));
            else gram_block_batch(cur_w, cols_base, sb, batch_means(cols_base, sb, sidx, pool == d_Dpool.p ? 0 : 1),
                                  pool + size_t(j0) * SL * SL, sidx);
            if (vb_vars && !multi()) // IRLS, groups of one: the variances of a block's coordinates are its diagonal (vars_from_blocks)
                launch_block_diag_vars<T>(pool + size_t(j0) * SL * SL, sb, SL, int32_t(cols_base - vb_cols_all), vb_list, vb_vars,
                                          sidx == 0 ? st : (sidx >= 2 ? st_x[sidx - 2] : st2));
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
    // Means mode (solver_screen.hpp::step_means_now): nobody sweeps the screen columns for their means any more, and a block is
    // centred with the means of ITS weights -- so a batch of builds first takes the weighted means of its own columns (one sweep
    // per run of consecutive blocks, on the build's stream) into a by-column vector of its table (screen order / activation
    // order: the two tables build concurrently and share columns).  Only stale blocks pay: ~a fifth of the columns the per-
    // iteration sweep went over.
    DevBuf<T> d_xmb[2], d_xmb_tmp[2 + kMaxExtra + 1], d_xmb_work[2 + kMaxExtra + 1];
    const T* batch_means(const int32_t* cols_base, const SyrkBatch& sb, int side, int table) {
        hipStream_t gs = side == 0 ? st : (side >= 2 ? st_x[side - 2] : st2);
        const int SL = cd_block_size();
        T* xmb = d_xmb[table].reserve(size_t(p) + 8);
        T* tmp = d_xmb_tmp[side].reserve(size_t(16) * SL + 8);
        T* work = d_xmb_work[side].reserve(size_t(sweep_work_elems(n, int64_t(16) * SL)));
        int y = 0;
        while (y < sb.count) { // runs of blocks that stand next to each other in the pass's column list
            int y1 = y + 1;
            int64_t ncols = sb.nb[y];
            while (y1 < sb.count && sb.off[y1] == sb.off[y1 - 1] + sb.nb[y1 - 1]) ncols += sb.nb[y1++];
            const int32_t* cols = cols_base + sb.off[y];
            launch_sweep_snp<T>(D->snp(), static_cast<const T*>(D->impute), cur_w, tmp, 0, ncols, cols, nullptr, nullptr, false, work, gs);
            launch_scatter<T>(tmp, cols, ncols, xmb, gs);
            y = y1;
        }
        return xmb;
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
    // Round 5 (profiles/r05_irls_reuse.txt, same script): theta = 0 / 0.01 / 0.1 / 0.2 / 0.4 build 54.4 / 20.9 / 10.7 / 7.3 / 4.5 k
    // blocks, 5.69 / 4.21 / 3.78 / 3.60 / 3.48 s, max |delta beta| against theta = 0 over the path - / 1.1e-9 / 2.3e-8 / 5.2e-8 /
    // 1.1e-7 on coefficients of size 0.25, the same 222 IRLS iterations and 585 passes throughout; at 0.2 one group enters the
    // screen set one lambda apart.  Default 0.1: identical sets, 40 times below the stated tolerance of 1e-6.
    double irls_reuse = 0.1;
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
    // sequential panel form: the stand-alone solve sums the step's slice partials itself instead of a panel_reduce launch per
    // block.  Measured SLOWER on config 4 (cd 2787-2794 against 2737-2739 ms with 489 partials per column, 2710-2729 against
    // 2667-2677 ms with 245): one compute unit pulls the 125-250 KB of partials at ~60 GB/s, which costs more than the
    // 64-workgroup reduce launch (4.7 us) and its boundary.  Off; hook ADELIE_HIP_SOLVE_SUMS=1 (A/B).
    bool plain_solve_sums = false;
    // sequential panel form on a 2-bit design: the step's last eight workgroups sum its partials (StepTail) -- no panel_reduce
    // launch per block.  A/B hook ADELIE_HIP_STEP_TAIL=0.
    bool step_tail = true;
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
    bool strips_apply() const { return strip_builds && dense() && !is_glm() && D->std_center == nullptr; }
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
    // sparse design kept sparse (adelie_hip_design_create_csc): sweeps, Gram rows and residual updates walk its compressed
    // forms; the solve runs on the full-Gram engines (the panel engines stream dense column slices)
    bool sparse() const { return D->kind == 3; }
    DevBuf<T> d_sp_delta; // p zeros between two residual updates
    // standardized view over a dense or 2-bit design (adelie_hip_design_create_standardized): the base kernels run on the raw
    // matrix, centring and scaling are applied around them (solver_screen.hpp: sweep / gram / axpy_cols); Gram engines only
    bool std_generic() const { return D->std_center != nullptr && D->kind != 3; }
    DevBuf<T> d_std_tmp, d_std_coef;
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

