// solver_screen.hpp — part of `template <class T> struct Solver` (solver.hip includes this file INSIDE the struct body, in this order:
// solver_builds, solver_screen, solver_panel, solver_fit, solver_path; one translation unit, several readable files).
// Contents: device primitives (sweep, panel step, Gram blocks), update_abs_grad, the screen-derived quantities of new screen groups
// (solver_gaussian_naive.hpp:41-176), the partition of a visiting list into blocks, screening (solver_base.hpp:273-403) and KKT.
    // ---------------------------------------------------------------------------------------------------------
    void sweep(const T* v, T* out, const int32_t* cols, idx ncols, const T* sub_scale, const T* sub_vec,
               bool square = false) {
        if (multi()) { // only the full sweep of the Gaussian path is needed on the view (intercept off: no centring epilogue)
            if (cols || ncols != p || sub_vec || square) throw make_core_error("unsupported sweep on a multi-response view.");
            const MultiView<T> mv = D->multi<T>();
            launch_multi_sweep<T>(mv, v, out, d_work_sweep.reserve(size_t(multi_sweep_work_elems<T>(mv))), st);
            return;
        }
        if (std_generic()) { // raw sweep(s) of the base design, then the view's epilogue (kernels_sparse.hip, header)
            const T* ce = static_cast<const T*>(D->std_center);
            const T* is = static_cast<const T*>(D->std_iscale);
            T* tmp = d_std_tmp.reserve(size_t(2 * ncols + kVecSumScratch));
            T *raw = tmp, *raw_plain = tmp + ncols, *vsum = tmp + 2 * ncols;
            T* work = d_work_sweep.reserve(size_t(sweep_work_elems(n, ncols)));
            auto base = [&](T* dst, bool sq) {
                if (dense()) launch_sweep<T>(D->dense<T>(), v, dst, 0, ncols, cols, nullptr, nullptr, sq, work, st);
                else launch_sweep_snp<T>(D->snp(), static_cast<const T*>(D->impute), v, dst, 0, ncols, cols, nullptr, nullptr, sq, work, st);
            };
            base(raw, square);
            if (square) base(raw_plain, false);
            launch_vec_sum<T>(v, n, vsum, st);
            launch_std_sweep_epilogue<T>(ce, is, raw, raw_plain, vsum, square, out, 0, ncols, cols, sub_scale, sub_vec, st);
            return;
        }
        if (batcher && dense() && !cols && ncols == p && !square &&
            batcher->template sweep<T>(D->dense<T>(), v, out, sub_scale, sub_vec, st)) {
            ++cnt.n_sweeps_shared;
            return;
        }
        if (sparse()) { // one wavefront per column over its stored entries (kernels_sparse.hip)
            launch_sweep_csc<T>(D->csc<T>(), v, out, 0, ncols, cols, sub_scale, sub_vec, square,
                                d_work_sweep.reserve(size_t(sweep_work_elems_csc(D->sp_parts(), ncols))), st);
            return;
        }
        T* work = d_work_sweep.reserve(size_t(sweep_work_elems(n, ncols)));
        if (dense()) launch_sweep<T>(D->dense<T>(), v, out, 0, ncols, cols, sub_scale, sub_vec, square, work, st);
        else launch_sweep_snp<T>(D->snp(), static_cast<const T*>(D->impute), v, out, 0, ncols, cols, sub_scale, sub_vec, square, work, st);
    }
    // `want_tail`: the step may leave the block's gradient in d_gblk itself (2-bit designs, StepTail; kernels.hpp) -- then
    // step_tailed is set and the caller skips panel_reduce; tail_xm = the by-column means of the intercept term (or nullptr)
    DevBuf<int32_t> d_tail_ctr;
    int32_t tail_base = 0;
    bool step_tailed = false;
    // Means mode (step_means_now; IRLS on a 2-bit design with an intercept): the step also produces the CURRENT weighted means of
    // the block's columns and leaves them in d_irls_xm (by column) and d_sxm (by screen value: `list` / `pos0` locate the block)
    // before the block's solve reads them -- glm_fit then runs no mean sweep over the screen columns per IRLS iteration.
    // Measured on config 4 (profiles/r06_cfg4_ab.txt): parity, 2.91-2.94 s with it against 2.90-2.95 s without -- the mean sweeps it
    // removes (0.15 s of the main stream) the step gives back (cd 2.43 against 2.29 s: phase (B)'s second accumulation, butterfly
    // and tail sum, 2.5 us per step); the builds take the means of their own columns inside the MFMA kernel at no cost (with a
    // sweep per build batch instead it was 3.03-3.06 against 2.98-3.00 s).  Off; hook ADELIE_HIP_STEP_MEANS=1.
    bool step_means_opt = false;
    bool step_means_now = false;  // (set per IRLS iteration by glm_fit)
    bool step_means_possible() const {
        return step_means_opt && step_tail && !plain_solve_sums && is_glm() && intercept && all_scalar && !multi() && !sparse() &&
               !dense() && D->std_center == nullptr && !cov_mode && panel_step_snp_has_tail(D->snp());
    }
    int panel_step(const T* w, T* r, const int32_t* dcol, const T* dlt, const int32_t* nz_dev, const int32_t* cols, int nb,
                   bool slice_major = false, bool want_tail = false, const T* tail_xm = nullptr, const int32_t* list = nullptr,
                   int pos0 = 0) {
        step_tailed = false;
        if (multi()) return launch_multi_panel_step<T>(D->multi<T>(), w, r, dcol, dlt, nz_dev, cols, nb, d_part.p, st);
        const T* kappa = nullptr;
        const bool stdv = D->std_center != nullptr; // a standardized view (of a dense / 2-bit design, or of compressed columns)
        if (stdv) { // the changes over the scales into the base design's step, then kappa = sum c delta / s off every row
            T* c2 = d_std_coef.reserve(size_t(cd_block_size()) + 16);
            T* kp = c2 + cd_block_size();
            launch_std_scale_coef<T>(static_cast<const T*>(D->std_center), static_cast<const T*>(D->std_iscale), dcol, dlt, nz_dev, 0, c2,
                                     kp, st);
            dlt = c2;
            kappa = kp;
            // (before the step: its phase (B) must see the residual with the whole change applied)
            launch_vec_shift<T>(r, n, kappa, T(-1), nz_dev, st);
        }
        if (sparse()) { // compressed columns: the gradient goes straight to d_gblk (returns 0: no partials to reduce)
            launch_panel_step_csc<T>(D->csc<T>(), w, r, dcol, dlt, nz_dev, cols, nb, &d_blk.p->resid_sum,
                                     (intercept && !stdv) ? cur_xm : nullptr, d_gblk.p, st);
            if (stdv && nb > 0) // (raw sums of the stored entries -> the view's gradient)
                launch_std_fix_gblk<T>(d_gblk.p, cols, nb, static_cast<const T*>(D->std_center), static_cast<const T*>(D->std_iscale),
                                       &d_blk.p->resid_sum, intercept ? cur_xm : nullptr, st);
            return 0;
        }
        if (dense()) return launch_panel_step<T>(D->dense<T>(), w, r, dcol, dlt, nz_dev, cols, nb, d_part.p, st, slice_major);
        StepTail<T> tl{};
        const bool tail_ok = want_tail && !stdv && nb > 0 && !slice_major;
        if (tail_ok) {
            if (!d_tail_ctr.p) { // (monotone over the launches of this solver; zeroed once)
                d_tail_ctr.reserve(4);
                AHIP_CHECK(hipMemsetAsync(d_tail_ctr.p, 0, 4 * sizeof(int32_t), st));
                tail_base = 0;
            }
            tl.counter = d_tail_ctr.p; tl.base = tail_base; tl.g = d_gblk.p; tl.rsum = &d_blk.p->resid_sum; tl.xm = tail_xm;
            if (step_means_now) {
                tl.xm_col = d_irls_xm.p; tl.sxm = d_sxm.p; tl.list = list; tl.pos0 = pos0;
            }
        }
        if (step_means_now && nb > 0 && !tail_ok) throw make_core_error("means mode without a step tail (internal error).");
        const int nsl = launch_panel_step_snp<T>(D->snp(), static_cast<const T*>(D->impute), w, r, dcol, dlt, nz_dev, cols, nb,
                                                 d_part.p, st, slice_major, tail_ok ? &tl : nullptr, &step_tailed);
        if (step_tailed) {
            tail_base += nsl;
            if (tail_base > (int32_t(1) << 30)) { // (far from overflow: start over behind everything enqueued so far)
                AHIP_CHECK(hipMemsetAsync(d_tail_ctr.p, 0, 4 * sizeof(int32_t), st));
                tail_base = 0;
            }
        }
        (void)kappa;
        return nsl;
    }
    // slice partials -> gradient of the block; on a standardized view the raw sums are corrected (kernels_sparse.hip)
    void panel_reduce(int nsl, int nb, const int32_t* cols, const T* xm_c, T* gblk) {
        if (nsl <= 0) return; // (compressed columns: the step wrote the gradient itself)
        if (!std_generic()) {
            launch_panel_reduce<T>(d_part.p, nsl, nb, cols, &d_blk.p->resid_sum, xm_c, gblk, st);
            return;
        }
        launch_panel_reduce<T>(d_part.p, nsl, nb, cols, &d_blk.p->resid_sum, static_cast<const T*>(nullptr), gblk, st);
        launch_std_fix_gblk<T>(gblk, cols, nb, static_cast<const T*>(D->std_center), static_cast<const T*>(D->std_iscale),
                               &d_blk.p->resid_sum, xm_c, st);
    }
    // standardized view: the blocks of a batch from the raw X' W X just built on stream gs (sum of the weights per call: two tiny
    // launches; one scratch per build stream)
    DevBuf<T> d_std_ws[2 + kMaxExtra + 1];
    void std_block_fix(const T* w, const int32_t* cols_base, const SyrkBatch& sb, const T* xm_view, T* D0, int side, hipStream_t gs) {
        T* ws = d_std_ws[side].reserve(size_t(kVecSumScratch) + 8);
        launch_vec_sum<T>(w, n, ws, gs);
        launch_std_block_fix<T>(D0, sb, cols_base, cd_block_size(), static_cast<const T*>(D->std_center),
                                static_cast<const T*>(D->std_iscale), xm_view, ws, intercept, gs);
    }
    // `sb.count` diagonal blocks in one launch (non-multi designs): block y = columns cols_base[sb.off[y] ...], into
    // D0 + sb.dst[y] (ld = B)
    void gram_block_batch(const T* w, const int32_t* cols_base, const SyrkBatch& sb, const T* xm, T* D0, int side, T* xm_build = nullptr) {
        const int B = cd_block_size();
        hipStream_t gs = side == 0 ? st : (side >= 2 ? st_x[side - 2] : st2);
        if (sparse()) { // compressed columns: one thread per pair of columns merges the two row lists (kernels_sparse.hip)
            t_gram.begin(gs);
            const bool stdv = D->std_center != nullptr;
            launch_block_gram_csc<T>(D->csc<T>(), w, cols_base, sb, xm, intercept && !stdv, D0, B, gs);
            if (stdv) std_block_fix(w, cols_base, sb, xm, D0, side, gs);
            t_gram.end(gs);
            for (int y = 0; y < sb.count; ++y) cnt.n_gram_col_reads += 2 * sb.nb[y];
            return;
        }
        T* work = (side == 0 ? d_work_gram : (side >= 2 ? d_work_x[side - 2] : d_work_gram2))
                      .reserve(size_t(std::max(syrk_batch_work_elems(n, sb.count), syrk_work_elems(n, 128))));
        t_gram.begin(gs);
        const bool stdv = std_generic(); // raw X' W X of the base design, then the view's corrections (kernels_sparse.hip)
        if (dense()) launch_syrk_batch<T>(D->dense<T>(), w, cols_base, sb, xm, intercept && !stdv, D0, B, work, gs);
        else launch_syrk_batch_snp<T>(D->snp(), static_cast<const T*>(D->impute), w, cols_base, sb, xm, intercept && !stdv, D0, B, work, gs,
                                      stdv ? nullptr : xm_build);
        if (stdv) std_block_fix(w, cols_base, sb, xm, D0, side, gs);
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
        if (sparse()) { // compressed columns (kernels_sparse.hip: LDS hash join per column of the block)
            SyrkBatch sb{};
            sb.count = 1; sb.off[0] = 0; sb.nb[0] = nb; sb.dst[0] = 0;
            const bool stdv = D->std_center != nullptr;
            launch_block_gram_csc<T>(D->csc<T>(), w, cols, sb, xm, intercept && !stdv, Dptr, B, gs);
            if (stdv) std_block_fix(w, cols, sb, xm, Dptr, side, gs);
            t_gram.end(gs);
            cnt.n_gram_col_reads += 2 * nb;
            return;
        }
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
            const bool stdv = std_generic();
            if (dense()) launch_syrk<T>(D->dense<T>(), w, cols, nb, xm, intercept && !stdv, Dptr, B, work, gs);
            else launch_syrk_snp<T>(D->snp(), static_cast<const T*>(D->impute), w, cols, nb, xm, intercept && !stdv, Dptr, B, work, gs);
            if (stdv) {
                SyrkBatch sb{};
                sb.count = 1; sb.off[0] = 0; sb.nb[0] = nb; sb.dst[0] = 0;
                std_block_fix(w, cols, sb, xm, Dptr, side, gs);
            }
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
        if (sparse()) { // coefficients scattered into a p-vector that is all zero between calls, then one CSR pass
            if (d_sp_delta.cap < size_t(p) + 8) {
                d_sp_delta.reserve(size_t(p) + 8);
                AHIP_CHECK(hipMemsetAsync(d_sp_delta.p, 0, (size_t(p) + 8) * sizeof(T), st));
            }
            launch_axpy_cols_csc<T>(D->csc<T>(), cols, coef, cnt_dev, count, sign, out, d_sp_delta.p, st);
            return;
        }
        if (std_generic()) { // coefficients over the scales into the base design's update, then kappa off every row
            const T* ce = static_cast<const T*>(D->std_center);
            const T* is = static_cast<const T*>(D->std_iscale);
            const size_t cap = size_t(std::max<idx>(nv, idx(count))) + 8;
            T* c2 = d_std_coef.reserve(cap + 8);
            T* kappa = c2 + cap;
            launch_std_scale_coef<T>(ce, is, cols, coef, cnt_dev, count, c2, kappa, st);
            if (dense()) launch_axpy_cols<T>(D->dense<T>(), cols, c2, cnt_dev, count, sign, out, st);
            else launch_axpy_cols_snp<T>(D->snp(), static_cast<const T*>(D->impute), cols, c2, cnt_dev, count, sign, out, st);
            launch_vec_shift<T>(out, n, kappa, sign, cnt_dev, st);
            return;
        }
        if (dense()) launch_axpy_cols<T>(D->dense<T>(), cols, coef, cnt_dev, count, sign, out, st);
        else launch_axpy_cols_snp<T>(D->snp(), static_cast<const T*>(D->impute), cols, coef, cnt_dev, count, sign, out, st);
    }
    void gram(const T* w, idx M, idx pos0, idx N, const T* xm, bool center) {
        T* work = d_work_gram.reserve(size_t(sparse() ? gram_work_elems_csc(n, M, N, D->sp_parts()) : gram_work_elems(n, M, N)));
        t_gram.begin(st);
        if (std_generic()) { // raw X^T W X of the base design in place, then the view's rank-one corrections over the panel
            const T* ce = static_cast<const T*>(D->std_center);
            const T* is = static_cast<const T*>(D->std_iscale);
            if (dense())
                launch_gram<T>(D->dense<T>(), w, d_vcol.p, int32_t(M), 0, d_vcol.p + pos0, int32_t(N), int32_t(pos0), xm, false, d_C.p,
                               ldc, work, st);
            else
                launch_gram_snp<T>(D->snp(), static_cast<const T*>(D->impute), w, d_vcol.p, int32_t(M), 0, d_vcol.p + pos0, int32_t(N),
                                   int32_t(pos0), xm, false, d_C.p, ldc, work, st);
            T* tmp = d_std_tmp.reserve(size_t(M + kVecSumScratch));
            T *mv = tmp, *wsum = tmp + M;
            T* swork = d_work_sweep.reserve(size_t(sweep_work_elems(n, M)));
            if (dense()) launch_sweep<T>(D->dense<T>(), w, mv, 0, M, d_vcol.p, nullptr, nullptr, false, swork, st);
            else launch_sweep_snp<T>(D->snp(), static_cast<const T*>(D->impute), w, mv, 0, M, d_vcol.p, nullptr, nullptr, false, swork, st);
            launch_vec_sum<T>(w, n, wsum, st);
            launch_std_gram_fix<T>(ce, is, d_C.p, ldc, int32_t(M), int32_t(pos0), int32_t(N), d_vcol.p, mv, wsum, xm, center, st);
        } else if (sparse())
            launch_gram_csc<T>(D->csc<T>(), w, d_vcol.p, int32_t(M), 0, d_vcol.p + pos0, int32_t(N), int32_t(pos0), xm, center, d_C.p,
                               ldc, work, st);
        else if (dense())
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
            const T regul = ((1 - alpha) * lm) * penalty2[i];
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
            if (cons_dev) // box / one-sided objects on several coefficients: their gradient term / solve_zero (kernels_cons.hip)
                launch_cons_abs_grad<T>(d_devcons_list.p, int(devcons_list.size()), d_cons_native.p, d_groups.p, d_gsizes.p, d_slot.p,
                                        d_grad.p, d_beta.p, d_penalty.p, (1 - alpha) * lm, d_cons_va.p, d_cons_vb.p, d_cons_mu.p, d_absgrad.p, st);
        } else {
            launch_abs_grad<T>(d_grad.p, d_groups.p, d_gsizes.p, G, d_slot.p, d_beta.p, has_pen2 ? d_penalty2.p : d_penalty.p,
                               (1 - alpha) * lm, d_absgrad.p, st);
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
        if (has_pen2) { // (groups of one coefficient: screen group == screen value; the host copy outlives the upload)
            h_spen2.reserve(size_t(G));
            h_spen2.resize(size_t(ns_dev));
            for (idx ss = ns_dev; ss < ns; ++ss) h_spen2.push_back(penalty2[screen_set[ss]]);
            d_spen2.upload(h_spen2.data() + ns_dev, size_t(ns - ns_dev), st, size_t(ns_dev));
        }
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
        const bool vfb = vars_from_blocks(); // IRLS, groups of one: the variances come with the blocks' builds (solver_builds.hpp)
        if (!vfb) sweep(w_dev, d_vars.p + pos0, d_vcol.p + pos0, N, nullptr, nullptr, true);
        for (idx ss = idx(g_begin); ss < ns; ++ss) {
            const idx g = screen_set[ss], b = screen_begins[ss];
            screen_X_means[b] = xm_host[groups[g]];
            if (screen_transforms[ss].size() != 1) screen_transforms[ss].assign(1, T(1));
        }
        // by-value means on the device straight from the by-column vector; the host copy of the variances is only an output
        // (finalize() downloads it), so no synchronisation here
        launch_gather<T>(xm_dev, d_vcol.p + pos0, N, d_sxm.p + pos0, st);
        if (!vfb) launch_center_vars<T>(d_vars.p + pos0, d_sxm.p + pos0, int(N), intercept, st);
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
    std::vector<std::pair<T, idx>> screen_keyed; // (kept between calls: no allocation per lambda)
    T screen_thr = 0;                            // see the pivot rule below
    bool screen_thr_valid = false;
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
                const int subset_size =
                    std::min<int>(std::max<int>(int(old_size * (1 + pivot_subset_ratio)), int(pivot_subset_min)), Gi);
                // The rule reads the sorted scores only from the top: the `subset_size` largest for the pivot search, and below
                // the pivot as many more as it takes to find slack * n_new_active groups outside the screen set — at most
                // `need` positions in all.  With a lower bound on the need-th largest score (what the previous lambda's list
                // had at twice that depth: the scores of unscreened groups grow as lambda falls) one pass collects every
                // group at or above it, in group order, and only those are sorted; whenever they are fewer than `need`
                // (or there is no bound yet) all G are sorted as before.  Same pairs in the same order either way.
                const int64_t need = int64_t(subset_size) + int64_t(std::ceil(pivot_slack_ratio * n_new_active)) + old_size + 2;
                auto score = [&](int i) {
                    return (penalty[i] <= 0) ? alpha * lmda : std::min(abs_grad[i] / penalty[i], alpha * lmda);
                };
                // (score, group) pairs, sorted in place: contiguous keys instead of an indirect comparator
                std::vector<std::pair<T, idx>>& keyed = screen_keyed;
                keyed.clear();
                bool partial = false;
                if (screen_thr_valid && need * 4 < Gi) {
                    for (int i = 0; i < Gi; ++i) {
                        const T wt = score(i);
                        if (wt >= screen_thr) keyed.emplace_back(wt, idx(i));
                    }
                    partial = int64_t(keyed.size()) >= need;
                }
                if (!partial) {
                    keyed.resize(size_t(Gi));
                    for (int i = 0; i < Gi; ++i) keyed[size_t(i)] = std::make_pair(score(i), idx(i));
                }
                // The reference sorts with `weights[i] < weights[j]` only (solver_base.hpp:320-326): every group whose score is
                // capped at alpha*lmda ties exactly, and std::sort leaves the order of ties unspecified.  Ties are broken by
                // group index here (pair comparison) so that the screen insertion order (= CD visiting order) is reproducible.
                sort_keyed(keyed);
                const int M = int(keyed.size()); // position ii of the full order is keyed[ii - (Gi - M)]
                const int base = Gi - M;
                {   // bound for the next lambda: the score twice as deep as this call could have read
                    const int64_t depth = std::min<int64_t>(2 * need, M);
                    screen_thr = keyed[size_t(M - depth)].first;
                    screen_thr_valid = depth >= need;
                }
                std::vector<T> sub(subset_size), mses(subset_size), ind(subset_size);
                for (int i = 0; i < subset_size; ++i) {
                    sub[i] = keyed[size_t(Gi - subset_size + i - base)].first;
                    ind[i] = T(i);
                }
                const int pivot_idx = search_pivot(ind, sub, mses);
                const int full_pivot_idx = Gi - subset_size + pivot_idx;
                for (int ii = Gi - 1; ii >= full_pivot_idx; --ii) {
                    const idx i = keyed[size_t(ii - base)].second;
                    if (is_screen(i)) continue;
                    screen_set.push_back(i);
                }
                int count = 0;
                for (int ii = full_pivot_idx - 1; ii >= base; --ii) {
                    if (count >= pivot_slack_ratio * n_new_active) break;
                    const idx i = keyed[size_t(ii - base)].second;
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

