// solver_panel.hpp — part of `template <class T> struct Solver` (solver.hip includes this file INSIDE the struct body, in this order:
// solver_builds, solver_screen, solver_panel, solver_fit, solver_path; one translation unit, several readable files).
// Contents: the pin solver's passes spread over the chip: full-Gram block passes (covariance method), the lasso panel passes with
// their look-ahead form (run_panel_passes), the group panel passes and the host visits of constraint objects.
    // ---------------------------------------------------------------------------------------------------------
    // Lasso pin solve as a sequence of block passes spread over the chip (kernels_cd_block.hip).  The pass structure
    // (solve_active until convergence, one screen pass, repeat; pin_naive:317-357) is driven from the host, which reads
    // one small scalar block per pass.  Fills `sc` like the single-workgroup kernel does.
    // `r_stale` != nullptr (IRLS with a Gram kept from earlier weights, Solver::gram_stale): every pass but the first of the fit
    // first brings the residual up to date with the changes made so far and takes the exact gradient X_S' W r from it.
    void run_block_passes(const CdParams<T>& cp, CdScalars<T>& sc, T* r_stale = nullptr) {
        const int B = cd_block_size();
        if (r_stale) {
            d_beta_ref.reserve(size_t(cp.nv) + 8);
            AHIP_CHECK(hipMemcpyAsync(d_beta_ref.p, cp.beta, size_t(cp.nv) * sizeof(T), hipMemcpyDeviceToDevice, st));
        }
        int64_t n_pass = 0;
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
        bp.nv = cp.nv; bp.C = cp.C; bp.ldc = cp.ldc; bp.vars = cp.vars; bp.xmean = cp.xmean; bp.spen = cp.spen; bp.spen2 = cp.spen2;
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
            if (r_stale && n_pass++ > 0) {
                launch_cd_compact<T>(cp.beta, d_beta_ref.p, cp.vcol, cp.nv, cp.dcols, cp.dvals, &cp.sc->n_delta, st);
                t_axpy.begin(st);
                axpy_cols(cp.dcols, cp.dvals, &cp.sc->n_delta, 0, T(-1), r_stale);
                t_axpy.end(st);
                AHIP_CHECK(hipMemcpyAsync(d_beta_ref.p, cp.beta, size_t(cp.nv) * sizeof(T), hipMemcpyDeviceToDevice, st));
                launch_vmul<T>(cur_w, r_stale, d_v.p, n, st);
                sweep(d_v.p, cp.g, d_vcol.p, cp.nv, &d_blk.p->resid_sum, intercept ? cur_xm : nullptr);
            }
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
        launch_cd_compact<T>(cp.beta, r_stale ? d_beta_ref.p : cp.beta0, cp.vcol, cp.nv, cp.dcols, cp.dvals, &cp.sc->n_delta, st);
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
        bp.nv = cp.nv; bp.vars = cp.vars; bp.xmean = cp.xmean; bp.spen = cp.spen; bp.spen2 = cp.spen2;
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
        const bool la = lookahead && !is_glm() && B == SL && !std_generic() && !sparse();
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
            // The stand-alone solve sums the slice partials itself (blk_solve_la_body, second round trip of its prologue): no
            // panel_reduce launch between step and solve.  Config 4: 54.8 k blocks per path x (reduce 4.85 us + a boundary).
            // Not with constraints (their solve kernel has the plain prologue), views and compressed columns (their gradient
            // is corrected / written by other launches), the multi-response view (its own partial layout).
            bool step_tailed_of[2] = {false, false};
            const bool solve_sums = plain_solve_sums && !cons_on && !multi() && !std_generic() && !sparse() && D->std_center == nullptr;
            auto step_of = [&](int j) {
                const int nb = std::min(B, count - j * B);
                const int32_t* cols = cols_all + size_t(j) * B;
                const int ps = (j == 0) ? pending_slot : -1;
                if (time_panel) t_step.begin(st);
                const int nsl = panel_step(cur_w, r_dev, ps < 0 ? d_dcolblk.p : d_la_dcol.p + size_t(ps) * SL,
                                           ps < 0 ? d_dlt.p : d_la_dlt.p + size_t(ps) * SL,
                                           ps < 0 ? &d_blk.p->nz : d_la_nz.p + ps, cols, nb, solve_sums,
                                           step_tail && !solve_sums && !multi() && !sparse(), xm_c, bp.list, j * B);
                if (time_panel) t_step.end(st);
                step_tailed_of[j & 1] = step_tailed;
                return nsl;
            };

            record_pass_e0();
            t_cd.begin(st);
            const int nsl0 = step_of(0);
            build_stale_strips(nblk, tab_nb, tab_ver, nullptr, pool, static_cast<T*>(nullptr),
                               [&](int j) { return std::min(B, count - j * B); }, [&](int j) { return cols_all + size_t(j) * B; });
            const bool vfb = vars_from_blocks(); // IRLS: the variances of a block's coordinates come with its build
            if (vfb) {
                if (d_vars_act.cap < size_t(p) + 8) {
                    d_vars_act.reserve(size_t(p) + 8);
                    AHIP_CHECK(hipMemsetAsync(d_vars_act.p, 0, (size_t(p) + 8) * sizeof(T), st));
                }
                vb_vars = screen_pass ? d_vars.p : d_vars_act.p;
                vb_cols_all = cols_all;
                vb_list = screen_pass ? nullptr : cp.active_set;
            }
            build_stale_blocks(nblk, tab_nb, tab_ver, pool, [&](int j) { return std::min(B, count - j * B); },
                               [&](int j) { return cols_all + size_t(j) * B; }, false, screen_pass);
            vb_vars = nullptr;
            merge_strip_events(false);
            pass_e0_valid = false;
            if (!screen_pass && prebuild_screen && !screen_prebuilt && side_grams && st2) {
                // IRLS: every screen-order block is stale as well (new weights) and the screen pass follows the active-set
                // passes of this fit: enqueue those builds now, behind the ones this pass waits for, so that they run while
                // the active-set passes iterate
                screen_prebuilt = true;
                const int cnt_s = cp.nv, nblk_s = (cnt_s + B - 1) / B;
                if (vfb) { vb_vars = d_vars.p; vb_cols_all = d_vcol.p; vb_list = nullptr; }
                build_stale_blocks(nblk_s, dscr_nb, dscr_ver, d_Dpool.p, [&](int j) { return std::min(B, cnt_s - j * B); },
                                   [&](int j) { return d_vcol.p + size_t(j) * B; }, true);
                vb_vars = nullptr;
            }
            // (a look-ahead pass may have run before: plain buffers for the solves, its pending changes for the first step)
            bp.gblk = d_gblk.p; bp.dlt = d_dlt.p; bp.dcol = d_dcolblk.p;
            bp.vars = vfb ? (screen_pass ? d_vars.p : d_vars_act.p) : cp.vars;
            bp.Cprev = nullptr; bp.dpos = nullptr; bp.nz_out = nullptr; bp.rsum_out = nullptr;
            bp.part = nullptr; bp.pdd = nullptr; bp.dd = nullptr;
            for (int j = 0; j < nblk; ++j) {
                const int nb = std::min(B, count - j * B);
                const int32_t* cols = cols_all + size_t(j) * B;
                T* Dptr = pool + size_t(j) * SL * SL;
                const int nsl = (j == 0) ? nsl0 : step_of(j);
                pending_slot = -1;
                cnt.n_panel_cols += nb;
                if (solve_sums) {
                    bp.part = d_part.p; bp.part_ld = 0; bp.part_n = nsl;
                    bp.part_rsum = xm_c ? &d_blk.p->resid_sum : nullptr; // (the residual sum the step's residual belongs to)
                } else if (!step_tailed_of[j & 1]) { // (else the step's last workgroups left the gradient in d_gblk)
                    panel_reduce(nsl, nb, cols, xm_c, d_gblk.p);
                }
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
    DevBuf<T> d_la_corr;                      // look-ahead corrections left by the previous solve (CdGrpBlkParams::corr_out), by block parity
    bool group_next_corr = true;              // A/B: ADELIE_HIP_GROUP_NEXT_CORR=0
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
        if (cons_on) { bp.clo = d_clo.p; bp.chi = d_chi.p; bp.cmu = d_cmu.p; } // one-coefficient closed forms (grp_clip_1d)
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
            if (!cons_host) {
                launch_cd_group_block_pass<T>(bp, st);
            } else { // blocks that are one group with a constraint object on the caller's side are visited on the host
                int j0 = 0;
                for (int j = 0; j < nblk; ++j) {
                    const idx ss0 = screen_pass ? idx(part_host[size_t(j)]) : act_host[size_t(part_host[size_t(j)])];
                    if (!(part_host[size_t(j) + 1] - part_host[size_t(j)] == 1 && host_cons(screen_set[ss0]))) continue;
                    launch_cd_group_block_range<T>(bp, j0, j, st);
                    if (on_device(screen_set[ss0])) dev_group_visit(cp, ss0, screen_pass, j == 0, true, false);
                    else (void)host_group_visit(cp, ss0, screen_pass, j == 0, true);
                    launch_cd_group_block_update<T>(bp, j, st);
                    j0 = j + 1;
                }
                launch_cd_group_block_range<T>(bp, j0, nblk, st);
            }
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
    // `gram`: the covariance method's engine — the group's gradient is its slice of the screen gradient d_g (kept current by the
    // Gram updates), there is no residual and no intercept, and the changes go out as (screen value, delta) pairs for
    // grp_update_kernel (solver_gaussian_pin_cov.hpp:287-355) instead of (design column, delta) pairs for the next panel step.
    CdBlkState<T> host_group_visit(const CdParams<T>& cp, idx ss, bool mark, bool first_of_pass, bool gram = false) {
        const idx g = screen_set[ss], q = group_sizes[g], b = screen_begins[ss];
        const size_t uq = static_cast<size_t>(q);
        const bool trace_hv = hooks.trace >= 1;
        ++n_host_cons_visits;
        if (trace_hv) std::fprintf(stderr, "[host visit] ss=%lld g=%lld q=%lld b=%lld voff=%lld v_used=%zu nv=%lld\n", (long long)ss, (long long)g,
                                   (long long)q, (long long)b, (long long)(size_t(ss) < h_voff.size() ? h_voff[size_t(ss)] : -1), v_used, (long long)nv);
        std::vector<T> gk(uq), ak(uq), Ak(uq), Vk(uq * uq, T(1));
        CdBlkState<T> bs{};
        int8_t was_active = 0;
        if (gram) d_g.download(gk.data(), size_t(q), st, size_t(b));
        else d_gblk.download(gk.data(), size_t(q), st);
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
                dcol[size_t(i)] = gram ? int32_t(b + i) : int32_t(groups[g] + i);
                if (!gram) rsum += double(screen_X_means[size_t(b + i)]) * double(ak[size_t(i)] - a_new[size_t(i)]);
            }
            bs.resid_sum += T(rsum);
            bs.n_updates += 1;
            bs.nz = int32_t(q);
            d_beta.upload(a_new.data(), size_t(q), st, size_t(b));
            if (gram) d_didx.upload(dcol.data(), size_t(q), st);
            else d_dcolblk.upload(dcol.data(), size_t(q), st);
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
    // The same visit for a box / one-sided object that runs on the device (on_device(g); kernels_cons.hip): one launch on the
    // solver's stream, no host round trip.  `report`: this block is the last one of the pass, the kernel publishes the pass state.
    void dev_group_visit(const CdParams<T>& cp, idx ss, bool mark, bool first_of_pass, bool gram, bool report) {
        const idx g = screen_set[ss], q = group_sizes[g], b = screen_begins[ss];
        ConsVisitParams<T> vp{};
        vp.q = int32_t(q); vp.ss = int32_t(ss); vp.b = int32_t(b); vp.col0 = int32_t(groups[g]); vp.native = cons_native[g];
        vp.gsrc = gram ? d_g.p + b : d_gblk.p;
        vp.beta = d_beta.p + b;
        vp.vars = d_vars.p + b;
        vp.V = q > 1 ? d_V.p + h_voff[size_t(ss)] : nullptr;
        vp.sxm = gram ? nullptr : cp.xmean + b;
        vp.is_active = d_isact.p;
        vp.active_set = d_actset.p;
        vp.st = d_blk.p;
        const T pk = penalty[g];
        vp.l1 = double(cp.lmda * cp.alpha) * double(pk);
        vp.l2 = double(cp.lmda * (T(1) - cp.alpha)) * double(pk);
        vp.dbeta_tol = double(g_dbeta_tol);
        vp.mark = mark ? 1 : 0; vp.first_of_pass = first_of_pass ? 1 : 0; vp.gram = gram ? 1 : 0;
        vp.max_active_size = int32_t(std::min<size_t>(max_active_size, size_t(std::numeric_limits<int32_t>::max())));
        vp.dcol = gram ? d_didx.p : d_dcolblk.p;
        vp.dlt = d_dlt.p;
        vp.va = d_cons_va.p + groups[g]; vp.vb = d_cons_vb.p + groups[g]; vp.mu = d_cons_mu.p + groups[g];
        for (int e = 0; e < 5; ++e) vp.cfg[e] = cons_cfg[size_t(g) * 5 + e];
        vp.host_st = rep_st_dev; vp.host_seq = rep_seq_dev;
        vp.report_seq = (report && h_report) ? ++report_seq : 0;
        vp.n_visits = d_cons_nvis.p;
        launch_grp_cons_visit<T>(vp, st);
    }
    // abs_grad of the groups with host constraint objects (solver_base.hpp:62-93): the constraint's gradient for screened groups,
    // its solve_zero for the others; overrides what the device kernel wrote for them (it knows no bounds for these groups)
    void host_cons_abs_grad(T lm) {
        if (!cons_host) return;
        if (cons_dev && devcons_list.size() == n_host_cons) return; // every object runs on the device
        d_grad.download(grad.data(), size_t(p), st);
        sync();
        std::vector<double> v, out;
        std::vector<idx> begin_of(static_cast<size_t>(G), idx(-1));
        for (size_t ss = 0; ss < screen_set.size() && ss < screen_begins.size(); ++ss) begin_of[size_t(screen_set[ss])] = screen_begins[ss];
        for (idx g = 0; g < G; ++g) {
            if (!host_cons(g) || on_device(g)) continue; // (device objects: cons_abs_grad_kernel, device_abs_grad)
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

    // The same for the objects that run on the device, on the HOST mirrors (state construction only: abs_grad of the state that
    // was handed in decides lmda_max and the first screening step; afterwards cons_abs_grad_kernel does this on the device).
    // Leaves the multipliers solve_zero gives the groups outside the screen set, as the reference's objects would hold them.
    void dev_cons_abs_grad_host(T lm) {
        if (!cons_dev) return;
        const double M = 1e100;
        std::vector<idx> begin_of(static_cast<size_t>(G), idx(-1));
        for (size_t ss = 0; ss < screen_set.size() && ss < screen_begins.size(); ++ss) begin_of[size_t(screen_set[ss])] = screen_begins[ss];
        std::vector<T> va(static_cast<size_t>(p), T(0)), vb(static_cast<size_t>(p), T(0));
        d_cons_va.download(va.data(), size_t(p), st);
        d_cons_vb.download(vb.data(), size_t(p), st);
        sync();
        for (int32_t g : devcons_list) {
            const idx q = group_sizes[g], k = groups[g];
            const bool box = cons_native[g] == ADELIE_HIP_NATIVE_BOX;
            double acc = 0;
            if (begin_of[size_t(g)] >= 0) {
                const idx b = begin_of[size_t(g)];
                const double regul = double((1 - alpha) * lm) * double(penalty[g]);
                for (idx t = 0; t < q; ++t) {
                    const double cg = box ? double(cons_vmu[size_t(k + t)]) : double(va[size_t(k + t)]) * double(cons_vmu[size_t(k + t)]);
                    const double e = double(grad[size_t(k + t)]) - regul * double(screen_beta[size_t(b + t)]) - cg;
                    acc += e * e;
                }
            } else {
                for (idx t = 0; t < q; ++t) {
                    const double v = double(grad[size_t(k + t)]);
                    double m, e;
                    if (box) {
                        const double lo = -std::max(double(va[size_t(k + t)]), -M), up = std::min(double(vb[size_t(k + t)]), M);
                        m = std::min(std::max(v, (lo <= 0) ? -M : 0.0), (up <= 0) ? M : 0.0);
                        e = v - m;
                    } else {
                        const double sg = double(va[size_t(k + t)]), bb = std::min(double(vb[size_t(k + t)]), M);
                        m = std::min(std::max(sg * v, 0.0), (bb <= 0) ? M : 0.0);
                        e = v - sg * m;
                    }
                    cons_vmu[size_t(k + t)] = T(m);
                    acc += e * e;
                }
            }
            abs_grad[size_t(g)] = T(std::sqrt(acc));
        }
        d_cons_mu.upload(cons_vmu.data(), size_t(p), st);
        sync();
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
        const bool la = lookahead && !is_glm() && (!multi() || multi_w_uniform) && !std_generic() && !sparse();
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
            d_tail_counter.reserve(2); // (arrivals, finished tail workgroups)
            AHIP_CHECK(hipMemsetAsync(d_tail_counter.p, 0, 2 * sizeof(int32_t), st));
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
            // the correction of block j + 1 formed by the idle waves of solve j (CdGrpBlkParams::Cnext): fused launches of the
            // rotated single-response form
            const bool next_corr = group_next_corr && bp.rot && !multi();
            if (next_corr) d_la_corr.reserve(size_t(2) * SL);
            bool prev_made_corr = false;
            for (int j = 0; j < nblk; ++j) {
                const int slot = j & 1, pslot = slot ^ 1;
                bp.gblk = d_la_g.p + size_t(slot) * SL;
                bp.Dptr = pool + size_t(j) * SL * SL;
                bp.Cprev = j > 0 ? xpool + size_t(j) * SL * SL : nullptr;
                bp.corr_in = (prev_made_corr && j > 0) ? d_la_corr.p + size_t(pslot) * SL : nullptr;
                const bool fused_j = !(j == 0 && !fr_open);
                bp.Cnext = (next_corr && fused_j && j + 1 < nblk) ? xpool + size_t(j + 1) * SL * SL : nullptr;
                bp.corr_out = d_la_corr.p ? d_la_corr.p + size_t(slot) * SL : nullptr;
                prev_made_corr = bp.Cnext != nullptr;
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
                if (bp.Cnext) { // the next cross block is read by THIS launch: whatever builds or extends it (strips record into either list)
                    if (x_ev[size_t(j) + 1]) AHIP_CHECK(hipStreamWaitEvent(st, x_ev[size_t(j) + 1], 0));
                    if (blk_ev[size_t(j) + 1]) AHIP_CHECK(hipStreamWaitEvent(st, blk_ev[size_t(j) + 1], 0));
                }
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
                panel_reduce(nsl, nval, cols, xm_c, d_gblk.p);
                if (cons_host) { // a block that is one group with a constraint object on the caller's side: visited on the host
                    const idx ss0 = screen_pass ? idx(part_host[size_t(j)]) : act_host[size_t(part_host[size_t(j)])];
                    if (part_host[size_t(j) + 1] - part_host[size_t(j)] == 1 && host_cons(screen_set[ss0])) {
                        if (blk_ev[size_t(j)]) AHIP_CHECK(hipStreamWaitEvent(st, blk_ev[size_t(j)], 0)); // (its eigenbasis)
                        if (on_device(screen_set[ss0])) { // one launch, the pass report included when it is the last block
                            dev_group_visit(cp, ss0, screen_pass, j == 0, false, j == nblk - 1);
                            last_on_host = false;
                            continue;
                        }
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

