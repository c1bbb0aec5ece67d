// grp_solve_body.hpp — the one-workgroup solve of a block of consecutive groups (<= 128 values), see
// kernels_cd_block_group.hip; a device function so that it can run as its own kernel and as workgroup 0 of the fused
// look-ahead step (kernels_cd_panel.hip).  Must be called by threads 0..255 of a workgroup (uses threadIdx.x).
#pragma once
#include <type_traits>
#include "kernels.hpp"

namespace ahip {
namespace {

constexpr int GBLK = 128;
constexpr int VPOOL = 1280; // LDS pool (elements) for the eigenbases of a block's groups (12 groups of 10: 1200)

// hardware reciprocal approximation (v_rcp_f64 / v_rcp_f32): for starting values only
__device__ __forceinline__ double fast_rcp(double x) { return __builtin_amdgcn_rcp(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

template <class T>
__device__ __forceinline__ T gwsum(T x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

// Sum over the first 16 lanes with DPP (no LDS crossbar): xor-1, xor-2 by quad_perm, then row_half_mirror and
// row_mirror; the result is made wave-uniform with readfirstlane.  Lanes >= 16 must not contribute.
// (bound_ctrl: these four patterns give every lane a source inside its row, so the "old" operand is never used -- saying so
// spares the two zero moves and the wait state the compiler otherwise puts in front of every dpp move: 6 -> 3 instructions per
// step of a double-precision sum, 29 -> 17 per reduction, on a wave that is bound by the instructions it issues.)
template <int CTRL>
__device__ __forceinline__ double dpp_move(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float dpp_move(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ double first_lane(double x) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)),
                            __builtin_amdgcn_readfirstlane(__double2loint(x)));
}
__device__ __forceinline__ float first_lane(float x) {
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x)));
}
__device__ __forceinline__ double rdl(double x, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}
__device__ __forceinline__ float rdl(float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); }
template <class T>
__device__ __forceinline__ T row16_sum(T x) {
    x += dpp_move<0xB1>(x);  // quad_perm [1,0,3,2]
    x += dpp_move<0x4E>(x);  // quad_perm [2,3,0,1]
    x += dpp_move<0x141>(x); // row_half_mirror
    x += dpp_move<0x140>(x); // row_mirror
    return first_lane(x);
}
// sum of per-lane partials of a q-long group (lanes >= q hold 0)
template <class T>
__device__ __forceinline__ T group_sum(T x, int q) {
    return q <= 16 ? row16_sum(x) : gwsum(x);
}

// two sums at once: the two chains interleave (each step's add waits on its dpp moves and the next moves on the add: alone a
// chain leaves wait states between them)
template <class T>
__device__ __forceinline__ void group_sum2(T& a, T& b, int q) {
    if (q <= 16) {
        T x = a, y = b;
        x += dpp_move<0xB1>(x);  y += dpp_move<0xB1>(y);
        x += dpp_move<0x4E>(x);  y += dpp_move<0x4E>(y);
        x += dpp_move<0x141>(x); y += dpp_move<0x141>(y);
        x += dpp_move<0x140>(x); y += dpp_move<0x140>(y);
        a = first_lane(x);
        b = first_lane(y);
    } else {
        a = gwsum(a);
        b = gwsum(b);
    }
}

// Fills vmap[0..nval) with the global screen-value index of every value of block j, plus the group tables (all in LDS).
// Cooperative: every thread of the (256-thread) workgroup must call it.  The per-group lookups (list -> begin / size)
// are two dependent global loads; doing them one group per thread costs two round trips for the whole block instead of
// two per group in a serial loop (which was most of a 12-group block's latency).
template <class T>
__device__ __forceinline__ void block_layout(const CdGrpBlkParams<T>& p, int j, int32_t* vmap, int32_t* goff,
                                             int32_t* gq, int32_t* gss, int32_t* meta /* [0]=ngrp [1]=nval */) {
    __shared__ int32_t gb_[GBLK];
    const int tid = threadIdx.x;
    const int g0 = p.blk_g0[j], g1 = p.blk_g0[j + 1];
    const int ng = g1 - g0;
    if (tid < GBLK) {
        int ss = 0, b = 0, q = 0;
        if (tid < ng) {
            ss = p.list ? p.list[g0 + tid] : g0 + tid;
            b = p.sbegin[ss];
            q = p.ssize[ss];
        }
        gss[tid] = ss;
        gb_[tid] = b;
        gq[tid] = q;
    }
    __syncthreads();
    if (tid == 0) {
        int o = 0;
        for (int k = 0; k < ng; ++k) { goff[k] = o; o += gq[k]; }
        goff[ng] = o;
        meta[0] = ng;
        meta[1] = o;
    }
    __syncthreads();
    for (int k = tid >> 1; k < ng; k += 128) { // two threads per group
        const int o = goff[k], q = gq[k], b = gb_[k];
        for (int t = tid & 1; t < q; t += 2) vmap[o + t] = b + t;
    }
    __syncthreads();
}

template <class T>
__device__ __forceinline__ void gather_group_block(const CdGrpBlkParams<T>& p, int j, const int32_t* vmap, int nval,
                                                   int gtid, int gthreads) {
    T* D = p.Dbuf + size_t(j & 1) * GBLK * GBLK;
    for (int e = gtid; e < nval * nval; e += gthreads) {
        const int i = e % nval, m = e / nval;
        D[i + m * GBLK] = p.C[int64_t(vmap[i]) + int64_t(vmap[m]) * p.ldc];
    }
}

// One-coefficient box constraint of a group of size one inside a grouped problem (CdGrpBlkParams::clo / chi / cmu, indexed by
// screen value; see blk_solve_body<.., CONS>): clips the unconstrained update and leaves the multiplier mu_+ - mu_- of
// constraint_box.ipp:66-95 (A = Q(0,0) = 1) in cmu.  Uniform over the wave; lane 0 stores.
template <class T>
__device__ __forceinline__ T grp_clip_1d(const CdGrpBlkParams<T>& p, int a, T ak, T gk, T l1, T den, int lane) {
    const T INF = T(1) / T(0);
    const T lo = p.clo[a], hi = p.chi[a];
    if (lo == -INF && hi == INF) return ak;
    const T x0 = fmax(fmin(ak, hi), lo);
    const T mp0 = (hi > T(0)) ? T(0) : fmax(gk, T(0));
    const T mn0 = (lo < T(0)) ? T(0) : fmax(-gk, T(0));
    T mu;
    if (fabs(gk - (mp0 - mn0)) <= l1) {
        mu = mp0 - mn0;
    } else {
        const T full = gk - (den * x0 + copysign(l1, x0));
        mu = ((x0 < hi) ? T(0) : fmax(full, T(0))) - ((x0 > lo) ? T(0) : fmax(-full, T(0)));
    }
    if (lane == 0) p.cmu[a] = mu;
    return x0;
}

// LDS of the solve proper; the look-ahead correction partials (2 * GBLK values) follow it
template <class T>
__host__ __device__ constexpr size_t grp_solve_lds() {
    return size_t(GBLK) * GBLK * sizeof(T) + size_t(GBLK) * (5 + 8) * sizeof(T) + (size_t(GBLK) * 4 + 8) * sizeof(int32_t) + 16 +
           size_t(GBLK) * sizeof(T) + size_t(GBLK) * 2 * sizeof(int32_t) + size_t(VPOOL) * sizeof(T);
}
template <class T>
__host__ __device__ constexpr size_t grp_solve_lds_total() {
    return ((grp_solve_lds<T>() + 15) / 16) * 16 + size_t(2) * GBLK * sizeof(T) + size_t(GBLK) * sizeof(int32_t);
}

// The panel solve in the eigen-coordinates of the block's groups (CdGrpBlkParams::rot): p.Dptr holds R^T D R.  Same iterates
// as grp_solve_body in exact arithmetic — the reference rotates the gradient and the coefficients of a group into the
// eigenbasis at every visit (pin_naive:123-140) and the new coefficients back (:156-157); here every VALUE of the block is
// rotated once, lane-parallel, before the sequential loop, the loop keeps the rotated gradient current with the rotated
// block, and the coefficients of the groups that changed are rotated back once, lane-parallel, after it.  The dependent
// chain of a visit shrinks to: Newton root find -> change test -> gradient update of the following groups.
// Prologue: the workgroup (nt = 256 threads as its own kernel, all 1024 of workgroup 0 in the fused launches) fetches with as
// few dependent round trips as the data allow — (1) the block's layout descriptor (launch_grp_layout, once per pass), the
// gradient handed over, the previous block's dense changes; the rotated block D~ and the cross block stream in meanwhile, 16
// independent loads per thread and batch; (2) per-value / per-group constants through the descriptor's indices; (3) the
// eigenbasis column of every value — then rotates lane-parallel in LDS.  (The first version derived the layout in every solve
// and gathered the correction through the changes' positions: ~28 dependent round trips, 17.5 of the launch's 52.9 us on
// config 3, scripts/grp_profile.py.)
template <class T, bool FRP = false>
__device__ __forceinline__ void grp_solve_body_rot(const CdGrpBlkParams<T>& p, int j, char* smem_raw, int nt) {
#ifdef AHIP_GRP_PROFILE
    const long long t_entry = __builtin_readcyclecounter();
#endif
    T* D = reinterpret_cast<T*>(smem_raw); // GBLK*GBLK, rotated
    T* gT = D + GBLK * GBLK;  // rotated gradient of the block's values, kept current
    T* bT = gT + GBLK;        // rotated coefficients, current
    T* b0B = bT + GBLK;       // coefficients at block entry (original coordinates)
    T* AB = b0B + GBLK;
    T* xmT = AB + GBLK;       // rotated column means
    T* scr = xmT + GBLK;      // 8 * GBLK scratch
    T* gO = scr;              // gradient as handed over (original coordinates)
    T* xmO = scr + GBLK;
    T* delT = scr + 2 * GBLK; // rotated change of the group just visited
    T* gk_t = scr + 3 * GBLK; // (groups wider than a wavefront only)
    T* ako_t = scr + 4 * GBLK;
    T* ak_t = scr + 5 * GBLK;
    T* buf1 = scr + 6 * GBLK;
    T* buf2 = scr + 7 * GBLK;
    T* dl = buf2;             // prologue only: the previous block's dense changes
    int32_t* vmap = reinterpret_cast<int32_t*>(scr + 8 * GBLK);
    int32_t* goff = vmap + GBLK;
    int32_t* gq = goff + GBLK + 1;
    int32_t* gss = gq + GBLK;
    int32_t* meta = gss + GBLK;
    T* gpenB = reinterpret_cast<T*>(meta + 5);
    int32_t* gactB = reinterpret_cast<int32_t*>(gpenB + GBLK);
    int32_t* vgrp = gactB + GBLK;                       // value -> group of the block
    int32_t* chg = vgrp + GBLK;                         // group changed in this pass
    // partial sums of the look-ahead correction, nt / GBLK <= 8 parts (the eigenbasis pool of grp_solve_body is free here)
    T* corr = reinterpret_cast<T*>(chg + GBLK);

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int32_t* dsc = p.desc + size_t(j) * GDESC_STRIDE;
    const int ngrp = dsc[GDESC_NG], nval = dsc[GDESC_NVAL];
    const bool wide0 = nt == 1024;
    const bool has_cin = wide0 && p.corr_in != nullptr;  // the previous solve left the look-ahead correction (CdGrpBlkParams::Cnext)
    const bool has_corr = p.Cprev != nullptr && !has_cin;
    const int pnv = has_corr ? dsc[GDESC_NVAL - GDESC_STRIDE] : 0; // values of the previous block = columns of Cprev
    // (1) layout descriptor, the gradient handed over (or nothing: has_part), the previous block's dense changes; the rotated
    //     block and - fused launch, 1024 threads - the whole cross block, 16 values per thread, stream in meanwhile
    const bool wide = nt == 1024;
    const bool has_part = FRP && p.part != nullptr && wide; // (FRP: only the fused launch of kernels_cd_panel.hip compiles this in)
    int vm = 0, vg = 0, go = 0, gqv = 0, gs = 0, vof = 0, vps = 0, vq = 1;
    T gb = T(0), dlv = T(0);
    if (tid < GBLK) {
        vm = dsc[GDESC_VMAP + tid]; vg = dsc[GDESC_VGRP + tid];
        go = dsc[GDESC_GOFF + tid]; gqv = dsc[GDESC_GQ + tid]; gs = dsc[GDESC_GSS + tid];
        vof = dsc[GDESC_VOFF + tid]; vps = dsc[GDESC_VPOS + tid]; vq = dsc[GDESC_VQ + tid];
        if (!has_part) gb = p.gblk[tid];
        if (has_corr) dlv = p.pdd[tid];
        if (has_cin) dlv = p.corr_in[tid]; // (the correction itself; dl is not used on this route)
    }
    const T prs = (has_part && p.part_rsum) ? p.part_rsum[0] : T(0);
    // fused launch (1024 threads): the whole cross block, 16 values per thread (row tid % 128, columns 16 * part + u, part =
    // tid / 128 uniform per wavefront), against the previous block's dense changes read as scalars: the partial sums are
    // formed as the values arrive, no LDS staging and no barrier in between
    constexpr int CW = 16;
    T creg[CW];
    T dls[CW];
    const int part_u = __builtin_amdgcn_readfirstlane(tid >> 7);
    if (wide && has_corr) {
        const T* Cp = p.Cprev + (tid & (GBLK - 1)) + size_t(part_u) * CW * GBLK;
#pragma unroll
        for (int u = 0; u < CW; ++u) creg[u] = Cp[size_t(u) * GBLK];
#pragma unroll
        for (int u = 0; u < CW; ++u) dls[u] = p.pdd[part_u * CW + u];
    }
    {   // D~: columns [0, nval)
        const T* src = p.Dptr;
        const int NE = nval * GBLK;
        // Only entries BELOW the diagonal are ever read (the loop updates the gradient of the groups still to come: rows past
        // the visited group's columns): the prologue is bound by what one CU pulls while 196 others stream (D~ and the cross
        // block, 256 KB at ~60 GB/s), so the upper triangle is not fetched
        for (int e0 = tid; e0 < NE; e0 += nt * 16) {
            T v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int e = min(e0 + u * nt, NE - 1);
                v[u] = ((e & (GBLK - 1)) > (e / GBLK)) ? src[e] : T(0);
            }
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (e0 + u * nt < NE) D[e0 + u * nt] = v[u];
        }
    }
    if (wide && has_corr) {
        T acc = T(0);
#pragma unroll
        for (int u = 0; u < CW; ++u)
            if (part_u * CW + u < pnv) acc = fma(creg[u], dls[u], acc);
        corr[part_u * GBLK + (tid & (GBLK - 1))] = acc;
    }
    if (tid < GBLK) {
        vmap[tid] = vm; vgrp[tid] = vg; goff[tid] = go; gq[tid] = gqv; gss[tid] = gs;
        if (tid == 0) goff[GBLK] = dsc[GDESC_GOFF + GBLK];
        dl[tid] = dlv;
        if (!has_part) gO[tid] = tid < nval ? gb : T(0);
    }
    // (2) per-value / per-group constants and the eigenbasis column of every value through the descriptor's indices; the
    //     slice partials of the block's gradient when the reduction is fused into this launch: eight adjacent lanes per column,
    //     k = k0, k0 + 8, ... each, combined by a fixed shuffle tree
    T b0v = T(0), Av = T(0), xmv = T(0);
    if (tid < nval) { b0v = p.beta[vm]; Av = p.vars[vm]; xmv = p.xmean[vm]; }
    T penv = T(0);
    int actv = 0;
    if (tid < ngrp) { penv = p.spen[gs]; actv = p.is_active[gs]; }
    constexpr int QM = 16;
    T vcol_r[QM];
    const int o_v = tid - vps, q_v = vq;
    auto load_vcol = [&]() {
        if (tid < nval && q_v > 1 && q_v <= QM) {
            const T* Vt = p.V + int64_t(vof) + int64_t(vps) * q_v;
#pragma unroll
            for (int u = 0; u < QM; ++u) vcol_r[u] = u < q_v ? Vt[u] : T(0);
        }
    };
    if (!has_part) load_vcol();
    if (has_part) {
        constexpr int PBN = 25;
        const int c = tid >> 3, k0 = tid & 7;
        const T* pc = p.part + c;
        T pw[PBN];
#pragma unroll
        for (int u = 0; u < PBN; ++u) pw[u] = pc[int64_t(min(k0 + 8 * u, p.part_n - 1)) * GBLK];
        T ps = T(0);
#pragma unroll
        for (int u = 0; u < PBN; ++u) ps += (k0 + 8 * u < p.part_n) ? pw[u] : T(0);
        for (int k = k0 + 8 * PBN; k < p.part_n; k += 8) ps += pc[int64_t(k) * GBLK];
        ps += __shfl_xor(ps, 1, 64);
        ps += __shfl_xor(ps, 2, 64);
        ps += __shfl_xor(ps, 4, 64);
        if (k0 == 0) gO[c] = c < nval ? ps : T(0);
        load_vcol(); // (behind the partials: the register budget of the 1024-thread workgroup; lands during the barriers below)
    }
    __syncthreads(); // dl, the tables, gO
    // look-ahead correction in original coordinates: corr = Cprev[:, 0:pnv] dl[0:pnv], the columns split over nt / GBLK parts
    const int nparts = nt / GBLK;
    if (has_corr) {
        const int row = tid & (GBLK - 1), part = tid / GBLK;
        T acc = T(0);
        if (!wide) {
            const int per = (pnv + nparts - 1) / nparts;
            const int c0 = part * per, c1 = min(pnv, c0 + per);
            const T* Cp = p.Cprev + row;
            for (int c = c0; c < c1; c += 16) {
                T cv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) cv[u] = Cp[size_t(min(c + u, c1 - 1)) * GBLK];
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (c + u < c1) acc = fma(cv[u], dl[c + u], acc);
            }
            corr[part * GBLK + row] = acc;
        }
    }
    if (tid < GBLK) {
        b0B[tid] = b0v; AB[tid] = Av; xmO[tid] = xmv;
        if (tid >= nval) { gT[tid] = 0; bT[tid] = 0; xmT[tid] = 0; }
        if (tid < ngrp) { gpenB[tid] = penv; gactB[tid] = actv; chg[tid] = 0; }
    }
    __syncthreads(); // corr parts, b0B / xmO
    if (tid < nval && (has_corr || has_part || has_cin)) {
        T cs = T(0);
        if (has_corr)
            for (int q = 0; q < nparts; ++q) cs += corr[q * GBLK + tid]; // fixed order
        if (has_cin) cs = dlv;
        gO[tid] = (has_part ? gO[tid] - prs * xmO[tid] : gO[tid]) - cs;
    }
    __syncthreads();
    // every value into the eigenbasis of its group: (g V)_t, (beta V)_t, (xbar V)_t   (pin_naive:123-135)
    if (tid < nval) {
        const int i = tid, o = o_v, q = q_v;
        if (q == 1) {
            gT[i] = gO[i]; bT[i] = b0B[i]; xmT[i] = xmO[i];
        } else if (q <= QM) {
            T s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
            for (int u = 0; u < QM; ++u) {
                if (u < q) {
                    s1 = fma(gO[o + u], vcol_r[u], s1);
                    s2 = fma(b0B[o + u], vcol_r[u], s2);
                    s3 = fma(xmO[o + u], vcol_r[u], s3);
                }
            }
            gT[i] = s1; bT[i] = s2; xmT[i] = s3;
        } else {
            const T* Vt = p.V + int64_t(vof) + int64_t(i - o) * q;
            T s1 = 0, s2 = 0, s3 = 0;
#pragma unroll 4
            for (int u = 0; u < q; ++u) {
                const T v = Vt[u];
                s1 = fma(gO[o + u], v, s1);
                s2 = fma(b0B[o + u], v, s2);
                s3 = fma(xmO[o + u], v, s3);
            }
            gT[i] = s1; bT[i] = s2; xmT[i] = s3;
        }
    }
    // per-group constants of the sequential loop, one group per thread (the correction partials are consumed: their space is
    // free): 1 / max b_i, b_i = A_i + l2 (start of the root find), 1 / q (convergence measure), (dbeta_tol * sqrt(q))^2 (change test)
    T* gbmx = corr;
    T* grq = corr + GBLK;
    T* gsq = corr + 2 * GBLK;
    if (tid < ngrp) {
        const int o = go, q = gqv;
        const T l2p = p.l2 * penv;
        T mx = T(0);
        for (int t = 0; t < q; ++t) { const T b = AB[o + t] + l2p; mx = b > mx ? b : mx; }
        gbmx[tid] = mx > T(0) ? T(1) / mx : T(0); // (reciprocal: the root find's start is (||v|| - l1) / max b_i)
        grq[tid] = T(1) / T(q);
        const T thr = p.dbeta_tol * sqrt(T(q));
        gsq[tid] = thr * thr;                     // ||delta||^2 against the squared threshold of pin_naive:144
    }
    __syncthreads();
    // correction of the NEXT block by the idle waves (CdGrpBlkParams::Cnext): partition and order of blk.. the prologue's own
    // form above — thread (part = tid / 128, row = tid % 128) owns columns 16 part .. 16 part + 15; wavefront 1 also takes the
    // rows of wavefront 0, which is busy visiting
    const bool mk_next = wide0 && p.Cnext != nullptr;
    T* dnx = corr + 8 * GBLK;    // this block's dense changes (written by the visiting wave's epilogue)
    if (wv != 0) {
        if (!mk_next) return;
        const int row = tid & (GBLK - 1), part = __builtin_amdgcn_readfirstlane(tid >> 7);
        T cn[CW], cn0[CW];
        const T* Cp = p.Cnext + row + size_t(part) * CW * GBLK;
#pragma unroll
        for (int u = 0; u < CW; ++u) cn[u] = Cp[size_t(u) * GBLK];
        if (wv == 1) {
#pragma unroll
            for (int u = 0; u < CW; ++u) cn0[u] = Cp[size_t(u) * GBLK - 64];
        }
        __syncthreads(); // (A) the visits are over: dnx is complete, the loop's per-group constants in `corr` are dead
        T acc = T(0), acc0 = T(0);
#pragma unroll
        for (int u = 0; u < CW; ++u)
            if (part * CW + u < nval) acc = fma(cn[u], dnx[part * CW + u], acc);
        if (wv == 1) {
#pragma unroll
            for (int u = 0; u < CW; ++u)
                if (u < nval) acc0 = fma(cn0[u], dnx[u], acc0);
        }
        __builtin_amdgcn_wave_barrier();
        corr[part * GBLK + row] = acc;
        if (wv == 1) corr[row - 64] = acc0;
        __syncthreads(); // (B)
        if (tid < GBLK) {
            T cs = T(0);
            for (int q = 0; q < 8; ++q) cs += corr[q * GBLK + tid]; // fixed order, as the prologue's
            p.corr_out[tid] = cs;
        }
        return;
    }
    __builtin_amdgcn_s_setprio(3);
#ifdef AHIP_GRP_PROFILE
    // cycle profile of the wave (scripts/grp_profile.py): 0 prologue (kernel entry to here), 1 operand loads + norm, 2 Newton,
    // 3 change test, 4 marking + gradient update of the groups to come, 5 epilogue
    long long tmark = t_entry;
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define RP_MARK(k) { const long long tn = __builtin_readcyclecounter(); tacc[k] += tn - tmark; tmark = tn; }
    RP_MARK(0)
#else
#define RP_MARK(k)
#endif

    CdBlkState<T>* st = p.st;
    T rsq = st->rsq, rsum = st->resid_sum, cm = (j == 0) ? T(0) : st->cm;
    int asz = st->active_size, status = st->status;
    int64_t n_upd = st->n_updates;
    T rs_acc = T(0), xs_acc = T(0); // lane partials of the rsq / resid_sum updates of the groups with q > 1 (summed once, below)

    // operands of a group that do not depend on the groups before it (layout, penalty, variances, old coefficients) are read
    // one group ahead, so that their LDS latency is not on the chain
    // The rotated gradient of the next group's values travels the same way: lane t holds the value o_next + t (the whole
    // 64-lane chunk behind the current group); the update of a changed group (q <= 64) leaves exactly that chunk in its
    // accumulator registers, so the next visit starts from registers instead of an LDS store -> load round trip.
    int o_n = 0, q_n = 0, ss_n = 0;
    T pk_n = T(0), A_n = T(0), ako_n = T(0), g_n = T(0);
    auto prefetch = [&](int kk) {
        if (kk < ngrp) {
            o_n = goff[kk]; q_n = gq[kk]; ss_n = gss[kk];
            pk_n = gpenB[kk];
            const bool onn = lane < q_n;
            A_n = onn ? AB[o_n + lane] : T(0);
            ako_n = onn ? bT[o_n + lane] : T(0);
            g_n = (o_n + lane < nval) ? gT[o_n + lane] : T(0);
        }
    };
    prefetch(0);
    for (int k = 0; k < ngrp && status == CD_OK; ++k) {
        const int o = o_n, q = q_n, ss = ss_n;
        const T pk = pk_n;
        const T A_c = A_n, ako_c = ako_n, g_c = g_n;
        prefetch(k + 1);
        const T l1p = p.l1 * pk, l2p = p.l2 * pk;
        bool changed = false;
        T d_reg = T(0); // q <= 64: lane t holds the rotated change of the group's value t (q == 1: every lane)
        if (q == 1) {
            const T gcur = rdl(g_c, 0), bi = rdl(ako_c, 0), A = rdl(A_c, 0);
            const T gk = fma(bi, A, gcur);                       // pin_naive:85-89
            const T v = fabs(gk) - l1p;                          // pin_base:181-195
            T ak = (v > T(0)) ? copysign(v, gk) / (A + l2p) : T(0);
            if (p.clo) ak = grp_clip_1d(p, vmap[o], ak, gk, l1p, A + l2p, lane); // constraint->solve, pin_naive:419-437
            if (ak != bi) {                                      // pin_naive:97
                changed = true;
                const T d = ak - bi;
                const T c1 = A * d * d;
                cm = c1 > cm ? c1 : cm;
                rsq += d * (T(2) * gcur - d * A);
                rsum -= xmT[o] * d;
                if (lane == 0) bT[o] = ak;
                d_reg = d; // the same in every lane
            }
        } else if (q <= 64) {
            const bool on = lane < q;
            const T A_r = A_c;
            const T ako_r = ako_c;
            const T gk_r = on ? g_c + A_r * ako_r : T(0); // gk_t += A * ak_old_t   (pin_naive:139-140)
            // newton_solver (bcd/unconstrained/newton.hpp:35-142): v = gk_t, L = A
            const T gk2 = on ? gk_r * gk_r : T(0);
            T nrm2 = gk2;
            // (double precision only, see below; the second reduction is independent of the first: they overlap)
            constexpr bool kMeanStart = sizeof(T) == 8;
            T sb1 = T(0);
            if constexpr (kMeanStart) {
                sb1 = on ? gk2 * (A_r + l2p) : T(0);
                group_sum2(nrm2, sb1, q);
            } else {
                nrm2 = group_sum(gk2, q);
            }
            const T nrm = sqrt(nrm2);
            T akt_r = T(0);
            RP_MARK(1)
            if (nrm <= l1p) {
                akt_r = T(0);
            } else if (l1p <= T(0)) {
                akt_r = on ? gk_r / (A_r + l2p) : T(0);
            } else {
                const T b1 = A_r + l2p;
                T h = 0, fh, tt = T(0), sxx = T(0), b2 = T(0);
                // Start of the root find.  The reference starts at h = 0 (bcd/unconstrained/newton.hpp:137); the secular
                // function  f(h) = sum v_i^2 / (b_i h + l1)^2 - 1  is convex and decreasing, and
                //     h_lb = (||v|| - l1) / max_i b_i
                // has f(h_lb) >= 0 (replace every b_i by the largest), so Newton from h_lb converges monotonically from the
                // left like from 0, only from much closer — the same root to newton_tol in fewer of the (strictly
                // sequential) evaluations; an isotropic block (all b_i equal) starts AT its root.  Same stopping test, same
                // error when newton_tol is unreachable.
                h = (nrm - l1p) * gbmx[k]; // (1 / max_i b_i from the prologue; 0 when there is none)
                // Round 4: start at (||v|| - l1) / b_eff with the v^2-weighted mean  b_eff = sum v_i^2 b_i / sum v_i^2  instead of
                // the largest b_i.  Exact for an isotropic block like h_lb, and second-order accurate in the spread of the b_i
                // (h_lb is first-order): on groups of 10 columns of a random design (spread ~ sqrt(q / n) = 1 %) the start is
                // 1e-4 instead of 1e-2 from the root, one evaluation less of the strictly sequential three to four
                // (config 3: 602.7 -> 581 ms; a third-order start with the weighted variance of the b_i on top measured slower,
                // 586 ms: its extra reduction costs more than it saves).  It may lie
                // to the RIGHT of the root; the iteration below is safe from either side (one step brings it to the left, from
                // where it is monotone; h is clamped at 0 as in the reference).  Same stopping test, same result to newton_tol.
                // Double precision only: in single precision newton_tol has to be loose (1e-5) and a start that may already
                // pass it from the right accepts a different point of the tolerance interval than the reference's approach from
                // the left — on constrained problems enough to move multipliers (tests/test_constraint.py, f32).
                if constexpr (kMeanStart) {
                    if (sb1 > T(0)) {
                        const T rb = nrm2 * fast_rcp(sb1); // 1 / b_eff (a start only has to be close: the hardware approximation,
                                                           // off the IEEE division's dependent chain: 581 -> 569 ms on config 3)
                        h = (nrm - l1p) * rb;
                    }
                }
                auto step = [&](T hh) {
                    T t = 0, sx = 0;
                    if (on) {
                        b2 = T(1) / (b1 * hh + l1p); // (v_rcp + refinements instead of the IEEE division: -1 % on config 3, not kept)
                        const T z = gk_r * b2;
                        const T x = z * z;
                        t = x;
                        sx = x * b1 * b2;
                    }
                    group_sum2(t, sx, q);
                    tt = t;
                    sxx = sx;
                    fh = tt - T(1);
                };
                step(h);
                int iters = 0;
                while ((fabs(fh) > p.newton_tol) && (iters < p.newton_max_iters)) {
                    // h -= f / f'  with  f' = -sx (1 + sqrt t) / t  (newton.hpp:83-93):  f / f' = -t (sqrt t - 1) / sx, one
                    // division instead of two on the dependent chain
                    h += tt * (sqrt(tt) - T(1)) / sxx;
                    h = h > T(0) ? h : T(0);
                    step(h);
                    ++iters;
                }
                akt_r = on ? h * gk_r * b2 : T(0);
                if (iters >= p.newton_max_iters) { status = CD_NEWTON; break; }
            }
            RP_MARK(2)
            // changed? ; convergence / rsq in rotated coordinates (pin_naive:144-154)
            T d = T(0), rs = T(0), dn = T(0), c1 = T(0);
            if (on) {
                const T gg = gk_r - A_r * ako_r;
                d = akt_r - ako_r;
                dn = d * d;
                c1 = (A_r * d) * d;
                rs = d * (T(2) * gg - d * A_r);
            }
            group_sum2(dn, c1, q);
            if (!(dn <= gsq[k])) {
                changed = true;
                c1 *= grq[k];
                cm = c1 > cm ? c1 : cm;
                rs_acc += rs;
                if (on) {
                    xs_acc = fma(xmT[o + lane], d, xs_acc);   // resid_sum -= xbar . del = (xbar V) . del_t   (pin_naive:161-163)
                    bT[o + lane] = akt_r;
                }
                d_reg = d;
            }
        } else {
            const T* A = AB + o;
            for (int i = lane; i < q; i += 64) {
                ako_t[i] = bT[o + i];
                gk_t[i] = gT[o + i] + A[i] * bT[o + i];
            }
            __builtin_amdgcn_wave_barrier();
            T nrm2 = 0;
            for (int i = lane; i < q; i += 64) nrm2 = fma(gk_t[i], gk_t[i], nrm2);
            nrm2 = gwsum(nrm2);
            if (sqrt(nrm2) <= l1p) {
                for (int i = lane; i < q; i += 64) ak_t[i] = 0;
            } else if (l1p <= T(0)) {
                for (int i = lane; i < q; i += 64) ak_t[i] = gk_t[i] / (A[i] + l2p);
            } else {
                for (int i = lane; i < q; i += 64) buf1[i] = A[i] + l2p;
                T h = 0, fh, dfh;
                auto step = [&](T hh) {
                    T t = 0, sx = 0;
                    for (int i = lane; i < q; i += 64) {
                        const T b2 = T(1) / (buf1[i] * hh + l1p);
                        const T z = gk_t[i] * b2;
                        const T x = z * z;
                        buf2[i] = b2;
                        t += x;
                        sx += x * buf1[i] * b2;
                    }
                    t = gwsum(t);
                    sx = gwsum(sx);
                    fh = t - T(1);
                    dfh = -sx * (T(1) + sqrt(t)) / t;
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
                if (iters >= p.newton_max_iters) { status = CD_NEWTON; break; }
            }
            __builtin_amdgcn_wave_barrier();
            T dn = 0, c1 = 0, rs = 0;
            for (int i = lane; i < q; i += 64) {
                const T gg = gk_t[i] - A[i] * ako_t[i];
                const T d = ak_t[i] - ako_t[i];
                dn = fma(d, d, dn);
                c1 = fma(A[i] * d, d, c1);
                rs += d * (T(2) * gg - d * A[i]);
            }
            dn = gwsum(dn);
            c1 = gwsum(c1);
            if (!(sqrt(dn) <= p.dbeta_tol * sqrt(T(q)))) {
                changed = true;
                c1 /= T(q);
                cm = c1 > cm ? c1 : cm;
                rs_acc += rs;
                for (int i = lane; i < q; i += 64) {
                    const T d = ak_t[i] - ako_t[i];
                    xs_acc = fma(xmT[o + i], d, xs_acc);
                    bT[o + i] = ak_t[i];
                    delT[i] = d;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        RP_MARK(3)
        if (changed) {
            if (lane == 0) chg[k] = 1;
            if (p.mark && gactB[k] == 0) {                         // add_active_set, pin_naive:294-304
                if (asz >= p.max_active_size) { status = CD_MAX_ACTIVE; break; }
                if (lane == 0) { p.is_active[ss] = 1; p.active_set[asz] = ss; }
                ++asz;
            }
            // keep the rotated gradient of the groups STILL TO COME current: gT[l] -= D~[l, o:o+q] del_t for l >= o + q (the
            // values before that were visited already and are not read again in this solve).  q <= 64: the changes come out
            // of the lanes' registers with v_readlane (uniform index), no LDS round trip on the chain.
            if (q > 1 && q <= 16) {
                // Every entry of the block a chunk needs is requested before the first multiply-add (one LDS round trip per chunk
                // instead of one per four columns), without a predicate per column: the trip count is q rounded up to a multiple
                // of four, the surplus columns (clamped to the block's last value: finite entries) meet the zero changes of lanes >= q.
                auto upd = [&](auto qb_c) {
                    constexpr int QB = decltype(qb_c)::value;
                    T dv[QB];
#pragma unroll
                    for (int t = 0; t < QB; ++t) dv[t] = rdl(d_reg, t);
                    bool first = true;
                    for (int l = o + q + lane; l < nval; l += 64) {
                        T dd[QB];
#pragma unroll
                        for (int t = 0; t < QB; ++t) dd[t] = D[l + min(o + t, nval - 1) * GBLK]; // (columns >= nval were never written)
                        T acc = first ? g_n : gT[l]; // (the first chunk is the one prefetched for the next group)
#pragma unroll
                        for (int t = 0; t < QB; ++t) acc = fma(-dd[t], dv[t], acc);
                        gT[l] = acc;
                        if (first) g_n = acc;
                        first = false;
                    }
                };
                if (q <= 4) upd(std::integral_constant<int, 4>{});
                else if (q <= 8) upd(std::integral_constant<int, 8>{});
                else if (q <= 12) upd(std::integral_constant<int, 12>{});
                else upd(std::integral_constant<int, 16>{});
            } else if (q <= 64) {
                bool first = true;
                for (int l = o + q + lane; l < nval; l += 64) {
                    T acc = first ? g_n : gT[l]; // (the first chunk is the one prefetched for the next group)
#pragma unroll 4
                    for (int t = 0; t < q; ++t) acc = fma(-D[l + (o + t) * GBLK], rdl(d_reg, q == 1 ? 0 : t), acc);
                    gT[l] = acc;
                    if (first) g_n = acc;
                    first = false;
                }
            } else {
                for (int l = o + q + lane; l < nval; l += 64) {
                    T acc = gT[l];
#pragma unroll 4
                    for (int t = 0; t < q; ++t) acc = fma(-D[l + (o + t) * GBLK], delT[t], acc);
                    gT[l] = acc;
                }
                __builtin_amdgcn_wave_barrier();
                if (k + 1 < ngrp) g_n = (o_n + lane < nval) ? gT[o_n + lane] : T(0);
            }
            __builtin_amdgcn_wave_barrier();
            ++n_upd;
        }
        RP_MARK(4)
    }
    rsq += gwsum(rs_acc);
    rsum -= gwsum(xs_acc);
    __builtin_amdgcn_wave_barrier();
    // back into original coordinates: ak = ak_t V^T for the groups that changed (pin_naive:156-157); the others keep their
    // coefficients bit for bit.  Then the compacted non-zero value changes for the update kernel, as in grp_solve_body.
    int nz = 0;
    for (int i0 = 0; i0 < GBLK; i0 += 64) {
        const int i = i0 + lane;
        T bnew = T(0), b0 = T(0);
        bool ch = false;
        if (i < nval) {
            const int k = vgrp[i], o = goff[k], q = gq[k];
            b0 = b0B[i];
            bnew = b0;
            if (chg[k]) {
                if (q == 1) {
                    bnew = bT[i];
                } else {
                    const T* Vg = p.V + p.voff[gss[k]];
                    T s = 0;
#pragma unroll 4
                    for (int jj = 0; jj < q; ++jj) s = fma(bT[o + jj], Vg[(i - o) + int64_t(jj) * q], s);
                    bnew = s;
                }
            }
            ch = bnew != b0;
        }
        if (ch) p.beta[vmap[i]] = bnew;
        if (p.dd) p.dd[i] = bnew - b0; // dense form for the next block's look-ahead correction (0 beyond the block)
        if (mk_next) dnx[i] = bnew - b0;
        const unsigned long long m = __ballot(ch);
        const int pos = nz + __popcll(m & ((1ull << lane) - 1ull));
        if (ch) {
            p.dcol[pos] = p.vcol[vmap[i]];
            if (p.dpos) p.dpos[pos] = i;
            p.dlt[pos] = bnew - b0;
        }
        nz += __popcll(m);
    }
    if (lane == 0) {
        st->rsq = rsq;
        st->resid_sum = rsum;
        st->cm = cm;
        st->active_size = asz;
        st->status = status;
        st->n_updates = n_upd;
        st->nz = nz;
        if (p.nz_out) {
            p.nz_out[0] = nz;
            p.rsum_out[0] = rsum;
        }
#ifdef AHIP_GRP_PROFILE
        RP_MARK(5)
        if (p.dbg) for (int k = 0; k < 6; ++k) atomicAdd(reinterpret_cast<unsigned long long*>(p.dbg) + k, (unsigned long long)tacc[k]);
        if (p.dbg) atomicAdd(reinterpret_cast<unsigned long long*>(p.dbg) + 7, 1ull);
#endif
        if (p.host_st && j == p.report_j) {
            CdBlkState<T> out;
            out.rsq = rsq; out.resid_sum = rsum; out.cm = cm; out.n_updates = n_upd;
            out.active_size = asz; out.status = status; out.nz = nz; out._pad = 0;
            *p.host_st = out;
            __threadfence_system();
            __hip_atomic_store(p.host_seq, p.report_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (mk_next) { // the visiting wave's side of the two barriers above; its rows are summed by wavefront 1, stored here
        __syncthreads(); // (A)
        __syncthreads(); // (B)
        T cs = T(0);
        for (int q = 0; q < 8; ++q) cs += corr[q * GBLK + lane];
        p.corr_out[lane] = cs;
    }
}

#undef RP_MARK

// nt: threads of the workgroup that call (>= 256, a multiple of 128); only the rotated form uses more than the first 256
template <class T, bool NAIVE, bool FRP = false>
__device__ __forceinline__ void grp_solve_body(const CdGrpBlkParams<T>& p, int j, char* smem_raw, int nt = 256) {
    if constexpr (NAIVE) {
        if (p.rot) { // uniform over the workgroup
            grp_solve_body_rot<T, FRP>(p, j, smem_raw, nt);
            return;
        }
    }
    if (threadIdx.x >= 256) return; // before any barrier: ended waves do not take part in them
    T* D = reinterpret_cast<T*>(smem_raw); // GBLK*GBLK
    T* gB = D + GBLK * GBLK;
    T* bB = gB + GBLK;     // current beta of the block's values
    T* b0B = bB + GBLK;    // beta at block entry
    T* AB = b0B + GBLK;
    T* xmB = AB + GBLK;
    T* scr = xmB + GBLK;   // 8 * GBLK group scratch
    int32_t* vmap = reinterpret_cast<int32_t*>(scr + 8 * GBLK);
    int32_t* goff = vmap + GBLK;
    int32_t* gq = goff + GBLK + 1;
    int32_t* gss = gq + GBLK;
    int32_t* meta = gss + GBLK;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    block_layout(p, j, vmap, goff, gq, gss, meta);
    const int ngrp = meta[0], nval = meta[1];
    // per-group constants and (when they fit) all eigenbases of the block, fetched by the whole workgroup up front so that
    // the sequential wave never waits on a global load
    T* gpenB = reinterpret_cast<T*>(meta + 5); // int index 518 of the region: 8-byte aligned
    int32_t* gactB = reinterpret_cast<int32_t*>(gpenB + GBLK);
    int32_t* gvoB = gactB + GBLK;          // offset of the group's eigenbasis in Vpool, or -1
    T* Vpool = reinterpret_cast<T*>(gvoB + GBLK);
    if (tid < GBLK && tid < ngrp) {
        const int ss = gss[tid];
        gpenB[tid] = p.spen[ss];
        gactB[tid] = p.is_active[ss];
    }
    if (tid == 0) {
        int used = 0;
        for (int k = 0; k < ngrp; ++k) {
            const int q = gq[k];
            if (q > 1 && used + q * q <= VPOOL) { gvoB[k] = used; used += q * q; }
            else gvoB[k] = -1;
        }
    }
    __syncthreads();
    // gidB[k] != 0: the eigenbasis of group k is exactly the identity (isotropic blocks, e.g. every group of a multi-response
    // view): the two rotations of its visit reduce to copies -- bit-identical, the general loops would add exact zeros
    int32_t* gidB = reinterpret_cast<int32_t*>(smem_raw + ((grp_solve_lds<T>() + 15) / 16) * 16 + size_t(2) * GBLK * sizeof(T));
    for (int k = wv; k < ngrp; k += 4) { // one wave per group
        const int vo = gvoB[k];
        if (vo < 0) {
            if (lane == 0) gidB[k] = 0;
            continue;
        }
        const int q = gq[k];
        const T* Vg = p.V + p.voff[gss[k]];
        bool ident = true;
        for (int e = lane; e < q * q; e += 64) {
            const T v = Vg[e];
            Vpool[vo + e] = v;
            ident = ident && (v == ((e / q == e % q) ? T(1) : T(0)));
        }
        const bool all_ident = __ballot(!ident) == 0ull;
        if (lane == 0) gidB[k] = all_ident ? 1 : 0;
    }
    if (tid < GBLK) {
        const int i = tid;
        if (i < nval) {
            const int a = vmap[i];
            gB[i] = NAIVE ? p.gblk[i] : p.g[a];
            bB[i] = p.beta[a];
            b0B[i] = bB[i];
            AB[i] = p.vars[a];
            xmB[i] = p.xmean[a];
        } else {
            gB[i] = 0; bB[i] = 0; b0B[i] = 0; AB[i] = 0; xmB[i] = 0;
        }
    }
    // look-ahead correction (CdGrpBlkParams::Cprev): gB was taken from a residual without the previous block's changes;
    // two threads per value sum  Cprev[value, ppos[m]] * pdlt[m]  over halves of the m-range into LDS, combined below
    T* corr = reinterpret_cast<T*>(smem_raw + ((grp_solve_lds<T>() + 15) / 16) * 16);
    const bool has_corr = NAIVE && p.Cprev != nullptr;
    if (has_corr) {
        const int row = tid & (GBLK - 1), half = tid >> 7;
        const int nzp = p.pnz[0];
        const int per = (nzp + 1) / 2;
        const int m0 = half * per, m1 = min(nzp, m0 + per);
        const T* Cp = p.Cprev + row;
        T acc = T(0);
        int m = m0;
        for (; m + 8 <= m1; m += 8) {
            T c[8], d[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                c[u] = Cp[size_t(p.ppos[m + u]) * GBLK];
                d[u] = p.pdlt[m + u];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = fma(c[u], d[u], acc);
        }
        for (; m < m1; ++m) acc = fma(Cp[size_t(p.ppos[m]) * GBLK], p.pdlt[m], acc);
        corr[half * GBLK + row] = acc;
    }
    {
        const T* src = NAIVE ? p.Dptr : p.Dbuf + size_t(j & 1) * GBLK * GBLK;
        // 16 loads in flight per lane; columns [0, nval) of the slot
        const int NE = nval * GBLK;
        for (int e0 = tid; e0 < NE; e0 += 256 * 16) {
            T v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = src[min(e0 + u * 256, NE - 1)];
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (e0 + u * 256 < NE) D[e0 + u * 256] = v[u];
        }
    }
    __syncthreads();
    if (has_corr) { // uniform over the workgroup
        if (tid < nval) gB[tid] -= corr[tid] + corr[GBLK + tid];
        __syncthreads();
    }
    if (wv != 0) return;
    __builtin_amdgcn_s_setprio(3); // critical path of the pass: win the issue arbitration on this CU

#ifdef AHIP_GRP_PROFILE
    long long tmark = __builtin_readcyclecounter();
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define GP_MARK(k) { const long long tn = __builtin_readcyclecounter(); tacc[k] += tn - tmark; tmark = tn; }
    GP_MARK(0) /* prologue (since wave start is not visible: from here) */
#else
#define GP_MARK(k)
#endif
    CdBlkState<T>* st = p.st;
    T rsq = st->rsq, rsum = st->resid_sum, cm = (j == 0) ? T(0) : st->cm;
    int asz = st->active_size, status = st->status;
    int64_t n_upd = st->n_updates;
    T* gk_t = scr;
    T* ak_old_t = scr + GBLK;
    T* ak_t = scr + 2 * GBLK;
    T* buf1 = scr + 3 * GBLK;
    T* buf2 = scr + 4 * GBLK;
    T* del = scr + 5 * GBLK;

    for (int k = 0; k < ngrp && status == CD_OK; ++k) {
        const int o = goff[k], q = gq[k], ss = gss[k];
        const T pk = gpenB[k];
        const T l1p = p.l1 * pk, l2p = p.l2 * pk;
        bool changed = false;
        if (q == 1) {
            const T gcur = gB[o], bi = bB[o], A = AB[o];
            const T gk = fma(bi, A, gcur);                       // pin_naive:85-89
            const T v = fabs(gk) - l1p;                          // pin_base:181-195
            T ak = (v > T(0)) ? copysign(v, gk) / (A + l2p) : T(0);
            if (p.clo) ak = grp_clip_1d(p, vmap[o], ak, gk, l1p, A + l2p, lane); // constraint->solve, pin_naive:419-437 / pin_cov:723-741
            if (ak != bi) {                                      // pin_naive:97
                changed = true;
                const T d = ak - bi;
                const T c1 = A * d * d;
                cm = c1 > cm ? c1 : cm;
                rsq += d * (T(2) * gcur - d * A);
                rsum -= xmB[o] * d;
                if (lane == 0) { bB[o] = ak; del[0] = d; }
            }
        } else {
            // stage the (q,q) eigenbasis in LDS (one coalesced read) when it fits the scratch; else read it in place
            const T* V;
            if (gvoB[k] >= 0) {
                V = Vpool + gvoB[k];
            } else {
                const T* Vg = p.V + p.voff[ss];
                V = Vg;
                if (q * q <= 2 * GBLK) {
                    T* Vl = scr + 6 * GBLK;
                    for (int e = lane; e < q * q; e += 64) Vl[e] = Vg[e];
                    V = Vl;
                    __builtin_amdgcn_wave_barrier();
                }
            }
            const T* A = AB + o;
            T dn = 0, c1 = 0, rs = 0;
            const bool idV = q <= 64 && gvoB[k] >= 0 && gidB[k] != 0;
            if (q <= 64) {
                // One element per lane: the group's rotated vectors stay in registers from the rotation to the change test
                // (same operations in the same order as the general path below, so the results are bit-identical; only the
                // LDS round trips between the phases are gone).
                const bool on = lane < q;
                const T A_r = on ? A[lane] : T(0);
                T gk_r = T(0), ako_r = T(0);
                if (on) { // gk_t = gk V ; ak_old_t = ak_old V ; gk_t += A * ak_old_t   (pin_naive:123-140)
                    T s1 = 0, s2 = 0;
                    if (idV) {
                        s1 = gB[o + lane];
                        s2 = bB[o + lane];
                    } else {
                        const T* Vj = V + int64_t(lane) * q;
#pragma unroll 4
                        for (int i = 0; i < q; ++i) {
                            s1 = fma(gB[o + i], Vj[i], s1);
                            s2 = fma(bB[o + i], Vj[i], s2);
                        }
                    }
                    ako_r = s2;
                    gk_r = s1 + A_r * s2;
                }
                GP_MARK(1) /* rotation */
                // newton_solver (bcd/unconstrained/newton.hpp:35-142): v = gk_t, L = A
                const T nrm2 = group_sum(on ? gk_r * gk_r : T(0), q);
                T akt_r = T(0);
                if (sqrt(nrm2) <= l1p) {
                    akt_r = T(0);
                } else if (l1p <= T(0)) {
                    akt_r = on ? gk_r / (A_r + l2p) : T(0);
                } else {
                    const T b1 = A_r + l2p;
                    T h = 0, fh, dfh, b2 = T(0);
                    // Isotropic block (all eigenvalues equal: every group of a multi-response view, X_g^T W X_g (x) I_K): the
                    // root of the secular equation is known in closed form, (||v|| - l1) / (A + l2).  The iteration of the
                    // reference (start h = 0, bcd/unconstrained/newton.hpp:60-129) is kept, but started there: it then stops
                    // after its first evaluation whenever newton_tol is reachable, and behaves as in the reference
                    // (iterates to newton_max_iters -> the same error) when it is not.
                    {
                        const T a0 = first_lane(A_r);
                        const bool iso = q > 1 && __ballot(on && A_r != a0) == 0ull && (a0 + l2p) > T(0);
                        if (iso) h = (sqrt(nrm2) - l1p) / (a0 + l2p);
                    }
                    auto step = [&](T hh) {
                        T t = 0, sx = 0;
                        if (on) {
                            b2 = T(1) / (b1 * hh + l1p);
                            const T z = gk_r * b2;
                            const T x = z * z;
                            t = x;
                            sx = x * b1 * b2;
                        }
                        t = group_sum(t, q);
                        sx = group_sum(sx, q);
                        fh = t - T(1);
                        dfh = -sx * (T(1) + sqrt(t)) / t;
                    };
                    step(h);
                    int iters = 0;
                    while ((fabs(fh) > p.newton_tol) && (iters < p.newton_max_iters)) {
                        h -= fh / dfh;
                        h = h > T(0) ? h : T(0);
                        step(h);
                        ++iters;
                    }
                    akt_r = on ? h * gk_r * b2 : T(0);
                    if (iters >= p.newton_max_iters) { status = CD_NEWTON; break; }
                }
                if (on) ak_t[lane] = akt_r; // read by every lane in the back rotation
                __builtin_amdgcn_wave_barrier();
                GP_MARK(2) /* norm + newton */
                // changed? ; convergence / rsq in rotated coordinates (pin_naive:144-154)
                if (on) {
                    const T gg = gk_r - A_r * ako_r;
                    const T d = akt_r - ako_r;
                    dn = d * d;
                    c1 = (A_r * d) * d;
                    rs = d * (T(2) * gg - d * A_r);
                }
                dn = group_sum(dn, q);
                c1 = group_sum(c1, q);
                rs = group_sum(rs, q);
            } else {
                const T* A = AB + o;
                // gk_t = gk V ; ak_old_t = ak_old V ; gk_t += A * ak_old_t   (pin_naive:123-140)
                for (int jj = lane; jj < q; jj += 64) {
                    T s1 = 0, s2 = 0;
                    const T* Vj = V + int64_t(jj) * q;
#pragma unroll 4
                    for (int i = 0; i < q; ++i) {
                        s1 = fma(gB[o + i], Vj[i], s1);
                        s2 = fma(bB[o + i], Vj[i], s2);
                    }
                    ak_old_t[jj] = s2;
                    gk_t[jj] = s1 + A[jj] * s2;
                }
                __builtin_amdgcn_wave_barrier();
                GP_MARK(1) /* rotation */
                // newton_solver (bcd/unconstrained/newton.hpp:35-142): v = gk_t, L = A
                T nrm2 = 0;
                for (int i = lane; i < q; i += 64) nrm2 = fma(gk_t[i], gk_t[i], nrm2);
                nrm2 = group_sum(nrm2, q);
                if (sqrt(nrm2) <= l1p) {
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
                        t = group_sum(t, q);
                        s = group_sum(s, q);
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
                    if (iters >= p.newton_max_iters) { status = CD_NEWTON; break; }
                }
                __builtin_amdgcn_wave_barrier();
                GP_MARK(2) /* norm + newton */
                // changed? ; convergence / rsq in rotated coordinates (pin_naive:144-154)
                for (int i = lane; i < q; i += 64) {
                    const T gg = gk_t[i] - A[i] * ak_old_t[i];
                    const T d = ak_t[i] - ak_old_t[i];
                    dn = fma(d, d, dn);
                    c1 = fma(A[i] * d, d, c1);
                    rs += d * (T(2) * gg - d * A[i]);
                }
                dn = group_sum(dn, q);
                c1 = group_sum(c1, q);
                rs = group_sum(rs, q);
            }
            if (!(sqrt(dn) <= p.dbeta_tol * sqrt(T(q)))) {
                changed = true;
                c1 /= T(q);
                cm = c1 > cm ? c1 : cm;
                rsq += rs;
                // ak = ak_t V^T ; del = ak - ak_old ; resid_sum -= xbar . del   (pin_naive:156-163)
                T rsd = 0;
                for (int i = lane; i < q; i += 64) {
                    T s = 0;
                    if (idV) {
                        s = ak_t[i];
                    } else {
#pragma unroll 4
                        for (int jj = 0; jj < q; ++jj) s = fma(ak_t[jj], V[i + int64_t(jj) * q], s);
                    }
                    const T d = s - bB[o + i];
                    del[i] = d;
                    bB[o + i] = s;
                    rsd = fma(xmB[o + i], d, rsd);
                }
                rsum -= group_sum(rsd, q);
            }
        }
        __builtin_amdgcn_wave_barrier();
        GP_MARK(3) /* changed test + back rotation */
        if (changed) {
            if (p.mark && gactB[k] == 0) {                         // add_active_set, pin_naive:294-304
                if (asz >= p.max_active_size) { status = CD_MAX_ACTIVE; break; }
                if (lane == 0) { p.is_active[ss] = 1; p.active_set[asz] = ss; }
                ++asz;
            }
            // keep the block's gradient current: gB -= D[:, o:o+q] del
            for (int l = lane; l < GBLK; l += 64) {
                T acc = gB[l];
#pragma unroll 4
                for (int t = 0; t < q; ++t) acc = fma(-D[l + (o + t) * GBLK], del[t], acc);
                gB[l] = acc;
            }
            __builtin_amdgcn_wave_barrier();
            ++n_upd;
        }
        GP_MARK(4) /* active marking + block gradient update */
    }
    // write back beta and the compacted non-zero value changes for the update kernel
    int nz = 0;
    for (int i0 = 0; i0 < GBLK; i0 += 64) {
        const int i = i0 + lane;
        const T d = (i < nval) ? (bB[i] - b0B[i]) : T(0);
        const bool ch = (i < nval) && (bB[i] != b0B[i]);
        if (ch) p.beta[vmap[i]] = bB[i];
        const unsigned long long m = __ballot(ch);
        const int pos = nz + __popcll(m & ((1ull << lane) - 1ull));
        if (ch) {
            if (NAIVE) {
                p.dcol[pos] = p.vcol[vmap[i]];
                if (p.dpos) p.dpos[pos] = i;
            } else {
                p.didx[pos] = vmap[i];
            }
            p.dlt[pos] = d;
        }
        nz += __popcll(m);
    }
    if (lane == 0) {
        st->rsq = rsq;
        st->resid_sum = rsum;
        st->cm = cm;
        st->active_size = asz;
        st->status = status;
        st->n_updates = n_upd;
        st->nz = nz;
        if (NAIVE && p.nz_out) {
            p.nz_out[0] = nz;
            p.rsum_out[0] = rsum;
        }
#ifdef AHIP_GRP_PROFILE
        GP_MARK(5) /* epilogue */
        if (p.dbg) for (int k = 0; k < 8; ++k) atomicAdd(reinterpret_cast<unsigned long long*>(p.dbg) + k, (unsigned long long)tacc[k]);
        if (p.dbg) atomicAdd(reinterpret_cast<unsigned long long*>(p.dbg) + 7, 1ull);
#endif
        if (NAIVE && p.host_st && j == p.report_j) {
            CdBlkState<T> out;
            out.rsq = rsq; out.resid_sum = rsum; out.cm = cm; out.n_updates = n_upd;
            out.active_size = asz; out.status = status; out.nz = nz; out._pad = 0;
            *p.host_st = out;
            __threadfence_system();
            __hip_atomic_store(p.host_seq, p.report_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

#undef GP_MARK

} // namespace
} // namespace ahip
