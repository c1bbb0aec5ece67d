// blk_solve_body.hpp — the one-workgroup solve of a block of <= 128 lasso visits (see kernels_cd_block.hip), as a device
// function so that it can run both as its own kernel and as workgroup 0 of the fused look-ahead step (kernels_cd_panel.hip).
#pragma once
#include <type_traits>
#include "kernels.hpp"

namespace ahip {
namespace {

constexpr int BLK = 128; // visits per block

typedef double cb_d2 __attribute__((ext_vector_type(2)));
typedef float cb_f4 __attribute__((ext_vector_type(4)));
template <class T> struct CbVec;
template <> struct CbVec<double> { using type = cb_d2; static constexpr int N = 2; };
template <> struct CbVec<float> { using type = cb_f4; static constexpr int N = 4; };

// v_readlane with a wave-uniform lane index (SGPR): a register-to-scalar move, no LDS crossbar as __shfl would use
__device__ __forceinline__ double rdlane(double x, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}
__device__ __forceinline__ float rdlane(float x, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l));
}
__device__ __forceinline__ int rdlane(int x, int l) { return __builtin_amdgcn_readlane(x, l); }
// a * b + x[l] for a wave-uniform lane index l: the addend goes from v_readlane's scalar registers straight into the multiply-add
// (the compiler moves it back into vector registers first: three moves on the dependent chain of a coordinate visit)
__device__ __forceinline__ double fma_lane(double a, double b, double x, int l) {
    const double s = rdlane(x, l);
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(s));
    return r;
}
__device__ __forceinline__ float fma_lane(float a, float b, float x, int l) {
    const float s = rdlane(x, l);
    float r;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(s));
    return r;
}

template <class T>
__device__ __forceinline__ int blk_index(const CdBlkParams<T>& p, int pos) {
    return p.list ? p.list[pos] : pos;
}

// `tid` in [0, 256): the calling workgroup's first 256 threads; smem_raw holds blk_solve_lds<T>() bytes, 16-byte aligned.
// Sum of the slice partials of the block's columns (CdBlkParams::part) by the first 1024 / `nthreads` threads of the fused
// workgroup: 8 threads per column, each over every 8th slice, combined through LDS in a fixed order.  gsum[c] (LDS, BLK
// values) is complete after the next workgroup barrier.
template <class T>
__device__ __forceinline__ void blk_part_sum(const CdBlkParams<T>& p, int nb, T* gsum8 /* [8][BLK] in LDS */, int wtid) {
    const int c = wtid & (BLK - 1), k0 = wtid >> 7; // wtid in [0, 1024)
    T acc = T(0);
    if (p.part_ld == 0) {
        // slice-major partials part[k * BLK + c]: the 128 threads of a row read 1 KB contiguously; every thread issues its
        // loads in batches of eight before it sums them (fixed order k0, k0 + 8, ...)
        const T* pc = p.part + c;
        int k = k0;
        for (; k + 56 < p.part_n; k += 64) {
            T v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = pc[int64_t(k + 8 * u) * BLK];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; k < p.part_n; k += 8) acc += pc[int64_t(k) * BLK];
        if (c >= nb) acc = T(0);
    } else if (c < nb) {
        const T* pc = p.part + int64_t(c) * p.part_ld;
        int k = k0;
        for (; k + 24 < p.part_n; k += 32) {
            const T v0 = pc[k], v1 = pc[k + 8], v2 = pc[k + 16], v3 = pc[k + 24];
            acc += v0; acc += v1; acc += v2; acc += v3;
        }
        for (; k < p.part_n; k += 8) acc += pc[k];
    }
    gsum8[k0 * BLK + c] = acc;
}

// CONS: one-coefficient box constraints lo_i <= beta_i <= hi_i (CdBlkParams::clo / chi / cmu; ConstraintBox::solve_1d,
// constraint_box.ipp:51-96, and ConstraintOneSided, constraint_one_sided.ipp:12-49, whose solution is the box form with one
// side at infinity).  The minimiser of the one-dimensional problem over an interval is the unconstrained one clipped to it,
// so a visit only gains a min/max; the multiplier  mu = mu_+ - mu_-  of every visited coordinate is computed after the loop,
// lane-parallel, from the gradient its visit saw — the same quantities the reference's solve_1d leaves in the object.
template <class T, bool NAIVE, bool CONS = false>
__device__ __forceinline__ void blk_solve_body(const CdBlkParams<T>& p, int j, char* smem_raw, int tid,
                                               const T* corr = nullptr, int ncorr = 0, const T* gsum8 = nullptr) {
    T* D = reinterpret_cast<T*>(smem_raw);   // BLK*BLK
    T* gB = D + BLK * BLK;
    T* bB = gB + BLK;
    T* AB = bB + BLK;
    T* l1B = AB + BLK;
    T* denB = l1B + BLK;
    T* rdenB = denB + BLK;
    T* xmB = rdenB + BLK;
    T* dB = xmB + BLK;                        // net change of each coordinate of the block
    int32_t* idxB = reinterpret_cast<int32_t*>(dB + BLK);
    int32_t* actB = idxB + BLK;

    const int lane = tid & 63, wv = tid >> 6;
    const int base = j * p.bsz;
    const int nb = min(p.bsz, p.count - base);

    if (tid < BLK) {
        const int i = tid;
        if (i < nb) {
            const int idx = blk_index(p, base + i);
            const T A = p.vars[idx], pk = p.spen[idx];
            const T den = A + p.l2 * (p.spen2 ? p.spen2[idx] : pk); // (penalty_l2, ABI 8)
            idxB[i] = idx;
            gB[i] = NAIVE ? ((p.part && gsum8) ? T(0) : p.gblk[i]) : p.g[idx];
            bB[i] = p.beta[idx];
            AB[i] = A;
            l1B[i] = p.l1 * pk;
            denB[i] = den;
            rdenB[i] = T(1) / den;
            xmB[i] = p.xmean[idx];
            actB[i] = p.is_active[idx];
        } else {
            idxB[i] = 0; gB[i] = 0; bB[i] = 0; AB[i] = 0; l1B[i] = 0; denB[i] = 1; rdenB[i] = 1; xmB[i] = 0; actB[i] = 1;
        }
        dB[i] = 0;
    }
    // look-ahead correction, see CdBlkParams.  `corr` != nullptr: helper threads of the fused kernel compute it meanwhile
    // (blk_corr_helper below: ncorr partial sums per coordinate in LDS, complete at the barrier below)
    if (NAIVE && p.Cprev != nullptr && corr == nullptr && tid < nb) {
        const int nzp = p.pnz[0];
        const T* Cp = p.Cprev + tid;
        T acc = T(0);
        int m = 0;
        for (; m + 8 <= nzp; m += 8) {
            T c[8], d[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                c[u] = Cp[size_t(p.ppos[m + u]) * BLK];
                d[u] = p.pdlt[m + u];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = fma(c[u], d[u], acc);
        }
        for (; m < nzp; ++m) acc = fma(Cp[size_t(p.ppos[m]) * BLK], p.pdlt[m], acc);
        gB[tid] -= acc;
    }
    {
        using V = typename CbVec<T>::type;
        constexpr int VEC = CbVec<T>::N;
        const V* src = reinterpret_cast<const V*>(NAIVE ? p.Dptr : p.Dbuf + size_t(j & 1) * BLK * BLK);
        V* dst = reinterpret_cast<V*>(D);
        // 8 loads in flight per lane: the 128 KB block arrives in ~4 round trips instead of 32
        const int NE = nb * BLK / VEC; // columns [0, nb) of the slot
        for (int e0 = tid; e0 < NE; e0 += 256 * 8) {
            V v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[min(e0 + u * 256, NE - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (e0 + u * 256 < NE) dst[e0 + u * 256] = v[u];
        }
    }
    __syncthreads();
    if (wv != 0) return;
    // the chain below is the critical path of the whole pass: let this wave win the issue arbitration against whatever
    // else is resident on the CU (diagonal-block builds of the side stream)
    __builtin_amdgcn_s_setprio(3);

    // ---- one wavefront: the block's visits, strictly in order ------------------------------------------------------
    CdBlkState<T>* st = p.st;
    T rsq = st->rsq, rsum = st->resid_sum, cm = (j == 0) ? T(0) : st->cm;
    int asz = st->active_size, status = st->status;
    int64_t n_upd = st->n_updates;
    // Lane l keeps coordinates l and l+64 of the block in registers (g and the per-coordinate constants); a visit reads
    // its operands with v_readlane (uniform lane index), so the dependent chain of a visit is register-only:
    // readlane g_i -> soft-threshold -> fma into the two g registers.  The D column comes from LDS (2 x ds_read_b64,
    // address independent of the chain).  The single wave is bound by its instruction count, so everything that is not on
    // the chain (rsq, convergence measure, resid_sum, active-set marking, counters) is left out of the loop: a visit
    // only records the new coefficient and the gradient it saw in the owning lane, and the bookkeeping of the whole block
    // is done lane-parallel afterwards (same quantities; the sums are wave reductions instead of running sums).
    static_assert(BLK == 128, "two coordinates per lane");
    T g0 = gB[lane], g1 = gB[lane + 64];
    if (NAIVE && gsum8 != nullptr && p.part != nullptr) { // gradient from the slice partials (fixed order), intercept term
        T s0 = T(0), s1 = T(0);
        for (int q = 0; q < 8; ++q) {
            s0 += gsum8[q * BLK + lane];
            s1 += gsum8[q * BLK + lane + 64];
        }
        const T rs = p.part_rsum ? p.part_rsum[0] : T(0);
        g0 = (lane < nb) ? s0 - rs * xmB[lane] : T(0);
        g1 = (lane + 64 < nb) ? s1 - rs * xmB[lane + 64] : T(0);
    }
    if (NAIVE && corr != nullptr && p.Cprev != nullptr) {
        T c0 = T(0), c1 = T(0);
        for (int q = 0; q < ncorr; ++q) { // fixed order
            c0 += corr[q * BLK + lane];
            c1 += corr[q * BLK + lane + 64];
        }
        g0 -= c0;
        g1 -= c1;
    }
    const T b0 = bB[lane], b1 = bB[lane + 64];
    const T A0 = AB[lane], A1 = AB[lane + 64];
    const T L0 = l1B[lane], L1 = l1B[lane + 64];
    const T N0 = denB[lane], N1 = denB[lane + 64];
    const T R0 = rdenB[lane], R1 = rdenB[lane + 64];
    const T X0 = xmB[lane], X1 = xmB[lane + 64];
    const int a0 = actB[lane], a1 = actB[lane + 64];
    // CONS: bounds of this lane's two coordinates (+-inf where there is none; lanes beyond nb never visit)
    const T INF = T(1) / T(0);
    T lo0 = -INF, lo1 = -INF, hi0 = INF, hi1 = INF;
    if (CONS) {
        if (lane < nb) { lo0 = p.clo[idxB[lane]]; hi0 = p.chi[idxB[lane]]; }
        if (lane + 64 < nb) { lo1 = p.clo[idxB[lane + 64]]; hi1 = p.chi[idxB[lane + 64]]; }
    }
    T nb0 = b0, nb1 = b1; // new coefficients of this lane's two coordinates
    T gc0 = T(0), gc1 = T(0); // gradient seen by the visit of this lane's coordinates (only meaningful if they changed)
#define AHIP_BLK_VISIT(GREG, BREG, AREG, LREG, NREG, RREG, NBREG, GCREG, LOREG, HIREG, IL)                            \
    {                                                                                                                  \
        const T dc0 = D[i * BLK + lane], dc1 = D[i * BLK + lane + 64]; /* off the dependent chain: issued first */  \
        const T gcur = rdlane(GREG, IL);                                                                           \
        const T bi = rdlane(BREG, IL), A = rdlane(AREG, IL);                                                   \
        const T gk = fma(bi, A, gcur);                    /* pin_naive:85-89 */                                       \
        const T v = fabs(gk) - rdlane(LREG, IL);      /* pin_base:181-195 */                                      \
        T ak = T(0);                                                                                                   \
        if (v > T(0)) {                                                                                                \
            const T x = copysign(v, gk);                                                                               \
            const T den = rdlane(NREG, IL), rden = rdlane(RREG, IL);                                           \
            const T q0 = x * rden;                                                                                     \
            const T r = fma(-q0, den, x);                                                                              \
            ak = fma(r, rden, q0);                                                                                     \
        }                                                                                                              \
        if (CONS) {                                       /* constraint_box.ipp:75-80: clip to [lo, hi] */           \
            ak = fmax(fmin(ak, rdlane(HIREG, IL)), rdlane(LOREG, IL));                                         \
            if (lane == (IL)) GCREG = gcur;               /* the multiplier needs it for unchanged coordinates too */ \
        }                                                                                                              \
        if (ak != bi) {                                   /* pin_naive:97 */                                          \
            const T del = ak - bi;                                                                                     \
            g0 = fma(-del, dc0, g0);                                                                                   \
            g1 = fma(-del, dc1, g1);                                                                                   \
            if (lane == (IL)) { NBREG = ak; GCREG = gcur; }                                                            \
        }                                                                                                              \
    }
    {
        const int n0 = nb < 64 ? nb : 64;
        for (int i = 0; i < n0; ++i) AHIP_BLK_VISIT(g0, b0, A0, L0, N0, R0, nb0, gc0, lo0, hi0, i)
        for (int i = 64; i < nb; ++i) AHIP_BLK_VISIT(g1, b1, A1, L1, N1, R1, nb1, gc1, lo1, hi1, i - 64)
    }
#undef AHIP_BLK_VISIT
    if (CONS) {
        // multipliers (constraint_box.ipp:66-95 with A = Q(0,0) = 1): v = gradient seen + beta_old * A_kk (pin_naive:85-89)
        auto multiplier = [&](T gc, T bold, T A, T l1, T den, T x0, T lo, T hi) -> T {
            const T v = fma(bold, A, gc);
            const T mp0 = (hi > T(0)) ? T(0) : fmax(v, T(0));
            const T mn0 = (lo < T(0)) ? T(0) : fmax(-v, T(0));
            if (fabs(v - (mp0 - mn0)) <= l1) return mp0 - mn0;         // x = 0 is optimal
            const T full = v - (den * x0 + copysign(l1, x0));
            const T mp = (x0 < hi) ? T(0) : fmax(full, T(0));
            const T mn = (x0 > lo) ? T(0) : fmax(-full, T(0));
            return mp - mn;
        };
        if (lane < nb && (lo0 != -INF || hi0 != INF)) p.cmu[idxB[lane]] = multiplier(gc0, b0, A0, L0, N0, nb0, lo0, hi0);
        if (lane + 64 < nb && (lo1 != -INF || hi1 != INF))
            p.cmu[idxB[lane + 64]] = multiplier(gc1, b1, A1, L1, N1, nb1, lo1, hi1);
    }
    // ---- bookkeeping of the block, lane-parallel --------------------------------------------------------------------
    {
        const T d0 = nb0 - b0, d1 = nb1 - b1; // lanes beyond nb hold b = nb = 0
        const bool ch0 = d0 != T(0), ch1 = d1 != T(0);
        // pin_base:136-146 (rsq), pin_naive:107 (resid_sum), pin_base:112-122 (convergence measure)
        const T rs = (ch0 ? d0 * (T(2) * gc0 - d0 * A0) : T(0)) + (ch1 ? d1 * (T(2) * gc1 - d1 * A1) : T(0));
        const T xs = X0 * d0 + X1 * d1;
        const T c0 = A0 * d0 * d0, c1 = A1 * d1 * d1;
        T cmx = c0 > c1 ? c0 : c1;
        T rs_t = rs, xs_t = xs;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            rs_t += __shfl_xor(rs_t, off, 64);
            xs_t += __shfl_xor(xs_t, off, 64);
            const T o = __shfl_xor(cmx, off, 64);
            cmx = o > cmx ? o : cmx;
        }
        rsq += rs_t;
        rsum -= xs_t;
        cm = cmx > cm ? cmx : cm;
        const unsigned long long m0 = __ballot(ch0), m1 = __ballot(ch1);
        n_upd += __popcll(m0) + __popcll(m1);
        if (p.mark) { // add_active_set in visiting order, pin_naive:294-304
            const bool new0 = ch0 && a0 == 0, new1 = ch1 && a1 == 0;
            const unsigned long long q0 = __ballot(new0), q1 = __ballot(new1);
            const int cnt = __popcll(q0) + __popcll(q1);
            if (asz + cnt > p.max_active_size) {
                status = CD_MAX_ACTIVE;
            } else {
                const unsigned long long lt = (1ull << lane) - 1ull;
                if (new0) { const int pos = asz + __popcll(q0 & lt); p.is_active[idxB[lane]] = 1; p.active_set[pos] = idxB[lane]; }
                if (new1) {
                    const int pos = asz + __popcll(q0) + __popcll(q1 & lt);
                    p.is_active[idxB[lane + 64]] = 1;
                    p.active_set[pos] = idxB[lane + 64];
                }
                asz += cnt;
            }
        }
    }
    // net changes of the block (a coordinate is visited once per pass, so delta = new - old)
    bB[lane] = nb0; bB[lane + 64] = nb1;
    dB[lane] = nb0 - b0; dB[lane + 64] = nb1 - b1;
    if (NAIVE && p.dd != nullptr) { p.dd[lane] = nb0 - b0; p.dd[lane + 64] = nb1 - b1; }
    // ---- write back the block: beta, and the compacted non-zero changes for the update kernel ---------------------------
    int nz = 0;
    for (int i0 = 0; i0 < BLK; i0 += 64) {
        const int i = i0 + lane;
        const T d = (i < nb) ? dB[i] : T(0);
        const bool ch = d != T(0);
        if (ch) p.beta[idxB[i]] = bB[i];
        const unsigned long long m = __ballot(ch);
        const int pos = nz + __popcll(m & ((1ull << lane) - 1ull));
        if (ch) {
            if (NAIVE) {
                p.dcol[pos] = p.vcol[idxB[i]];
                if (p.dpos) p.dpos[pos] = i;
            } else {
                p.didx[pos] = idxB[i];
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

// ---------------------------------------------------------------------------------------------------------------------
// Solve of block j inside the fused look-ahead launch, for ALL 1024 threads of workgroup 0 (lasso, no constraints).
// Same visits and bookkeeping as blk_solve_body; what differs is the prologue.  While the other workgroups of the launch
// stream 190 MB, a dependent global round trip of this workgroup takes 3-5 us (it queues behind the step's loads), and
// blk_solve_body + blk_corr_helper make four to five of them in a row (indices -> constants, the diagonal block in four
// batches, change positions -> cross-block columns in three): the solve, not the step, set the length of the launch.  Here
// every load of the prologue is independent of the others and issued up front by all sixteen waves:
//   * the diagonal block, 128 KB, one batch of 16-byte loads per thread;
//   * the cross block C_{j,j-1}, all of it, 128 KB likewise - the previous solve left its changes in DENSE form (pdd, zeros
//     where nothing changed), so the correction is C * pdd with no index loads; each thread multiplies the pieces it holds by
//     the entries of pdd (from LDS) and the 1024 / (BLK / VEC) partial sums per row are combined in a fixed order;
//   * the slice partials of the block's gradient when the solver fused the reduction into this launch (p.part, slice-major),
//     in two batches (register budget of a 1024-thread workgroup: 128 VGPRs);
//   * the per-coordinate constants, straight into the registers of the visiting wave (two coordinates per lane).
// LDS: D | corr[NG][BLK] | gsum[8][BLK] | bB | dB | idxB  (blk_solve_lds_la).
template <class T> struct LaShape {
    static constexpr int VEC = CbVec<T>::N;
    static constexpr int CPC = BLK / VEC;            // 16-byte chunks per column
    static constexpr int ND = BLK * BLK / VEC / 1024; // chunks per thread and block (f64: 8, f32: 4)
    static constexpr int NG = 1024 / CPC;            // column groups = partial sums per row (f64: 16, f32: 32)
};
template <class T>
__host__ __device__ constexpr size_t blk_solve_lds_la() {
    return size_t(BLK) * BLK * sizeof(T) + size_t(LaShape<T>::NG) * BLK * sizeof(T) + size_t(8) * BLK * sizeof(T) +
           size_t(2) * BLK * sizeof(T) + size_t(BLK) * sizeof(int32_t) + 16;
}
template <class T>
__device__ __forceinline__ void blk_solve_la_body(const CdBlkParams<T>& p, int j, char* smem_raw, int tid) {
    using V = typename CbVec<T>::type;
    constexpr int VEC = LaShape<T>::VEC, CPC = LaShape<T>::CPC, ND = LaShape<T>::ND, NG = LaShape<T>::NG;
    T* D = reinterpret_cast<T*>(smem_raw);
    T* corr = D + BLK * BLK;       // [NG][BLK]
    T* gsum = corr + NG * BLK;     // [8][BLK]
    T* bB = gsum + 8 * BLK;
    T* dB = bB + BLK;
    int32_t* idxB = reinterpret_cast<int32_t*>(dB + BLK);
    const int lane = tid & 63, wv = tid >> 6;
    const int base = j * p.bsz;
    const int nb = min(p.bsz, p.count - base);
    const bool has_c = p.Cprev != nullptr;
    const bool has_part = p.part != nullptr;

    // ---- every load of the prologue ----------------------------------------------------------------------------------
    V dreg[ND], creg[ND];
    {
        const V* dsrc = reinterpret_cast<const V*>(p.Dptr);
#pragma unroll
        for (int u = 0; u < ND; ++u) // (columns [0, nb) of the slot; blocks of 64 visits fetch half of it)
            if ((tid + u * 1024) / CPC < nb) dreg[u] = dsrc[tid + u * 1024];
        if (has_c) {
            const V* csrc = reinterpret_cast<const V*>(p.Cprev);
#pragma unroll
            for (int u = 0; u < ND; ++u) creg[u] = csrc[tid + u * 1024];
        }
    }
    T pd = T(0); // previous block's dense changes (first BLK threads)
    if (has_c && tid < BLK) pd = p.pdd[tid];
    // visiting wave: indices, then the per-coordinate constants (two coordinates per lane)
    int i0 = 0, i1 = 0;
    T g0 = T(0), g1 = T(0), b0 = T(0), b1 = T(0), A0 = T(0), A1 = T(0), L0 = T(0), L1 = T(0), N0 = T(1), N1 = T(1), X0 = T(0), X1 = T(0);
    int a0 = 1, a1 = 1;
    T rsq = T(0), rsum = T(0), cm = T(0), prs = T(0);
    int asz = 0, status = 0;
    int64_t n_upd = 0;
    CdBlkState<T>* st = p.st;
    if (wv == 0) {
        const bool v0 = lane < nb, v1 = lane + 64 < nb;
        i0 = v0 ? blk_index(p, base + lane) : 0;
        i1 = v1 ? blk_index(p, base + lane + 64) : 0;
        if (!has_part) {
            g0 = v0 ? p.gblk[lane] : T(0);
            g1 = v1 ? p.gblk[lane + 64] : T(0);
        } else if (p.part_rsum) {
            prs = p.part_rsum[0];
        }
        rsq = st->rsq; rsum = st->resid_sum; cm = (j == 0) ? T(0) : st->cm;
        asz = st->active_size; status = st->status; n_upd = st->n_updates;
        if (v0) {
            const T pk = p.spen[i0];
            A0 = p.vars[i0]; b0 = p.beta[i0]; X0 = p.xmean[i0]; a0 = p.is_active[i0];
            L0 = p.l1 * pk; N0 = A0 + p.l2 * (p.spen2 ? p.spen2[i0] : pk);
        }
        if (v1) {
            const T pk = p.spen[i1];
            A1 = p.vars[i1]; b1 = p.beta[i1]; X1 = p.xmean[i1]; a1 = p.is_active[i1];
            L1 = p.l1 * pk; N1 = A1 + p.l2 * (p.spen2 ? p.spen2[i1] : pk);
        }
    }

    // ---- diagonal block and the previous changes into LDS ---------------------------------------------------------------
    {
        V* dst = reinterpret_cast<V*>(D);
#pragma unroll
        for (int u = 0; u < ND; ++u)
            if ((tid + u * 1024) / CPC < nb) dst[tid + u * 1024] = dreg[u];
    }
    if (tid < BLK) dB[tid] = pd; // (dB doubles as the staging of pdd until the loop is over)
    __syncthreads();
    // ---- correction C * pdd: this thread's rows rowc * VEC .. + VEC, columns cg + NG * u --------------------------------
    if (has_c) {
        const int rowc = tid % CPC, cg = tid / CPC;
        T acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = T(0);
#pragma unroll
        for (int u = 0; u < ND; ++u) {
            const T dl = dB[cg + NG * u];
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] = fma(creg[u][e], dl, acc[e]);
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) corr[cg * BLK + rowc * VEC + e] = acc[e];
    }
    // ---- slice partials of the block's gradient (solver fused the reduction into this launch): a second round trip, with
    //      the registers of the two blocks free again; k0, k0 + 8, ... in a fixed order per thread, eight threads per column
    // Blocks of 64 visits (IRLS) give every column sixteen threads instead of eight; beyond PBN partials per thread (long
    // designs: 489 slices of 1024 rows at 500k rows) the rest follows in batches of eight loads in flight.
    const int pc_sh = (p.bsz <= 64) ? 6 : 7; // log2 of the columns a row of gsum holds
    if (has_part) {
        constexpr int PBN = 25;
        const int tpc = 1024 >> pc_sh;
        const int pc_c = tid & ((1 << pc_sh) - 1), pc_k0 = tid >> pc_sh;
        const T* pc = p.part + pc_c;
        T pw[PBN];
#pragma unroll
        for (int u = 0; u < PBN; ++u) pw[u] = pc[int64_t(min(pc_k0 + tpc * u, p.part_n - 1)) * BLK];
        T psum = T(0);
#pragma unroll
        for (int u = 0; u < PBN; ++u) psum += (pc_k0 + tpc * u < p.part_n) ? pw[u] : T(0);
        for (int k = pc_k0 + tpc * PBN; k < p.part_n; k += tpc * 8) {
            T qw[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) qw[u] = pc[int64_t(min(k + tpc * u, p.part_n - 1)) * BLK];
#pragma unroll
            for (int u = 0; u < 8; ++u) psum += (k + tpc * u < p.part_n) ? qw[u] : T(0);
        }
        gsum[(pc_k0 << pc_sh) + pc_c] = (pc_c < nb) ? psum : T(0);
    }
    __syncthreads();
    if (wv != 0) return;
    __builtin_amdgcn_s_setprio(3);

    if (has_part) {
        T s0 = T(0), s1 = T(0);
        if (pc_sh == 7) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                s0 += gsum[q * BLK + lane];
                s1 += gsum[q * BLK + lane + 64];
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) s0 += gsum[q * 64 + lane];
        }
        g0 = (lane < nb) ? s0 - prs * X0 : T(0);
        g1 = (lane + 64 < nb) ? s1 - prs * X1 : T(0);
    }
    if (has_c) {
        T c0 = T(0), c1 = T(0);
#pragma unroll
        for (int q = 0; q < NG; ++q) { // fixed order
            c0 += corr[q * BLK + lane];
            c1 += corr[q * BLK + lane + 64];
        }
        g0 -= c0;
        g1 -= c1;
    }
    const T R0 = T(1) / N0, R1 = T(1) / N1;
    T nb0 = b0, nb1 = b1;
    T gc0 = T(0), gc1 = T(0);
    // The per-coordinate constants of a visit (variance, threshold, denominator and its reciprocal, old coefficient) go through
    // LDS — six values per coordinate in the space of the consumed correction partials, read by all lanes at one address
    // (broadcast) ONE VISIT AHEAD — instead of ten v_readlane out of the lanes' registers on the dependent chain of the visit;
    // only the current gradient, which the visits before it changed, still comes out of a register.  Same arithmetic, bit for
    // bit (config 4: cd 3488 -> 3444 ms; the 128-visit launches of the Gaussian paths are bound elsewhere: unchanged).
    T* cst = corr; // [BLK][6]
    static_assert(6 * (BLK + 1) <= NG * BLK, "the constants (and the row the last visit prefetches) fit where the correction partials were");
    {
        T* c0p = cst + lane * 6;
        c0p[0] = A0; c0p[1] = L0; c0p[2] = N0; c0p[3] = R0; c0p[4] = b0;
        T* c1p = cst + (lane + 64) * 6;
        c1p[0] = A1; c1p[1] = L1; c1p[2] = N1; c1p[3] = R1; c1p[4] = b1;
    }
    __builtin_amdgcn_wave_barrier();
    // ---- the visits -------------------------------------------------------------------------------------------------------
    // One wavefront, one coordinate after the other: the wave is bound by the instructions it issues (about 5 cycles each,
    // dependent or not), so the loop is written for their number -- 30 per visit where the first form had 45:
    //   * branch-free (round 6): the quotient is formed whatever the sign of the threshold test and selected away, an unchanged
    //     coordinate updates the gradients with del = 0 -- the same values bit for bit without two exec-mask round trips;
    //   * the visit's gradient goes from v_readlane's scalar registers straight into the first multiply-add (fma_lane);
    //   * the constants of visit i + 1 are fetched during visit i into a second register set (two visits per trip, the sets
    //     swapping roles: no moves), through one vector-register offset (immediate offsets in the LDS reads);
    //   * only the gradient a visit saw is kept (lane i of the gradient register before the update; blocks of <= 64 visits
    //     write the whole register to LDS, one instruction, and pick the diagonal behind the loop): the new coefficients are
    //     formed again from it behind the loop, lane-parallel, by the visit's own operations;
    //   * a visit updates only the gradients still to be read: blocks of <= 64 visits (IRLS) carry no second half, the second
    //     half of a longer block no first.
    {
        struct Cst { T A, thr, den, rden, bi; };
        // byte offset (from the start of the LDS block) of the next row of constants to fetch, in a vector register
        int coff = int(reinterpret_cast<const char*>(cst) - smem_raw);
        asm volatile("" : "+v"(coff));
        const char* cbase = smem_raw;
        auto fetch = [&]() {
            const T* c = reinterpret_cast<const T*>(cbase + coff);
            Cst k{c[0], c[1], c[2], c[3], c[4]};
            coff += int(6 * sizeof(T));
            return k;
        };
        const T* dp = D + lane; // column i of the diagonal block, this lane's row(s)
        T* rec = D + 64 * BLK + lane; // blocks of <= 64 visits: columns 64.. of the diagonal block's space are free
        auto visit = [&](auto mode_c, T& G, T& GC, int il, const Cst k) { // mode: which halves of the gradient still matter
            constexpr int MODE = decltype(mode_c)::value;                   //   1 = lanes' first, 2 = second, 3 = both
            T dc0 = T(0), dc1 = T(0);
            if constexpr (MODE & 1) dc0 = dp[0];
            if constexpr (MODE & 2) dc1 = dp[64];
            dp += BLK;
            const T gk = fma_lane(k.bi, k.A, G, il);      // pin_naive:85-89
            // pin_base:181-195; a coordinate under its threshold goes through the quotient as x = +-0 and comes out as +0
            // (-q0 * den + x = -+0 + +-0 = +0, then +0 * rden + +-0 = +0): one v_max instead of a compare and two selects
            const T v = fmax(fabs(gk) - k.thr, T(0));
            const T x = copysign(v, gk);
            const T q0 = x * k.rden;
            const T r = fma(-q0, k.den, x);
            const T ak = fma(r, k.rden, q0);
            const T del = ak - k.bi;                      // 0 for an unchanged coordinate, pin_naive:97
            if constexpr (MODE == 1) { // the gradients as visit il saw them, all lanes: lane il's entry is read behind the loop
                rec[0] = G;
                rec += 64;
            } else {
                GC = (lane == il) ? G : GC;
            }
            if constexpr (MODE & 1) g0 = fma(-del, dc0, g0);
            if constexpr (MODE & 2) g1 = fma(-del, dc1, g1);
        };
        auto half = [&](auto two_c, T& G, T& GC, int cnt) { // `cnt` visits of the coordinates in the lanes of G
            Cst a = fetch(), b;
            int il = 0;
            for (; il + 1 < cnt; il += 2) {
                b = fetch();
                visit(two_c, G, GC, il, a);
                a = fetch();                              // (the last trip fetches a row nobody uses: row <= BLK exists)
                visit(two_c, G, GC, il + 1, b);
            }
            if (il < cnt) visit(two_c, G, GC, il, a);
        };
        if (nb <= 64) {
            half(std::integral_constant<int, 1>{}, g0, gc0, nb);
            gc0 = (lane < nb) ? D[64 * BLK + lane * 64 + lane] : T(0);
        } else {
            half(std::integral_constant<int, 3>{}, g0, gc0, 64);
            coff = int(reinterpret_cast<const char*>(cst + 64 * 6) - smem_raw);
            asm volatile("" : "+v"(coff));
            half(std::integral_constant<int, 2>{}, g1, gc1, nb - 64);
        }
        // the coefficients the visits left, from the gradients they saw
        auto visit_result = [](T bi, T A, T thr, T den, T rden, T gcur) {
            const T gk = fma(bi, A, gcur);
            const T v = fabs(gk) - thr;
            const T x = copysign(v, gk);
            const T q0 = x * rden;
            const T r = fma(-q0, den, x);
            const T akq = fma(r, rden, q0);
            return v > T(0) ? akq : T(0);
        };
        if (lane < nb) nb0 = visit_result(b0, A0, L0, N0, R0, gc0);
        if (lane + 64 < nb) nb1 = visit_result(b1, A1, L1, N1, R1, gc1);
    }
    // ---- bookkeeping of the block, lane-parallel (as blk_solve_body) ----------------------------------------------------
    const T d0 = nb0 - b0, d1 = nb1 - b1; // lanes beyond nb hold b = nb = 0
    const bool ch0 = d0 != T(0), ch1 = d1 != T(0);
    {
        const T rs = (ch0 ? d0 * (T(2) * gc0 - d0 * A0) : T(0)) + (ch1 ? d1 * (T(2) * gc1 - d1 * A1) : T(0));
        const T xs = X0 * d0 + X1 * d1;
        const T c0 = A0 * d0 * d0, c1 = A1 * d1 * d1;
        T cmx = c0 > c1 ? c0 : c1;
        T rs_t = rs, xs_t = xs;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            rs_t += __shfl_xor(rs_t, off, 64);
            xs_t += __shfl_xor(xs_t, off, 64);
            const T o = __shfl_xor(cmx, off, 64);
            cmx = o > cmx ? o : cmx;
        }
        rsq += rs_t;
        rsum -= xs_t;
        cm = cmx > cm ? cmx : cm;
        const unsigned long long m0 = __ballot(ch0), m1 = __ballot(ch1);
        n_upd += __popcll(m0) + __popcll(m1);
        if (p.mark) { // add_active_set in visiting order, pin_naive:294-304
            const bool new0 = ch0 && a0 == 0, new1 = ch1 && a1 == 0;
            const unsigned long long q0 = __ballot(new0), q1 = __ballot(new1);
            const int cnt = __popcll(q0) + __popcll(q1);
            if (asz + cnt > p.max_active_size) {
                status = CD_MAX_ACTIVE;
            } else {
                const unsigned long long lt = (1ull << lane) - 1ull;
                if (new0) { const int pos = asz + __popcll(q0 & lt); p.is_active[i0] = 1; p.active_set[pos] = i0; }
                if (new1) {
                    const int pos = asz + __popcll(q0) + __popcll(q1 & lt);
                    p.is_active[i1] = 1;
                    p.active_set[pos] = i1;
                }
                asz += cnt;
            }
        }
    }
    // ---- write back: beta, the dense and the compacted changes ---------------------------------------------------------
    if (ch0) p.beta[i0] = nb0;
    if (ch1) p.beta[i1] = nb1;
    if (p.dd != nullptr) { p.dd[lane] = d0; p.dd[lane + 64] = d1; }
    int nz = 0;
    {
        const unsigned long long m0 = __ballot(ch0), m1 = __ballot(ch1);
        const unsigned long long lt = (1ull << lane) - 1ull;
        if (ch0) {
            const int pos = __popcll(m0 & lt);
            p.dcol[pos] = p.vcol[i0];
            if (p.dpos) p.dpos[pos] = lane;
            p.dlt[pos] = d0;
        }
        const int n0c = __popcll(m0);
        if (ch1) {
            const int pos = n0c + __popcll(m1 & lt);
            p.dcol[pos] = p.vcol[i1];
            if (p.dpos) p.dpos[pos] = lane + 64;
            p.dlt[pos] = d1;
        }
        nz = n0c + __popcll(m1);
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
        if (p.host_st && j == p.report_j) {
            CdBlkState<T> out;
            out.rsq = rsq; out.resid_sum = rsum; out.cm = cm; out.n_updates = n_upd;
            out.active_size = asz; out.status = status; out.nz = nz; out._pad = 0;
            *p.host_st = out;
            __threadfence_system();
            __hip_atomic_store(p.host_seq, p.report_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Helper threads of the fused kernel (htid in [0, BLK * NCORR)): partial sums of  Cprev[:, ppos[m]] * pdlt[m]  over the
// m-range of part htid / BLK for coordinate htid % BLK, into corr[part * BLK + coordinate].  Ends with the workgroup barrier
// that pairs with the one inside blk_solve_body.
constexpr int NCORR = 6;
template <class T>
__device__ __forceinline__ void blk_corr_helper(const CdBlkParams<T>& p, T* corr, int htid) {
    const int row = htid & (BLK - 1), part = htid / BLK;
    T acc = T(0);
    if (p.Cprev != nullptr) {
        const int nzp = p.pnz[0];
        const int per = (nzp + NCORR - 1) / NCORR;
        const int m0 = part * per, m1 = min(nzp, m0 + per);
        const T* Cp = p.Cprev + row;
        int m = m0;
        for (; m + 8 <= m1; m += 8) {
            T c[8], d[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                c[u] = Cp[size_t(p.ppos[m + u]) * BLK];
                d[u] = p.pdlt[m + u];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = fma(c[u], d[u], acc);
        }
        for (; m < m1; ++m) acc = fma(Cp[size_t(p.ppos[m]) * BLK], p.pdlt[m], acc);
    }
    corr[part * BLK + row] = acc;
    __syncthreads();
}

// LDS of the solve (+ the helper partials of the fused kernel)
template <class T>
__host__ __device__ constexpr size_t blk_solve_lds_fused() {
    return size_t(BLK) * BLK * sizeof(T) + size_t(BLK) * 8 * sizeof(T) + size_t(BLK) * 2 * sizeof(int32_t) + 16 +
           size_t(NCORR) * BLK * sizeof(T) + size_t(8) * BLK * sizeof(T);
}

template <class T>
__host__ __device__ constexpr size_t blk_solve_lds() {
    return size_t(BLK) * BLK * sizeof(T) + size_t(BLK) * 8 * sizeof(T) + size_t(BLK) * 2 * sizeof(int32_t) + 16;
}

} // namespace
} // namespace ahip
