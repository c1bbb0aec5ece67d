// kernels_cd_panel.hip — the "panel" form of a lasso coordinate-descent pass: residual based, exactly the naive
// method's data flow (reference solver_gaussian_pin_naive.hpp:16-168: gradient of a coordinate from the residual,
// residual updated with the coordinate's column), but organised per block of B = 128 consecutive visits so the whole
// chip streams the block's columns while one wavefront does the (inherently sequential) updates:
//
//   panel_step_kernel    (n / rows-per-slice workgroups, row sliced)
//                        (A) applies the previous block's changes to its slice of the residual,
//                            r -= sum_m del_m X[:, col_m]            (ctmul of pin_naive:104-108, 128 columns at once)
//                        (B) partial gradients of the next block on the same slice,
//                            part[c][slice] = X[slice, col_c] . (w*r)[slice]   (cmul of pin_naive:84-86)
//                        r is read and written once per block; the block's columns are read once for (B) and, if they
//                        changed, once more for (A) of the following step (second read served by the 256 MB MALL).
//   panel_reduce_kernel  sums the slice partials in a fixed order and applies the intercept term  - resid_sum * xbar_c.
//   blk_solve_kernel     (kernels_cd_block.hip, NAIVE variant, ONE workgroup) runs the block's visits in order against
//                        the B x B block  D = X_B^T W X_B - xbar xbar^T  (computed once per block and weight vector by
//                        the MFMA Gram kernel and cached), which carries the within-block coupling exactly.
//
// Compared with keeping the full |S| x |S| Gram matrix current, the MFMA work drops from n|S|^2/2 to 128 n |S| MACs and
// nothing has to be recomputed per IRLS iteration except the blocks that are actually visited.  The iterates are the
// Gauss-Seidel sequence of the reference in exact arithmetic.
#include <cstdlib>
#include "kernels.hpp"
#include "accessors.hpp"
#include "wavered.hpp"
#include "blk_solve_body.hpp"
#include "grp_solve_body.hpp"

namespace ahip {

namespace {

constexpr int PB = 128;
constexpr int PT = 256;

// Raw (undecoded) row-slice loads: VEC consecutive rows of one column per lane.  Dense: the values themselves, with
// temporal (cache-allocating) loads because the block's columns are read again by the next step.  SNP: the byte
// holding the four 2-bit calls, decoded at use (keeps 16 loads in flight within the register budget).
template <class T, int VEC>
struct RawDense { Pack<T, VEC> v; };
struct RawSnp { unsigned byte; };

// `full` == false (only in the ragged last slice): lanes whose rows lie beyond n read from row 0 instead (a valid
// address) and every out-of-range element is zeroed by a select — no branches, so the loads of a batch stay in flight
// together (a branchy tail path made the one ragged workgroup the slowest of the launch by ~10 us).
// Load policy of the two phases (dense designs).  Phase (B) is the FIRST read of a block's columns, phase (A) of the step two
// launches later the SECOND and last one: 1 = non-temporal (streaming) load, 0 = plain (allocating) load.
#ifndef AHIP_PANEL_NT_A
#define AHIP_PANEL_NT_A 1
#endif
#ifndef AHIP_PANEL_NT_B
#define AHIP_PANEL_NT_B 1
#endif
template <class T, int VEC, bool NTL = true>
__device__ __forceinline__ RawDense<T, VEC> praw(const DenseAcc<T>& X, int64_t j, int64_t i, int64_t n, bool full) {
    RawDense<T, VEC> r;
    const T* col = X.colptr(j);
    if constexpr (VEC == 1) {
        const T x = col[(full || i < n) ? i : 0];
        r.v.v[0] = (full || i < n) ? x : T(0);
    } else {
        using V = typename VecOf<T>::type;
        static_assert(VEC == VecOf<T>::N, "dense vector width");
        // vector path requires ld % VEC == 0, so a lane starting below n may read up to VEC-1 pad elements: in bounds
        const int64_t ii = (full || i < n) ? i : 0;
        V x;
        if constexpr (NTL) x = __builtin_nontemporal_load(reinterpret_cast<const V*>(col + ii));
        else x = *reinterpret_cast<const V*>(col + ii);
#pragma unroll
        for (int e = 0; e < VEC; ++e) r.v.v[e] = (full || i + e < n) ? x[e] : T(0);
    }
    return r;
}
template <class T, int VEC, bool NTL = true>
__device__ __forceinline__ RawSnp praw(const SnpAcc<T>& X, int64_t j, int64_t i, int64_t n, bool full) {
    static_assert(VEC == 4 || VEC == 16, "one byte (4 calls) or one 32-bit word (16 calls) per lane");
    RawSnp r;
    if constexpr (VEC == 4) {
        const unsigned b = unsigned(X.colptr(j)[((full || i < n) ? i : 0) >> 2]);
        r.byte = (full || i < n) ? b : 0u;
    } else {
        // columns are 64-byte aligned and padded (SnpView::ldb), i is a multiple of 16: an aligned word inside the column
        const unsigned b = reinterpret_cast<const unsigned*>(X.colptr(j))[((full || i < n) ? i : 0) >> 4];
        r.byte = (full || i < n) ? b : 0u;
    }
    return r;
}
template <class T, int VEC>
__device__ __forceinline__ Pack<T, VEC> pdecode(const DenseAcc<T>&, const RawDense<T, VEC>& r, int64_t, int64_t, int64_t) {
    return r.v;
}
template <class T, int VEC>
__device__ __forceinline__ Pack<T, VEC> pdecode(const SnpAcc<T>& X, const RawSnp& r, int64_t j, int64_t i, int64_t n) {
    Pack<T, VEC> o;
    const T imp = X.impute[j];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        const unsigned c = (r.byte >> (2 * e)) & 3u;
        o.v[e] = (i + e < n) ? (c == 3u ? imp : T(c)) : T(0);
    }
    return o;
}
template <class T, class Acc, int VEC> struct RawOf;
template <class T, int VEC> struct RawOf<T, DenseAcc<T>, VEC> { using type = RawDense<T, VEC>; };
template <class T, int VEC> struct RawOf<T, SnpAcc<T>, VEC> { using type = RawSnp; };

// Latency structure (the kernel is bound by dependent HBM round trips otherwise): column indices and coefficients are
// wave-uniform scalar loads; each wave keeps 16 (A) + 8 (B) column-slice loads in flight: the first batch of phase (B)
// is issued before phase (A) so that its round trip overlaps (A)'s.  The register budget is held at 128 VGPRs so that
// 4 workgroups fit per CU: 1024 resident slots cover the 782 row slices of n = 100k in ONE round (at 3 per CU a
// handful of stragglers cost a whole second round).
// WGRED: the slice's partial gradients go to `red[0][c]` (LDS, free after phase (A)) instead of `part`: the fused kernel sums
// the four slices of its workgroup and writes ONE partial per column and workgroup.
template <class T, class Acc, int VEC, bool FULL, bool WGRED = false>
__device__ __forceinline__ void panel_step_body(const Acc& X, int64_t n, const T* __restrict__ w, T* __restrict__ r,
                                                const int32_t* __restrict__ dcol, const T* __restrict__ dlt, int nz,
                                                const int32_t* __restrict__ cols, int nb, T* __restrict__ part,
                                                int64_t part_ld, T (*red)[64 * VEC], T* wrs, int tid, int64_t slice) {
    constexpr int U = 16;
    using Raw = typename RawOf<T, Acc, VEC>::type;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t i = slice * (64 * VEC) + int64_t(lane) * VEC;
    const bool full = FULL ? true : (i + VEC <= n);

    // first batch of (B): columns wv, wv+4, ... of the next block
    constexpr int UB = 8;
    Raw xb[UB];
    int jb[UB];
    if (nb > 0) {
#pragma unroll
        for (int u = 0; u < UB; ++u) jb[u] = cols[min(wv + 4 * u, nb - 1)];
#pragma unroll
        for (int u = 0; u < UB; ++u) xb[u] = praw<T, VEC, AHIP_PANEL_NT_B != 0>(X, jb[u], i, n, full);
    }

    // ---- (A) residual slice -= X[slice, changed columns] * del ---------------------------------------------------------
    if (nz > 0) {
        T acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = T(0);
        for (int m0 = wv; m0 < nz; m0 += 4 * U) {
            Raw xa[U];
            int ja[U];
            T cf[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int m = m0 + 4 * u;
                ja[u] = dcol[min(m, nz - 1)];
                cf[u] = m < nz ? dlt[min(m, nz - 1)] : T(0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) xa[u] = praw<T, VEC, AHIP_PANEL_NT_A != 0>(X, ja[u], i, n, full);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const Pack<T, VEC> xx = pdecode<T, VEC>(X, xa[u], ja[u], i, n);
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] = fma(cf[u], xx.v[e], acc[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) red[wv][lane * VEC + e] = acc[e];
        __syncthreads();
        if (wv == 0) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const int q = lane * VEC + e;
                T wr = T(0);
                if (FULL || i + e < n) {
                    const T rr = r[i + e] - ((red[0][q] + red[1][q]) + (red[2][q] + red[3][q]));
                    r[i + e] = rr;
                    wr = w[i + e] * rr;
                }
                wrs[q] = wr;
            }
        }
    } else if (wv == 0) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) wrs[lane * VEC + e] = (FULL || i + e < n) ? w[i + e] * r[i + e] : T(0);
    }
    if (nb <= 0) return;
    __syncthreads();

    // ---- (B) partial gradients of the next block on this slice ----------------------------------------------------------
    T wr[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) wr[e] = wrs[lane * VEC + e];
    for (int c0 = wv; c0 < nb; c0 += 4 * UB) {
        if (c0 != wv) {
#pragma unroll
            for (int u = 0; u < UB; ++u) jb[u] = cols[min(c0 + 4 * u, nb - 1)];
#pragma unroll
            for (int u = 0; u < UB; ++u) xb[u] = praw<T, VEC, AHIP_PANEL_NT_B != 0>(X, jb[u], i, n, full);
        }
        T pu[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const Pack<T, VEC> xx = pdecode<T, VEC>(X, xb[u], jb[u], i, n);
            T sacc = T(0);
#pragma unroll
            for (int e = 0; e < VEC; ++e) sacc = fma(xx.v[e], wr[e], sacc);
            pu[u] = sacc;
        }
        static_assert(UB == 8, "reduce8");
        const T tot = reduce8(pu, lane);
        if (lane < UB && c0 + 4 * lane < nb) {
            if constexpr (WGRED) red[0][c0 + 4 * lane] = tot; // (64 * VEC >= 128 columns)
            else part[part_ld > 0 ? int64_t(c0 + 4 * lane) * part_ld + slice : slice * PB + (c0 + 4 * lane)] = tot;
        }
    }
}

template <class T, class Acc, int VEC>
__global__ __launch_bounds__(PT, (VEC >= 16 ? 2 : 4)) void panel_step_kernel(Acc X, int64_t n, const T* __restrict__ w, T* __restrict__ r,
                                                        const int32_t* __restrict__ dcol, const T* __restrict__ dlt,
                                                        const int32_t* __restrict__ nz_dev,
                                                        const int32_t* __restrict__ cols, int nb, T* __restrict__ part,
                                                        int64_t part_ld) {
    constexpr int RS = 64 * VEC;
    __shared__ T red[4][RS];
    __shared__ T wrs[RS];
    const int nz = nz_dev[0];
    // only the last slice can be ragged; every other workgroup runs the branch-free body
    if ((int64_t(blockIdx.x) + 1) * RS <= n)
        panel_step_body<T, Acc, VEC, true>(X, n, w, r, dcol, dlt, nz, cols, nb, part, part_ld, red, wrs, threadIdx.x, blockIdx.x);
    else
        panel_step_body<T, Acc, VEC, false>(X, n, w, r, dcol, dlt, nz, cols, nb, part, part_ld, red, wrs, threadIdx.x, blockIdx.x);
}

// ---- the step on a 2-bit design, one 32-bit word (16 calls) per lane and column --------------------------------------------
// The generic body above makes three dependent round trips (columns of (A) + first half of (B) -> residual / weight rows by wave
// 0 alone -> second half of (B)) and funnels the residual update through one wavefront, whose LDS accesses at a stride of 16
// values hit one bank.  With one register per column and lane everything a workgroup needs can be in flight at once: here every
// load of the step is issued up front — 16 words of (A), 16 words of (B), and a quarter of the slice's residual / weight rows
// per wavefront —, the four waves of a slice update a quarter of its rows each, and the LDS rows are laid out element-major
// (index e * 64 + lane: conflict-free).  A workgroup carries TWO slices of 1024 rows (512 threads): 245 workgroups at 500k rows,
// one per compute unit, and half as many partials for whoever sums them.  Same sums in the same order as panel_step_body for
// the residual; the partial gradients are summed over the lanes by a 16-value butterfly.
constexpr int S16_RS = 1024; // rows of a slice
constexpr int S16_NSUB = 2;  // slices per workgroup

// Sums sixteen per-lane values over the 64 lanes at once: lane l ends with the total of value (l & 15); fixed order.
template <class T>
__device__ __forceinline__ T reduce16(const T (&v)[16], int lane) {
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
    T a[8], c[4], d[2];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const T keep = b0 ? v[2 * m + 1] : v[2 * m], send = b0 ? v[2 * m] : v[2 * m + 1];
        a[m] = keep + pdpp<0xB1>(send); // xor 1
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const T keep = b1 ? a[2 * m + 1] : a[2 * m], send = b1 ? a[2 * m] : a[2 * m + 1];
        c[m] = keep + pdpp<0x4E>(send); // xor 2
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const T keep = b2 ? c[2 * m + 1] : c[2 * m], send = b2 ? c[2 * m] : c[2 * m + 1];
        d[m] = keep + __shfl_xor(send, 4, 64);
    }
    const T keep = b3 ? d[1] : d[0], send = b3 ? d[0] : d[1];
    T t = keep + pdpp<0x128>(send); // row_ror:8 == xor 8 inside a row of 16
    t += __shfl_xor(t, 16, 64);
    t += __shfl_xor(t, 32, 64);
    return t;
}

template <class T, bool FULL>
__device__ __forceinline__ void snp16_decode(unsigned word, T imp, int64_t i, int64_t n, T (&o)[16]) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const unsigned c = (word >> (2 * e)) & 3u;
        const T x = c == 3u ? imp : T(c);
        o[e] = (FULL || i + e < n) ? x : T(0);
    }
}

template <class T, bool FULL, bool MEANS>
__device__ __forceinline__ void panel_step_snp16_body(const SnpAcc<T>& X, int64_t n, const T* __restrict__ w, T* __restrict__ r,
                                                      const int32_t* __restrict__ dcol, const T* __restrict__ dlt, int nz,
                                                      const int32_t* __restrict__ cols, int nb, T* __restrict__ part,
                                                      int64_t part_ld, T (*red)[S16_RS], T* wrs, T* wsl, T* psum, T* psum2,
                                                      const T* ptab, int t, int64_t slice) {
    constexpr int U = 16;
    const int lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int64_t i = slice * S16_RS + int64_t(lane) * 16;
    const bool in = FULL || i < n;
    const int64_t wofs = in ? (i >> 4) : 0; // (columns are 64-byte aligned and padded: an aligned word inside the column)
    auto word = [&](int j) -> unsigned {
        const unsigned x = reinterpret_cast<const unsigned*>(X.colptr(j))[wofs];
        return in ? x : 0u;
    };
    // ---- every load of the step ------------------------------------------------------------------------------------------
    unsigned xa[U], xb[U];
    int ja[U], jb[U];
    if (nz > 0) {
#pragma unroll
        for (int u = 0; u < U; ++u) ja[u] = dcol[min(wv + 4 * u, nz - 1)];
#pragma unroll
        for (int u = 0; u < U; ++u) xa[u] = word(ja[u]);
    }
    // calls of this lane's word that lie below n (the ragged last slice; the table lookups have no other guard)
    unsigned vmask = 0xFFFFFFFFu;
    if (!FULL) {
        const int64_t left = n - i;
        vmask = left >= 16 ? 0xFFFFFFFFu : (left <= 0 ? 0u : ((1u << (2 * int(left))) - 1u));
    }
    if (nb > 0) {
#pragma unroll
        for (int u = 0; u < U; ++u) jb[u] = cols[min(wv + 4 * u, nb - 1)];
#pragma unroll
        for (int u = 0; u < U; ++u) xb[u] = word(jb[u]);
    }
    // this wave's quarter of the slice's rows: elements 4 wv .. 4 wv + 3 of every lane
    const int64_t q0 = i + 4 * wv;
    T rq[4], wq[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool ok = FULL || q0 + k < n;
        rq[k] = ok ? r[q0 + k] : T(0);
        wq[k] = ok ? w[q0 + k] : T(0);
    }
    // ---- (A) ----------------------------------------------------------------------------------------------------------------
    if (nz > 0) {
        __syncthreads(); // the pair tables are complete (uniform: nz is the same for the whole workgroup)
        T acc[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = T(0);
        // Two columns at a time through a 16-entry table per column PAIR (ptab, built by the whole workgroup from the pair's
        // coefficients and impute values: entry c1 + 4 c2 = cf_a x_a(c1) + cf_b x_b(c2)): the calls of the two columns are
        // interleaved into nibbles once per word pair, and an element costs two integer instructions, one LDS read (a table is
        // one row of banks: conflict-free for any mix of indices) and one addition, where decoding both calls and two
        // multiply-adds cost twelve.
        for (int m0 = wv, bt = 0; m0 < nz; m0 += 4 * U, ++bt) {
            if (m0 != wv) {
#pragma unroll
                for (int u = 0; u < U; ++u) ja[u] = dcol[min(m0 + 4 * u, nz - 1)];
#pragma unroll
                for (int u = 0; u < U; ++u) xa[u] = word(ja[u]);
            }
#pragma unroll
            for (int k = 0; k < U / 2; ++k) {
                const unsigned w1 = xa[2 * k] & vmask, w2 = xa[2 * k + 1] & vmask;
                const unsigned wi = (w1 & 0x33333333u) | ((w2 & 0x33333333u) << 2);  // nibble j: calls 2j of both columns
                const unsigned wo = ((w1 >> 2) & 0x33333333u) | (w2 & 0xCCCCCCCCu);  // nibble j: calls 2j + 1
                const T* tb = ptab + (bt * 32 + wv * 8 + k) * 16;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    acc[2 * j] += tb[(wi >> (4 * j)) & 15u];
                    acc[2 * j + 1] += tb[(wo >> (4 * j)) & 15u];
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) red[wv][e * 64 + lane] = acc[e];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int q = (4 * wv + k) * 64 + lane;
        const bool ok = FULL || q0 + k < n;
        T rr = rq[k];
        if (nz > 0) {
            rr -= ((red[0][q] + red[1][q]) + (red[2][q] + red[3][q]));
            if (ok) r[q0 + k] = rr;
        }
        wrs[q] = ok ? wq[k] * rr : T(0);
        if constexpr (MEANS) wsl[q] = ok ? wq[k] : T(0);
    }
    if (nb <= 0) return;
    __syncthreads();
    // ---- (B) ----------------------------------------------------------------------------------------------------------------
    T wr[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) wr[e] = wrs[e * 64 + lane];
    T wl[16];
    if constexpr (MEANS) {
#pragma unroll
        for (int e = 0; e < 16; ++e) wl[e] = wsl[e * 64 + lane];
    }
    for (int c0 = wv; c0 < nb; c0 += 4 * U) {
        if (c0 != wv) {
#pragma unroll
            for (int u = 0; u < U; ++u) jb[u] = cols[min(c0 + 4 * u, nb - 1)];
#pragma unroll
            for (int u = 0; u < U; ++u) xb[u] = word(jb[u]);
        }
        if constexpr (!MEANS) {
            T pu[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                T xx[16];
                snp16_decode<T, FULL>(xb[u], X.impute[jb[u]], i, n, xx);
                T sacc = T(0);
#pragma unroll
                for (int e = 0; e < 16; ++e) sacc = fma(xx[e], wr[e], sacc);
                pu[u] = sacc;
            }
            const T tot = reduce16(pu, lane);
            if (lane < U && c0 + 4 * lane < nb) psum[c0 + 4 * lane] = tot;
        } else {
            // eight columns at a time: their gradient sums and their weighted sums (the column means under the CURRENT weights,
            // from the same decoded calls) share one sixteen-value butterfly -- lanes 0-7 end with the former, 8-15 the latter
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                T pv[16];
#pragma unroll
                for (int u8 = 0; u8 < 8; ++u8) {
                    const int u = 8 * h + u8;
                    T xx[16];
                    snp16_decode<T, FULL>(xb[u], X.impute[jb[u]], i, n, xx);
                    T sacc = T(0), macc = T(0);
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        sacc = fma(xx[e], wr[e], sacc);
                        macc = fma(xx[e], wl[e], macc);
                    }
                    pv[u8] = sacc;
                    pv[8 + u8] = macc;
                }
                const T tot = reduce16(pv, lane);
                const int c = c0 + 4 * (8 * h + (lane & 7));
                if (lane < 16 && c < nb) (lane < 8 ? psum : psum2)[c] = tot;
            }
        }
    }
    (void)part; (void)part_ld;
}

template <class T, bool MEANS>
__global__ __launch_bounds__(256 * S16_NSUB) void panel_step_snp16_kernel(SnpAcc<T> X, int64_t n, const T* __restrict__ w,
                                                                          T* __restrict__ r, const int32_t* __restrict__ dcol,
                                                                          const T* __restrict__ dlt,
                                                                          const int32_t* __restrict__ nz_dev,
                                                                          const int32_t* __restrict__ cols, int nb,
                                                                          T* __restrict__ part, int64_t part_ld, StepTail<T> tail) {
    __shared__ T red[S16_NSUB][4][S16_RS];
    __shared__ T wrs[S16_NSUB][S16_RS];
    __shared__ T psum[S16_NSUB][PB];
    __shared__ T wsl[MEANS ? S16_NSUB : 1][MEANS ? S16_RS : 1]; // means mode: the slice's weights
    __shared__ T psum2[S16_NSUB][PB];                           // means mode: the columns' weighted sums
    __shared__ T ptab[64 * 16]; // pair tables of phase (A): [batch of 64 columns][wave][pair of the wave's columns][16]
    const int nz = nz_dev[0];
    const int sub = threadIdx.x >> 8, t = threadIdx.x & 255;
    const int64_t slice = int64_t(blockIdx.x) * S16_NSUB + sub;
    if (nz > 0) { // pair q = bt * 32 + wave * 8 + k holds columns m_a = wave + 64 bt + 8 k and m_a + 4 of the change list
        const int e = threadIdx.x & 15, c1 = e & 3, c2 = e >> 2;
        const int npairs = ((min(nz, PB) + 63) / 64) * 32;
        for (int q = threadIdx.x >> 4; q < npairs; q += (256 * S16_NSUB) / 16) {
            const int ma = ((q >> 3) & 3) + 64 * (q >> 5) + 8 * (q & 7), mb = ma + 4;
            T va = T(0), vb = T(0);
            if (ma < nz) {
                const T cfa = dlt[ma];
                va = cfa * (c1 == 3 ? X.impute[dcol[ma]] : T(c1));
            }
            if (mb < nz) {
                const T cfb = dlt[mb];
                vb = cfb * (c2 == 3 ? X.impute[dcol[mb]] : T(c2));
            }
            ptab[q * 16 + e] = va + vb;
        }
    }
    constexpr bool means = MEANS; // (the launcher picks the instantiation: tail.xm_col != nullptr; implies a tail)
    if ((int64_t(blockIdx.x) + 1) * S16_NSUB * S16_RS <= n)
        panel_step_snp16_body<T, true, MEANS>(X, n, w, r, dcol, dlt, nz, cols, nb, part, part_ld, red[sub], wrs[sub], wsl[MEANS ? sub : 0],
                                              psum[sub], psum2[sub], ptab, t, slice);
    else
        panel_step_snp16_body<T, false, MEANS>(X, n, w, r, dcol, dlt, nz, cols, nb, part, part_ld, red[sub], wrs[sub], wsl[MEANS ? sub : 0],
                                               psum[sub], psum2[sub], ptab, t, slice);
    if (nb <= 0) return; // (uniform)
    __syncthreads();
    const bool do_tail = tail.counter != nullptr; // (then part_ld > 0: column-major partials)
    T* part2 = part + int64_t(PB) * part_ld;      // means mode: the weighted column sums, same layout behind the gradients'
    if (int(threadIdx.x) < nb) { // one partial per column and workgroup: the two slices in a fixed order
        const int c = threadIdx.x;
        const T tot = psum[0][c] + psum[1][c];
        T* dst = part + (part_ld > 0 ? int64_t(c) * part_ld + blockIdx.x : int64_t(blockIdx.x) * PB + c);
        // tail: device-coherent (written through) -- workgroups of this launch on other XCDs read it
        if (do_tail) __hip_atomic_store(dst, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *dst = tot;
        if (means)
            __hip_atomic_store(part2 + int64_t(c) * part_ld + blockIdx.x, psum2[0][c] + psum2[1][c], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!do_tail) return;
    // ---- tail: the LAST EIGHT workgroups to get here sum the partials, eight columns per workgroup and round ----------------
    // One workgroup alone pulls the 125 KB of partials at a single compute unit's ~60 GB/s (measured: slower than the reduce
    // launch it replaces); eight of them take one round trip.  They wait for the stragglers on the arrival counter (they are
    // resident and nobody waits for them: no deadlock); same sums in the same order whoever computes them.
    __shared__ int s_rank;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) s_rank = __hip_atomic_fetch_add(tail.counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - tail.base;
    __syncthreads();
    const int nwg = int(gridDim.x), NT = nwg < 8 ? nwg : 8;
    const int rank = s_rank;
    if (rank < nwg - NT) return;
    const int tidx = rank - (nwg - NT);
    if (threadIdx.x == 0) {
        while (__hip_atomic_load(tail.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - tail.base < nwg) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const int lane = threadIdx.x & 63, wvg = threadIdx.x >> 6; // 8 wavefronts: one column each per round
    for (int c = tidx * 8 + wvg; c < nb; c += NT * 8) {
        const T* pc = part + int64_t(c) * part_ld;
        T sacc = T(0);
        for (int k0 = lane; k0 < nwg; k0 += 4 * 64) { // (fixed order: lane, lane + 64, ...)
            T v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                v[u] = (k0 + 64 * u < nwg) ? __hip_atomic_load(pc + k0 + 64 * u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : T(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) sacc += v[u];
        }
        sacc = wave_sum64(sacc);
        T macc = T(0);
        if (means) {
            const T* pm = part2 + int64_t(c) * part_ld;
            for (int k0 = lane; k0 < nwg; k0 += 4 * 64) {
                T v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    v[u] = (k0 + 64 * u < nwg) ? __hip_atomic_load(pm + k0 + 64 * u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : T(0);
#pragma unroll
                for (int u = 0; u < 4; ++u) macc += v[u];
            }
            macc = wave_sum64(macc);
        }
        if (lane == 0) {
            T g = sacc;
            if (means) {
                g -= tail.rsum[0] * macc;
                tail.xm_col[cols[c]] = macc;
                tail.sxm[tail.list ? tail.list[tail.pos0 + c] : tail.pos0 + c] = macc;
            } else if (tail.xm) {
                g -= tail.rsum[0] * tail.xm[cols[c]];
            }
            tail.g[c] = g;
        }
    }
}

// Fused look-ahead step (solver.hip::run_panel_passes): workgroup 0 runs the solve of block j (blk_solve_body, one wavefront
// against the diagonal block in LDS) while workgroups 1.. run the panel step that prepares block j+1 -- phase (A) with the
// changes of block j-1, phase (B) on a residual that does not contain block j's changes yet (the next solve corrects with
// the cached cross block).  One launch, one stream: no cross-queue dependency (measured at ~35 us per hop on this
// platform, which is why solves on a second stream lose).  Workgroups are 1024 threads = four 256-thread step slices, so
// that the 160 KB LDS request of the solve (one workgroup per CU for everybody) costs the step nothing: 196 step
// workgroups of 16 waves occupy the CUs exactly like 782 workgroups of 4 waves did.
constexpr int FS = 4; // step slices per fused workgroup
// LDS of a fused launch: the solve's request, which also makes every workgroup of the launch the only one on its CU (a step
// slice needs 5 * 64 * VEC values)
template <class T>
constexpr size_t fused_lds() {
    return blk_solve_lds_la<T>() > size_t(148) * 1024 ? blk_solve_lds_la<T>() : size_t(148) * 1024;
}
// The step part of a fused launch, workgroups 1.. (1024 threads = four 256-thread slices each): phase (A) with the changes of
// the previous block, phase (B) partial gradients of the next one; the four slices of a workgroup are summed in LDS and ONE
// partial per column and workgroup goes out (column-major part[c * part_ld + g], or slice-major part[g * 128 + c] when
// part_ld == 0: for a solve that sums the partials itself).
template <class T, class Acc, int VEC>
__device__ __forceinline__ void fused_step_part(char* smem_raw, const Acc& X, int64_t n, const T* __restrict__ w,
                                                T* __restrict__ r, const int32_t* __restrict__ dcol,
                                                const T* __restrict__ dlt, const int32_t* __restrict__ nz_dev,
                                                const int32_t* __restrict__ cols, int nb, T* __restrict__ part,
                                                int64_t part_ld, int fsa, bool coh = false) {
    constexpr int RS = 64 * VEC;
    const int sub = threadIdx.x >> 8, tid = threadIdx.x & 255;
    T* base = reinterpret_cast<T*>(smem_raw) + size_t(sub) * 5 * RS;
    T (*red)[RS] = reinterpret_cast<T (*)[RS]>(base);
    T* wrs = base + 4 * RS;
    const int nz = nz_dev[0];
    static_assert(RS >= PB || VEC == 1, "a slice's LDS row holds the block's column partials");
    // `fsa` (1, 2 or 4) of the workgroup's four 256-thread parts carry a row slice each; the others only keep the barriers.
    // The launch wants about one workgroup per CU: with 256-row slices (single precision, 2-bit designs of moderate n) or few
    // rows, four slices per workgroup would leave half of the chip or more without work (fused_fsa).
    const int64_t grp = int64_t(blockIdx.x) - 1;
    const int64_t slice = grp * fsa + sub;
    if constexpr (RS >= PB) {
        if (sub >= fsa) { // idle part: the barriers of panel_step_body, zeros where the workgroup's sum reads
            if (nz > 0) __syncthreads();
            if (nb <= 0) return;
            __syncthreads();
            if (tid < PB) red[0][tid] = T(0);
        } else if ((grp + 1) * fsa * RS <= n) {
            panel_step_body<T, Acc, VEC, true, true>(X, n, w, r, dcol, dlt, nz, cols, nb, part, part_ld, red, wrs, tid, slice);
        } else {
            panel_step_body<T, Acc, VEC, false, true>(X, n, w, r, dcol, dlt, nz, cols, nb, part, part_ld, red, wrs, tid, slice);
        }
        if (nb <= 0) return; // uniform: the step bodies returned before phase (B) as well
        __syncthreads();
        // one partial per column for the whole workgroup: the four parts in a fixed order
        if (threadIdx.x < nb) {
            const T* b0 = reinterpret_cast<const T*>(smem_raw);
            const int c = threadIdx.x;
            // part_ld == 0: slice-major layout part[k * PB + c] (coalesced here and in the solve that sums the partials itself)
            const T tot = (b0[c] + b0[size_t(5) * RS + c]) + (b0[size_t(10) * RS + c] + b0[size_t(15) * RS + c]);
            T* dst = part + (part_ld > 0 ? int64_t(c) * part_ld + grp : grp * PB + c);
            // coh: device-coherent store (written through: another workgroup of this launch reads it, see the tail reduce)
            if (coh) __hip_atomic_store(dst, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *dst = tot;
        }
    } else {
        const int64_t slice4 = grp * FS + sub; // (unaligned designs, one row per lane: always four slices, one partial per slice)
        if ((grp + 1) * FS * RS <= n)
            panel_step_body<T, Acc, VEC, true>(X, n, w, r, dcol, dlt, nz, cols, nb, part, part_ld, red, wrs, tid, slice4);
        else
            panel_step_body<T, Acc, VEC, false>(X, n, w, r, dcol, dlt, nz, cols, nb, part, part_ld, red, wrs, tid, slice4);
    }
}

// active parts per fused workgroup: the largest of 4, 2, 1 that still gives the launch at most ~250 workgroups ... and the
// smallest that does not exceed them (one round of whole-CU workgroups); beyond 4 x 250 slices: 4
inline int fused_fsa(int64_t ns) {
    if ((ns + 3) / 4 > 250) return 4;
    if (ns <= 250) return 1;
    if ((ns + 1) / 2 <= 250) return 2;
    return 4;
}

template <class T, class Acc, int VEC>
__global__ __launch_bounds__(256 * FS) void panel_fused_kernel(CdBlkParams<T> sp, int j, Acc X, int64_t n,
                                                              const T* __restrict__ w, T* __restrict__ r,
                                                              const int32_t* __restrict__ dcol, const T* __restrict__ dlt,
                                                              const int32_t* __restrict__ nz_dev,
                                                              const int32_t* __restrict__ cols, int nb,
                                                              T* __restrict__ part, int64_t part_ld, int fsa) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int RS = 64 * VEC;
    if (blockIdx.x == 0) { // the solve of block j: all 1024 threads fetch, one wavefront visits (blk_solve_la_body)
        if (j < 0) { // opening launch of a pass: no solve; both look-ahead slots get the residual sum the pass starts from
            if (threadIdx.x == 0 && sp.part_rsum) {
                const T v = sp.part_rsum[0];
                sp.rsum_out[0] = v;
                sp.rsum_out[1] = v;
            }
            return;
        }
        blk_solve_la_body<T>(sp, j, smem_raw, threadIdx.x);
        return;
    }
    fused_step_part<T, Acc, VEC>(smem_raw, X, n, w, r, dcol, dlt, nz_dev, cols, nb, part, part_ld, fsa);
}

template <class T>
__global__ __launch_bounds__(PT) void panel_reduce_kernel(const T* __restrict__ part, int64_t part_ld, int nslices,
                                                          const int32_t* __restrict__ cols,
                                                          const T* __restrict__ rsum, const T* __restrict__ xm_by_col,
                                                          T* __restrict__ gblk) {
    __shared__ T red[4];
    const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const T* pc = part + int64_t(c) * part_ld;
    T s = T(0);
    for (int k = tid; k < nslices; k += PT) s += pc[k];
    s = wave_sum64(s);
    if (lane == 0) red[wv] = s;
    __syncthreads();
    if (tid == 0) {
        T g = (red[0] + red[1]) + (red[2] + red[3]);
        if (xm_by_col) g -= rsum[0] * xm_by_col[cols[c]];
        gblk[c] = g;
    }
}

// vars[a] = max(vars[a] - xm[a]^2, 0)   (solver_gaussian_naive.hpp:99-111 for groups of size one)
template <class T>
__global__ void center_vars_kernel(T* __restrict__ vars, const T* __restrict__ xm, int cnt, int center) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= cnt) return;
    T v = vars[a];
    if (center) v -= xm[a] * xm[a];
    vars[a] = v > T(0) ? v : T(0);
}

__global__ void gather_i32_kernel(const int32_t* __restrict__ src, const int32_t* __restrict__ idx, int cnt,
                                  int32_t* __restrict__ out) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a < cnt) out[a] = src[idx[a]];
}

template <class T, class Acc, int VEC>
int step_launch(const Acc& acc, int64_t n, const T* w, T* r, const int32_t* dcol, const T* dlt, const int32_t* nz_dev,
                const int32_t* cols, int nb, T* part, bool slice_major, hipStream_t s) {
    constexpr int RS = 64 * VEC;
    const int64_t ns = (n + RS - 1) / RS;
    // slice_major: part[slice * 128 + c] for a solve that sums the partials itself (blk_solve_la_body), else part[c * ns + slice]
    hipLaunchKernelGGL((panel_step_kernel<T, Acc, VEC>), dim3((unsigned)ns), dim3(PT), 0, s, acc, n, w, r, dcol, dlt,
                       nz_dev, cols, nb, part, slice_major ? int64_t(0) : ns);
    return int(ns);
}

// same with the group solve (grp_solve_body) in workgroup 0
template <class T, class Acc, int VEC>
__global__ __launch_bounds__(256 * FS) void panel_fused_grp_kernel(CdGrpBlkParams<T> sp, int j, Acc X, int64_t n,
                                                                  const T* __restrict__ w, T* __restrict__ r,
                                                                  const int32_t* __restrict__ dcol,
                                                                  const T* __restrict__ dlt,
                                                                  const int32_t* __restrict__ nz_dev,
                                                                  const int32_t* __restrict__ cols, int nb,
                                                                  T* __restrict__ part, int64_t part_ld, int fsa) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int RS = 64 * VEC;
    if (blockIdx.x == 0) { // all 256 * FS threads: the rotated solve spreads its prologue's loads over them
        if (j < 0) { // opening launch of a pass: no solve; both look-ahead slots get the residual sum the pass starts from
            if (threadIdx.x == 0 && sp.part_rsum) {
                const T v = sp.part_rsum[0];
                sp.rsum_out[0] = v;
                sp.rsum_out[1] = v;
            }
            return;
        }
        grp_solve_body<T, true, true>(sp, j, smem_raw, 256 * FS);
        return;
    }
    fused_step_part<T, Acc, VEC>(smem_raw, X, n, w, r, dcol, dlt, nz_dev, cols, nb, part, part_ld, fsa, sp.tail_counter != nullptr);
    if constexpr (RS >= PB) {
        if (sp.tail_counter == nullptr || nb <= 0) return; // (uniform)
        // ---- tail: the last EIGHT step workgroups to get here sum the partials of all of them, sixteen columns each ----
        // The partials were stored device-coherently (written through) and are read back the same way, so no cache
        // write-back / invalidate is needed (an agent-scope fence per workgroup costs more than the reduce launch it replaces:
        // measured +15 us per fused launch); the workgroup-scope fence + barrier make the stores complete before the counter
        // moves.  One workgroup alone (rounds 4-5) pulled the 200 KB at a single compute unit's ~60 GB/s, 3 us at the end of
        // every launch -- hidden while the launch was bound by its solve, on the critical path since the solve got shorter;
        // eight pull 25 KB each.  They wait for the stragglers on the arrival counter (every workgroup of the launch is
        // resident, one per compute unit, and nobody waits for them); the last of the eight to finish resets both counters
        // (tail_counter[0]: arrivals, [1]: finished tail workgroups).  Same sums in the same order as the single tail: thread
        // (column, q) adds the partials q, q + 8, ..., the eight of a column are combined pairwise.
        __shared__ int s_rank;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        if (threadIdx.x == 0) s_rank = __hip_atomic_fetch_add(sp.tail_counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int nwg = int(gridDim.x) - 1, NT = nwg < 8 ? 1 : 8;
        const int rank = s_rank;
        if (rank < nwg - NT) return;
        const int tidx = rank - (nwg - NT);
        if (threadIdx.x == 0) {
            while (__hip_atomic_load(sp.tail_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nwg) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        T* red = reinterpret_cast<T*>(smem_raw); // [8][128]
        const int cpw = PB / NT;                 // columns of this workgroup: [tidx * cpw, (tidx + 1) * cpw)
        const int cl = int(threadIdx.x) % cpw, q = int(threadIdx.x) / cpw;
        const int c = tidx * cpw + cl;
        if (q < 8) {
            T sacc = T(0);
            for (int k0 = q; k0 < nwg; k0 += 8 * 32) { // 32 loads in flight per thread: up to 256 workgroups in ONE round trip
                T v[32];
#pragma unroll
                for (int u = 0; u < 32; ++u) {
                    const int k = k0 + 8 * u;
                    v[u] = k < nwg ? __hip_atomic_load(part + int64_t(k) * PB + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : T(0);
                }
#pragma unroll
                for (int u = 0; u < 32; ++u) sacc += v[u];
            }
            red[q * PB + c] = sacc;
        }
        __syncthreads();
        if (int(threadIdx.x) < cpw && c < nb) {
            T g = ((red[c] + red[PB + c]) + (red[2 * PB + c] + red[3 * PB + c])) +
                  ((red[4 * PB + c] + red[5 * PB + c]) + (red[6 * PB + c] + red[7 * PB + c]));
            if (sp.tail_xm) g -= sp.tail_rsum[0] * sp.tail_xm[cols[c]];
            sp.tail_g[c] = g;
        }
        if (threadIdx.x == 0) {
            const int d = __hip_atomic_fetch_add(sp.tail_counter + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (d == NT - 1) {
                __hip_atomic_store(sp.tail_counter + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(sp.tail_counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

template <class T, class Acc, int VEC>
int fused_grp_launch(const CdGrpBlkParams<T>& sp, int j, const Acc& acc, int64_t n, const T* w, T* r, const int32_t* dcol,
                     const T* dlt, const int32_t* nz_dev, const int32_t* cols, int nb, T* part, bool tr, hipStream_t s) {
    constexpr int RS = 64 * VEC;
    const int64_t ns = (n + RS - 1) / RS;
    const int fsa = (64 * VEC >= PB) ? fused_fsa(ns) : FS;
    const int64_t nwg = (ns + fsa - 1) / fsa;
    const int64_t part_ld = (64 * VEC >= PB) ? nwg : nwg * FS; // (one partial per column and workgroup, as fused_launch)
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(panel_fused_grp_kernel<T, Acc, VEC>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, int(grp_solve_lds_total<T>()));
        attr_done = true;
    }
    hipLaunchKernelGGL((panel_fused_grp_kernel<T, Acc, VEC>), dim3((unsigned)(nwg + 1)), dim3(256 * FS),
                       grp_solve_lds_total<T>(), s, sp, j, acc, n, w, r, dcol, dlt, nz_dev, cols, nb, part,
                       tr ? int64_t(0) : part_ld, fsa);
    return int(part_ld);
}

template <class T, class Acc, int VEC>
int fused_launch(const CdBlkParams<T>& sp, int j, const Acc& acc, int64_t n, const T* w, T* r, const int32_t* dcol,
                 const T* dlt, const int32_t* nz_dev, const int32_t* cols, int nb, T* part, bool tr, hipStream_t s) {
    constexpr int RS = 64 * VEC;
    const int64_t ns = (n + RS - 1) / RS;
    const int fsa = (64 * VEC >= PB) ? fused_fsa(ns) : FS;
    const int64_t nwg = (ns + fsa - 1) / fsa;
    // 64 * VEC >= 128: one partial per column and WORKGROUP (the four slices are summed in LDS); else one per slice
    // (slices past the end write zeros)
    const int64_t part_ld = (64 * VEC >= PB) ? nwg : nwg * FS;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(panel_fused_kernel<T, Acc, VEC>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, int(fused_lds<T>()));
        attr_done = true;
    }
    hipLaunchKernelGGL((panel_fused_kernel<T, Acc, VEC>), dim3((unsigned)(nwg + 1)), dim3(256 * FS),
                       fused_lds<T>(), s, sp, j, acc, n, w, r, dcol, dlt, nz_dev, cols, nb, part,
                       tr ? int64_t(0) : part_ld, fsa);
    return int(part_ld);
}

} // namespace

template <class T>
int launch_panel_fused(const CdBlkParams<T>& sp, int j, const DenseView<T>& X, const T* w, T* r, const int32_t* dcol,
                       const T* dlt, const int32_t* nz_dev, const int32_t* cols, int nb, T* part, bool tr, hipStream_t s) {
    DenseAcc<T> acc{X.X, X.ld};
    constexpr int V = VecOf<T>::N;
    const bool vecok = (X.ld % V == 0) && ((reinterpret_cast<uintptr_t>(X.X) % 16) == 0);
    if (vecok) return fused_launch<T, DenseAcc<T>, V>(sp, j, acc, X.n, w, r, dcol, dlt, nz_dev, cols, nb, part, tr, s);
    return fused_launch<T, DenseAcc<T>, 1>(sp, j, acc, X.n, w, r, dcol, dlt, nz_dev, cols, nb, part, tr, s);
}
template <class T>
int launch_panel_fused_snp(const CdBlkParams<T>& sp, int j, const SnpView& X, const T* impute, const T* w, T* r,
                           const int32_t* dcol, const T* dlt, const int32_t* nz_dev, const int32_t* cols, int nb, T* part,
                           bool tr, hipStream_t s) {
    SnpAcc<T> acc{X.bits, X.ldb, impute};
    return fused_launch<T, SnpAcc<T>, 4>(sp, j, acc, X.n, w, r, dcol, dlt, nz_dev, cols, nb, part, tr, s);
}

template <class T>
int launch_panel_fused_grp(const CdGrpBlkParams<T>& sp, int j, const DenseView<T>& X, const T* w, T* r, const int32_t* dcol,
                           const T* dlt, const int32_t* nz_dev, const int32_t* cols, int nb, T* part, bool tr, hipStream_t s) {
    DenseAcc<T> acc{X.X, X.ld};
    constexpr int V = VecOf<T>::N;
    const bool vecok = (X.ld % V == 0) && ((reinterpret_cast<uintptr_t>(X.X) % 16) == 0);
    if (vecok) return fused_grp_launch<T, DenseAcc<T>, V>(sp, j, acc, X.n, w, r, dcol, dlt, nz_dev, cols, nb, part, tr, s);
    return fused_grp_launch<T, DenseAcc<T>, 1>(sp, j, acc, X.n, w, r, dcol, dlt, nz_dev, cols, nb, part, tr, s);
}
template <class T>
int launch_panel_fused_grp_snp(const CdGrpBlkParams<T>& sp, int j, const SnpView& X, const T* impute, const T* w, T* r,
                               const int32_t* dcol, const T* dlt, const int32_t* nz_dev, const int32_t* cols, int nb,
                               T* part, bool tr, hipStream_t s) {
    SnpAcc<T> acc{X.bits, X.ldb, impute};
    return fused_grp_launch<T, SnpAcc<T>, 4>(sp, j, acc, X.n, w, r, dcol, dlt, nz_dev, cols, nb, part, tr, s);
}

namespace {
template <class T>
__global__ void la_open_kernel(const T* __restrict__ grad, const int32_t* __restrict__ cols, int nb, T* __restrict__ g,
                               const T* __restrict__ rsum_src, T* __restrict__ rsum_out) {
    const int c = threadIdx.x;
    if (c < nb) g[c] = grad[cols[c]];
    if (c == 0 && rsum_src) {
        const T v = rsum_src[0];
        rsum_out[0] = v;
        rsum_out[1] = v;
    }
}
} // namespace
template <class T>
void launch_la_open_from_grad(const T* grad, const int32_t* cols, int nb, T* g, const T* rsum_src, T* rsum_out, hipStream_t s) {
    hipLaunchKernelGGL((la_open_kernel<T>), dim3(1), dim3(PB), 0, s, grad, cols, nb, g, rsum_src, rsum_out);
}
template void launch_la_open_from_grad<double>(const double*, const int32_t*, int, double*, const double*, double*, hipStream_t);
template void launch_la_open_from_grad<float>(const float*, const int32_t*, int, float*, const float*, float*, hipStream_t);

int64_t panel_part_elems(int64_t n) { return int64_t(PB) * ((n + 63) / 64) + 16; }

template <class T>
int launch_panel_step(const DenseView<T>& X, const T* w, T* r, const int32_t* dcol, const T* dlt, const int32_t* nz_dev,
                      const int32_t* cols, int nb, T* part, hipStream_t s, bool slice_major) {
    DenseAcc<T> acc{X.X, X.ld};
    constexpr int V = VecOf<T>::N;
    const bool vecok = (X.ld % V == 0) && ((reinterpret_cast<uintptr_t>(X.X) % 16) == 0);
    if (vecok) return step_launch<T, DenseAcc<T>, V>(acc, X.n, w, r, dcol, dlt, nz_dev, cols, nb, part, slice_major, s);
    return step_launch<T, DenseAcc<T>, 1>(acc, X.n, w, r, dcol, dlt, nz_dev, cols, nb, part, slice_major, s);
}
namespace {
inline bool snp16_old_form() {
    static const bool v = std::getenv("ADELIE_HIP_SNP16_OLD") != nullptr; // (A/B of the round-6 kernel)
    return v;
}
inline bool snp16_shape_ok(const SnpView& X) { return X.n >= 16384 && X.ldb % 4 == 0 && (reinterpret_cast<uintptr_t>(X.bits) % 4) == 0; }
} // namespace
bool panel_step_snp_has_tail(const SnpView& X) { return snp16_shape_ok(X) && !snp16_old_form(); }

template <class T>
int launch_panel_step_snp(const SnpView& X, const T* impute, const T* w, T* r, const int32_t* dcol, const T* dlt,
                          const int32_t* nz_dev, const int32_t* cols, int nb, T* part, hipStream_t s, bool slice_major,
                          const StepTail<T>* tail, bool* tailed) {
    SnpAcc<T> acc{X.bits, X.ldb, impute};
    if (tailed) *tailed = false;
    // 16 calls (one 32-bit word) per lane and column instead of 4 (one byte): a quarter of the workgroups, four times the
    // bytes per load instruction - the byte form is bound by the number of workgroups and load instructions, not by bytes.
    if (snp16_shape_ok(X)) {
        // (nb > 128: the opening step of a look-ahead pass prepares two blocks at once; the generic body has no per-block tables)
        if (snp16_old_form() || nb > PB) return step_launch<T, SnpAcc<T>, 16>(acc, X.n, w, r, dcol, dlt, nz_dev, cols, nb, part, slice_major, s);
        constexpr int64_t RW = int64_t(S16_RS) * S16_NSUB;
        const int64_t nwg = (X.n + RW - 1) / RW;
        StepTail<T> tl{};
        if (tail && nb > 0 && !slice_major) {
            tl = *tail;
            if (tailed) *tailed = true;
        }
        if (tl.xm_col != nullptr)
            hipLaunchKernelGGL((panel_step_snp16_kernel<T, true>), dim3((unsigned)nwg), dim3(256 * S16_NSUB), 0, s, acc, X.n, w, r,
                               dcol, dlt, nz_dev, cols, nb, part, nwg, tl);
        else
            hipLaunchKernelGGL((panel_step_snp16_kernel<T, false>), dim3((unsigned)nwg), dim3(256 * S16_NSUB), 0, s, acc, X.n, w, r,
                               dcol, dlt, nz_dev, cols, nb, part, slice_major ? int64_t(0) : nwg, tl);
        return int(nwg);
    }
    return step_launch<T, SnpAcc<T>, 4>(acc, X.n, w, r, dcol, dlt, nz_dev, cols, nb, part, slice_major, s);
}
template <class T>
void launch_panel_reduce(const T* part, int nslices, int nb, const int32_t* cols, const T* rsum_dev, const T* xm_by_col,
                         T* gblk, hipStream_t s) {
    if (nb <= 0) return;
    hipLaunchKernelGGL((panel_reduce_kernel<T>), dim3((unsigned)nb), dim3(PT), 0, s, part, int64_t(nslices), nslices, cols,
                       rsum_dev, xm_by_col, gblk);
}
// same with the partials' leading dimension given separately (fused step: padded to whole workgroups)
template <class T>
void launch_panel_reduce_ld(const T* part, int64_t part_ld, int nslices, int nb, const int32_t* cols, const T* rsum_dev,
                            const T* xm_by_col, T* gblk, hipStream_t s) {
    if (nb <= 0) return;
    hipLaunchKernelGGL((panel_reduce_kernel<T>), dim3((unsigned)nb), dim3(PT), 0, s, part, part_ld, nslices, cols, rsum_dev,
                       xm_by_col, gblk);
}
template <class T>
void launch_center_vars(T* vars, const T* xm, int cnt, bool center, hipStream_t s) {
    if (cnt <= 0) return;
    hipLaunchKernelGGL((center_vars_kernel<T>), dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s, vars, xm, cnt,
                       center ? 1 : 0);
}
void launch_gather_i32(const int32_t* src, const int32_t* idx, int cnt, int32_t* out, hipStream_t s) {
    if (cnt <= 0) return;
    hipLaunchKernelGGL(gather_i32_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s, src, idx, cnt, out);
}

#define INST(T)                                                                                                        \
    template int launch_panel_step<T>(const DenseView<T>&, const T*, T*, const int32_t*, const T*, const int32_t*,     \
                                      const int32_t*, int, T*, hipStream_t, bool);                                     \
    template int launch_panel_step_snp<T>(const SnpView&, const T*, const T*, T*, const int32_t*, const T*,            \
                                          const int32_t*, const int32_t*, int, T*, hipStream_t, bool,                  \
                                          const StepTail<T>*, bool*);                                                  \
    template void launch_panel_reduce<T>(const T*, int, int, const int32_t*, const T*, const T*, T*, hipStream_t);     \
    template void launch_panel_reduce_ld<T>(const T*, int64_t, int, int, const int32_t*, const T*, const T*, T*,       \
                                            hipStream_t);                                                              \
    template int launch_panel_fused<T>(const CdBlkParams<T>&, int, const DenseView<T>&, const T*, T*, const int32_t*,  \
                                       const T*, const int32_t*, const int32_t*, int, T*, bool, hipStream_t);          \
    template int launch_panel_fused_grp<T>(const CdGrpBlkParams<T>&, int, const DenseView<T>&, const T*, T*,           \
                                           const int32_t*, const T*, const int32_t*, const int32_t*, int, T*, bool,    \
                                           hipStream_t);                                                               \
    template int launch_panel_fused_grp_snp<T>(const CdGrpBlkParams<T>&, int, const SnpView&, const T*, const T*, T*,  \
                                               const int32_t*, const T*, const int32_t*, const int32_t*, int, T*, bool,\
                                               hipStream_t);                                                           \
    template int launch_panel_fused_snp<T>(const CdBlkParams<T>&, int, const SnpView&, const T*, const T*, T*,         \
                                           const int32_t*, const T*, const int32_t*, const int32_t*, int, T*, bool,    \
                                           hipStream_t);                                                               \
    template void launch_center_vars<T>(T*, const T*, int, bool, hipStream_t);
INST(double)
INST(float)
#undef INST

} // namespace ahip
