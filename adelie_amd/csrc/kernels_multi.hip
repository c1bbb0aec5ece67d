// kernels_multi.hip — the multi-response view [1 (x) I_K, X (x) I_K] of a resident dense design (gfx950 / CDNA4).
//
// Replaces, for StateMultiGaussianNaive (reference adelie/state.py:2027-2391, solver_multigaussian_naive.hpp:9-52), the
// composition MatrixNaiveConcatenate([MatrixNaiveKroneckerEye(ones), MatrixNaiveKroneckerEye(X)])
// (matrix_naive_kronecker_eye.ipp:27-245, matrix_naive_concatenate.ipp): there every cmul / ctmul of a view column copies
// the strided response column of v and w into a buffer and calls the base matrix, so a group of K view columns streams
// the same base column K times.  Here the residual and the weights live response-major in HBM (element (i, l) at
// [l*nb + i]) and every kernel reads a slice of a base column ONCE and applies it to all K responses from registers:
//
//   multi_sweep        out[u*K + l] = x_u . v_l for every extended feature u   (invariance / KKT sweep: X read once, not K times)
//   multi_panel_step   the panel step of kernels_cd_panel.hip on view columns: r_l -= sum Delta[u, l] x_u, then the partial
//                      gradients of the next block; entries of one feature are merged into one column-slice load
//   multi_expand       Gram block over extended features (MFMA syrk, kernels_gram.hip) -> block over view columns
//                      (entries between different responses are zero)
//   multi_axpy_cols    rollback of a failed fit
//   multi_{to,from}_major   (n, K) row-major <-> response-major, once per solve on the way in / out
//
// All of these are HBM-bound streaming kernels; the bytes per view column visited drop by K against the single-response
// kernels because the K columns of a feature share one read.
#include "kernels.hpp"
#include "accessors.hpp"
#include "wavered.hpp"
#include "grp_solve_body.hpp"

#include <algorithm>
#include <cstdlib>

namespace ahip {

namespace {

constexpr int MT = 256;  // threads of the sweep kernel
// Features per sweep block.  The K vectors are re-read (from L2) once per block: with 4 features a K = 8 sweep pulls twice as
// many bytes of v through L2 as it streams of X from HBM and ends up bound by L2 (1.73 ms = 4.6 TB/s at 100k x 10k f64); with
// 8 the two are equal.  (ADELIE_HIP_MULTI_SWEEP_MCB=4 selects the old shape for an A/B run.)
constexpr int MCB_MAX = 8;
constexpr int MAXB = 128; // entries of a panel block (== cd_block_size())

template <class T>
__device__ __forceinline__ T wsum(T x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    return x;
}

// ---- sweep ------------------------------------------------------------------------------------------------------------
// grid (feature panels, row splits, response chunks of KT).  A block owns MCB features and walks its rows once; the KT
// response vectors v_l are re-read per panel but from L2 (blocks of one row split are scheduled together and share them).
// WPC ("wave per column group"): the four wavefronts of a block own MCB features each and all walk the same rows, so a
// block covers 4 * MCB features and its waves read the same addresses of the K vectors at about the same time — the vectors
// then cross L2 -> L1 once per 32 features instead of once per 8 (the kernel's time goes with that traffic, see MCB above).
template <class T, class Acc, int VEC, int KT, int MCB, bool WPC = false>
__global__ __launch_bounds__(MT) void multi_sweep_kernel(Acc X, const T* __restrict__ v, T* __restrict__ part,
                                                        int64_t nb, int64_t nfeat, int K, int64_t rows_per_split) {
    const int tid = threadIdx.x;
    const int64_t cb = WPC ? int64_t(blockIdx.x) * (MT / 64) + (tid >> 6) : int64_t(blockIdx.x);
    constexpr int RT = WPC ? 64 : MT;         // threads that share the rows of one feature group
    const int rtid = WPC ? (tid & 63) : tid;
    const int split = blockIdx.y;
    const int l0 = blockIdx.z * KT;
    const int64_t r0 = int64_t(split) * rows_per_split;
    const int64_t r1 = min(nb, r0 + rows_per_split);
    decltype(X.colptr(0)) cp[MCB]; // dense: const T*; 2-bit SNP: const uint8_t*
    int64_t cj[MCB];
#pragma unroll
    for (int k = 0; k < MCB; ++k) {
        cj[k] = min(cb * MCB + k, nfeat - 1);
        cp[k] = X.colptr(cj[k]);
    }
    const T* vp[KT];
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) vp[kk] = v + int64_t(min(l0 + kk, K - 1)) * nb; // clamped: duplicates are discarded below
    T acc[MCB][KT];
#pragma unroll
    for (int k = 0; k < MCB; ++k)
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) acc[k][kk] = T(0);

    const int64_t body_end = r0 + ((r1 - r0) / VEC) * VEC;
    for (int64_t i = r0 + int64_t(rtid) * VEC; i < body_end; i += int64_t(RT) * VEC) {
        Pack<T, VEC> xx[MCB];
#pragma unroll
        for (int k = 0; k < MCB; ++k) xx[k] = X.template load<VEC>(cp[k], i, cj[k]);
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const T vv = vp[kk][i + e];
#pragma unroll
                for (int k = 0; k < MCB; ++k) acc[k][kk] = fma(xx[k].v[e], vv, acc[k][kk]);
            }
        }
    }
    for (int64_t i = body_end + rtid; i < r1; i += RT) {
        T x1[MCB];
#pragma unroll
        for (int k = 0; k < MCB; ++k) x1[k] = X.template load<1>(cp[k], i, cj[k]).v[0];
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) {
            const T vv = vp[kk][i];
#pragma unroll
            for (int k = 0; k < MCB; ++k) acc[k][kk] = fma(x1[k], vv, acc[k][kk]);
        }
    }

    if (WPC) { // every wavefront has its own features: no exchange between them
        const int lane = tid & 63;
#pragma unroll
        for (int k = 0; k < MCB; ++k)
#pragma unroll
            for (int kk = 0; kk < KT; ++kk) {
                const T s = wsum(acc[k][kk]);
                const int64_t u = cb * MCB + k;
                const int l = l0 + kk;
                if (lane == 0 && u < nfeat && l < K) part[(int64_t(split) * nfeat + u) * K + l] = s;
            }
        return;
    }
    __shared__ T red[MT / 64][MCB * KT];
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int k = 0; k < MCB; ++k)
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) {
            const T s = wsum(acc[k][kk]);
            if (lane == 0) red[wv][k * KT + kk] = s;
        }
    __syncthreads();
    if (tid < MCB * KT) {
        const int k = tid / KT, kk = tid % KT;
        const int64_t u = cb * MCB + k;
        const int l = l0 + kk;
        if (u < nfeat && l < K) {
            T s = T(0);
#pragma unroll
            for (int w = 0; w < MT / 64; ++w) s += red[w][tid];
            part[(int64_t(split) * nfeat + u) * K + l] = s;
        }
    }
}

// The K = 8 sweep with the vectors staged through LDS.  The plain kernel's time goes with the bytes of the K vectors that
// cross L2 -> L1 (measured: 1.73 ms with 4 features per block, 1.28-1.38 ms with 8; re-mapping the four waves onto feature
// groups of their own without sharing changes nothing, i.e. waves do not meet in L1).  Here a block of four waves covers
// 32 features: all waves walk the same rows, the block fetches each 8 KB tile of the vectors (64*VEC rows x 8 vectors) ONCE,
// double-buffered in LDS with one barrier per tile, and every wave reads it from there — the vectors cross L2 -> L1 once per
// 32 features instead of once per 8.
template <class T, class Acc, int VEC>
__global__ __launch_bounds__(MT) void multi_sweep_lds_kernel(Acc X, const T* __restrict__ v, T* __restrict__ part,
                                                            int64_t nb, int64_t nfeat, int K, int64_t rows_per_split) {
    constexpr int KT = 8, MCB = 8, TR = 64 * VEC;
    using V = typename VecOf<T>::type; // 16 bytes = VEC rows
    static_assert(MT == 256 && sizeof(V) == 16, "staging map below: 256 threads x 32 bytes = one 8 KB tile");
    __shared__ V tile[2][KT][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t cb = int64_t(blockIdx.x) * (MT / 64) + (tid >> 6);
    const int split = blockIdx.y;
    const int l0 = blockIdx.z * KT;
    const int64_t r0 = int64_t(split) * rows_per_split;
    const int64_t r1 = min(nb, r0 + rows_per_split);
    decltype(X.colptr(0)) cp[MCB];
    int64_t cj[MCB];
#pragma unroll
    for (int k = 0; k < MCB; ++k) {
        cj[k] = min(cb * MCB + k, nfeat - 1);
        cp[k] = X.colptr(cj[k]);
    }
    T acc[MCB][KT];
#pragma unroll
    for (int k = 0; k < MCB; ++k)
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) acc[k][kk] = T(0);

    const int64_t n_tiles = (r1 - r0) / TR;
    // staging role of this thread: vector sk, the two 16-byte pieces sc, sc + 1 of its tile row
    const int sk = tid >> 5, sc = (tid & 31) * 2;
    const T* svp = v + int64_t(min(l0 + sk, K - 1)) * nb; // clamped: duplicates are discarded below
    V s0, s1;
    if (n_tiles > 0) {
        s0 = *reinterpret_cast<const V*>(svp + r0 + int64_t(sc) * VEC);
        s1 = *reinterpret_cast<const V*>(svp + r0 + int64_t(sc + 1) * VEC);
        tile[0][sk][sc] = s0;
        tile[0][sk][sc + 1] = s1;
    }
    __syncthreads();
    for (int64_t it = 0; it < n_tiles; ++it) {
        const int64_t base = r0 + it * TR, i = base + int64_t(lane) * VEC;
        const bool more = it + 1 < n_tiles;
        if (more) {
            s0 = *reinterpret_cast<const V*>(svp + base + TR + int64_t(sc) * VEC);
            s1 = *reinterpret_cast<const V*>(svp + base + TR + int64_t(sc + 1) * VEC);
        }
        Pack<T, VEC> xx[MCB];
#pragma unroll
        for (int k = 0; k < MCB; ++k) xx[k] = X.template load<VEC>(cp[k], i, cj[k]);
        const int b = int(it & 1);
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) {
            const V vv = tile[b][kk][lane];
#pragma unroll
            for (int e = 0; e < VEC; ++e)
#pragma unroll
                for (int k = 0; k < MCB; ++k) acc[k][kk] = fma(xx[k].v[e], vv[e], acc[k][kk]);
        }
        if (more) {
            tile[b ^ 1][sk][sc] = s0;
            tile[b ^ 1][sk][sc + 1] = s1;
        }
        __syncthreads();
    }
    for (int64_t i = r0 + n_tiles * TR + lane; i < r1; i += 64) { // rows beyond the last whole tile
        T x1[MCB];
#pragma unroll
        for (int k = 0; k < MCB; ++k) x1[k] = X.template load<1>(cp[k], i, cj[k]).v[0];
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) {
            const T vv = v[int64_t(min(l0 + kk, K - 1)) * nb + i];
#pragma unroll
            for (int k = 0; k < MCB; ++k) acc[k][kk] = fma(x1[k], vv, acc[k][kk]);
        }
    }
#pragma unroll
    for (int k = 0; k < MCB; ++k)
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) {
            const T s = wsum(acc[k][kk]);
            const int64_t u = cb * MCB + k;
            const int l = l0 + kk;
            if (lane == 0 && u < nfeat && l < K) part[(int64_t(split) * nfeat + u) * K + l] = s;
        }
}

template <class T>
__global__ void multi_sweep_reduce_kernel(const T* __restrict__ part, T* __restrict__ out, int64_t ncols, int nsplit) {
    const int64_t c = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= ncols) return;
    T s = T(0);
    for (int r = 0; r < nsplit; ++r) s += part[int64_t(r) * ncols + c];
    out[c] = s;
}

inline int msweep_mcb() {
    return 8; // features per block (4 was the round-1 shape: the K vectors re-read from L2 twice as often)
}
// 0: one feature group per block; 1: four waves x 8 features; 2: four waves x 4 features (half the accumulators, twice the
// waves per SIMD: 2.2 ms, the vectors' L2 traffic doubles); 3: four waves x 8 features with the vectors staged through LDS
inline int msweep_wpc() {
    return 3; // (the unstaged forms 1 / 2 measured 1.36 / 2.23 ms against 1.335 ms at K = 8)
}
inline void msweep_shape(int64_t nb, int64_t nfeat, int vec, int64_t& blocks_c, int& nsplit, int64_t& rows_per_split,
                         int wpc = 0) {
    const int MCB = wpc == 2 ? 4 * (MT / 64) : wpc ? 8 * (MT / 64) : msweep_mcb();
    blocks_c = (nfeat + MCB - 1) / MCB;
    const int64_t unit = int64_t(MT) * vec;
    const int64_t max_split = std::max<int64_t>(1, (nb + unit * 4 - 1) / (unit * 4));
    int64_t ns = std::max<int64_t>(1, (1024 + blocks_c - 1) / blocks_c);
    ns = std::min<int64_t>(std::min<int64_t>(ns, max_split), 65535);
    rows_per_split = (nb + ns - 1) / ns;
    rows_per_split = ((rows_per_split + unit - 1) / unit) * unit;
    ns = std::max<int64_t>(1, (nb + rows_per_split - 1) / rows_per_split);
    nsplit = int(ns);
}

template <class T>
bool multi_vecok(const MultiView<T>& X) {
    constexpr int V = VecOf<T>::N;
    if (X.bits) return true; // (2-bit base: a lane's V calls lie in one byte whatever the column's address)
    return (X.ld % V == 0) && ((reinterpret_cast<uintptr_t>(X.X) % 16) == 0) &&
           ((reinterpret_cast<uintptr_t>(X.ones) % 16) == 0);
}

// ---- panel step -------------------------------------------------------------------------------------------------------
// One wave per workgroup owns 64*VEC base rows and a chunk of KT responses.  Prologue (LDS): the entry lists (view
// columns) are merged into per-feature slots -- consecutive entries of one extended feature share a slot -- with the
// coefficients spread over the KT responses; then phase (A)/(B) of kernels_cd_panel.hip run per slot, with the residual
// slices of the KT responses in registers.
template <class T, int VEC>
__device__ __forceinline__ Pack<T, VEC> mload(const T* col, int64_t i, int64_t nb, bool full) {
    Pack<T, VEC> r;
    if constexpr (VEC == 1) {
        const T x = col[(full || i < nb) ? i : 0];
        r.v[0] = (full || i < nb) ? x : T(0);
    } else {
        using V = typename VecOf<T>::type;
        const int64_t ii = (full || i < nb) ? i : 0; // ld % VEC == 0: a lane starting below nb reads at most pad elements
        const V x = *reinterpret_cast<const V*>(col + ii);
#pragma unroll
        for (int e = 0; e < VEC; ++e) r.v[e] = (full || i + e < nb) ? x[e] : T(0);
    }
    return r;
}

// VEC rows of extended feature u starting at row i (see mload for `full`)
template <class T, int VEC>
__device__ __forceinline__ Pack<T, VEC> mcol(const DenseOnesAcc<T>& X, int u, int64_t i, int64_t nb, bool full) {
    return mload<T, VEC>(X.colptr(u), i, nb, full);
}
template <class T, int VEC>
__device__ __forceinline__ Pack<T, VEC> mcol(const SnpOnesAcc<T>& X, int u, int64_t i, int64_t nb, bool full) {
    // (i is a multiple of VEC: the 2 or 4 calls of a lane lie in one byte)
    const int64_t ii = (full || i < nb) ? i : 0;
    const Pack<T, VEC> x = X.template load<VEC>(X.colptr(u), ii, u);
    Pack<T, VEC> r;
#pragma unroll
    for (int e = 0; e < VEC; ++e) r.v[e] = (full || i + e < nb) ? x.v[e] : T(0);
    return r;
}

// slots of a list of view columns: returns the number of slots; feat[slot] = extended feature; for every entry m the
// callback gets (m, slot, response)
template <class F>
__device__ __forceinline__ int build_slots(const int32_t* __restrict__ list, int cnt, int K, int* feat, int lane, F f) {
    int nslot = 0;
    for (int base = 0; base < cnt; base += 64) {
        const int m = base + lane;
        const bool valid = m < cnt;
        const int col = valid ? list[m] : 0;
        const int u = col / K, l = col - u * K;
        const int pu = (valid && m > 0) ? list[m - 1] / K : -1;
        const bool head = valid && (m == 0 || u != pu);
        const unsigned long long mask = __ballot(head);
        const int slot = nslot + __popcll(mask & ((2ull << lane) - 1ull)) - 1;
        if (head) feat[slot] = u;
        if (valid) f(m, slot, l);
        nslot += __popcll(mask);
    }
    return nslot;
}

// phases (A) and (B) of the multi-response panel step for one wave (64 * VEC rows starting at row i, responses l0..l0+KT)
template <class T, class Acc, int VEC, int KT>
__device__ __forceinline__ void multi_step_phases(const Acc& X, int64_t nb, int K, const T* __restrict__ w,
                                                  T* __restrict__ r, const int* featA, const int* featB, const T* dA,
                                                  const int* valB, int nsA, int nsB, int l0, int lane, int64_t i,
                                                  bool full, int64_t slice, T* __restrict__ part, int64_t part_ld) {
    constexpr int U = 8;
    // ---- (A) ---------------------------------------------------------------------------------------------------------
    T acc[KT][VEC];
#pragma unroll
    for (int kk = 0; kk < KT; ++kk)
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[kk][e] = T(0);
    for (int s0 = 0; s0 < nsA; s0 += U) {
        Pack<T, VEC> xa[U];
#pragma unroll
        for (int u = 0; u < U; ++u) xa[u] = mcol<T, VEC>(X, featA[min(s0 + u, nsA - 1)], i, nb, full);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (s0 + u < nsA) {
#pragma unroll
                for (int kk = 0; kk < KT; ++kk) {
                    const T d = dA[(s0 + u) * KT + kk];
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[kk][e] = fma(d, xa[u].v[e], acc[kk][e]);
                }
            }
        }
    }
    // residual slices of the chunk's responses; acc becomes w * r
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) {
        const int l = l0 + kk;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            T wr = T(0);
            if (l < K && i + e < nb) {
                const int64_t q = int64_t(l) * nb + i + e;
                T rr = r[q];
                if (nsA > 0) {
                    rr -= acc[kk][e];
                    r[q] = rr;
                }
                wr = w[q] * rr;
            }
            acc[kk][e] = wr;
        }
    }
    // ---- (B) ---------------------------------------------------------------------------------------------------------
    for (int s0 = 0; s0 < nsB; s0 += U) {
        Pack<T, VEC> xb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) xb[u] = mcol<T, VEC>(X, featB[min(s0 + u, nsB - 1)], i, nb, full);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (s0 + u < nsB) {
                T d8[8];
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    T d = T(0);
                    if (kk < KT) {
#pragma unroll
                        for (int e = 0; e < VEC; ++e) d = fma(xb[u].v[e], acc[kk][e], d);
                    }
                    d8[kk] = d;
                }
                const T tot = reduce8(d8, lane); // lane kk < 8: the total of response l0 + kk
                if (lane < KT) {
                    const int c = valB[(s0 + u) * KT + lane];
                    if (c >= 0) part[int64_t(c) * part_ld + slice] = tot;
                }
            }
        }
    }
}

template <class T, class Acc, int VEC, int KT>
__global__ __launch_bounds__(64) void multi_panel_step_kernel(Acc X, int64_t nb, int K,
                                                              const T* __restrict__ w, T* __restrict__ r,
                                                              const int32_t* __restrict__ dcol,
                                                              const T* __restrict__ dlt,
                                                              const int32_t* __restrict__ nz_dev,
                                                              const int32_t* __restrict__ cols, int nbc,
                                                              T* __restrict__ part, int64_t part_ld) {
    // the gradient list may span two blocks (first step of a look-ahead pass): up to 2 * MAXB entries
    __shared__ int featA[MAXB], featB[2 * MAXB];
    __shared__ T dA[MAXB * KT];
    __shared__ int valB[2 * MAXB * KT];
    const int lane = threadIdx.x;
    const int l0 = blockIdx.y * KT;
    const int64_t i = int64_t(blockIdx.x) * (64 * VEC) + int64_t(lane) * VEC;
    const bool full = (int64_t(blockIdx.x) + 1) * (64 * VEC) <= nb;
    const int nz = min(nz_dev[0], MAXB);
    nbc = min(nbc, 2 * MAXB);

    for (int q = lane; q < 2 * MAXB * KT; q += 64) {
        if (q < MAXB * KT) dA[q] = T(0);
        valB[q] = -1;
    }
    __syncthreads();
    const int nsA = build_slots(dcol, nz, K, featA, lane, [&](int m, int slot, int l) {
        if (l >= l0 && l < l0 + KT) dA[slot * KT + (l - l0)] = dlt[m];
    });
    const int nsB = build_slots(cols, nbc, K, featB, lane, [&](int m, int slot, int l) {
        if (l >= l0 && l < l0 + KT) valB[slot * KT + (l - l0)] = m;
    });
    __syncthreads();

    multi_step_phases<T, Acc, VEC, KT>(X, nb, K, w, r, featA, featB, dA, valB, nsA, nsB, l0, lane, i, full, blockIdx.x, part,
                                       part_ld);
}

// Fused look-ahead launch on the view (solver.hip::run_group_panel_passes): workgroup (0, 0) runs the group solve of block j,
// every other workgroup of row 0.. is 16 waves = 16 row slices of the multi-response step that prepares block j+1; the slot
// lists are built once per workgroup.  1024-thread workgroups for the same reason as panel_fused_kernel (the solve's LDS
// request makes it one workgroup per CU for everybody).
constexpr int MFS = 16; // waves (row slices) per fused step workgroup
template <class T, class Acc, int VEC, int KT>
__global__ __launch_bounds__(64 * MFS) void multi_fused_kernel(CdGrpBlkParams<T> sp, int j, Acc X, int64_t nb,
                                                              int K, const T* __restrict__ w, T* __restrict__ r,
                                                              const int32_t* __restrict__ dcol,
                                                              const T* __restrict__ dlt,
                                                              const int32_t* __restrict__ nz_dev,
                                                              const int32_t* __restrict__ cols, int nbc,
                                                              T* __restrict__ part, int64_t part_ld) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    if (blockIdx.x == 0) {
        if (blockIdx.y == 0) grp_solve_body<T, true>(sp, j, smem_raw, 64 * MFS);
        return;
    }
    T* dA = reinterpret_cast<T*>(smem_raw);                 // MAXB * KT
    int* valB = reinterpret_cast<int*>(dA + MAXB * KT);     // MAXB * KT
    int* featA = valB + MAXB * KT;                          // MAXB
    int* featB = featA + MAXB;                              // MAXB
    int* cnts = featB + MAXB;                               // nsA, nsB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l0 = blockIdx.y * KT;
    const int nz = min(nz_dev[0], MAXB);
    for (int q = threadIdx.x; q < MAXB * KT; q += 64 * MFS) {
        dA[q] = T(0);
        valB[q] = -1;
    }
    __syncthreads();
    if (wave == 0) {
        const int a = build_slots(dcol, nz, K, featA, lane, [&](int m, int slot, int l) {
            if (l >= l0 && l < l0 + KT) dA[slot * KT + (l - l0)] = dlt[m];
        });
        const int b = build_slots(cols, nbc, K, featB, lane, [&](int m, int slot, int l) {
            if (l >= l0 && l < l0 + KT) valB[slot * KT + (l - l0)] = m;
        });
        if (lane == 0) { cnts[0] = a; cnts[1] = b; }
    }
    __syncthreads();
    const int nsA = cnts[0], nsB = cnts[1];
    const int64_t slice = (int64_t(blockIdx.x) - 1) * MFS + wave;
    const int64_t i = slice * (64 * VEC) + int64_t(lane) * VEC;
    const bool full = (slice + 1) * (64 * VEC) <= nb;
    multi_step_phases<T, Acc, VEC, KT>(X, nb, K, w, r, featA, featB, dA, valB, nsA, nsB, l0, lane, i, full, slice, part, part_ld);
}

// cross block over view columns from the Gram of the two blocks' distinct features: rows = block b, columns = block b-1
template <class T>
__global__ void multi_expand_cross_kernel(const T* __restrict__ G, int64_t ldg, const int32_t* __restrict__ slot_r,
                                          const int32_t* __restrict__ resp_r, int nr, const int32_t* __restrict__ slot_c,
                                          const int32_t* __restrict__ resp_c, int nc, T* __restrict__ C, int64_t ldc) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (a >= nr || b >= nc) return;
    C[a + int64_t(b) * ldc] = (resp_r[a] == resp_c[b]) ? G[slot_r[a] + int64_t(slot_c[b]) * ldg] : T(0);
}

template <class T>
__global__ void multi_expand_kernel(const T* __restrict__ C, int64_t ldc, const int32_t* __restrict__ slot,
                                    const int32_t* __restrict__ resp, int nv, int lsel, T* __restrict__ D, int64_t ldd) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (a >= nv || b >= nv) return;
    const int la = resp[a], lb = resp[b];
    if (la == lb && (lsel < 0 || la == lsel)) D[a + int64_t(b) * ldd] = C[slot[a] + int64_t(slot[b]) * ldc];
    else if (lsel <= 0) D[a + int64_t(b) * ldd] = T(0);
}

// ulist / slot / resp of a block's view columns: distinct extended features in order of first appearance (one workgroup
// of MAXB threads; the host counts the same way to size the syrk launch)
__global__ __launch_bounds__(MAXB) void multi_block_lists_kernel(const int32_t* __restrict__ cols, int nv, int K,
                                                                 int32_t* __restrict__ ulist, int32_t* __restrict__ slot,
                                                                 int32_t* __restrict__ resp) {
    __shared__ int us[MAXB], first[MAXB], rank[MAXB];
    const int a = threadIdx.x;
    const int u = a < nv ? cols[a] / K : -1;
    us[a] = u;
    __syncthreads();
    int f = a;
    for (int b = 0; b < a; ++b)
        if (us[b] == u) { f = b; break; }
    first[a] = f;
    __syncthreads();
    int rk = 0;
    for (int b = 0; b < a; ++b) rk += (b < nv && first[b] == b) ? 1 : 0;
    rank[a] = rk;
    __syncthreads();
    if (a < nv) {
        if (first[a] == a) ulist[rank[a]] = u;
        slot[a] = rank[first[a]];
        resp[a] = cols[a] - u * K;
    }
}

template <class T, class Acc>
__global__ void multi_axpy_kernel(Acc X, int64_t nb, int K, const int32_t* __restrict__ cols,
                                  const T* __restrict__ coef, const int32_t* __restrict__ cnt_dev, T sign,
                                  T* __restrict__ out) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= nb) return;
    const int cnt = cnt_dev[0];
    for (int m = 0; m < cnt; ++m) {
        const int col = cols[m];
        const int u = col / K, l = col - u * K;
        out[int64_t(l) * nb + i] += sign * coef[m] * mcol<T, 1>(X, u, i, nb, true).v[0];
    }
}

// out[u] = src[u*K + slot] - (sub_vec ? sub_scale[0] * sub_vec[u] : 0): one solver's share of a batched K-vector sweep
template <class T>
__global__ void batch_pick_kernel(const T* __restrict__ src, int64_t nfeat, int K, int slot, const T* __restrict__ sub_scale,
                                  const T* __restrict__ sub_vec, T* __restrict__ out) {
    const int64_t u = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (u >= nfeat) return;
    T s = src[u * K + slot];
    if (sub_vec) s -= sub_scale[0] * sub_vec[u];
    out[u] = s;
}
template <class T>
__global__ void multi_to_major_kernel(const T* __restrict__ src, int64_t nb, int K, T* __restrict__ dst) {
    const int64_t q = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (q >= nb * K) return;
    const int64_t l = q / nb, i = q - l * nb;
    dst[q] = src[i * K + l];
}
template <class T>
__global__ void multi_from_major_kernel(const T* __restrict__ src, int64_t nb, int K, T* __restrict__ dst) {
    const int64_t q = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (q >= nb * K) return;
    const int64_t i = q / K, l = q - i * K;
    dst[q] = src[l * nb + i];
}

inline int kt_of(int K) { return K > 4 ? 8 : (K > 2 ? 4 : 2); }

} // namespace

template <class T>
int64_t multi_sweep_work_elems(const MultiView<T>& X) {
    int64_t bc, rps;
    int ns;
    msweep_shape(X.nb, X.pb + X.icpt, VecOf<T>::N, bc, ns, rps);
    int64_t bc2, rps2;
    int ns2;
    msweep_shape(X.nb, X.pb + X.icpt, VecOf<T>::N, bc2, ns2, rps2, 1);
    ns = std::max(ns, ns2);
    msweep_shape(X.nb, X.pb + X.icpt, VecOf<T>::N, bc2, ns2, rps2, 2);
    return int64_t(std::max(ns, ns2)) * (X.pb + X.icpt) * X.K + 16;
}

namespace {
template <class T, class Acc>
void multi_sweep_launch(const Acc& acc, const MultiView<T>& X, const T* v, T* out, T* work, hipStream_t s) {
    const int64_t nfeat = X.pb + X.icpt;
    constexpr int V = VecOf<T>::N;
    int64_t bc, rps;
    int ns;
    const int KT = kt_of(X.K);
    const int wpc = (KT == 8 && msweep_mcb() == 8) ? msweep_wpc() : 0;
    msweep_shape(X.nb, nfeat, V, bc, ns, rps, wpc); // shape for the vector width (also valid for scalar loads)
    const dim3 grid((unsigned)bc, (unsigned)ns, (unsigned)((X.K + KT - 1) / KT));
    const bool vok = multi_vecok(X);
#define AHIP_MS(VV, KK)                                                                                                 \
    do {                                                                                                                \
        if (wpc == 2)                                                                                                   \
            hipLaunchKernelGGL((multi_sweep_kernel<T, Acc, VV, KK, 4, true>), grid, dim3(MT), 0, s, acc, v, work, X.nb, nfeat, int(X.K), rps); \
        else if (wpc)                                                                                                   \
            hipLaunchKernelGGL((multi_sweep_kernel<T, Acc, VV, KK, 8, true>), grid, dim3(MT), 0, s, acc, v, work, X.nb, nfeat, int(X.K), rps); \
        else if (msweep_mcb() == 8)                                                                                     \
            hipLaunchKernelGGL((multi_sweep_kernel<T, Acc, VV, KK, 8>), grid, dim3(MT), 0, s, acc, v, work, X.nb, nfeat, int(X.K), rps); \
        else                                                                                                            \
            hipLaunchKernelGGL((multi_sweep_kernel<T, Acc, VV, KK, 4>), grid, dim3(MT), 0, s, acc, v, work, X.nb, nfeat, int(X.K), rps); \
    } while (0)
    const bool v_al = X.nb % V == 0 && (reinterpret_cast<uintptr_t>(v) % 16) == 0; // each of the K vectors starts 16-byte aligned
    if (vok && v_al && wpc == 3) {
        hipLaunchKernelGGL((multi_sweep_lds_kernel<T, Acc, V>), grid, dim3(MT), 0, s, acc, v, work, X.nb, nfeat, int(X.K), rps);
    } else if (vok) {
        if (KT == 8) AHIP_MS(V, 8); else if (KT == 4) AHIP_MS(V, 4); else AHIP_MS(V, 2);
    } else {
        if (KT == 8) AHIP_MS(1, 8); else if (KT == 4) AHIP_MS(1, 4); else AHIP_MS(1, 2);
    }
#undef AHIP_MS
    const int64_t ncols = nfeat * X.K;
    hipLaunchKernelGGL((multi_sweep_reduce_kernel<T>), dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, s, work, out,
                       ncols, ns);
}
} // namespace
template <class T>
void launch_multi_sweep(const MultiView<T>& X, const T* v, T* out, T* work, hipStream_t s) {
    if (X.pb + X.icpt <= 0 || X.K <= 0) return;
    if (X.bits) multi_sweep_launch<T, SnpOnesAcc<T>>(SnpOnesAcc<T>{X.bits, X.ldb, X.impute, int64_t(X.icpt)}, X, v, out, work, s);
    else multi_sweep_launch<T, DenseOnesAcc<T>>(DenseOnesAcc<T>{X.X, X.ld, X.ones, int64_t(X.icpt)}, X, v, out, work, s);
}

// the same sweep over a 2-bit SNP design (no ones column): out[u*K + l] = x_u . v_l; the calls of a column are decoded once
// for all K vectors.  `work` holds multi_sweep_work_elems of the equivalent view (nb = n, pb = p, icpt = 0).
template <class T>
void launch_multi_sweep_snp(const SnpView& X, const T* impute, int K, const T* v, T* out, T* work, hipStream_t s) {
    const int64_t nfeat = X.p;
    if (nfeat <= 0 || K <= 0) return;
    SnpAcc<T> acc{X.bits, X.ldb, impute};
    constexpr int V = VecOf<T>::N; // 2 or 4 calls per load: a row split starts at a multiple of MT * V, so loads stay inside a byte
    int64_t bc, rps;
    int ns;
    msweep_shape(X.n, nfeat, V, bc, ns, rps);
    const int KT = kt_of(K);
    const dim3 grid((unsigned)bc, (unsigned)ns, (unsigned)((K + KT - 1) / KT));
#define AHIP_MSS(KK)                                                                                                    \
    do {                                                                                                                \
        if (msweep_mcb() == 8)                                                                                          \
            hipLaunchKernelGGL((multi_sweep_kernel<T, SnpAcc<T>, V, KK, 8>), grid, dim3(MT), 0, s, acc, v, work, X.n, nfeat, K, rps); \
        else                                                                                                            \
            hipLaunchKernelGGL((multi_sweep_kernel<T, SnpAcc<T>, V, KK, 4>), grid, dim3(MT), 0, s, acc, v, work, X.n, nfeat, K, rps); \
    } while (0)
    if (KT == 8) AHIP_MSS(8); else if (KT == 4) AHIP_MSS(4); else AHIP_MSS(2);
#undef AHIP_MSS
    const int64_t ncols = nfeat * K;
    hipLaunchKernelGGL((multi_sweep_reduce_kernel<T>), dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, s, work, out,
                       ncols, ns);
}

int64_t multi_panel_part_elems(int64_t nb) { return int64_t(MAXB) * ((nb + 63) / 64) + 16; }

namespace {
template <class T, class Acc>
int multi_step_launch(const Acc& acc, const MultiView<T>& X, const T* w, T* r, const int32_t* dcol, const T* dlt,
                      const int32_t* nz_dev, const int32_t* cols, int nb_cols, T* part, hipStream_t s) {
    constexpr int V = VecOf<T>::N;
    const bool vok = multi_vecok(X);
    const int RS = 64 * (vok ? V : 1);
    const int64_t nsl = (X.nb + RS - 1) / RS;
    const int KT = kt_of(X.K);
    const dim3 grid((unsigned)nsl, (unsigned)((X.K + KT - 1) / KT));
#define AHIP_MP(VV, KK)                                                                                                 \
    hipLaunchKernelGGL((multi_panel_step_kernel<T, Acc, VV, KK>), grid, dim3(64), 0, s, acc, X.nb, int(X.K), w, r, dcol, dlt, \
                       nz_dev, cols, nb_cols, part, nsl)
    if (vok) {
        if (KT == 8) AHIP_MP(V, 8); else if (KT == 4) AHIP_MP(V, 4); else AHIP_MP(V, 2);
    } else {
        if (KT == 8) AHIP_MP(1, 8); else if (KT == 4) AHIP_MP(1, 4); else AHIP_MP(1, 2);
    }
#undef AHIP_MP
    return int(nsl);
}
template <class T, class Acc>
int multi_fused_launch(const Acc& acc, const CdGrpBlkParams<T>& sp, int j, const MultiView<T>& X, const T* w, T* r,
                       const int32_t* dcol, const T* dlt, const int32_t* nz_dev, const int32_t* cols, int nb_cols, T* part,
                       hipStream_t s) {
    constexpr int V = VecOf<T>::N;
    const bool vok = multi_vecok(X);
    const int RS = 64 * (vok ? V : 1);
    const int64_t nsl = (X.nb + RS - 1) / RS;
    const int64_t nwg = (nsl + MFS - 1) / MFS;
    const int64_t part_ld = nwg * MFS;
    const int KT = kt_of(X.K);
    const dim3 grid((unsigned)(nwg + 1), (unsigned)((X.K + KT - 1) / KT));
    const size_t lds = grp_solve_lds_total<T>();
#define AHIP_MF(VV, KK)                                                                                                 \
    {                                                                                                                  \
        static bool done = false;                                                                                      \
        if (!done) {                                                                                                   \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(multi_fused_kernel<T, Acc, VV, KK>),               \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));                           \
            done = true;                                                                                               \
        }                                                                                                              \
        hipLaunchKernelGGL((multi_fused_kernel<T, Acc, VV, KK>), grid, dim3(64 * MFS), lds, s, sp, j, acc, X.nb, int(X.K), w, r, \
                           dcol, dlt, nz_dev, cols, nb_cols, part, part_ld);                                           \
    }
    if (vok) {
        if (KT == 8) AHIP_MF(V, 8) else if (KT == 4) AHIP_MF(V, 4) else AHIP_MF(V, 2)
    } else {
        if (KT == 8) AHIP_MF(1, 8) else if (KT == 4) AHIP_MF(1, 4) else AHIP_MF(1, 2)
    }
#undef AHIP_MF
    return int(part_ld);
}
} // namespace

template <class T>
int launch_multi_panel_step(const MultiView<T>& X, const T* w, T* r, const int32_t* dcol, const T* dlt,
                            const int32_t* nz_dev, const int32_t* cols, int nb_cols, T* part, hipStream_t s) {
    if (X.bits)
        return multi_step_launch<T, SnpOnesAcc<T>>(SnpOnesAcc<T>{X.bits, X.ldb, X.impute, int64_t(X.icpt)}, X, w, r, dcol, dlt,
                                                   nz_dev, cols, nb_cols, part, s);
    return multi_step_launch<T, DenseOnesAcc<T>>(DenseOnesAcc<T>{X.X, X.ld, X.ones, int64_t(X.icpt)}, X, w, r, dcol, dlt, nz_dev,
                                                 cols, nb_cols, part, s);
}

template <class T>
int launch_multi_panel_fused(const CdGrpBlkParams<T>& sp, int j, const MultiView<T>& X, const T* w, T* r,
                             const int32_t* dcol, const T* dlt, const int32_t* nz_dev, const int32_t* cols, int nb_cols,
                             T* part, hipStream_t s) {
    if (X.bits)
        return multi_fused_launch<T, SnpOnesAcc<T>>(SnpOnesAcc<T>{X.bits, X.ldb, X.impute, int64_t(X.icpt)}, sp, j, X, w, r, dcol,
                                                    dlt, nz_dev, cols, nb_cols, part, s);
    return multi_fused_launch<T, DenseOnesAcc<T>>(DenseOnesAcc<T>{X.X, X.ld, X.ones, int64_t(X.icpt)}, sp, j, X, w, r, dcol, dlt,
                                                  nz_dev, cols, nb_cols, part, s);
}

template <class T>
void launch_multi_expand_cross(const T* G, int64_t ldg, const int32_t* slot_r, const int32_t* resp_r, int nr,
                               const int32_t* slot_c, const int32_t* resp_c, int nc, T* C, int64_t ldc, hipStream_t s) {
    if (nr <= 0 || nc <= 0) return;
    hipLaunchKernelGGL((multi_expand_cross_kernel<T>), dim3((unsigned)((nr + 63) / 64), (unsigned)nc), dim3(64), 0, s, G, ldg,
                       slot_r, resp_r, nr, slot_c, resp_c, nc, C, ldc);
}

template <class T>
void launch_multi_axpy_cols(const MultiView<T>& X, const int32_t* cols, const T* coef, const int32_t* cnt_dev, T sign,
                            T* out, hipStream_t s) {
    const dim3 grid((unsigned)((X.nb + 255) / 256));
    if (X.bits) {
        SnpOnesAcc<T> acc{X.bits, X.ldb, X.impute, int64_t(X.icpt)};
        hipLaunchKernelGGL((multi_axpy_kernel<T, SnpOnesAcc<T>>), grid, dim3(256), 0, s, acc, X.nb, int(X.K), cols, coef, cnt_dev,
                           sign, out);
        return;
    }
    DenseOnesAcc<T> acc{X.X, X.ld, X.ones, int64_t(X.icpt)};
    hipLaunchKernelGGL((multi_axpy_kernel<T, DenseOnesAcc<T>>), grid, dim3(256), 0, s, acc, X.nb, int(X.K), cols, coef, cnt_dev,
                       sign, out);
}

template <class T>
void launch_multi_expand(const T* C, int64_t ldc, const int32_t* slot, const int32_t* resp, int nv, int lsel, T* D,
                         int64_t ldd, hipStream_t s) {
    if (nv <= 0) return;
    hipLaunchKernelGGL((multi_expand_kernel<T>), dim3((unsigned)((nv + 63) / 64), (unsigned)nv), dim3(64), 0, s, C, ldc,
                       slot, resp, nv, lsel, D, ldd);
}

void launch_multi_block_lists(const int32_t* cols, int nv, int K, int32_t* ulist, int32_t* slot, int32_t* resp,
                              hipStream_t s) {
    if (nv <= 0) return;
    hipLaunchKernelGGL(multi_block_lists_kernel, dim3(1), dim3(MAXB), 0, s, cols, nv, K, ulist, slot, resp);
}

template <class T>
void launch_batch_pick(const T* src, int64_t nfeat, int K, int slot, const T* sub_scale, const T* sub_vec, T* out,
                       hipStream_t s) {
    hipLaunchKernelGGL((batch_pick_kernel<T>), dim3((unsigned)((nfeat + 255) / 256)), dim3(256), 0, s, src, nfeat, K, slot,
                       sub_scale, sub_vec, out);
}
template <class T>
void launch_multi_to_major(const T* src, int64_t nb, int K, T* dst, hipStream_t s) {
    const int64_t tot = nb * K;
    if (tot <= 0) return;
    hipLaunchKernelGGL((multi_to_major_kernel<T>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, src, nb, K, dst);
}
template <class T>
void launch_multi_from_major(const T* src, int64_t nb, int K, T* dst, hipStream_t s) {
    const int64_t tot = nb * K;
    if (tot <= 0) return;
    hipLaunchKernelGGL((multi_from_major_kernel<T>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, src, nb, K, dst);
}

#define INST(T)                                                                                                        \
    template int64_t multi_sweep_work_elems<T>(const MultiView<T>&);                                                   \
    template void launch_multi_sweep<T>(const MultiView<T>&, const T*, T*, T*, hipStream_t);                           \
    template void launch_multi_sweep_snp<T>(const SnpView&, const T*, int, const T*, T*, T*, hipStream_t);             \
    template int launch_multi_panel_step<T>(const MultiView<T>&, const T*, T*, const int32_t*, const T*, const int32_t*, \
                                            const int32_t*, int, T*, hipStream_t);                                     \
    template int launch_multi_panel_fused<T>(const CdGrpBlkParams<T>&, int, const MultiView<T>&, const T*, T*,         \
                                             const int32_t*, const T*, const int32_t*, const int32_t*, int, T*,        \
                                             hipStream_t);                                                             \
    template void launch_multi_expand_cross<T>(const T*, int64_t, const int32_t*, const int32_t*, int, const int32_t*, \
                                               const int32_t*, int, T*, int64_t, hipStream_t);                         \
    template void launch_multi_axpy_cols<T>(const MultiView<T>&, const int32_t*, const T*, const int32_t*, T, T*,      \
                                            hipStream_t);                                                              \
    template void launch_multi_expand<T>(const T*, int64_t, const int32_t*, const int32_t*, int, int, T*, int64_t,     \
                                         hipStream_t);                                                                 \
    template void launch_batch_pick<T>(const T*, int64_t, int, int, const T*, const T*, T*, hipStream_t);              \
    template void launch_multi_to_major<T>(const T*, int64_t, int, T*, hipStream_t);                                   \
    template void launch_multi_from_major<T>(const T*, int64_t, int, T*, hipStream_t);
INST(double)
INST(float)
#undef INST

} // namespace ahip
