// kernels_sparse.hip — a sparse design kept sparse in HBM (reference matrix_naive_sparse.ipp: MatrixNaiveSparse walks the CSC
// arrays per operation).  The design holds the matrix twice, column-compressed (CSC: gradients, Gram rows) and row-compressed
// (CSR: residual updates, X beta), 12 bytes per stored entry each way: every operation below is one stream over the entries
// it needs, HBM-bound on nnz, with a fixed summation order (no atomics on the data path of a solve).
//
//   sweep      out[k] = sum_t val[t](^2) v[row[t]]          full design: a tile-major copy of the entries, tiles of v in LDS;
//                                                           column lists: 64 / 16 / 4 lanes per (column, row block) segment,
//                                                           row blocks keep the gathered slice of v in L2
//   Gram       C[a, b] = sum_i w_i x_ia x_ib - xm_a xm_b    up to 8 columns b scattered (times w) into a dense (n, 8) slab,
//                                                           then one 8-wide sweep over the columns a
//   axpy       out[i] += sign sum_m coef[m] x[i, cols[m]]   coefficients scattered into a dense p-vector, one CSR pass
//   sp_tmul    out[l, i] = sum_j V[l, j] x[i, j]            8 rows l per CSR pass over a dense (p, 8) slab
//
// Standardized view (CscView::center / inv_scale non-null; reference matrix_naive_standardize.ipp over a sparse matrix): the
// design is  x~_ij = (x_ij - c_j) / s_j  with the stored entries untouched — centring would fill every cell.  Every operation
// is the raw one plus a rank-one correction in its epilogue:
//   sweep      (x_j . v - c_j sum(v)) / s_j ;   squares: (sum v x^2 - 2 c_j sum v x + c_j^2 sum(v)) / s_j^2
//   Gram       (C_ab - c_a m_b - c_b m_a + c_a c_b W) / (s_a s_b),  m = X^T w (raw),  W = sum(w)
//   axpy       coefficients scaled by 1 / s_j on their way into the p-vector, and  kappa = sum_m coef_m c_m / s_m  off every row
#include <stdexcept>
#include <string>
#include "kernels.hpp"
#include "wavered.hpp"

#include <algorithm>

namespace ahip {

namespace {
constexpr int KB = 8; // right-hand sides per pass of the slab kernels

// Row blocks (CscView::bptr): blockIdx.y = row block.  Workgroups are dispatched x-fastest, so the whole chip works on one
// block of rows at a time and the slice of v it gathers from (rb rows, about 1 MB) stays in every XCD's L2; without blocks
// every stored entry pulls its own 128-byte line of an n-vector that does not fit there (1M x 100k, 1e8 entries: 1.06 ms per
// sweep unblocked, 0.68 ms with 1 MB slices; 0.70 / 0.71 / 0.77 ms at 0.5 / 2 / 4 MB).
// LPC lanes per (column, row block) segment, 64 / LPC segments per wavefront, chosen from the mean segment length (measured on
// the same design: 0.65 / 0.65 / 0.68 ms at 64 / 16 / 4 lanes — the kernel is bound by the rate at which L2 answers one gather
// per stored entry, about 150 G/s, not by wavefront launches or the 1.2 GB stream).
template <class T, bool SQ, int LPC>
__global__ __launch_bounds__(256) void csc_sweep_kernel(CscView<T> X, const T* __restrict__ v, T* __restrict__ out, int64_t c0,
                                                         int64_t ncols, const int32_t* __restrict__ cols,
                                                         const T* __restrict__ sub_scale, const T* __restrict__ sub_vec,
                                                         T* __restrict__ part) {
    constexpr int CPW = 64 / LPC;
    const int lane = threadIdx.x & 63, sub = lane % LPC;
    const int64_t k = (int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6)) * CPW + lane / LPC;
    const bool valid = k < ncols;
    int64_t c = 0, b = 0, e = 0;
    if (valid) {
        c = cols ? int64_t(cols[k]) : c0 + k;
        if (X.nb > 1) {
            const int64_t* bp = X.bptr + c * (X.nb + 1) + blockIdx.y;
            b = bp[0];
            e = bp[1];
        } else {
            b = X.cptr[c];
            e = X.cptr[c + 1];
        }
    }
    T acc = T(0);
#pragma unroll 4
    for (int64_t t = b + sub; t < e; t += LPC) {
        T x = X.cval[t];
        if (SQ) x *= x;
        acc = fma(x, v[X.cidx[t]], acc);
    }
    if constexpr (LPC == 64) {
        acc = wave_sum64(acc);
    } else { // fixed-order sum inside the segment's lanes (a quad, or a row of 16)
        acc += pdpp<0xB1>(acc); // quad_perm [1,0,3,2]
        acc += pdpp<0x4E>(acc); // quad_perm [2,3,0,1]
        if constexpr (LPC == 16) {
            acc += pdpp<0x141>(acc); // row_half_mirror
            acc += pdpp<0x140>(acc); // row_mirror
        }
    }
    if (valid && sub == 0) {
        if (X.nb > 1) part[int64_t(blockIdx.y) * ncols + k] = acc;
        else out[k] = sub_vec ? acc - sub_scale[0] * sub_vec[c] : acc;
    }
}
// out[k] = sum over the row blocks, in block order, minus the centring term
template <class T>
__global__ __launch_bounds__(256) void csc_sweep_reduce_kernel(const T* __restrict__ part, int nb, T* __restrict__ out, int64_t c0,
                                                                int64_t ncols, const int32_t* __restrict__ cols,
                                                                const T* __restrict__ sub_scale, const T* __restrict__ sub_vec) {
    const int64_t k = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (k >= ncols) return;
    T acc = part[k];
    for (int b = 1; b < nb; ++b) acc += part[int64_t(b) * ncols + k];
    const int64_t c = cols ? int64_t(cols[k]) : c0 + k;
    out[k] = sub_vec ? acc - sub_scale[0] * sub_vec[c] : acc;
}
// bptr[c * (nb + 1) + b] = first stored entry of column c with row >= b * rb  (b = nb: the column's end)
__global__ __launch_bounds__(256) void csc_block_ptr_kernel(const int64_t* __restrict__ cptr, const int32_t* __restrict__ cidx,
                                                             int64_t p, int nb, int64_t rb, int64_t* __restrict__ bptr) {
    const int64_t id = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (id >= p * (nb + 1)) return;
    const int64_t c = id / (nb + 1);
    const int b = int(id - c * (nb + 1));
    int64_t lo = cptr[c], hi = cptr[c + 1];
    const int64_t key = int64_t(b) * rb;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (cidx[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    bptr[id] = lo;
}

// Full sweep from the tile-major copy: a workgroup keeps one tile of v (th rows, 128 KB) in LDS and streams the tile's entries,
// which stand together by column: the gather that bounds csc_sweep_kernel (one L2 request per stored entry) becomes an LDS
// read, and the stream is perfectly coalesced.  16 lanes per (tile, column) segment, 4 segments per wavefront step; the
// per-tile partial sums are added in tile order by csc_sweep_reduce_kernel.
template <class T, bool SQ, int U>
__global__ __launch_bounds__(1024) void csc_tile_sweep_kernel(CscView<T> X, const T* __restrict__ v, T* __restrict__ part) {
    extern __shared__ unsigned char lds_raw[];
    T* vt = reinterpret_cast<T*>(lds_raw);
    const int t = blockIdx.y;
    const int64_t r0 = int64_t(t) * X.th, rn = (X.n - r0 < X.th) ? X.n - r0 : X.th;
    for (int64_t i = threadIdx.x; i < X.th; i += 1024) vt[i] = i < rn ? v[r0 + i] : T(0);
    __syncthreads();
    const int64_t per = (X.p + gridDim.x - 1) / gridDim.x;
    const int64_t c0 = int64_t(blockIdx.x) * per, c1 = (c0 + per < X.p) ? c0 + per : X.p;
    const int lane = threadIdx.x & 63, sub = lane & 15, q = lane >> 4, w = threadIdx.x >> 6;
    const int64_t* tp = X.tptr + int64_t(t) * (X.p + 1);
    // U groups of 64 columns per round: the loads of all U segments go out together (one wavefront in 16 per SIMD has to keep
    // kilobytes in flight: with one segment at a time a workgroup streamed 6 GB/s)
    for (int64_t cb = c0 + int64_t(w) * 4; cb < c1; cb += 64 * U) { // (uniform per wavefront: the DPP sums below need all lanes)
        int64_t k[U], e[U];
        T acc[U], x[U];
        int r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t c = cb + int64_t(u) * 64 + q;
            int64_t b = 0;
            e[u] = 0;
            if (c < c1) { b = tp[c]; e[u] = tp[c + 1]; }
            k[u] = b + sub;
            acc[u] = T(0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            x[u] = T(0);
            r[u] = 0;
            if (k[u] < e[u]) { x[u] = X.tval[k[u]]; r[u] = X.trow[k[u]]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (SQ) x[u] *= x[u];
            acc[u] = fma(x[u], vt[r[u]], acc[u]); // (x = 0 beyond the segment)
            for (int64_t kk = k[u] + 16; kk < e[u]; kk += 16) { // segments of more than 16 entries
                T y = X.tval[kk];
                if (SQ) y *= y;
                acc[u] = fma(y, vt[X.trow[kk]], acc[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            T a = acc[u];
            a += pdpp<0xB1>(a);  // quad_perm [1,0,3,2]
            a += pdpp<0x4E>(a);  // quad_perm [2,3,0,1]
            a += pdpp<0x141>(a); // row_half_mirror
            a += pdpp<0x140>(a); // row_mirror
            const int64_t c = cb + int64_t(u) * 64 + q;
            if (c < c1 && sub == 0) part[int64_t(t) * X.p + c] = a;
        }
    }
}
// one wavefront per column: its entries, tile by tile, to their tile-major places
template <class T>
__global__ __launch_bounds__(256) void csc_tile_scatter_kernel(const int64_t* __restrict__ colptr, const int32_t* __restrict__ cidx,
                                                                const T* __restrict__ cval, int64_t p, int nt, int64_t th,
                                                                const int64_t* __restrict__ tptr, uint16_t* __restrict__ trow,
                                                                T* __restrict__ tval) {
    const int lane = threadIdx.x & 63;
    const int64_t c = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (c >= p) return;
    for (int t = 0; t < nt; ++t) {
        const int64_t b = colptr[c * (nt + 1) + t], e = colptr[c * (nt + 1) + t + 1];
        const int64_t d = tptr[int64_t(t) * (p + 1) + c];
        for (int64_t k = lane; k < e - b; k += 64) {
            trow[d + k] = uint16_t(int64_t(cidx[b + k]) - int64_t(t) * th);
            tval[d + k] = cval[b + k];
        }
    }
}

// out[0] = sum_i v[i] in a fixed order: 256 workgroups sum a contiguous chunk each (tree over their 256 threads' strided
// partial sums) into out[8 + b], then one workgroup adds the 256 chunk sums.  (A single workgroup over a million values took
// 0.4 ms, as long as a sweep of 1e8 stored entries.)
template <class T>
__global__ __launch_bounds__(256) void vec_sum_kernel(const T* __restrict__ v, int64_t n, T* __restrict__ out, int stage) {
    __shared__ T sh[256];
    T a = T(0);
    if (stage == 0) {
        const int64_t chunk = (n + 255) / 256, b0 = int64_t(blockIdx.x) * chunk, b1 = b0 + chunk < n ? b0 + chunk : n;
        for (int64_t i = b0 + threadIdx.x; i < b1; i += 256) a += v[i];
    } else {
        a = out[8 + threadIdx.x];
    }
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (int(threadIdx.x) < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[stage == 0 ? 8 + blockIdx.x : 0] = sh[0];
}
// the standardized view's sweep from the raw dot products (see the header)
template <class T>
__global__ __launch_bounds__(256) void std_sweep_epilogue_kernel(const T* __restrict__ center, const T* __restrict__ inv_scale,
                                                                  const T* __restrict__ raw, const T* __restrict__ raw_plain,
                                                                  const T* __restrict__ vsum, bool square, T* __restrict__ out,
                                                                  int64_t c0, int64_t ncols, const int32_t* __restrict__ cols,
                                                                  const T* __restrict__ sub_scale, const T* __restrict__ sub_vec) {
    const int64_t k = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (k >= ncols) return;
    const int64_t c = cols ? int64_t(cols[k]) : c0 + k;
    const T ce = center[c], is = inv_scale[c], s0 = vsum[0];
    T val;
    if (square) val = ((raw[k] - T(2) * ce * raw_plain[k]) + ce * ce * s0) * (is * is);
    else val = (raw[k] - ce * s0) * is;
    out[k] = sub_vec ? val - sub_scale[0] * sub_vec[c] : val;
}

// slab[row * KB + kb] = w[row] * x[row, ncols[b0 + kb]]   (or 0 with CLEAR: puts the slab back to all zeros)
template <class T, bool CLEAR>
__global__ __launch_bounds__(256) void csc_slab_kernel(CscView<T> X, const T* __restrict__ w, const int32_t* __restrict__ ncols,
                                                        int32_t nb, T* __restrict__ slab) {
    const int kb = blockIdx.y;
    if (kb >= nb) return;
    const int64_t c = ncols[kb];
    const int64_t b = X.cptr[c], e = X.cptr[c + 1];
    for (int64_t t = b + int64_t(blockIdx.x) * 256 + threadIdx.x; t < e; t += int64_t(gridDim.x) * 256) {
        const int64_t r = X.cidx[t];
        slab[r * KB + kb] = CLEAR ? T(0) : w[r] * X.cval[t];
    }
}

// one wavefront per row column a of the Gram panel: KB dots against the slab
template <class T>
__global__ __launch_bounds__(256) void csc_gram_kernel(CscView<T> X, const T* __restrict__ slab, const int32_t* __restrict__ mcols,
                                                        int32_t M, int32_t m_pos0, const int32_t* __restrict__ ncols, int32_t nb,
                                                        int32_t n_pos, int32_t n_lo, int32_t n_hi, const T* __restrict__ xm,
                                                        bool center, T* __restrict__ C, int64_t ldc, const T* __restrict__ mM,
                                                        const T* __restrict__ mN, const T* __restrict__ wsum) {
    const int lane = threadIdx.x & 63;
    const int64_t a = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (a >= M) return;
    const int64_t c = mcols[a];
    const int64_t b = X.cptr[c], e = X.cptr[c + 1];
    T acc[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) acc[k] = T(0);
    for (int64_t t = b + lane; t < e; t += 64) {
        const T x = X.cval[t];
        const T* s = slab + int64_t(X.cidx[t]) * KB;
#pragma unroll
        for (int k = 0; k < KB; ++k) acc[k] = fma(x, s[k], acc[k]);
    }
    const T tot = reduce8(acc, lane); // lane l: total of right-hand side l & 7
    if (lane < nb) {
        const int64_t rp = m_pos0 + a, cp = n_pos + lane;
        const int64_t cb = ncols[lane];
        T raw = tot;
        if (X.center) { // standardized view: rank-one corrections from the raw weighted column sums
            const T ca = X.center[c], cbv = X.center[cb];
            raw = (((tot - ca * mN[lane]) - cbv * mM[a]) + ca * cbv * wsum[0]) * (X.inv_scale[c] * X.inv_scale[cb]);
        }
        const T val = center ? raw - xm[c] * xm[cb] : raw;
        // Entries whose mirror image is itself computed by this call (both positions inside both ranges) are written by the
        // pair below the diagonal only, to both places: the panel is exactly symmetric and no entry is written twice.
        const bool both = rp >= n_lo && rp < n_hi && cp >= m_pos0 && cp < int64_t(m_pos0) + M;
        if (!both || rp >= cp) {
            C[rp + cp * ldc] = val;
            C[cp + rp * ldc] = val;
        }
    }
}

template <class T>
__global__ __launch_bounds__(256) void vec_scatter_kernel(const int32_t* __restrict__ cols, const T* __restrict__ coef,
                                                           const int32_t* __restrict__ count_dev, int32_t count, bool clear,
                                                           const T* __restrict__ inv_scale, T* __restrict__ delta) {
    const int32_t cnt = count_dev ? count_dev[0] : count;
    for (int32_t m = blockIdx.x * 256 + threadIdx.x; m < cnt; m += gridDim.x * 256)
        delta[cols[m]] = clear ? T(0) : (inv_scale ? coef[m] * inv_scale[cols[m]] : coef[m]);
}
// kappa[0] = sum_m coef[m] center[cols[m]] inv_scale[cols[m]]  (one workgroup, fixed order)
template <class T>
__global__ __launch_bounds__(256) void std_kappa_kernel(CscView<T> X, const int32_t* __restrict__ cols, const T* __restrict__ coef,
                                                         const int32_t* __restrict__ count_dev, int32_t count, T* __restrict__ kappa) {
    __shared__ T sh[256];
    const int32_t cnt = count_dev ? count_dev[0] : count;
    T a = T(0);
    for (int32_t m = threadIdx.x; m < cnt; m += 256) a = fma(coef[m] * X.inv_scale[cols[m]], X.center[cols[m]], a);
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (int(threadIdx.x) < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) kappa[0] = sh[0];
}

// out[i] += sign * sum_t rval[t] delta[rcol[t]]: 8 lanes per row.  Almost every delta is zero (a fit changes the coefficients
// of its active set), and one gather per stored entry is what bounds a pass (see the sweep).  BITS: the workgroup first marks
// the changed columns in a bitmap in LDS (p bits) and only entries of marked columns load their value and gather their
// coefficient: the pass streams the column indices (4 of the 12 bytes per entry) and nothing else.  Skipped terms are exact
// zeros, so the sums are those of the plain pass bit for bit.
template <class T, bool BITS>
__global__ __launch_bounds__(512) void csr_axpy_kernel(CscView<T> X, const T* __restrict__ delta, const int32_t* __restrict__ cols,
                                                        const int32_t* __restrict__ count_dev, int32_t count, T sign,
                                                        const T* __restrict__ kappa, T* __restrict__ out) {
    extern __shared__ uint32_t bm[];
    const int32_t cnt = count_dev ? count_dev[0] : count;
    if (cnt <= 0) return;
    if constexpr (BITS) {
        const int64_t words = (X.p + 31) >> 5;
        for (int64_t wd = threadIdx.x; wd < words; wd += 512) bm[wd] = 0u;
        __syncthreads();
        for (int32_t m = threadIdx.x; m < cnt; m += 512) atomicOr(&bm[cols[m] >> 5], 1u << (cols[m] & 31));
        __syncthreads();
    }
    const int sub = threadIdx.x & 7;
    const int64_t groups = (X.n + 63) / 64; // 64 rows per workgroup step
    for (int64_t g = blockIdx.x; g < groups; g += gridDim.x) {
        const int64_t i = g * 64 + (threadIdx.x >> 3);
        T acc = T(0);
        if (i < X.n) {
            const int64_t b = X.rptr[i], e = X.rptr[i + 1];
#pragma unroll 4
            for (int64_t t = b + sub; t < e; t += 8) {
                const int32_t c = X.rcol[t];
                if constexpr (BITS) {
                    if ((bm[c >> 5] >> (c & 31)) & 1u) acc = fma(X.rval[t], delta[c], acc);
                } else {
                    acc = fma(X.rval[t], delta[c], acc);
                }
            }
        }
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 4, 64);
        if (kappa) acc -= kappa[0];
        if (i < X.n && sub == 0 && acc != T(0)) out[i] = fma(sign, acc, out[i]);
    }
}

// slab[j * KB + kb] = V[l0 + kb, j] for the stored entries of rows l0 .. l0 + nl of a host-made CSR (or 0 with CLEAR)
template <class T, bool CLEAR>
__global__ __launch_bounds__(256) void csr_rows_slab_kernel(const int64_t* __restrict__ indptr, const int64_t* __restrict__ indices,
                                                             const T* __restrict__ values, int nl, const T* __restrict__ inv_scale,
                                                             T* __restrict__ slab) {
    const int kb = blockIdx.y;
    if (kb >= nl) return;
    const int64_t b = indptr[kb], e = indptr[kb + 1];
    for (int64_t t = b + int64_t(blockIdx.x) * 256 + threadIdx.x; t < e; t += int64_t(gridDim.x) * 256)
        slab[indices[t] * KB + kb] = CLEAR ? T(0) : (inv_scale ? values[t] * inv_scale[indices[t]] : values[t]);
}
// kappa[kb] = sum_j V[l0 + kb, j] center[j] inv_scale[j]  (one workgroup per row, fixed order)
template <class T>
__global__ __launch_bounds__(256) void std_rows_kappa_kernel(CscView<T> X, const int64_t* __restrict__ indptr,
                                                              const int64_t* __restrict__ indices, const T* __restrict__ values,
                                                              T* __restrict__ kappa) {
    __shared__ T sh[256];
    const int kb = blockIdx.x;
    const int64_t b = indptr[kb], e = indptr[kb + 1];
    T a = T(0);
    for (int64_t t = b + threadIdx.x; t < e; t += 256) a = fma(values[t] * X.inv_scale[indices[t]], X.center[indices[t]], a);
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (int(threadIdx.x) < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) kappa[kb] = sh[0];
}

// out[kb * n + i] = sum_t rval[t] slab[rcol[t] * KB + kb]: 8 lanes per row, KB results each
template <class T>
__global__ __launch_bounds__(256) void csr_tmul_kernel(CscView<T> X, const T* __restrict__ slab, int nl, const T* __restrict__ kappa,
                                                        T* __restrict__ out) {
    const int sub = threadIdx.x & 7;
    const int64_t i = (int64_t(blockIdx.x) * 256 + threadIdx.x) >> 3;
    T acc[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) acc[k] = T(0);
    if (i < X.n) {
        const int64_t b = X.rptr[i], e = X.rptr[i + 1];
        for (int64_t t = b + sub; t < e; t += 8) {
            const T x = X.rval[t];
            const T* s = slab + int64_t(X.rcol[t]) * KB;
#pragma unroll
            for (int k = 0; k < KB; ++k) acc[k] = fma(x, s[k], acc[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < KB; ++k) {
        acc[k] += __shfl_xor(acc[k], 1, 64);
        acc[k] += __shfl_xor(acc[k], 2, 64);
        acc[k] += __shfl_xor(acc[k], 4, 64);
    }
    if (i < X.n && sub == 0)
        for (int k = 0; k < nl; ++k) out[int64_t(k) * X.n + i] = kappa ? acc[k] - kappa[k] : acc[k];
}

inline unsigned blocks_for(int64_t items, int per_block) {
    const int64_t g = (items + per_block - 1) / per_block;
    return unsigned(std::max<int64_t>(1, std::min<int64_t>(g, int64_t(1) << 30)));
}
} // namespace

// ---- standardized view over a dense or 2-bit design (solver_screen.hpp composes the base kernels with these) ----------------
namespace {
// C[a, b] (a < M rows by screen position, b in [pos0, pos0 + N)) and its mirror image, from the raw X^T W X the base Gram kernel
// left there: (C_ab - c_a m_b - c_b m_a + c_a c_b W) / (s_a s_b) - xm_a xm_b.  Pairs inside the new x new square are handled by
// the one below the diagonal only.
template <class T>
__global__ __launch_bounds__(256) void std_gram_fix_kernel(const T* __restrict__ center, const T* __restrict__ inv_scale,
                                                            T* __restrict__ C, int64_t ldc, int32_t M, int32_t pos0, int32_t N,
                                                            const int32_t* __restrict__ vcol, const T* __restrict__ m,
                                                            const T* __restrict__ wsum, const T* __restrict__ xm, bool centered) {
    const int64_t id = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (id >= int64_t(M) * N) return;
    const int64_t a = id % M, b = pos0 + id / M;
    if (a >= pos0 && a < int64_t(pos0) + N && a < b) return;
    const int64_t ca_ = vcol[a], cb_ = vcol[b];
    const T ca = center[ca_], cb = center[cb_];
    T val = (((C[a + b * ldc] - ca * m[b]) - cb * m[a]) + ca * cb * wsum[0]) * (inv_scale[ca_] * inv_scale[cb_]);
    if (centered) val -= xm[ca_] * xm[cb_];
    C[a + b * ldc] = val;
    C[b + a * ldc] = val;
}
// coef2[m] = coef[m] / s_col(m);  kappa[0] = sum_m coef[m] c_col(m) / s_col(m)   (one workgroup, fixed order)
template <class T>
__global__ __launch_bounds__(256) void std_scale_coef_kernel(const T* __restrict__ center, const T* __restrict__ inv_scale,
                                                              const int32_t* __restrict__ cols, const T* __restrict__ coef,
                                                              const int32_t* __restrict__ count_dev, int32_t count,
                                                              T* __restrict__ coef2, T* __restrict__ kappa) {
    __shared__ T sh[256];
    const int32_t cnt = count_dev ? count_dev[0] : count;
    T a = T(0);
    for (int32_t i = threadIdx.x; i < cnt; i += 256) {
        const T sc = coef[i] * inv_scale[cols[i]];
        coef2[i] = sc;
        a = fma(sc, center[cols[i]], a);
    }
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (int(threadIdx.x) < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) kappa[0] = sh[0];
}
template <class T>
__global__ __launch_bounds__(256) void vec_shift_kernel(T* __restrict__ out, int64_t n, const T* __restrict__ kappa, T sign,
                                                         const int32_t* __restrict__ count_dev) {
    if (count_dev && count_dev[0] <= 0) return;
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) out[i] -= sign * kappa[0];
}
} // namespace

namespace {
// Panel engine on a standardized view (Solver::panel_step / gram_block*): the gradient of a block from the raw partial sums,
//   g~_c = (raw_c - center_c * rsum) * inv_scale_c - rsum * xbar~_c        (rsum = sum_i w_i r_i, tracked by the solves)
// and a diagonal block from the raw X_b' W X_b of the base design (m_c = sum_i w_i x_ic = xbar~_c / inv_scale_c + center_c W),
//   D~_ab = (G_ab - center_a m_b - m_a center_b + center_a center_b W) inv_scale_a inv_scale_b - xbar~_a xbar~_b
// (the last term of either only with an intercept).
template <class T>
__global__ __launch_bounds__(128) void std_fix_gblk_kernel(T* __restrict__ gblk, const int32_t* __restrict__ cols, int nb,
                                                           const T* __restrict__ center, const T* __restrict__ inv_scale,
                                                           const T* __restrict__ rsum, const T* __restrict__ xm_view) {
    const int a = threadIdx.x;
    if (a >= nb) return;
    const int32_t c = cols[a];
    const T rs = rsum[0];
    T g = (gblk[a] - center[c] * rs) * inv_scale[c];
    if (xm_view) g -= rs * xm_view[c];
    gblk[a] = g;
}
template <class T>
__global__ __launch_bounds__(256) void std_block_fix_kernel(T* __restrict__ D0, SyrkBatch sb, const int32_t* __restrict__ cols_base,
                                                            int ldb, const T* __restrict__ center, const T* __restrict__ inv_scale,
                                                            const T* __restrict__ xm_view, const T* __restrict__ wsum, int centered) {
    const int y = blockIdx.y, nb = sb.nb[y];
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= nb * nb) return;
    const int a = t % nb, b = t / nb;
    if (a < b) return; // (lower triangle incl. the diagonal, mirrored)
    const int32_t* cols = cols_base + sb.off[y];
    const int32_t ca = cols[a], cb = cols[b];
    const T W = wsum[0];
    const T ia = inv_scale[ca], ib = inv_scale[cb], ea = center[ca], eb = center[cb];
    const T ma = xm_view[ca] / ia + ea * W, mb = xm_view[cb] / ib + eb * W;
    T* D = D0 + sb.dst[y];
    T v = (((D[a + int64_t(b) * ldb] - ea * mb) - ma * eb) + ea * eb * W) * (ia * ib);
    if (centered) v -= xm_view[ca] * xm_view[cb];
    D[a + int64_t(b) * ldb] = v;
    D[b + int64_t(a) * ldb] = v;
}
} // namespace
template <class T>
void launch_std_fix_gblk(T* gblk, const int32_t* cols, int nb, const T* center, const T* inv_scale, const T* rsum_dev,
                         const T* xm_view_or_null, hipStream_t s) {
    if (nb <= 0) return;
    hipLaunchKernelGGL((std_fix_gblk_kernel<T>), dim3(1), dim3(128), 0, s, gblk, cols, nb, center, inv_scale, rsum_dev, xm_view_or_null);
}
template <class T>
void launch_std_block_fix(T* D0, const SyrkBatch& sb, const int32_t* cols_base, int ldb, const T* center, const T* inv_scale,
                          const T* xm_view, const T* wsum_dev, bool centered, hipStream_t s) {
    if (sb.count <= 0) return;
    int mx = 0;
    for (int y = 0; y < sb.count; ++y) mx = std::max(mx, int(sb.nb[y]));
    if (mx <= 0) return;
    hipLaunchKernelGGL((std_block_fix_kernel<T>), dim3(unsigned((mx * mx + 255) / 256), unsigned(sb.count)), dim3(256), 0, s, D0, sb,
                       cols_base, ldb, center, inv_scale, xm_view, wsum_dev, centered ? 1 : 0);
}

template <class T>
void launch_vec_sum(const T* v, int64_t n, T* out, hipStream_t s) {
    hipLaunchKernelGGL((vec_sum_kernel<T>), dim3(256), dim3(256), 0, s, v, n, out, 0);
    hipLaunchKernelGGL((vec_sum_kernel<T>), dim3(1), dim3(256), 0, s, v, n, out, 1);
}
template <class T>
void launch_std_sweep_epilogue(const T* center, const T* inv_scale, const T* raw, const T* raw_plain, const T* vsum, bool square,
                               T* out, int64_t c0, int64_t ncols, const int32_t* cols, const T* sub_scale, const T* sub_vec,
                               hipStream_t s) {
    if (ncols <= 0) return;
    hipLaunchKernelGGL((std_sweep_epilogue_kernel<T>), dim3(blocks_for(ncols, 256)), dim3(256), 0, s, center, inv_scale, raw, raw_plain,
                       vsum, square, out, c0, ncols, cols, sub_scale, sub_vec);
}
template <class T>
void launch_std_gram_fix(const T* center, const T* inv_scale, T* C, int64_t ldc, int32_t M, int32_t pos0, int32_t N,
                         const int32_t* vcol, const T* m, const T* wsum, const T* xm, bool centered, hipStream_t s) {
    if (M <= 0 || N <= 0) return;
    hipLaunchKernelGGL((std_gram_fix_kernel<T>), dim3(blocks_for(int64_t(M) * N, 256)), dim3(256), 0, s, center, inv_scale, C, ldc, M,
                       pos0, N, vcol, m, wsum, xm, centered);
}
template <class T>
void launch_std_scale_coef(const T* center, const T* inv_scale, const int32_t* cols, const T* coef, const int32_t* count_dev,
                           int32_t count, T* coef2, T* kappa, hipStream_t s) {
    hipLaunchKernelGGL((std_scale_coef_kernel<T>), dim3(1), dim3(256), 0, s, center, inv_scale, cols, coef, count_dev, count, coef2, kappa);
}
template <class T>
void launch_vec_shift(T* out, int64_t n, const T* kappa, T sign, const int32_t* count_dev, hipStream_t s) {
    hipLaunchKernelGGL((vec_shift_kernel<T>), dim3(blocks_for(n, 256)), dim3(256), 0, s, out, n, kappa, sign, count_dev);
}

// [0, nb * ncols): per-block partials; then two raw vectors of ncols and the scratch of launch_vec_sum (standardized view)
int64_t sweep_work_elems_csc(int nb, int64_t ncols) { return int64_t(std::max(nb, 1)) * ncols + 2 * ncols + kVecSumScratch; }

namespace {
// raw dot products of the stored entries (no centring term of the caller, no standardization) into `dst`
template <class T>
void raw_sweep(const CscView<T>& X, const T* v, T* dst, int64_t c0, int64_t ncols, const int32_t* cols, const T* sub_scale,
               const T* sub_vec, bool square, T* part, hipStream_t s) {
    if (X.nt > 0 && !cols && c0 == 0 && ncols == X.p) { // the whole design: tiles of v in LDS
        static const bool raised = [] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(csc_tile_sweep_kernel<T, true, 8>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, int(kCscTileBytes));
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(csc_tile_sweep_kernel<T, false, 8>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, int(kCscTileBytes));
            return true;
        }();
        (void)raised;
        // about 1024 workgroups in all (one per CU at a time: the tile of v takes 128 of its 160 KB of LDS); measured on 1M x 100k
        // with 1e8 entries: 0.458 / 0.450 / 0.445 / 0.455 ms at 256 / 512 / 1024 / 2048 workgroups, 0.478 / 0.445 / 0.627 ms at
        // U = 4 / 8 / 16 segments in flight per lane group
        const unsigned chunks = unsigned(std::max<int64_t>(1, std::min<int64_t>(1024 / X.nt, (X.p + 255) / 256)));
        const dim3 grid(chunks, unsigned(X.nt));
        (void)hipGetLastError(); // (whatever an earlier call of this thread left behind)
        if (square) hipLaunchKernelGGL((csc_tile_sweep_kernel<T, true, 8>), grid, dim3(1024), size_t(kCscTileBytes), s, X, v, part);
        else hipLaunchKernelGGL((csc_tile_sweep_kernel<T, false, 8>), grid, dim3(1024), size_t(kCscTileBytes), s, X, v, part);
        if (const hipError_t e = hipGetLastError(); e != hipSuccess) // a refused launch must not pass for a sweep (stale partials)
            throw std::runtime_error(std::string("adelie_hip: csc_tile_sweep_kernel launch refused: ") + hipGetErrorString(e));
        hipLaunchKernelGGL((csc_sweep_reduce_kernel<T>), dim3(blocks_for(ncols, 256)), dim3(256), 0, s, part, X.nt, dst, c0, ncols, cols,
                           sub_scale, sub_vec);
        return;
    }
    // lanes per segment from the mean number of stored entries of a (column, row block) segment
    const int64_t seg = X.nnz / std::max<int64_t>(1, X.p * std::max(X.nb, 1));
    const int lpc = seg >= 192 ? 64 : (seg >= 24 ? 16 : 4);
    const dim3 grid(blocks_for(ncols, 4 * (64 / lpc)), unsigned(std::max(X.nb, 1))), wg(256);
#define AHIP_CSC_SWEEP(SQ, LPC) \
    hipLaunchKernelGGL((csc_sweep_kernel<T, SQ, LPC>), grid, wg, 0, s, X, v, dst, c0, ncols, cols, sub_scale, sub_vec, part)
    if (square) {
        if (lpc == 64) AHIP_CSC_SWEEP(true, 64); else if (lpc == 16) AHIP_CSC_SWEEP(true, 16); else AHIP_CSC_SWEEP(true, 4);
    } else {
        if (lpc == 64) AHIP_CSC_SWEEP(false, 64); else if (lpc == 16) AHIP_CSC_SWEEP(false, 16); else AHIP_CSC_SWEEP(false, 4);
    }
#undef AHIP_CSC_SWEEP
    if (X.nb > 1)
        hipLaunchKernelGGL((csc_sweep_reduce_kernel<T>), dim3(blocks_for(ncols, 256)), wg, 0, s, part, X.nb, dst, c0, ncols, cols,
                           sub_scale, sub_vec);
}
} // namespace

template <class T>
void launch_sweep_csc(const CscView<T>& X, const T* v, T* out, int64_t c0, int64_t ncols, const int32_t* cols, const T* sub_scale,
                      const T* sub_vec, bool square, T* work, hipStream_t s) {
    if (ncols <= 0) return;
    if (!X.center) {
        raw_sweep<T>(X, v, out, c0, ncols, cols, sub_scale, sub_vec, square, work, s);
        return;
    }
    T* raw = work + int64_t(std::max({X.nb, X.nt, 1})) * ncols;
    T* raw_plain = raw + ncols;
    T* vsum = raw_plain + ncols;
    raw_sweep<T>(X, v, raw, c0, ncols, cols, nullptr, nullptr, square, work, s);
    if (square) raw_sweep<T>(X, v, raw_plain, c0, ncols, cols, nullptr, nullptr, false, work, s);
    launch_vec_sum<T>(v, X.n, vsum, s);
    launch_std_sweep_epilogue<T>(X.center, X.inv_scale, raw, raw_plain, vsum, square, out, c0, ncols, cols, sub_scale, sub_vec, s);
}

template <class T>
void launch_csc_tile_scatter(const int64_t* colptr, const int32_t* cidx, const T* cval, int64_t p, int nt, int64_t th,
                             const int64_t* tptr, uint16_t* trow, T* tval, hipStream_t s) {
    hipLaunchKernelGGL((csc_tile_scatter_kernel<T>), dim3(blocks_for(p, 4)), dim3(256), 0, s, colptr, cidx, cval, p, nt, th, tptr, trow,
                       tval);
}

void csc_block_layout(int64_t n, size_t value_size, int* nb, int64_t* rb) {
    int64_t r = (int64_t(1) << 20) / int64_t(value_size); // a slice of v of about 1 MB
    while ((n + r - 1) / r > 64) r *= 2;
    *rb = r;
    *nb = int((n + r - 1) / r);
}
void launch_csc_block_ptr(const int64_t* cptr, const int32_t* cidx, int64_t p, int nb, int64_t rb, int64_t* bptr, hipStream_t s) {
    hipLaunchKernelGGL(csc_block_ptr_kernel, dim3(blocks_for(p * (nb + 1), 256)), dim3(256), 0, s, cptr, cidx, p, nb, rb, bptr);
}

// the (n, KB) slab; then (standardized view) the raw weighted sums of the row and the column members, eight scalars and
// the work of their sweeps
int64_t gram_work_elems_csc(int64_t n, int64_t M, int64_t N, int nb) {
    return n * KB + M + N + kVecSumScratch + sweep_work_elems_csc(nb, std::max(M, N));
}

template <class T>
void launch_gram_csc(const CscView<T>& X, const T* w, const int32_t* mcols, int32_t M, int32_t m_pos0, const int32_t* ncols,
                     int32_t N, int32_t n_pos0, const T* xm_by_col, bool center, T* C, int64_t ldc, T* work, hipStream_t s) {
    if (M <= 0 || N <= 0) return;
    (void)hipMemsetAsync(work, 0, size_t(X.n) * KB * sizeof(T), s);
    T *mM = nullptr, *mN = nullptr, *wsum = nullptr;
    if (X.center) {
        mM = work + X.n * KB;
        mN = mM + M;
        wsum = mN + N;
        T* sw = wsum + kVecSumScratch;
        raw_sweep<T>(X, w, mM, 0, M, mcols, nullptr, nullptr, false, sw, s);
        raw_sweep<T>(X, w, mN, 0, N, ncols, nullptr, nullptr, false, sw, s);
        launch_vec_sum<T>(w, X.n, wsum, s);
    }
    // entries per column are not known on the host: a fixed spread of workgroups strides each column
    const unsigned spread = unsigned(std::max<int64_t>(1, std::min<int64_t>(64, (X.nnz / std::max<int64_t>(X.p, 1) + 255) / 256)));
    for (int32_t b0 = 0; b0 < N; b0 += KB) {
        const int32_t nb = std::min<int32_t>(KB, N - b0);
        hipLaunchKernelGGL((csc_slab_kernel<T, false>), dim3(spread, unsigned(nb)), dim3(256), 0, s, X, w, ncols + b0, nb, work);
        hipLaunchKernelGGL((csc_gram_kernel<T>), dim3(blocks_for(M, 4)), dim3(256), 0, s, X, work, mcols, M, m_pos0, ncols + b0, nb,
                           n_pos0 + b0, n_pos0, n_pos0 + N, xm_by_col, center, C, ldc, mM, mN ? mN + b0 : nullptr, wsum);
        hipLaunchKernelGGL((csc_slab_kernel<T, true>), dim3(spread, unsigned(nb)), dim3(256), 0, s, X, w, ncols + b0, nb, work);
    }
}

// `delta_zeroed`: p + 8 elements, the first p all zero on entry and on exit (the rest is scratch)
template <class T>
void launch_axpy_cols_csc(const CscView<T>& X, const int32_t* cols, const T* coef, const int32_t* count_dev, int32_t count, T sign,
                          T* out, T* delta_zeroed, hipStream_t s) {
    if (!count_dev && count <= 0) return;
    const unsigned gs = count_dev ? 64u : blocks_for(count, 256);
    T* kappa = nullptr;
    if (X.center) {
        kappa = delta_zeroed + X.p;
        hipLaunchKernelGGL((std_kappa_kernel<T>), dim3(1), dim3(256), 0, s, X, cols, coef, count_dev, count, kappa);
    }
    hipLaunchKernelGGL((vec_scatter_kernel<T>), dim3(gs), dim3(256), 0, s, cols, coef, count_dev, count, false, X.inv_scale, delta_zeroed);
    const int64_t bm_bytes = ((X.p + 31) >> 5) * 4;
    const unsigned wgs = unsigned(std::min<int64_t>((X.n + 63) / 64, 2048));
    if (bm_bytes <= 128 * 1024) { // the bitmap of the changed columns fits in LDS (p <= 1M columns)
        static const bool raised = [] { // (thread-safe once-init, like raw_sweep: concurrent solves share the launcher)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(csr_axpy_kernel<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      128 * 1024);
            return true;
        }();
        (void)raised;
        (void)hipGetLastError();
        hipLaunchKernelGGL((csr_axpy_kernel<T, true>), dim3(wgs), dim3(512), size_t(bm_bytes), s, X, delta_zeroed, cols, count_dev,
                           count, sign, kappa, out);
        if (const hipError_t e = hipGetLastError(); e != hipSuccess) // (a refused launch would leave the residual stale)
            throw std::runtime_error(std::string("adelie_hip: csr_axpy_kernel launch refused: ") + hipGetErrorString(e));
    } else {
        hipLaunchKernelGGL((csr_axpy_kernel<T, false>), dim3(wgs), dim3(512), 0, s, X, delta_zeroed, cols, count_dev, count, sign,
                           kappa, out);
    }
    hipLaunchKernelGGL((vec_scatter_kernel<T>), dim3(gs), dim3(256), 0, s, cols, coef, count_dev, count, true, X.inv_scale, delta_zeroed);
}

// ---- panel engine over compressed columns (IRLS on a design kept sparse, Solver::run_panel_passes) ------------------------------
// The panel step of kernels_cd_panel.hip streams dense column slices; on compressed columns its two phases are
//   (A) r[row] -= x * delta over the stored entries of the changed columns of the previous block      (csc_panel_update_kernel)
//   (B) part[c] = sum_e x_e w[row_e] r[row_e] for the columns of the next block                       (csc_panel_grad_kernel)
// -- a thousand entries per column instead of a million rows.  Two changed columns of one block may share a row: phase (A) adds
// with hardware f64 atomics, so where that happens the order of the two additions is not fixed and runs agree to rounding, not
// bit for bit (everything else in this library is order-deterministic).  The diagonal blocks X_b' W X_b - xbar xbar' of 64
// visits come from csc_block_gram_kernel: one thread per pair of columns merges the two sorted row lists.
template <class T>
__global__ __launch_bounds__(256) void csc_panel_update_kernel(CscView<T> X, const int32_t* __restrict__ dcol,
                                                               const T* __restrict__ dlt, const int32_t* __restrict__ nz_dev,
                                                               T* __restrict__ r) {
    // one workgroup per changed column (a thousand entries: four rounds of 256 independent atomics instead of sixteen rounds of
    // one wavefront's 64: 12.8 -> 5.8 us per launch, the gradient kernel 18.0 -> 7.2 us; profiles/r05_sparse_rocprof_summary.txt)
    const int nz = nz_dev[0];
    for (int m = blockIdx.x; m < nz; m += gridDim.x) {
        const int32_t c = dcol[m];
        const T d = dlt[m];
        const int64_t e1 = X.cptr[c + 1];
        for (int64_t e = X.cptr[c] + threadIdx.x; e < e1; e += 256) unsafeAtomicAdd(r + X.cidx[e], -X.cval[e] * d);
    }
}
template <class T>
__global__ __launch_bounds__(256) void csc_panel_grad_kernel(CscView<T> X, const T* __restrict__ w, const T* __restrict__ r,
                                                             const int32_t* __restrict__ cols, int nb, const T* __restrict__ rsum,
                                                             const T* __restrict__ xm_by_col, T* __restrict__ gblk) {
    // one workgroup per column of the block; the four wavefronts' sums are combined in a fixed order
    __shared__ T red[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int a = blockIdx.x;
    const int32_t c = cols[a];
    const int64_t e1 = X.cptr[c + 1];
    T acc = T(0);
    for (int64_t e = X.cptr[c] + threadIdx.x; e < e1; e += 256) {
        const int32_t i = X.cidx[e];
        acc = fma(X.cval[e], w[i] * r[i], acc);
    }
    acc = wave_sum64(acc);
    if (lane == 0) red[wv] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const T g = (red[0] + red[1]) + (red[2] + red[3]);
        gblk[a] = xm_by_col ? g - rsum[0] * xm_by_col[c] : g; // (what panel_reduce_kernel does for the dense step)
    }
}
// the whole step: the gradient of the block's columns goes straight to gblk (no slice partials, no reduce launch)
template <class T>
void launch_panel_step_csc(const CscView<T>& X, const T* w, T* r, const int32_t* dcol, const T* dlt, const int32_t* nz_dev,
                           const int32_t* cols, int nb, const T* rsum_dev, const T* xm_by_col, T* gblk, hipStream_t s) {
    hipLaunchKernelGGL((csc_panel_update_kernel<T>), dim3(128), dim3(256), 0, s, X, dcol, dlt, nz_dev, r);
    if (nb > 0)
        hipLaunchKernelGGL((csc_panel_grad_kernel<T>), dim3(unsigned(nb)), dim3(256), 0, s, X, w, r, cols, nb, rsum_dev, xm_by_col,
                           gblk);
}

// diagonal blocks of the panel engine: block y = columns cols_base[sb.off[y] ...] (sb.nb[y] of them), X' W X - xm xm' into
// D0 + sb.dst[y] (leading dimension ldb, both triangles)
// One workgroup per (column b of the block, block y): the stored entries of column b go into an LDS hash table (row ->
// w[row] * value; columns longer than a table load go through it in chunks), then every column a >= b of the block probes it,
// one wavefront per column, lanes over its entries.  2 M LDS probes per block of 64 columns of a thousand entries.
constexpr int HJ_SLOTS = 4096, HJ_CHUNK = 2560; // (load factor <= 0.625)
template <class T>
__global__ __launch_bounds__(256) void csc_block_gram_kernel(CscView<T> X, const T* __restrict__ w, const int32_t* __restrict__ cols_base,
                                                             SyrkBatch sb, const T* __restrict__ xm_by_col, int center,
                                                             T* __restrict__ D0, int ldb) {
    __shared__ int32_t keys[HJ_SLOTS];
    __shared__ T vals[HJ_SLOTS];
    __shared__ T accs[128];
    const int y = blockIdx.y, b = blockIdx.x;
    const int nb = sb.nb[y];
    if (b >= nb) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int32_t* cols = cols_base + sb.off[y];
    const int32_t cb = cols[b];
    const int64_t eb0 = X.cptr[cb], eb1 = X.cptr[cb + 1];
    if (tid < 128) accs[tid] = T(0);
    for (int64_t c0 = eb0; c0 < eb1 || c0 == eb0; c0 += HJ_CHUNK) {
        __syncthreads();
        for (int k = tid; k < HJ_SLOTS; k += 256) keys[k] = -1;
        __syncthreads();
        const int64_t c1 = c0 + HJ_CHUNK < eb1 ? c0 + HJ_CHUNK : eb1;
        for (int64_t e = c0 + tid; e < c1; e += 256) {
            const int32_t row = X.cidx[e];
            unsigned slot = (unsigned(row) * 2654435761u) >> 20;
            while (atomicCAS(&keys[slot], -1, row) != -1) slot = (slot + 1) & (HJ_SLOTS - 1);
            vals[slot] = w[row] * X.cval[e];
        }
        __syncthreads();
        for (int a = b + wv; a < nb; a += 4) {
            const int32_t ca = cols[a];
            const int64_t ea1 = X.cptr[ca + 1];
            T acc = T(0);
            for (int64_t e = X.cptr[ca] + lane; e < ea1; e += 64) {
                const int32_t row = X.cidx[e];
                unsigned slot = (unsigned(row) * 2654435761u) >> 20;
                int32_t k;
                while ((k = keys[slot]) != -1) {
                    if (k == row) { acc = fma(X.cval[e], vals[slot], acc); break; }
                    slot = (slot + 1) & (HJ_SLOTS - 1);
                }
            }
            acc = wave_sum64(acc);
            if (lane == 0) accs[a] += acc; // (column a belongs to this wavefront in every chunk)
        }
        if (eb1 == eb0) break;
    }
    __syncthreads();
    T* D = D0 + sb.dst[y];
    for (int a = b + tid; a < nb; a += 256) {
        T v = accs[a];
        if (center) v -= xm_by_col[cols[a]] * xm_by_col[cb];
        D[a + int64_t(b) * ldb] = v;
        D[b + int64_t(a) * ldb] = v;
    }
}
template <class T>
void launch_block_gram_csc(const CscView<T>& X, const T* w, const int32_t* cols_base, const SyrkBatch& sb, const T* xm_by_col,
                           bool center, T* D0, int ldb, hipStream_t s) {
    if (sb.count <= 0) return;
    int mx = 0;
    for (int y = 0; y < sb.count; ++y) mx = std::max(mx, int(sb.nb[y]));
    if (mx <= 0) return;
    hipLaunchKernelGGL((csc_block_gram_kernel<T>), dim3(unsigned(mx), unsigned(sb.count)), dim3(256), 0, s, X, w, cols_base, sb,
                       xm_by_col, center ? 1 : 0, D0, ldb);
}

int64_t sp_tmul_work_elems_csc(int64_t p) { return p * KB + KB; }

template <class T>
void launch_sp_tmul_csc(const CscView<T>& X, int64_t L, const int64_t* indptr, const int64_t* indices, const T* values, T* out,
                        T* work, hipStream_t s) {
    if (L <= 0) return;
    (void)hipMemsetAsync(work, 0, size_t(X.p) * KB * sizeof(T), s);
    T* kappa = X.center ? work + X.p * KB : nullptr;
    for (int64_t l0 = 0; l0 < L; l0 += KB) {
        const int nl = int(std::min<int64_t>(KB, L - l0));
        if (kappa) hipLaunchKernelGGL((std_rows_kappa_kernel<T>), dim3(unsigned(nl)), dim3(256), 0, s, X, indptr + l0, indices, values, kappa);
        hipLaunchKernelGGL((csr_rows_slab_kernel<T, false>), dim3(16, unsigned(nl)), dim3(256), 0, s, indptr + l0, indices, values, nl,
                           X.inv_scale, work);
        hipLaunchKernelGGL((csr_tmul_kernel<T>), dim3(blocks_for(X.n * 8, 256)), dim3(256), 0, s, X, work, nl, kappa, out + l0 * X.n);
        hipLaunchKernelGGL((csr_rows_slab_kernel<T, true>), dim3(16, unsigned(nl)), dim3(256), 0, s, indptr + l0, indices, values, nl,
                           X.inv_scale, work);
    }
}

#define INST(T)                                                                                                                  \
    template void launch_sweep_csc<T>(const CscView<T>&, const T*, T*, int64_t, int64_t, const int32_t*, const T*, const T*, bool, \
                                      T*, hipStream_t);                                                                              \
    template void launch_gram_csc<T>(const CscView<T>&, const T*, const int32_t*, int32_t, int32_t, const int32_t*, int32_t,      \
                                     int32_t, const T*, bool, T*, int64_t, T*, hipStream_t);                                     \
    template void launch_axpy_cols_csc<T>(const CscView<T>&, const int32_t*, const T*, const int32_t*, int32_t, T, T*, T*,         \
                                          hipStream_t);                                                                          \
    template void launch_panel_step_csc<T>(const CscView<T>&, const T*, T*, const int32_t*, const T*, const int32_t*,             \
                                           const int32_t*, int, const T*, const T*, T*, hipStream_t);                            \
    template void launch_block_gram_csc<T>(const CscView<T>&, const T*, const int32_t*, const SyrkBatch&, const T*, bool, T*,     \
                                           int, hipStream_t);                                                                    \
    template void launch_sp_tmul_csc<T>(const CscView<T>&, int64_t, const int64_t*, const int64_t*, const T*, T*, T*, hipStream_t); \
    template void launch_csc_tile_scatter<T>(const int64_t*, const int32_t*, const T*, int64_t, int, int64_t, const int64_t*,     \
                                             uint16_t*, T*, hipStream_t);                                                       \
    template void launch_vec_sum<T>(const T*, int64_t, T*, hipStream_t);                                                         \
    template void launch_std_sweep_epilogue<T>(const T*, const T*, const T*, const T*, const T*, bool, T*, int64_t, int64_t,     \
                                               const int32_t*, const T*, const T*, hipStream_t);                                 \
    template void launch_std_gram_fix<T>(const T*, const T*, T*, int64_t, int32_t, int32_t, int32_t, const int32_t*, const T*,    \
                                         const T*, const T*, bool, hipStream_t);                                                 \
    template void launch_std_scale_coef<T>(const T*, const T*, const int32_t*, const T*, const int32_t*, int32_t, T*, T*,         \
                                           hipStream_t);                                                                         \
    template void launch_vec_shift<T>(T*, int64_t, const T*, T, const int32_t*, hipStream_t);                                     \
    template void launch_std_fix_gblk<T>(T*, const int32_t*, int, const T*, const T*, const T*, const T*, hipStream_t);           \
    template void launch_std_block_fix<T>(T*, const SyrkBatch&, const int32_t*, int, const T*, const T*, const T*, const T*, bool, \
                                          hipStream_t);
INST(double)
INST(float)
#undef INST

} // namespace ahip
