// kernels_sparse.hip — a sparse design kept sparse in HBM (reference matrix_naive_sparse.ipp: MatrixNaiveSparse walks the CSC
// arrays per operation).  The design holds the matrix twice, column-compressed (CSC: gradients, Gram rows) and row-compressed
// (CSR: residual updates, X beta), 12 bytes per stored entry each way: every operation below is one stream over the entries
// it needs, HBM-bound on nnz, with a fixed summation order (no atomics on the data path of a solve).
//
//   sweep      out[k] = sum_t val[t](^2) v[row[t]]          one wavefront per column, lanes stride the column's entries
//   Gram       C[a, b] = sum_i w_i x_ia x_ib - xm_a xm_b    up to 8 columns b scattered (times w) into a dense (n, 8) slab,
//                                                           then one 8-wide sweep over the columns a
//   axpy       out[i] += sign sum_m coef[m] x[i, cols[m]]   coefficients scattered into a dense p-vector, one CSR pass
//   sp_tmul    out[l, i] = sum_j V[l, j] x[i, j]            8 rows l per CSR pass over a dense (p, 8) slab
#include "kernels.hpp"
#include "wavered.hpp"

#include <algorithm>

namespace ahip {

namespace {
constexpr int KB = 8; // right-hand sides per pass of the slab kernels

template <class T, bool SQ>
__global__ __launch_bounds__(256) void csc_sweep_kernel(CscView<T> X, const T* __restrict__ v, T* __restrict__ out, int64_t c0,
                                                         int64_t ncols, const int32_t* __restrict__ cols,
                                                         const T* __restrict__ sub_scale, const T* __restrict__ sub_vec) {
    const int lane = threadIdx.x & 63;
    const int64_t k = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (k >= ncols) return;
    const int64_t c = cols ? int64_t(cols[k]) : c0 + k;
    const int64_t b = X.cptr[c], e = X.cptr[c + 1];
    T acc = T(0);
    for (int64_t t = b + lane; t < e; t += 64) {
        T x = X.cval[t];
        if (SQ) x *= x;
        acc = fma(x, v[X.cidx[t]], acc);
    }
    acc = wave_sum64(acc);
    if (lane == 0) out[k] = sub_vec ? acc - sub_scale[0] * sub_vec[c] : acc;
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
                                                        bool center, T* __restrict__ C, int64_t ldc) {
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
        const T val = center ? tot - xm[c] * xm[ncols[lane]] : tot;
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
                                                           T* __restrict__ delta) {
    const int32_t cnt = count_dev ? count_dev[0] : count;
    for (int32_t m = blockIdx.x * 256 + threadIdx.x; m < cnt; m += gridDim.x * 256) delta[cols[m]] = clear ? T(0) : coef[m];
}

// out[i] += sign * sum_t rval[t] delta[rcol[t]]: 8 lanes per row
template <class T>
__global__ __launch_bounds__(256) void csr_axpy_kernel(CscView<T> X, const T* __restrict__ delta, const int32_t* __restrict__ count_dev,
                                                        T sign, T* __restrict__ out) {
    if (count_dev && count_dev[0] <= 0) return;
    const int sub = threadIdx.x & 7;
    const int64_t i = (int64_t(blockIdx.x) * 256 + threadIdx.x) >> 3;
    T acc = T(0);
    if (i < X.n) {
        const int64_t b = X.rptr[i], e = X.rptr[i + 1];
        for (int64_t t = b + sub; t < e; t += 8) acc = fma(X.rval[t], delta[X.rcol[t]], acc);
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if (i < X.n && sub == 0 && acc != T(0)) out[i] = fma(sign, acc, out[i]);
}

// slab[j * KB + kb] = V[l0 + kb, j] for the stored entries of rows l0 .. l0 + nl of a host-made CSR (or 0 with CLEAR)
template <class T, bool CLEAR>
__global__ __launch_bounds__(256) void csr_rows_slab_kernel(const int64_t* __restrict__ indptr, const int64_t* __restrict__ indices,
                                                             const T* __restrict__ values, int nl, T* __restrict__ slab) {
    const int kb = blockIdx.y;
    if (kb >= nl) return;
    const int64_t b = indptr[kb], e = indptr[kb + 1];
    for (int64_t t = b + int64_t(blockIdx.x) * 256 + threadIdx.x; t < e; t += int64_t(gridDim.x) * 256)
        slab[indices[t] * KB + kb] = CLEAR ? T(0) : values[t];
}

// out[kb * n + i] = sum_t rval[t] slab[rcol[t] * KB + kb]: 8 lanes per row, KB results each
template <class T>
__global__ __launch_bounds__(256) void csr_tmul_kernel(CscView<T> X, const T* __restrict__ slab, int nl, T* __restrict__ out) {
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
        for (int k = 0; k < nl; ++k) out[int64_t(k) * X.n + i] = acc[k];
}

inline unsigned blocks_for(int64_t items, int per_block) {
    const int64_t g = (items + per_block - 1) / per_block;
    return unsigned(std::max<int64_t>(1, std::min<int64_t>(g, int64_t(1) << 30)));
}
} // namespace

template <class T>
void launch_sweep_csc(const CscView<T>& X, const T* v, T* out, int64_t c0, int64_t ncols, const int32_t* cols, const T* sub_scale,
                      const T* sub_vec, bool square, hipStream_t s) {
    if (ncols <= 0) return;
    const dim3 grid(blocks_for(ncols, 4)), wg(256);
    if (square) hipLaunchKernelGGL((csc_sweep_kernel<T, true>), grid, wg, 0, s, X, v, out, c0, ncols, cols, sub_scale, sub_vec);
    else hipLaunchKernelGGL((csc_sweep_kernel<T, false>), grid, wg, 0, s, X, v, out, c0, ncols, cols, sub_scale, sub_vec);
}

int64_t gram_work_elems_csc(int64_t n) { return n * KB; }

template <class T>
void launch_gram_csc(const CscView<T>& X, const T* w, const int32_t* mcols, int32_t M, int32_t m_pos0, const int32_t* ncols,
                     int32_t N, int32_t n_pos0, const T* xm_by_col, bool center, T* C, int64_t ldc, T* work, hipStream_t s) {
    if (M <= 0 || N <= 0) return;
    (void)hipMemsetAsync(work, 0, size_t(X.n) * KB * sizeof(T), s);
    // entries per column are not known on the host: a fixed spread of workgroups strides each column
    const unsigned spread = unsigned(std::max<int64_t>(1, std::min<int64_t>(64, (X.nnz / std::max<int64_t>(X.p, 1) + 255) / 256)));
    for (int32_t b0 = 0; b0 < N; b0 += KB) {
        const int32_t nb = std::min<int32_t>(KB, N - b0);
        hipLaunchKernelGGL((csc_slab_kernel<T, false>), dim3(spread, unsigned(nb)), dim3(256), 0, s, X, w, ncols + b0, nb, work);
        hipLaunchKernelGGL((csc_gram_kernel<T>), dim3(blocks_for(M, 4)), dim3(256), 0, s, X, work, mcols, M, m_pos0, ncols + b0, nb,
                           n_pos0 + b0, n_pos0, n_pos0 + N, xm_by_col, center, C, ldc);
        hipLaunchKernelGGL((csc_slab_kernel<T, true>), dim3(spread, unsigned(nb)), dim3(256), 0, s, X, w, ncols + b0, nb, work);
    }
}

template <class T>
void launch_axpy_cols_csc(const CscView<T>& X, const int32_t* cols, const T* coef, const int32_t* count_dev, int32_t count, T sign,
                          T* out, T* delta_zeroed, hipStream_t s) {
    if (!count_dev && count <= 0) return;
    const unsigned gs = count_dev ? 64u : blocks_for(count, 256);
    hipLaunchKernelGGL((vec_scatter_kernel<T>), dim3(gs), dim3(256), 0, s, cols, coef, count_dev, count, false, delta_zeroed);
    hipLaunchKernelGGL((csr_axpy_kernel<T>), dim3(blocks_for(X.n * 8, 256)), dim3(256), 0, s, X, delta_zeroed, count_dev, sign, out);
    hipLaunchKernelGGL((vec_scatter_kernel<T>), dim3(gs), dim3(256), 0, s, cols, coef, count_dev, count, true, delta_zeroed);
}

int64_t sp_tmul_work_elems_csc(int64_t p) { return p * KB; }

template <class T>
void launch_sp_tmul_csc(const CscView<T>& X, int64_t L, const int64_t* indptr, const int64_t* indices, const T* values, T* out,
                        T* work, hipStream_t s) {
    if (L <= 0) return;
    (void)hipMemsetAsync(work, 0, size_t(X.p) * KB * sizeof(T), s);
    for (int64_t l0 = 0; l0 < L; l0 += KB) {
        const int nl = int(std::min<int64_t>(KB, L - l0));
        hipLaunchKernelGGL((csr_rows_slab_kernel<T, false>), dim3(16, unsigned(nl)), dim3(256), 0, s, indptr + l0, indices, values, nl, work);
        hipLaunchKernelGGL((csr_tmul_kernel<T>), dim3(blocks_for(X.n * 8, 256)), dim3(256), 0, s, X, work, nl, out + l0 * X.n);
        hipLaunchKernelGGL((csr_rows_slab_kernel<T, true>), dim3(16, unsigned(nl)), dim3(256), 0, s, indptr + l0, indices, values, nl, work);
    }
}

#define INST(T)                                                                                                                  \
    template void launch_sweep_csc<T>(const CscView<T>&, const T*, T*, int64_t, int64_t, const int32_t*, const T*, const T*, bool, \
                                      hipStream_t);                                                                              \
    template void launch_gram_csc<T>(const CscView<T>&, const T*, const int32_t*, int32_t, int32_t, const int32_t*, int32_t,      \
                                     int32_t, const T*, bool, T*, int64_t, T*, hipStream_t);                                     \
    template void launch_axpy_cols_csc<T>(const CscView<T>&, const int32_t*, const T*, const int32_t*, int32_t, T, T*, T*,         \
                                          hipStream_t);                                                                          \
    template void launch_sp_tmul_csc<T>(const CscView<T>&, int64_t, const int64_t*, const int64_t*, const T*, T*, T*, hipStream_t);
INST(double)
INST(float)
#undef INST

} // namespace ahip
