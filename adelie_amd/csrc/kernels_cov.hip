// kernels_cov.hip — the covariance method's matrix A: a dense symmetric (p, p) matrix resident in HBM.
//
// Replaces MatrixCovDense (matrix/matrix_cov_dense.ipp:23-84).  The path solver never calls bmul per coordinate the way the
// reference's pin solver does (solver_gaussian_pin_cov.hpp:107-203): the rows / columns of A that belong to the screen set are
// gathered ONCE per screening step into the screen Gram matrix C = A[S, S] that the Gram coordinate-descent kernels
// (kernels_cd.hip, kernels_cd_lasso.hip, kernels_cd_block*.hip) iterate on — for the naive method that matrix is built by an
// MFMA kernel from X, here it is a gather: 8 bytes read and written per entry, HBM-bound, no flops.
// `tr` != 0: the stored matrix is the transpose of A (a row-major input uploaded as is), A(i, j) = S[j + i * lda].
#include "kernels.hpp"

namespace ahip {

namespace {

template <class T>
__device__ __forceinline__ T a_at(const T* __restrict__ S, int64_t lda, int tr, int64_t i, int64_t j) {
    return tr ? S[j + i * lda] : S[i + j * lda];
}

// C[a, pos0 + b] = A(vcol[a], vcol[pos0 + b]) for a < nv, b < N; mirrored into C[pos0 + b, a] for the old values a < pos0
// (the new x new square is covered from both sides by the threads themselves)
template <class T>
__global__ __launch_bounds__(256) void cov_gather_kernel(const T* __restrict__ S, int64_t lda, int tr,
                                                         const int32_t* __restrict__ vcol, int32_t nv, int32_t pos0,
                                                         int32_t N, T* __restrict__ C, int64_t ldc) {
    const int32_t a = blockIdx.x * blockDim.x + threadIdx.x; // fastest index = row of C = row of A's column: coalesced-ish
    if (a >= nv) return;
    const int64_t ra = int64_t(vcol[a]);
    for (int32_t b = blockIdx.y; b < N; b += gridDim.y) { // grid.y is capped at 65535: stride over the new values
        const T val = a_at(S, lda, tr, ra, int64_t(vcol[pos0 + b]));
        C[a + int64_t(pos0 + b) * ldc] = val;
        if (a < pos0) C[int64_t(pos0 + b) + int64_t(a) * ldc] = val;
    }
}

template <class T>
__global__ __launch_bounds__(256) void cov_bmul_kernel(const T* __restrict__ S, int64_t lda, int tr,
                                                       const int64_t* __restrict__ subset, int64_t ns,
                                                       const int64_t* __restrict__ indices, const T* __restrict__ values,
                                                       int64_t ni, T* __restrict__ out) {
    const int64_t jj = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (jj >= ns) return;
    const int64_t j = subset[jj];
    T acc = 0;
    for (int64_t ii = 0; ii < ni; ++ii) acc += values[ii] * a_at(S, lda, tr, indices[ii], j); // matrix_cov_dense.ipp:31-39
    out[jj] = acc;
}

// out[j] = sum_i values[i] * (column-major input: A(j, indices[i]); row-major input: A(indices[i], j)) — the reference adds
// whole columns resp. rows of its storage (matrix_cov_dense.ipp:52-60); in the stored layout both read S[j + indices[i]*lda]
template <class T>
__global__ __launch_bounds__(256) void cov_mul_kernel(const T* __restrict__ S, int64_t lda, int64_t p,
                                                      const int64_t* __restrict__ indices, const T* __restrict__ values,
                                                      int64_t ni, T* __restrict__ out) {
    const int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (j >= p) return;
    T acc = 0;
    for (int64_t ii = 0; ii < ni; ++ii) acc += values[ii] * S[j + indices[ii] * lda];
    out[j] = acc;
}

// grad = v - sum_k coef[k] * S[:, cols[k]]  for k < *cnt  (update_invariance of the covariance solver,
// solver_gaussian_cov.hpp:392-418: grad = v - A beta over the coefficients of the last fit)
template <class T>
__global__ __launch_bounds__(256) void cov_grad_kernel(const T* __restrict__ S, int64_t lda, int64_t p,
                                                       const T* __restrict__ v, const int32_t* __restrict__ cols,
                                                       const T* __restrict__ coef, const int32_t* __restrict__ cnt,
                                                       T* __restrict__ grad) {
    const int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (j >= p) return;
    const int32_t m = *cnt;
    T acc = 0;
    for (int32_t k = 0; k < m; ++k) acc += coef[k] * S[j + int64_t(cols[k]) * lda];
    grad[j] = v[j] - acc;
}

} // namespace

template <class T>
void launch_cov_gather(const T* S, int64_t lda, int tr, const int32_t* vcol, int32_t nv, int32_t pos0, int32_t N, T* C,
                       int64_t ldc, hipStream_t s) {
    if (nv <= 0 || N <= 0) return;
    hipLaunchKernelGGL((cov_gather_kernel<T>), dim3(unsigned((nv + 255) / 256), unsigned(N < 65535 ? N : 65535)), dim3(256),
                       0, s, S, lda, tr, vcol, nv, pos0, N, C, ldc);
}
template <class T>
void launch_cov_bmul(const T* S, int64_t lda, int tr, const int64_t* subset, int64_t ns, const int64_t* indices,
                     const T* values, int64_t ni, T* out, hipStream_t s) {
    if (ns <= 0) return;
    hipLaunchKernelGGL((cov_bmul_kernel<T>), dim3(unsigned((ns + 255) / 256)), dim3(256), 0, s, S, lda, tr, subset, ns, indices,
                       values, ni, out);
}
template <class T>
void launch_cov_mul(const T* S, int64_t lda, int64_t p, const int64_t* indices, const T* values, int64_t ni, T* out,
                    hipStream_t s) {
    hipLaunchKernelGGL((cov_mul_kernel<T>), dim3(unsigned((p + 255) / 256)), dim3(256), 0, s, S, lda, p, indices, values, ni, out);
}
template <class T>
void launch_cov_grad(const T* S, int64_t lda, int64_t p, const T* v, const int32_t* cols, const T* coef, const int32_t* cnt,
                     T* grad, hipStream_t s) {
    hipLaunchKernelGGL((cov_grad_kernel<T>), dim3(unsigned((p + 255) / 256)), dim3(256), 0, s, S, lda, p, v, cols, coef, cnt, grad);
}

#define INST(T)                                                                                                         \
    template void launch_cov_gather<T>(const T*, int64_t, int, const int32_t*, int32_t, int32_t, int32_t, T*, int64_t,  \
                                       hipStream_t);                                                                    \
    template void launch_cov_bmul<T>(const T*, int64_t, int, const int64_t*, int64_t, const int64_t*, const T*, int64_t, \
                                     T*, hipStream_t);                                                                  \
    template void launch_cov_mul<T>(const T*, int64_t, int64_t, const int64_t*, const T*, int64_t, T*, hipStream_t);    \
    template void launch_cov_grad<T>(const T*, int64_t, int64_t, const T*, const int32_t*, const T*, const int32_t*, T*, \
                                     hipStream_t);
INST(double)
INST(float)
#undef INST

} // namespace ahip
