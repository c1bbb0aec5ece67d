// kernels_glm.hip — fused elementwise + reduction kernels of the IRLS (proximal Newton) wrapper.
//
// GLM members restated from glm/glm_gaussian.ipp:15-63, glm/glm_binomial.ipp:37-99, glm/glm_base.ipp:23-37; the
// IRLS bookkeeping from solver/solver_glm_naive.hpp:199-231 (null model) and :336-348, :439-449 (fit).
// All of them are n-vector streams (<= 4 MB at n = 500k f64): each kernel fuses every elementwise step that the
// reference does as a separate Eigen expression, and reduces deterministically (fixed 256-block grid, block
// partials combined by one block).  `sums` layout: [0..3] results, [16 + 4*b + k] per-block partials.
#include "kernels.hpp"
#include "../../include/adelie_hip.h"

namespace ahip {

namespace {

constexpr int RB = 256; // reduction grid (blocks)
constexpr int RT = 256; // threads per block

template <class T>
__device__ __forceinline__ T wsum(T x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    return x;
}

template <class T, int K>
__device__ __forceinline__ void block_partials(T (&acc)[K], T* sums) {
    __shared__ T red[RT / 64][K];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const T s = wsum(acc[k]);
        if (lane == 0) red[wv][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < K) {
        T s = 0;
#pragma unroll
        for (int w = 0; w < RT / 64; ++w) s += red[w][threadIdx.x];
        sums[16 + 4 * blockIdx.x + threadIdx.x] = s;
    }
}

template <class T>
__global__ void final_reduce_kernel(T* sums, int K) {
    __shared__ T red[RT / 64];
    for (int k = 0; k < K; ++k) {
        T v = (threadIdx.x < RB) ? sums[16 + 4 * threadIdx.x + k] : T(0);
        v = wsum(v);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            T s = 0;
            for (int w = 0; w < RT / 64; ++w) s += red[w];
            sums[k] = s;
        }
        __syncthreads();
    }
}

// binomial probit helpers (glm_binomial.ipp:100-120)
template <class T> __device__ __forceinline__ T std_cdf(T x) { return T(0.5) * (T(1) + erf(x * T(0.70710678118654752440))); }
template <class T> __device__ __forceinline__ T std_pdf(T x) { return T(0.39894228040143267794) * exp(T(-0.5) * x * x); }
template <class T> __device__ __forceinline__ T fmax_of();
template <> __device__ __forceinline__ double fmax_of<double>() { return 1.7976931348623157e308; }
template <> __device__ __forceinline__ float fmax_of<float>() { return 3.402823466e38f; }

template <class T>
__device__ __forceinline__ T glm_grad(int kind, T y, T w, T eta) {
    if (kind == ADELIE_HIP_GLM_BINOMIAL_LOGIT) return w * (y - T(1) / (T(1) + exp(-eta)));
    if (kind == ADELIE_HIP_GLM_POISSON) return w * (y - exp(eta));                        // glm_poisson.ipp:14-23
    if (kind == ADELIE_HIP_GLM_BINOMIAL_PROBIT) {                                         // glm_binomial.ipp:131-143
        const T mx = fmax_of<T>(), c = std_cdf(eta);
        return w * std_pdf(eta) * (y * min(T(1) / c, mx) - (T(1) - y) * min(T(1) / (T(1) - c), mx));
    }
    return w * (y - eta);
}
// per-observation loss without the weight (glm_*.ipp loss members)
template <class T>
__device__ __forceinline__ T glm_loss_term(int kind, T y, T e) {
    if (kind == ADELIE_HIP_GLM_BINOMIAL_LOGIT) return (T(e > T(0)) - y) * e + log(T(1) + exp(-fabs(e)));
    if (kind == ADELIE_HIP_GLM_POISSON) return min(-e, fmax_of<T>()) * y + exp(e);        // glm_poisson.ipp:36-44
    if (kind == ADELIE_HIP_GLM_BINOMIAL_PROBIT) {                                         // glm_binomial.ipp:161-173
        const T mx = fmax_of<T>(), c = std_cdf(e);
        return -(y * max(log(c), -mx) + (T(1) - y) * max(log(T(1) - c), -mx));
    }
    return T(0.5) * e * e - y * e;
}
// multinomial (glm_multinomial.ipp:47-66): w is the observation weight repeated for every class, K the class count
template <class T>
__device__ __forceinline__ T glm_hess(int kind, T y, T w, T grad, T eta, int K = 1) {
    if (kind == ADELIE_HIP_GLM_POISSON) return w * y - grad;                              // glm_poisson.ipp:25-34
    if (kind == ADELIE_HIP_GLM_BINOMIAL_PROBIT) {                                         // glm_binomial.ipp:145-159
        const T mx = fmax_of<T>(), c = std_cdf(eta), pd = std_pdf(eta), c1 = T(1) - c;
        return w * (y * min(T(1) / (c * c), mx) + (T(1) - y) * min(T(1) / (c1 * c1), mx)) * pd * pd + eta * grad;
    }
    if (kind == ADELIE_HIP_GLM_MULTINOMIAL) {
        const T h = y * w / T(K) - grad;
        return h * (T(2) * (T(1) - T(K) * (h / (w + T(w <= T(0))))));
    }
    if (kind == ADELIE_HIP_GLM_BINOMIAL_LOGIT) {
        const T h = w * y - grad;
        return (h * (w - h)) / (w + T(w <= T(0)));
    }
    return w;
}

#define GRID_STRIDE(i, n) for (int64_t i = int64_t(blockIdx.x) * RT + threadIdx.x; i < (n); i += int64_t(RB) * RT)

template <class T>
__global__ __launch_bounds__(RT) void irls_prepare_kernel(int kind, const T* __restrict__ y, const T* __restrict__ w,
                                                          const T* __restrict__ eta, const T* __restrict__ resid,
                                                          const T* __restrict__ off, T hmin, int64_t n,
                                                          T* __restrict__ hess, T* __restrict__ irls_resid,
                                                          T* __restrict__ irls_y, T* sums, int K) {
    T acc[1] = {T(0)};
    // user-defined GLM (kind CALLBACK): hess and irls_resid arrive filled with glm.hessian() / glm.inv_hessian_gradient()
    const bool cb = kind == ADELIE_HIP_GLM_CALLBACK;
    GRID_STRIDE(i, n) {
        const T h0 = cb ? hess[i] : glm_hess(kind, y[i], w[i], resid[i], eta[i], K);
        const T h = (h0 > T(0) ? h0 : T(0)) + hmin * T(h0 <= T(0));
        const T z = cb ? irls_resid[i] : resid[i] / h; // inv_hessian_gradient uses the same raised hessian (glm_base.ipp:32-36)
        hess[i] = h;
        irls_resid[i] = z;
        irls_y[i] = z + eta[i] - off[i];
        acc[0] += h;
    }
    block_partials<T, 1>(acc, sums);
}

template <class T>
__global__ __launch_bounds__(RT) void irls_weights_kernel(const T* __restrict__ hess, T hess_sum,
                                                          const T* __restrict__ irls_y, T shift, int64_t n,
                                                          T* __restrict__ wts, T* __restrict__ irls_resid, T* sums) {
    T acc[3] = {T(0), T(0), T(0)};
    GRID_STRIDE(i, n) {
        const T wi = hess[i] / hess_sum;
        const T yi = irls_y[i];
        const T ri = irls_resid[i] + shift;
        wts[i] = wi;
        irls_resid[i] = ri;
        acc[0] += wi * yi;
        acc[1] += wi * yi * yi;
        acc[2] += wi * ri;
    }
    block_partials<T, 3>(acc, sums);
}

template <class T>
__global__ __launch_bounds__(RT) void irls_finish_kernel(int kind, const T* __restrict__ y, const T* __restrict__ w,
                                                         const T* __restrict__ irls_y, const T* __restrict__ off,
                                                         const T* __restrict__ irls_resid, T shift, int64_t n,
                                                         T* __restrict__ eta, T* __restrict__ resid) {
    GRID_STRIDE(i, n) {
        const T e = irls_y[i] + off[i] - irls_resid[i] + shift;
        eta[i] = e;
        // multinomial: row-coupled, below; a user-defined GLM's gradient is a host callback on the new eta
        if (kind != ADELIE_HIP_GLM_MULTINOMIAL && kind != ADELIE_HIP_GLM_CALLBACK) resid[i] = glm_grad(kind, y[i], w[i], e);
    }
}

template <class T>
__global__ __launch_bounds__(RT) void glm_gradient_kernel(int kind, const T* __restrict__ y, const T* __restrict__ w,
                                                          const T* __restrict__ eta, int64_t n, T* __restrict__ resid) {
    GRID_STRIDE(i, n) resid[i] = glm_grad(kind, y[i], w[i], eta[i]);
}

template <class T>
__global__ __launch_bounds__(RT) void glm_loss_kernel(int kind, const T* __restrict__ y, const T* __restrict__ w,
                                                      const T* __restrict__ eta, int64_t n, T* sums) {
    T acc[1] = {T(0)};
    GRID_STRIDE(i, n) {
        acc[0] += w[i] * glm_loss_term(kind, y[i], eta[i]);
    }
    block_partials<T, 1>(acc, sums);
}

// loss of eta = base + b0 + off under two weight vectors at once (cv_grpnet: full-data and training-fold weights)
template <class T>
__global__ __launch_bounds__(RT) void glm_loss2_kernel(int kind, const T* __restrict__ y, const T* __restrict__ wa,
                                                       const T* __restrict__ wb, const T* __restrict__ base, T b0,
                                                       const T* __restrict__ off, int64_t n, T* sums) {
    T acc[2] = {T(0), T(0)};
    GRID_STRIDE(i, n) {
        const T e = base[i] + b0 + off[i];
        const T l = glm_loss_term(kind, y[i], e);
        acc[0] += wa[i] * l;
        acc[1] += wb[i] * l;
    }
    block_partials<T, 2>(acc, sums);
}

template <class T>
__global__ __launch_bounds__(RT) void null_step_kernel(int kind, const T* __restrict__ y, const T* __restrict__ w,
                                                       const T* __restrict__ eta, const T* __restrict__ resid,
                                                       const T* __restrict__ off, T hmin, int64_t n, T* sums, int K,
                                                       const T* __restrict__ cb_hess, const T* __restrict__ cb_z) {
    T acc[2] = {T(0), T(0)};
    const bool cb = kind == ADELIE_HIP_GLM_CALLBACK;
    GRID_STRIDE(i, n) {
        const T h0 = cb ? cb_hess[i] : glm_hess(kind, y[i], w[i], resid[i], eta[i], K);
        const T h = (h0 > T(0) ? h0 : T(0)) + hmin * T(h0 <= T(0));
        const T z = cb ? cb_z[i] : resid[i] / h;
        acc[0] += h;
        acc[1] += h * (z + eta[i] - off[i]);
    }
    block_partials<T, 2>(acc, sums);
}

// multinomial gradient / loss (glm_multinomial.ipp:21-45, 68-87) on response-major (nb, K) arrays: element (i, k) at
// [k*nb + i]; one thread per observation walks its K classes (coalesced across threads for every k)
template <class T>
__global__ __launch_bounds__(RT) void multinomial_gradient_kernel(const T* __restrict__ y, const T* __restrict__ w,
                                                                  const T* __restrict__ eta, int64_t nb, int K,
                                                                  T* __restrict__ resid) {
    GRID_STRIDE(i, nb) {
        T mx = eta[i];
        for (int k = 1; k < K; ++k) mx = max(mx, eta[int64_t(k) * nb + i]);
        T sum = T(0);
        for (int k = 0; k < K; ++k) sum += exp(eta[int64_t(k) * nb + i] - mx);
        const T wk = w[i] / T(K);
        for (int k = 0; k < K; ++k) {
            const int64_t q = int64_t(k) * nb + i;
            resid[q] = (y[q] - exp(eta[q] - mx) / sum) * wk;
        }
    }
}
template <class T>
__global__ __launch_bounds__(RT) void multinomial_loss_kernel(const T* __restrict__ y, const T* __restrict__ w,
                                                              const T* __restrict__ eta, int64_t nb, int K, T* sums) {
    T acc[1] = {T(0)};
    GRID_STRIDE(i, nb) {
        T mx = eta[i];
        for (int k = 1; k < K; ++k) mx = max(mx, eta[int64_t(k) * nb + i]);
        T ye = T(0), se = T(0);
        for (int k = 0; k < K; ++k) {
            const int64_t q = int64_t(k) * nb + i;
            ye += y[q] * (eta[q] - mx);
            se += exp(eta[q] - mx);
        }
        acc[0] += w[i] * (-ye + log(se)) / T(K);
    }
    block_partials<T, 1>(acc, sums);
}

// multi-response loss of eta_k = base_k + b0[k] + off_k under two weight vectors (cv_grpnet over multigaussian / multinomial
// fits): response-major (nb, K) arrays as above; glm_multigaussian.ipp:60-72, glm_multinomial.ipp:68-87
template <class T>
__global__ __launch_bounds__(RT) void multi_loss2_kernel(int kind, const T* __restrict__ y, const T* __restrict__ wa,
                                                         const T* __restrict__ wb, const T* __restrict__ base,
                                                         const T* __restrict__ b0, const T* __restrict__ off, int64_t nb,
                                                         int K, T* sums) {
    T acc[2] = {T(0), T(0)};
    GRID_STRIDE(i, nb) {
        T l;
        if (kind == ADELIE_HIP_GLM_MULTINOMIAL) {
            T mx = base[i] + b0[0] + off[i];
            for (int k = 1; k < K; ++k) {
                const int64_t q = int64_t(k) * nb + i;
                mx = max(mx, base[q] + b0[k] + off[q]);
            }
            T ye = T(0), se = T(0);
            for (int k = 0; k < K; ++k) {
                const int64_t q = int64_t(k) * nb + i;
                const T e = base[q] + b0[k] + off[q] - mx;
                ye += y[q] * e;
                se += exp(e);
            }
            l = (-ye + log(se)) / T(K);
        } else {
            T a = T(0);
            for (int k = 0; k < K; ++k) {
                const int64_t q = int64_t(k) * nb + i;
                const T e = base[q] + b0[k] + off[q];
                a += T(0.5) * e * e - y[q] * e;
            }
            l = a / T(K);
        }
        acc[0] += wa[i] * l;
        acc[1] += wb[i] * l;
    }
    block_partials<T, 2>(acc, sums);
}

template <class T>
__global__ __launch_bounds__(RT) void set_eta_kernel(const T* __restrict__ off, T beta0, int64_t n, T* __restrict__ eta) {
    GRID_STRIDE(i, n) eta[i] = beta0 + off[i];
}

template <class T>
__global__ __launch_bounds__(RT) void dot_diff_kernel(const T* __restrict__ a, const T* __restrict__ a0,
                                                      const T* __restrict__ b, const T* __restrict__ b0, int64_t n,
                                                      T* sums) {
    T acc[1] = {T(0)};
    GRID_STRIDE(i, n) acc[0] += (a[i] - a0[i]) * (b[i] - b0[i]);
    block_partials<T, 1>(acc, sums);
}

template <class T>
__global__ void gather_kernel(const T* __restrict__ src, const int32_t* __restrict__ idx, int64_t cnt,
                              T* __restrict__ dst) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < cnt) dst[i] = src[idx[i]];
}

// out[0] = max_i |a_i - prev_i| / prev_i over prev_i > 0 (non-negative values compare like their bit patterns), then
// prev <- a: how far the IRLS weights moved since the iteration the cached diagonal blocks were built for
__device__ __forceinline__ void atomic_max_nonneg(double* out, double v) {
    atomicMax(reinterpret_cast<unsigned long long*>(out), static_cast<unsigned long long>(__double_as_longlong(v)));
}
__device__ __forceinline__ void atomic_max_nonneg(float* out, float v) {
    atomicMax(reinterpret_cast<unsigned int*>(out), __float_as_uint(v));
}
template <class T>
__global__ __launch_bounds__(RT) void rel_change_kernel(const T* __restrict__ a, T* __restrict__ prev, int64_t n, T* out) {
    T m = T(0);
    GRID_STRIDE(i, n) {
        const T x = a[i], y = prev[i];
        prev[i] = x;
        const T d = fabs(x - y);
        T r = y > T(0) ? d / y : (d > T(0) ? T(1e30) : T(0));
        if (!(r == r) || !(d == d)) r = T(1e30); // a non-finite weight is "moved by everything": never a reason to reuse a block
        m = r > m ? r : m;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const T o = __shfl_xor(m, off, 64);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0 && m > T(0)) atomic_max_nonneg(out, m);
}

template <class T>
void finish(T* sums, int K, hipStream_t s) {
    hipLaunchKernelGGL((final_reduce_kernel<T>), dim3(1), dim3(RT), 0, s, sums, K);
}

} // namespace

template <class T>
void launch_irls_prepare(int kind, const T* y, const T* w, const T* eta, const T* resid, const T* offsets, T hessian_min,
                         int64_t n, T* hess, T* irls_resid, T* irls_y, T* sums, hipStream_t s, int K) {
    hipLaunchKernelGGL((irls_prepare_kernel<T>), dim3(RB), dim3(RT), 0, s, kind, y, w, eta, resid, offsets, hessian_min,
                       n, hess, irls_resid, irls_y, sums, K);
    finish(sums, 1, s);
}
template <class T>
void launch_irls_weights(const T* hess, T hess_sum, const T* irls_y, T shift, int64_t n, T* wts, T* irls_resid, T* sums,
                         hipStream_t s) {
    hipLaunchKernelGGL((irls_weights_kernel<T>), dim3(RB), dim3(RT), 0, s, hess, hess_sum, irls_y, shift, n, wts,
                       irls_resid, sums);
    finish(sums, 3, s);
}
template <class T>
void launch_irls_finish(int kind, const T* y, const T* w, const T* irls_y, const T* offsets, const T* irls_resid, T shift,
                        int64_t n, T* eta, T* resid, T* /*sums*/, hipStream_t s, int K) {
    hipLaunchKernelGGL((irls_finish_kernel<T>), dim3(RB), dim3(RT), 0, s, kind, y, w, irls_y, offsets, irls_resid, shift,
                       n, eta, resid);
    if (kind == ADELIE_HIP_GLM_MULTINOMIAL)
        hipLaunchKernelGGL((multinomial_gradient_kernel<T>), dim3(RB), dim3(RT), 0, s, y, w, eta, n / K, K, resid);
}
template <class T>
void launch_glm_gradient(int kind, const T* y, const T* w, const T* eta, int64_t n, T* resid, hipStream_t s, int K) {
    if (kind == ADELIE_HIP_GLM_MULTINOMIAL) {
        hipLaunchKernelGGL((multinomial_gradient_kernel<T>), dim3(RB), dim3(RT), 0, s, y, w, eta, n / K, K, resid);
        return;
    }
    hipLaunchKernelGGL((glm_gradient_kernel<T>), dim3(RB), dim3(RT), 0, s, kind, y, w, eta, n, resid);
}
template <class T>
void launch_glm_loss(int kind, const T* y, const T* w, const T* eta, int64_t n, T* sums, hipStream_t s, int K) {
    if (kind == ADELIE_HIP_GLM_MULTINOMIAL)
        hipLaunchKernelGGL((multinomial_loss_kernel<T>), dim3(RB), dim3(RT), 0, s, y, w, eta, n / K, K, sums);
    else
        hipLaunchKernelGGL((glm_loss_kernel<T>), dim3(RB), dim3(RT), 0, s, kind, y, w, eta, n, sums);
    finish(sums, 1, s);
}
template <class T>
void launch_glm_loss2(int kind, const T* y, const T* wa, const T* wb, const T* base, T b0, const T* off, int64_t n, T* sums,
                      hipStream_t s) {
    hipLaunchKernelGGL((glm_loss2_kernel<T>), dim3(RB), dim3(RT), 0, s, kind, y, wa, wb, base, b0, off, n, sums);
    finish(sums, 2, s);
}
template <class T>
void launch_multi_loss2(int kind, const T* y, const T* wa, const T* wb, const T* base, const T* b0, const T* off, int64_t nb,
                        int K, T* sums, hipStream_t s) {
    hipLaunchKernelGGL((multi_loss2_kernel<T>), dim3(RB), dim3(RT), 0, s, kind, y, wa, wb, base, b0, off, nb, K, sums);
    finish(sums, 2, s);
}
template <class T>
void launch_null_step(int kind, const T* y, const T* w, const T* eta, const T* resid, const T* offsets, T hessian_min,
                      int64_t n, T* sums, hipStream_t s, int K, const T* cb_hess, const T* cb_z) {
    hipLaunchKernelGGL((null_step_kernel<T>), dim3(RB), dim3(RT), 0, s, kind, y, w, eta, resid, offsets, hessian_min, n,
                       sums, K, cb_hess, cb_z);
    finish(sums, 2, s);
}
template <class T>
void launch_set_eta(const T* offsets, T beta0, int64_t n, T* eta, hipStream_t s) {
    hipLaunchKernelGGL((set_eta_kernel<T>), dim3(RB), dim3(RT), 0, s, offsets, beta0, n, eta);
}
template <class T>
void launch_dot_diff(const T* a, const T* a0, const T* b, const T* b0, int64_t n, T* sums, hipStream_t s) {
    hipLaunchKernelGGL((dot_diff_kernel<T>), dim3(RB), dim3(RT), 0, s, a, a0, b, b0, n, sums);
    finish(sums, 1, s);
}
template <class T>
void launch_rel_change(const T* a, T* prev, int64_t n, T* out, hipStream_t s) {
    (void)hipMemsetAsync(out, 0, sizeof(T), s);
    hipLaunchKernelGGL((rel_change_kernel<T>), dim3(RB), dim3(RT), 0, s, a, prev, n, out);
}
namespace {
template <class T>
__global__ void scatter_kernel(const T* __restrict__ src, const int32_t* __restrict__ idx, int64_t cnt, T* __restrict__ dst) {
    const int64_t a = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (a < cnt) dst[idx[a]] = src[a];
}
} // namespace
// dst[idx[a]] = src[a]  (distinct indices)
template <class T>
void launch_scatter(const T* src, const int32_t* idx, int64_t cnt, T* dst, hipStream_t s) {
    if (cnt <= 0) return;
    hipLaunchKernelGGL((scatter_kernel<T>), dim3(unsigned((cnt + 255) / 256)), dim3(256), 0, s, src, idx, cnt, dst);
}
template <class T>
void launch_gather(const T* src, const int32_t* idx, int64_t cnt, T* dst, hipStream_t s) {
    if (cnt <= 0) return;
    hipLaunchKernelGGL((gather_kernel<T>), dim3(unsigned((cnt + 255) / 256)), dim3(256), 0, s, src, idx, cnt, dst);
}

#define INST(T)                                                                                                        \
    template void launch_irls_prepare<T>(int, const T*, const T*, const T*, const T*, const T*, T, int64_t, T*, T*,    \
                                         T*, T*, hipStream_t, int);                                                       \
    template void launch_irls_weights<T>(const T*, T, const T*, T, int64_t, T*, T*, T*, hipStream_t);                  \
    template void launch_irls_finish<T>(int, const T*, const T*, const T*, const T*, const T*, T, int64_t, T*, T*, T*, \
                                        hipStream_t, int);                                                                \
    template void launch_glm_gradient<T>(int, const T*, const T*, const T*, int64_t, T*, hipStream_t, int);              \
    template void launch_glm_loss<T>(int, const T*, const T*, const T*, int64_t, T*, hipStream_t, int);                  \
    template void launch_glm_loss2<T>(int, const T*, const T*, const T*, const T*, T, const T*, int64_t, T*,           \
                                      hipStream_t);                                                                    \
    template void launch_multi_loss2<T>(int, const T*, const T*, const T*, const T*, const T*, const T*, int64_t, int, T*, \
                                        hipStream_t);                                                                  \
    template void launch_null_step<T>(int, const T*, const T*, const T*, const T*, const T*, T, int64_t, T*,           \
                                      hipStream_t, int, const T*, const T*);                                                                \
    template void launch_set_eta<T>(const T*, T, int64_t, T*, hipStream_t);                                            \
    template void launch_dot_diff<T>(const T*, const T*, const T*, const T*, int64_t, T*, hipStream_t);                \
    template void launch_rel_change<T>(const T*, T*, int64_t, T*, hipStream_t);                                        \
    template void launch_gather<T>(const T*, const int32_t*, int64_t, T*, hipStream_t);                                \
    template void launch_scatter<T>(const T*, const int32_t*, int64_t, T*, hipStream_t);
INST(double)
INST(float)
#undef INST

} // namespace ahip
