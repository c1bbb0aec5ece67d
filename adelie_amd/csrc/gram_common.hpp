// gram_common.hpp — pieces shared by the MFMA Gram kernels (kernels_gram.hip, kernels_strip.hip): the 16x16x4 matrix-core
// instruction per element type and the per-thread column-run loader.
#pragma once
#include "kernels.hpp"
#include "accessors.hpp"

namespace ahip {
namespace {

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef unsigned u4v_t __attribute__((ext_vector_type(4)));
typedef float f4v_t __attribute__((ext_vector_type(4)));

template <class T> struct Mfma;
template <> struct Mfma<double> {
    using acc_t = d4_t;
    static __device__ __forceinline__ acc_t run(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};
template <> struct Mfma<float> {
    using acc_t = f4v_t;
    static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int lane, int reg) { return (lane >> 4) * 4 + reg; }
};

// Loads R consecutive rows [k, k+R) of column j (zero beyond kend / for an invalid column).
template <class T, class Acc, bool VECOK, int R>
__device__ __forceinline__ void load_rows(const Acc& X, int64_t j, bool valid, int64_t k, int64_t kend, T (&r)[R]) {
    if (!valid || k >= kend) {
#pragma unroll
        for (int e = 0; e < R; ++e) r[e] = T(0);
        return;
    }
    auto cp = X.colptr(j);
    constexpr int V = VecOf<T>::N;
    if (VECOK && k + R <= kend) {
#pragma unroll
        for (int u = 0; u < R / V; ++u) {
            const Pack<T, V> x = X.template load<V>(cp, k + u * V, j);
#pragma unroll
            for (int e = 0; e < V; ++e) r[u * V + e] = x.v[e];
        }
    } else {
#pragma unroll
        for (int e = 0; e < R; ++e) r[e] = (k + e < kend) ? X.template load<1>(cp, k + e, j).v[0] : T(0);
    }
}

} // namespace
} // namespace ahip
