// accessors.hpp — column accessors shared by the streaming and Gram kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace ahip {

typedef double d2_t __attribute__((ext_vector_type(2)));
typedef float f4_t __attribute__((ext_vector_type(4)));
template <class T> struct VecOf;
template <> struct VecOf<double> { using type = d2_t; static constexpr int N = 2; };
template <> struct VecOf<float> { using type = f4_t; static constexpr int N = 4; };

template <class T, int VEC> struct Pack { T v[VEC]; };

// ---- column accessors ---------------------------------------------------------------------------
template <class T>
struct DenseAcc {
    const T* X;
    int64_t ld;
    __device__ __forceinline__ const T* colptr(int64_t j) const { return X + j * ld; }
    template <int VEC>
    __device__ __forceinline__ Pack<T, VEC> load(const T* col, int64_t i, int64_t /*j*/) const {
        Pack<T, VEC> r;
        if constexpr (VEC == 1) {
            r.v[0] = col[i];
        } else {
            using V = typename VecOf<T>::type;
            V x = __builtin_nontemporal_load(reinterpret_cast<const V*>(col + i));
#pragma unroll
            for (int e = 0; e < VEC; ++e) r.v[e] = x[e];
        }
        return r;
    }
};

// Dense design preceded by `shift` (0 or 1) columns of ones: the "extended features" of the multi-response view
// [1 (x) I_K, X (x) I_K] (kernels_multi.hip).  Same interface as DenseAcc, so the MFMA syrk kernel takes it unchanged.
template <class T>
struct DenseOnesAcc {
    const T* X;
    int64_t ld;
    const T* ones;
    int64_t shift;
    __device__ __forceinline__ const T* colptr(int64_t u) const { return u < shift ? ones : X + (u - shift) * ld; }
    template <int VEC>
    __device__ __forceinline__ Pack<T, VEC> load(const T* col, int64_t i, int64_t /*j*/) const {
        Pack<T, VEC> r;
        if constexpr (VEC == 1) {
            r.v[0] = col[i];
        } else {
            using V = typename VecOf<T>::type;
            V x = __builtin_nontemporal_load(reinterpret_cast<const V*>(col + i));
#pragma unroll
            for (int e = 0; e < VEC; ++e) r.v[e] = x[e];
        }
        return r;
    }
};

// 2-bit SNP calls: value = code<3 ? code : impute[j]   (matrix_naive_snp_unphased.ipp:8-22 semantics)
template <class T>
struct SnpAcc {
    const uint8_t* bits;
    int64_t ldb;
    const T* impute;
    __device__ __forceinline__ const uint8_t* colptr(int64_t j) const { return bits + j * ldb; }
    template <int VEC>
    __device__ __forceinline__ Pack<T, VEC> load(const uint8_t* col, int64_t i, int64_t j) const {
        Pack<T, VEC> r;
        const T imp = impute[j];
        if constexpr (VEC == 1) {
            const unsigned c = (col[i >> 2] >> (2 * (i & 3))) & 3u;
            r.v[0] = c == 3u ? imp : T(c);
        } else if constexpr (VEC == 2) {
            const unsigned byte = col[i >> 2] >> (2 * (i & 3)); // i even
#pragma unroll
            for (int k = 0; k < 2; ++k) { const unsigned c = (byte >> (2 * k)) & 3u; r.v[k] = c == 3u ? imp : T(c); }
        } else {
            const unsigned byte = col[i >> 2]; // i multiple of 4
#pragma unroll
            for (int k = 0; k < 4; ++k) { const unsigned c = (byte >> (2 * k)) & 3u; r.v[k] = c == 3u ? imp : T(c); }
        }
        return r;
    }
};

// 2-bit SNP design preceded by `shift` (0 or 1) columns of ones: the extended features of the multi-response view over an
// SNP base.  Same interface as SnpAcc (the K-wide sweep and the MFMA Gram kernels take it unchanged); the ones columns are
// never dereferenced.
template <class T>
struct SnpOnesAcc {
    const uint8_t* bits;
    int64_t ldb;
    const T* impute;
    int64_t shift;
    __device__ __forceinline__ const uint8_t* colptr(int64_t u) const { return u < shift ? bits : bits + (u - shift) * ldb; }
    template <int VEC>
    __device__ __forceinline__ Pack<T, VEC> load(const uint8_t* col, int64_t i, int64_t u) const {
        if (u < shift) {
            Pack<T, VEC> r;
#pragma unroll
            for (int e = 0; e < VEC; ++e) r.v[e] = T(1);
            return r;
        }
        return SnpAcc<T>{bits, ldb, impute}.template load<VEC>(col, i, u - shift);
    }
};

} // namespace ahip
