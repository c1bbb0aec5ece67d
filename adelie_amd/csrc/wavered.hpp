// wavered.hpp — fixed-order sums over the 64 lanes of a wavefront (DPP / readlane; no LDS).
#pragma once
#include <hip/hip_runtime.h>

namespace ahip {

template <int CTRL>
__device__ __forceinline__ double pdpp(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float pdpp(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ double prdl(double x, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}
__device__ __forceinline__ float prdl(float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); }

// sum over the 64 lanes, fixed order: DPP inside each row of 16, then the four row sums via v_readlane
template <class T>
__device__ __forceinline__ T wave_sum64(T x) {
    x += pdpp<0xB1>(x);  // quad_perm [1,0,3,2]
    x += pdpp<0x4E>(x);  // quad_perm [2,3,0,1]
    x += pdpp<0x141>(x); // row_half_mirror
    x += pdpp<0x140>(x); // row_mirror
    return (prdl(x, 0) + prdl(x, 16)) + (prdl(x, 32) + prdl(x, 48));
}

// Sums eight per-lane values over the 64 lanes at once (butterfly with halving): after the xor-1 / xor-2 / xor-4 steps a
// lane carries one value (index lane & 7) summed over its group of 8, three more xor steps finish it.  ~10 exchange-adds
// instead of 8 x 7 for eight separate wave sums; the order is fixed, so the result is deterministic.
template <class T>
__device__ __forceinline__ T reduce8(const T (&v)[8], int lane) {
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
    T a[4], c[2];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const T keep = b0 ? v[2 * m + 1] : v[2 * m], send = b0 ? v[2 * m] : v[2 * m + 1];
        a[m] = keep + pdpp<0xB1>(send); // xor 1
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const T keep = b1 ? a[2 * m + 1] : a[2 * m], send = b1 ? a[2 * m] : a[2 * m + 1];
        c[m] = keep + pdpp<0x4E>(send); // xor 2
    }
    const T keep = b2 ? c[1] : c[0], send = b2 ? c[0] : c[1];
    T t = keep + __shfl_xor(send, 4, 64);
    t += pdpp<0x128>(t); // row_ror:8 == xor 8 inside a row of 16
    t += __shfl_xor(t, 16, 64);
    t += __shfl_xor(t, 32, 64);
    return t; // lane l: total of value (l & 7)
}

} // namespace ahip
