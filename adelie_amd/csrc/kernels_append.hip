// kernels_append.hip — new screen groups reach the device mirrors as ONE packed upload + one scatter launch
// (solver.hip::device_append_screen; the reference appends to its std::vectors in update_screen_derived, solver_base.hpp:120-153).
#include "kernels.hpp"

namespace ahip {
namespace {
template <class T>
__global__ void screen_append_kernel(const char* __restrict__ img, AppendDst<T> d) {
    const AppendImage<T> L(d.Ng, d.Nv, d.cons != 0);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < d.Ng) {
        const int32_t b = reinterpret_cast<const int32_t*>(img + L.begin)[t];
        d.spen[d.ns_old + t] = reinterpret_cast<const T*>(img + L.pen)[t];
        d.sbegin[d.ns_old + t] = b;
        d.ssize[d.ns_old + t] = reinterpret_cast<const int32_t*>(img + L.size)[t];
        d.isact[d.ns_old + t] = int8_t(reinterpret_cast<const int32_t*>(img + L.isact)[t]);
        d.slot[reinterpret_cast<const int32_t*>(img + L.group)[t]] = b;
    }
    if (t < d.Nv) {
        d.beta[d.nv_old + t] = reinterpret_cast<const T*>(img + L.beta)[t];
        d.vcol[d.nv_old + t] = reinterpret_cast<const int32_t*>(img + L.vcol)[t];
        if (d.cons) {
            d.clo[d.nv_old + t] = reinterpret_cast<const T*>(img + L.lo)[t];
            d.chi[d.nv_old + t] = reinterpret_cast<const T*>(img + L.hi)[t];
            d.cmu[d.nv_old + t] = reinterpret_cast<const T*>(img + L.mu)[t];
        }
    }
}
} // namespace

template <class T>
void launch_screen_append(const void* image_dev, const AppendDst<T>& d, hipStream_t s) {
    const int nt = std::max(d.Ng, d.Nv);
    if (nt <= 0) return;
    hipLaunchKernelGGL((screen_append_kernel<T>), dim3((nt + 255) / 256), dim3(256), 0, s, static_cast<const char*>(image_dev), d);
}
template void launch_screen_append<double>(const void*, const AppendDst<double>&, hipStream_t);
template void launch_screen_append<float>(const void*, const AppendDst<float>&, hipStream_t);

} // namespace ahip
