// Micro-benchmark: how fast can ONE workgroup (one CU) stream 64 KB "columns" from HBM, vs several workgroups?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int NT, int U>
__global__ __launch_bounds__(NT) void stream(const double* __restrict__ C, long ld, const int* __restrict__ cols, int ncols,
                                             int colbytes, double* out) {
    const int tid = threadIdx.x;
    d2 acc = {0, 0};
    const int per = colbytes / 16;  // 16B units per column
    for (int c = blockIdx.x; c < ncols; c += gridDim.x) {
        const double* p = C + (long)cols[c] * ld;
        for (int i0 = tid; i0 < per; i0 += NT * U) {
            d2 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                int i = i0 + u * NT;
                i = i < per ? i : per - 1;
                v[u] = __builtin_nontemporal_load(reinterpret_cast<const d2*>(p) + i);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u];
        }
    }
    if (acc[0] + acc[1] == 12345.678) out[0] = acc[0];
}
template <int NT, int U>
void run(const double* C, long ld, const int* cols, int ncols, int colbytes, double* out, int blocks) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    stream<NT, U><<<blocks, NT>>>(C, ld, cols, ncols, colbytes, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    stream<NT, U><<<blocks, NT>>>(C, ld, cols, ncols, colbytes, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("NT=%4d U=%2d blocks=%3d colKB=%3d : %8.1f GB/s  (%.2f us/col)\n", NT, U, blocks, colbytes / 1024,
           (double)ncols * colbytes / (ms * 1e-3) / 1e9, ms * 1e3 / ncols * blocks);
}
int main() {
    const long ld = 8192, ncap = 8192;  // 512 MB matrix
    double* C; hipMalloc(&C, ld * ncap * 8); hipMemset(C, 0, ld * ncap * 8);
    const int ncols = 4096;
    std::vector<int> h(ncols);
    unsigned s = 12345;
    for (int i = 0; i < ncols; ++i) { s = s * 1664525u + 1013904223u; h[i] = (s >> 8) % ncap; }
    int* cols; hipMalloc(&cols, ncols * 4); hipMemcpy(cols, h.data(), ncols * 4, hipMemcpyHostToDevice);
    double* out; hipMalloc(&out, 8);
    for (int kb : {64, 32}) {
        run<256, 8>(C, ld, cols, ncols, kb * 1024, out, 1);
        run<512, 8>(C, ld, cols, ncols, kb * 1024, out, 1);
        run<1024, 4>(C, ld, cols, ncols, kb * 1024, out, 1);
        run<1024, 8>(C, ld, cols, ncols, kb * 1024, out, 1);
        run<512, 16>(C, ld, cols, ncols, kb * 1024, out, 1);
    }
    for (int blocks : {2, 4, 8, 16, 32, 64, 256}) run<512, 8>(C, ld, cols, ncols, 64 * 1024, out, blocks);
    return 0;
}
