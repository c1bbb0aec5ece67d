// Micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 (and the f64 VALU fma for comparison) on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(double* out, int iters) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void fma_loop(double* out, int iters) {
    double acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = i;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = fma(a, acc[i], b);
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    double* out; hipMalloc(&out, 8 * 256 * 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int blocks_per_cu : {1, 2, 4}) {
        const int blocks = 256 * blocks_per_cu;
        auto timeit = [&](auto launch, double flops_per_thread_iter, const char* name) {
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double fl = flops_per_thread_iter * iters * blocks * 256.0;
            printf("%-22s blocks/CU=%d  %.1f TFLOP/s  (%.3f ms)\n", name, blocks_per_cu, fl / (ms * 1e-3) / 1e12, ms);
        };
        // one MFMA = 2048 flop per wave = 32 flop per lane
        timeit([&] { mfma_loop<4><<<blocks, 256>>>(out, iters); }, 4 * 32.0, "mfma_f64 4 acc");
        timeit([&] { mfma_loop<16><<<blocks, 256>>>(out, iters); }, 16 * 32.0, "mfma_f64 16 acc");
        timeit([&] { fma_loop<<<blocks, 256>>>(out, iters); }, 16 * 2.0, "v_fma_f64 16 chains");
    }
    return 0;
}
