// host-side cost per call of small copies vs kernel launches (round 3: the per-lambda copies of a path)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void copy_k(double* dst, const double* src, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) dst[i] = src[i]; }
__global__ void empty_k() {}
int main() {
    const int n = 1024, reps = 2000;
    double *d1, *d2, *hp, *hm, *hm_dev;
    CK(hipMalloc(&d1, n * 8)); CK(hipMalloc(&d2, n * 8));
    CK(hipHostMalloc(&hp, n * 8, hipHostMallocDefault));
    CK(hipHostMalloc(&hm, n * 8, hipHostMallocMapped)); CK(hipHostGetDevicePointer((void**)&hm_dev, hm, 0));
    std::vector<double> pg(n, 1.0);
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto run = [&](const char* name, auto fn) {
        for (int i = 0; i < 50; ++i) fn();
        (void)hipStreamSynchronize(s);
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; ++i) fn();
        auto t1 = std::chrono::steady_clock::now();
        (void)hipStreamSynchronize(s);
        auto t2 = std::chrono::steady_clock::now();
        printf("%-44s host %.2f us/call, incl. drain %.2f us/call\n", name, std::chrono::duration<double, std::micro>(t1 - t0).count() / reps,
               std::chrono::duration<double, std::micro>(t2 - t0).count() / reps);
    };
    run("empty kernel", [&] { hipLaunchKernelGGL(empty_k, dim3(1), dim3(64), 0, s); });
    run("copy kernel d2d 8KB", [&] { hipLaunchKernelGGL(copy_k, dim3(4), dim3(256), 0, s, d1, d2, n); });
    run("copy kernel mapped-host -> dev 8KB", [&] { hipLaunchKernelGGL(copy_k, dim3(4), dim3(256), 0, s, d1, hm_dev, n); });
    run("copy kernel dev -> mapped-host 8KB", [&] { hipLaunchKernelGGL(copy_k, dim3(4), dim3(256), 0, s, hm_dev, d1, n); });
    run("copy kernel pinned(default) -> dev 8KB", [&] { hipLaunchKernelGGL(copy_k, dim3(4), dim3(256), 0, s, d1, hp, n); });
    run("hipMemcpyAsync pageable H2D 8KB", [&] { (void)hipMemcpyAsync(d1, pg.data(), n * 8, hipMemcpyHostToDevice, s); });
    run("hipMemcpyAsync pinned H2D 8KB", [&] { (void)hipMemcpyAsync(d1, hp, n * 8, hipMemcpyHostToDevice, s); });
    run("hipMemcpyAsync pinned D2H 8KB", [&] { (void)hipMemcpyAsync(hp, d1, n * 8, hipMemcpyDeviceToHost, s); });
    run("hipMemcpyAsync pageable D2H 8KB", [&] { (void)hipMemcpyAsync(pg.data(), d1, n * 8, hipMemcpyDeviceToHost, s); });
    run("hipMemcpyAsync D2D 8KB", [&] { (void)hipMemcpyAsync(d1, d2, n * 8, hipMemcpyDeviceToDevice, s); });
    run("hipMemcpyAsync pinned H2D 48 B", [&] { (void)hipMemcpyAsync(d1, hp, 48, hipMemcpyHostToDevice, s); });
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    run("hipEventRecord", [&] { (void)hipEventRecord(ev, s); });
    hipStream_t s2; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    run("hipEventRecord + hipStreamWaitEvent(other)", [&] { (void)hipEventRecord(ev, s); (void)hipStreamWaitEvent(s2, ev, 0); });
    // one sync round trip
    {
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 500; ++i) { hipLaunchKernelGGL(empty_k, dim3(1), dim3(64), 0, s); (void)hipStreamSynchronize(s); }
        auto t1 = std::chrono::steady_clock::now();
        printf("%-44s %.2f us\n", "kernel + hipStreamSynchronize round trip", std::chrono::duration<double, std::micro>(t1 - t0).count() / 500);
        t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 500; ++i) { (void)hipMemcpyAsync(hp, d1, 48, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s); }
        t1 = std::chrono::steady_clock::now();
        printf("%-44s %.2f us\n", "D2H 48 B pinned + sync round trip", std::chrono::duration<double, std::micro>(t1 - t0).count() / 500);
        t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 500; ++i) { (void)hipMemcpyAsync(pg.data(), d1, 48, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s); }
        t1 = std::chrono::steady_clock::now();
        printf("%-44s %.2f us\n", "D2H 48 B pageable + sync round trip", std::chrono::duration<double, std::micro>(t1 - t0).count() / 500);
        t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 500; ++i) { hipLaunchKernelGGL(copy_k, dim3(1), dim3(64), 0, s, hm_dev, d1, 6); (void)hipStreamSynchronize(s); }
        t1 = std::chrono::steady_clock::now();
        printf("%-44s %.2f us\n", "kernel -> mapped host 48 B + sync round trip", std::chrono::duration<double, std::micro>(t1 - t0).count() / 500);
    }
    return 0;
}
