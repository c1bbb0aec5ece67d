// micro-benchmark of the panel step kernel vs the plain sweep / axpy kernels on the same 128 columns
#include "../../adelie_amd/csrc/kernels.hpp"
#include <cstdio>
#include <vector>
#include <random>
using namespace ahip;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 100000, p = argc > 2 ? atoll(argv[2]) : 4096, ld = argc > 3 ? atoll(argv[3]) : n;
    double* X; CK(hipMalloc(&X, ld * p * 8)); CK(hipMemset(X, 0, ld * p * 8));
    double *w, *r, *dlt, *part, *g, *work; int32_t *cols, *dcol, *nz;
    CK(hipMalloc(&w, n * 8)); CK(hipMalloc(&r, n * 8)); CK(hipMalloc(&dlt, 128 * 8)); CK(hipMalloc(&g, (p + 128) * 8));
    CK(hipMalloc(&part, panel_part_elems(n) * 8)); CK(hipMalloc(&work, (sweep_work_elems(n, p) + sweep_work_elems(n, 1024) + 16) * 8));
    CK(hipMalloc(&cols, 128 * 4)); CK(hipMalloc(&dcol, 128 * 4)); CK(hipMalloc(&nz, 4));
    CK(hipMemset(w, 0, n * 8)); CK(hipMemset(r, 0, n * 8)); CK(hipMemset(dlt, 0, 128 * 8));
    std::mt19937 rng(1);
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    DenseView<double> V{X, n, p, ld};
    auto newcols = [&](int32_t* dst) { std::vector<int32_t> h(128); for (auto& c : h) c = rng() % p; return hipMemcpy(dst, h.data(), 512, hipMemcpyHostToDevice); };
    struct Case { int nz, nb; const char* name; };
    Case cases[] = {{0, 128, "B only"}, {128, 0, "A only"}, {128, 128, "A+B"}, {16, 128, "A16+B"}};
    for (auto& c : cases) {
        CK(hipMemcpy(nz, &c.nz, 4, hipMemcpyHostToDevice));
        float tot = 0; const int reps = 20;
        for (int it = 0; it < reps + 3; ++it) {
            CK(newcols(cols)); CK(newcols(dcol));
            CK(hipEventRecord(a, s));
            launch_panel_step<double>(V, w, r, dcol, dlt, nz, cols, c.nb, part, s);
            CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); if (it >= 3) tot += ms;
        }
        const double bytes = double(c.nz + c.nb) * n * 8;
        printf("panel_step %-8s nz=%3d nb=%3d: %.1f us  %.2f TB/s\n", c.name, c.nz, c.nb, 1e3 * tot / reps, bytes / (tot / reps * 1e-3) / 1e12);
    }
    {   // adjacent columns (one contiguous 100 MB region) instead of 128 random ones
        int nzv = 0; CK(hipMemcpy(nz, &nzv, 4, hipMemcpyHostToDevice));
        float tot = 0; const int reps = 20;
        for (int it = 0; it < reps + 3; ++it) {
            std::vector<int32_t> h(128); const int c0 = rng() % (p - 128); for (int k = 0; k < 128; ++k) h[k] = c0 + k;
            CK(hipMemcpy(cols, h.data(), 512, hipMemcpyHostToDevice));
            CK(hipEventRecord(a, s));
            launch_panel_step<double>(V, w, r, dcol, dlt, nz, cols, 128, part, s);
            CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); if (it >= 3) tot += ms;
        }
        printf("panel_step B only, ADJACENT cols: %.1f us  %.2f TB/s\n", 1e3 * tot / reps, 128.0 * n * 8 / (tot / reps * 1e-3) / 1e12);
        tot = 0;
        for (int it = 0; it < reps + 3; ++it) {
            std::vector<int32_t> h(128); const int c0 = rng() % (p - 128); for (int k = 0; k < 128; ++k) h[k] = c0 + k;
            CK(hipMemcpy(cols, h.data(), 512, hipMemcpyHostToDevice));
            CK(hipEventRecord(a, s));
            launch_sweep<double>(V, w, g, 0, 128, cols, nullptr, nullptr, false, work, s);
            CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); if (it >= 3) tot += ms;
        }
        printf("sweep 128 ADJACENT cols: %.1f us  %.2f TB/s\n", 1e3 * tot / reps, 128.0 * n * 8 / (tot / reps * 1e-3) / 1e12);
        tot = 0;
        for (int it = 0; it < reps + 3; ++it) {
            CK(hipEventRecord(a, s));
            launch_sweep<double>(V, w, g, 0, (p < 1024 ? p : 1024), nullptr, nullptr, nullptr, false, work, s);
            CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); if (it >= 3) tot += ms;
        }
        printf("sweep 1024 cols (0..1023): %.1f us  %.2f TB/s\n", 1e3 * tot / reps, 1024.0 * n * 8 / (tot / reps * 1e-3) / 1e12);
        tot = 0;
        for (int it = 0; it < 8; ++it) {
            CK(hipEventRecord(a, s));
            launch_sweep<double>(V, w, g, 0, p, nullptr, nullptr, nullptr, false, work, s);
            CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); if (it >= 3) tot += ms;
        }
        printf("sweep ALL %lld cols: %.1f us  %.3f TB/s\n", (long long)p, 1e3 * tot / 5, double(p) * n * 8 / (tot / 5 * 1e-3) / 1e12);
    }
    {   // reuse: the (A) columns of a step are the (B) columns of the step before (as in a real pass)
        int nzv = 128; CK(hipMemcpy(nz, &nzv, 4, hipMemcpyHostToDevice));
        float tot = 0; const int reps = 40;
        CK(newcols(cols));
        for (int it = 0; it < reps + 3; ++it) {
            CK(hipMemcpy(dcol, cols, 512, hipMemcpyDeviceToDevice));
            CK(newcols(cols));
            CK(hipEventRecord(a, s));
            launch_panel_step<double>(V, w, r, dcol, dlt, nz, cols, 128, part, s);
            CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); if (it >= 3) tot += ms;
        }
        printf("panel_step A(reuse prev B)+B: %.1f us  %.2f TB/s\n", 1e3 * tot / reps, 256.0 * n * 8 / (tot / reps * 1e-3) / 1e12);
    }
    {   float tot = 0; const int reps = 20;
        for (int it = 0; it < reps + 3; ++it) {
            CK(newcols(cols));
            CK(hipEventRecord(a, s));
            launch_sweep<double>(V, w, g, 0, 128, cols, nullptr, nullptr, false, work, s);
            CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); if (it >= 3) tot += ms;
        }
        printf("sweep 128 cols (2 launches): %.1f us  %.2f TB/s\n", 1e3 * tot / reps, 128.0 * n * 8 / (tot / reps * 1e-3) / 1e12);
    }
    {   float tot = 0; const int reps = 20;
        for (int it = 0; it < reps + 3; ++it) {
            CK(newcols(cols));
            CK(hipEventRecord(a, s));
            launch_axpy_cols<double>(V, cols, dlt, nullptr, 128, -1.0, r, s);
            CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); if (it >= 3) tot += ms;
        }
        printf("axpy 128 cols: %.1f us  %.2f TB/s\n", 1e3 * tot / reps, 128.0 * n * 8 / (tot / reps * 1e-3) / 1e12);
    }
    return 0;
}
