// micro-benchmark + check of the strip build (kernels_strip.hip) against the staged kernels (syrk_batch + gram_batch) on the
// same block pair: n x 256 columns, the last m members of the second block are "new".
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench/strip.hip adelie_amd/csrc/build/kernels_strip.o \
//         adelie_amd/csrc/build/kernels_gram.o -o scripts/ubench/strip
#include "../../adelie_amd/csrc/kernels.hpp"
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
using namespace ahip;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <class T>
int run(int64_t n, int64_t p) {
    const int64_t ld = n;
    T* X; CK(hipMalloc(&X, ld * p * sizeof(T)));
    std::mt19937_64 rng(7);
    {
        std::vector<T> h(size_t(ld) * 512);
        std::normal_distribution<double> nd;
        for (auto& v : h) v = T(nd(rng));
        for (int64_t c = 0; c < p; c += 512) CK(hipMemcpy(X + c * ld, h.data(), size_t(std::min<int64_t>(512, p - c)) * ld * sizeof(T), hipMemcpyHostToDevice));
        // make the columns distinct: scale differently per chunk is not needed for the check (random column picks below)
    }
    std::vector<T> hw(n), hxm(p);
    for (auto& v : hw) v = T((0.5 + (rng() % 1000) / 1000.0) / n);
    for (auto& v : hxm) v = T((rng() % 1000) / 1000.0 - 0.5);
    T *w, *xm, *work, *D0, *X0, *D1, *X1; int32_t* cols;
    CK(hipMalloc(&w, n * sizeof(T))); CK(hipMalloc(&xm, p * sizeof(T)));
    CK(hipMemcpy(w, hw.data(), n * sizeof(T), hipMemcpyHostToDevice)); CK(hipMemcpy(xm, hxm.data(), p * sizeof(T), hipMemcpyHostToDevice));
    const int SL = 128;
    const size_t welems = size_t(std::max<int64_t>({strip_work_elems(n, 8, 64), syrk_batch_work_elems(n, 8), gram_batch_work_elems(n, 8)}));
    CK(hipMalloc(&work, welems * sizeof(T)));
    CK(hipMalloc(&D0, 8 * SL * SL * sizeof(T))); CK(hipMalloc(&X0, 8 * SL * SL * sizeof(T)));
    CK(hipMalloc(&D1, 8 * SL * SL * sizeof(T))); CK(hipMalloc(&X1, 8 * SL * SL * sizeof(T)));
    CK(hipMalloc(&cols, 2048 * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    DenseView<T> V{X, n, p, ld};
    auto newcols = [&]() {
        std::vector<int32_t> h(2048);
        for (auto& c : h) c = int32_t(rng() % p);
        return hipMemcpy(cols, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    };
    struct Case { int m, nbp, nb, count; };
    const Case cases[] = {{16, 128, 128, 1}, {13, 128, 77, 1}, {32, 128, 128, 1}, {64, 128, 128, 1}, {5, 0, 37, 1}, {16, 128, 128, 2}, {16, 128, 128, 4}, {32, 128, 128, 2}, {64, 128, 64, 1}, {64, 128, 128, 2}};
    for (const Case& c : cases) {
        float t_strip = 0, t_old = 0;
        const int reps = 10;
        double maxerr = 0, maxval = 0;
        for (int it = 0; it < reps + 2; ++it) {
            CK(newcols());
            CK(hipMemsetAsync(D0, 0, 8 * SL * SL * sizeof(T), s)); CK(hipMemsetAsync(X0, 0, 8 * SL * SL * sizeof(T), s));
            CK(hipMemsetAsync(D1, 0, 8 * SL * SL * sizeof(T), s)); CK(hipMemsetAsync(X1, 0, 8 * SL * SL * sizeof(T), s));
            // reference: full builds with the staged kernels (block pair y: previous block at cols[256 y], own at cols[256 y + 128])
            SyrkBatch sb{}; GramBatch gb{}; StripBatch st{};
            sb.count = gb.count = st.count = c.count;
            for (int y = 0; y < c.count; ++y) {
                sb.off[y] = 256 * y + 128; sb.nb[y] = c.nb; sb.dst[y] = int64_t(y) * SL * SL;
                gb.moff[y] = 256 * y + 128; gb.m[y] = c.nb; gb.noff[y] = 256 * y; gb.nn[y] = c.nbp; gb.dst[y] = int64_t(y) * SL * SL;
                st.voff[y] = 256 * y + 128 + (c.nb - c.m); st.m[y] = c.m; st.c0off[y] = 256 * y; st.c0n[y] = c.nbp;
                st.c1off[y] = 256 * y + 128; st.c1n[y] = c.nb; st.row0[y] = c.nb - c.m;
                st.dstX[y] = int64_t(y) * SL * SL; st.dstD[y] = int64_t(y) * SL * SL;
            }
            CK(hipEventRecord(a, s));
            launch_syrk_batch<T>(V, w, cols, sb, xm, true, D0, SL, work, s);
            if (c.nbp > 0) launch_gram_batch<T>(V, w, cols, gb, xm, true, X0, SL, work, s);
            CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); if (it >= 2) t_old += ms;
            CK(hipEventRecord(a, s));
            launch_strip_batch<T>(V, w, cols, st, xm, true, D1, X1, SL, work, s);
            CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
            CK(hipEventElapsedTime(&ms, a, b)); if (it >= 2) t_strip += ms;
            if (it == 0) {
                std::vector<T> d0(8 * SL * SL), x0(8 * SL * SL), d1(8 * SL * SL), x1(8 * SL * SL);
                CK(hipMemcpy(d0.data(), D0, d0.size() * sizeof(T), hipMemcpyDeviceToHost)); CK(hipMemcpy(x0.data(), X0, x0.size() * sizeof(T), hipMemcpyDeviceToHost));
                CK(hipMemcpy(d1.data(), D1, d1.size() * sizeof(T), hipMemcpyDeviceToHost)); CK(hipMemcpy(x1.data(), X1, x1.size() * sizeof(T), hipMemcpyDeviceToHost));
                for (int y = 0; y < c.count; ++y)
                    for (int r = 0; r < SL; ++r)
                        for (int q = 0; q < SL; ++q) {
                            const size_t o = size_t(y) * SL * SL + r + size_t(q) * SL;
                            const bool newrow = r >= c.nb - c.m && r < c.nb, newcol = q >= c.nb - c.m && q < c.nb;
                            // strip writes: D rows/cols of the new members (within nb), X rows of the new members (cols < nbp)
                            const bool dw = (newrow && q < c.nb) || (newcol && r < c.nb);
                            const double ed = dw ? std::fabs(double(d1[o]) - double(d0[o])) : std::fabs(double(d1[o]));
                            const bool xw = newrow && q < c.nbp;
                            const double ex = xw ? std::fabs(double(x1[o]) - double(x0[o])) : std::fabs(double(x1[o]));
                            maxerr = std::max(maxerr, std::max(ed, ex));
                            maxval = std::max(maxval, std::fabs(double(d0[o])));
                            if (dw && d1[o] != d1[size_t(y) * SL * SL + q + size_t(r) * SL]) maxerr = 1e30; // symmetry
                        }
            }
        }
        printf("%s m=%2d nbp=%3d nb=%3d count=%d: strip %.1f us   staged full (syrk+gram) %.1f us   max|diff| %.3g (max|D| %.3g)\n",
               sizeof(T) == 8 ? "f64" : "f32", c.m, c.nbp, c.nb, c.count, 1e3 * t_strip / reps, 1e3 * t_old / reps, maxerr, maxval);
    }
    return 0;
}
int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 100000, p = argc > 2 ? atoll(argv[2]) : 4096;
    if (const char* e = getenv("STRIP_LDS")) set_strip_lds(atoi(e) != 0);
    if (const char* e = getenv("STRIP_WGS")) set_strip_workgroups(atoi(e));
    if (run<double>(n, p)) return 1;
    if (run<float>(n, p)) return 1;
    return 0;
}
