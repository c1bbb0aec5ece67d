// snp_step.hip — the panel step on a 2-bit design in isolation, next to floors of its own access pattern.
//
// Config 4's sequential chain launches this step once per block of 64 visits (54.8 k times per path): (A) r -= X[:, 64 changed
// columns of the previous block] * delta, (B) partial gradients of the next block's 64 columns against w * r; 16 MB of 2-bit
// columns per launch at 500k rows.  One launch after another on one stream, block s reading the columns of block s (B) and of
// block s - 1 (A), as a pass does.  Arms (microseconds per launch, HIP events around `reps` back-to-back launches):
//   empty     a launch of the step's geometry that does nothing                       -> the launch boundary itself
//   loads     the step's loads only (words of A and B, the residual / weight rows), xor-folded, residual rewritten
//                                                                                     -> floor of the ACCESS PATTERN
//   decodeA / decodeB  the loads + the decode-and-multiply arithmetic of one phase, no LDS exchange, no reduction
//   old       panel_step_kernel<Snp, 16> (round 5: 489 workgroups x 256 threads, three dependent round trips)
//   new       panel_step_snp16_kernel (round 6: 245 x 512, every load up front)
//   new A / new B   the new kernel with nb = 0 / nz = 0
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I adelie_amd/csrc scripts/ubench/snp_step.hip -o scripts/ubench/snp_step
#include "../../adelie_amd/csrc/kernels_cd_panel.hip"
#include <cstdio>
#include <random>
#include <vector>
using namespace ahip;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(512) void empty_kernel(int* sink) {
    if (sink && threadIdx.x == 9999) sink[0] = 1;
}

// MODE 0: loads only; 1: + phase (A) arithmetic; 2: + phase (B) arithmetic (per-lane, no exchange)
template <int MODE>
__global__ __launch_bounds__(512) void floor_kernel(SnpAcc<double> X, int64_t n, const double* __restrict__ w, double* __restrict__ r,
                                                   const int32_t* __restrict__ dcol, const double* __restrict__ dlt,
                                                   const int32_t* __restrict__ cols, double* __restrict__ sink) {
    const int sub = threadIdx.x >> 8, t = threadIdx.x & 255, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int64_t slice = int64_t(blockIdx.x) * 2 + sub;
    const int64_t i = slice * 1024 + int64_t(lane) * 16;
    const bool in = i + 16 <= n;
    const int64_t wofs = in ? (i >> 4) : 0;
    unsigned xa[16], xb[16];
    int ja[16], jb[16];
    double cf[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { ja[u] = dcol[wv + 4 * u]; cf[u] = dlt[wv + 4 * u]; jb[u] = cols[wv + 4 * u]; }
#pragma unroll
    for (int u = 0; u < 16; ++u) xa[u] = reinterpret_cast<const unsigned*>(X.colptr(ja[u]))[wofs];
#pragma unroll
    for (int u = 0; u < 16; ++u) xb[u] = reinterpret_cast<const unsigned*>(X.colptr(jb[u]))[wofs];
    const int64_t q0 = i + 4 * wv;
    double rq[4], wq[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { rq[k] = in ? r[q0 + k] : 0.0; wq[k] = in ? w[q0 + k] : 0.0; }
    double acc = 0;
    if constexpr (MODE == 0) {
        unsigned f = 0;
#pragma unroll
        for (int u = 0; u < 16; ++u) f ^= xa[u] ^ (xb[u] >> 1);
        acc = double(f & 1u) * 1e-300;
    } else if constexpr (MODE == 1) {
        double a[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) a[e] = 0;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            double xx[16];
            snp16_decode<double, true>(xa[u], X.impute[ja[u]], i, n, xx);
#pragma unroll
            for (int e = 0; e < 16; ++e) a[e] = fma(cf[u], xx[e], a[e]);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) acc += a[e];
        acc = acc * 1e-300 + double(xb[3] & 1u) * 1e-300;
    } else {
        double wr[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) wr[e] = rq[e & 3] * wq[(e >> 2) & 3] + double(e);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            double xx[16];
            snp16_decode<double, true>(xb[u], X.impute[jb[u]], i, n, xx);
            double s = 0;
#pragma unroll
            for (int e = 0; e < 16; ++e) s = fma(xx[e], wr[e], s);
            acc += s;
        }
        acc = acc * 1e-300 + double(xa[3] & 1u) * 1e-300;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (in) r[q0 + k] = rq[k] + acc + wq[k] * 1e-300;
    if (sink && acc == 12345.678) sink[0] = acc;
}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 500000, p = argc > 2 ? atoll(argv[2]) : 16384;
    const int reps = argc > 3 ? atoi(argv[3]) : 400, B = 64;
    const int64_t ldb = (((n + 3) / 4 + 127) / 128) * 128;
    uint8_t* bits; CK(hipMalloc(&bits, size_t(ldb) * p));
    {
        std::vector<uint8_t> h(size_t(ldb) * 256);
        std::mt19937 rng(7);
        for (auto& b : h) { // 25 % ones, 5 % twos, 10 % missing per call
            unsigned v = 0;
            for (int k = 0; k < 4; ++k) { const unsigned u = rng() % 100; v |= (u < 25 ? 1u : u < 30 ? 2u : u < 40 ? 3u : 0u) << (2 * k); }
            b = uint8_t(v);
        }
        for (int64_t j = 0; j < p; j += 256) CK(hipMemcpy(bits + size_t(j) * ldb, h.data(), size_t(ldb) * std::min<int64_t>(256, p - j), hipMemcpyHostToDevice));
    }
    double *imp, *w, *r, *dlt, *part, *sink; int32_t *cols_all, *nz64, *nz0;
    CK(hipMalloc(&imp, p * 8)); CK(hipMalloc(&w, n * 8)); CK(hipMalloc(&r, n * 8)); CK(hipMalloc(&dlt, 128 * 8));
    CK(hipMalloc(&part, panel_part_elems(n) * 8)); CK(hipMalloc(&sink, 64));
    const int nblk = int(p / B);
    CK(hipMalloc(&cols_all, size_t(nblk) * B * 4)); CK(hipMalloc(&nz64, 4)); CK(hipMalloc(&nz0, 4));
    {
        std::vector<double> hi(p, 0.3889), hw(n, 1.0 / double(n)), hr(n, 0.25), hd(128, 1e-6);
        CK(hipMemcpy(imp, hi.data(), p * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), n * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(r, hr.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dlt, hd.data(), 128 * 8, hipMemcpyHostToDevice));
        std::vector<int32_t> hc(size_t(nblk) * B);
        std::mt19937 rng(3);
        for (auto& c : hc) c = int32_t(rng() % p);
        CK(hipMemcpy(cols_all, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
        const int v64 = B, v0 = 0;
        CK(hipMemcpy(nz64, &v64, 4, hipMemcpyHostToDevice)); CK(hipMemcpy(nz0, &v0, 4, hipMemcpyHostToDevice));
    }
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    SnpView V{bits, n, p, ldb};
    SnpAcc<double> acc{bits, ldb, imp};
    const unsigned nwg = unsigned((n + 2047) / 2048);
    const double mb = 2.0 * B * double(n) / 4 / 1e6;
    printf("# 2-bit panel step, n = %lld rows, %d + %d columns per launch = %.1f MB of columns, %d launches per arm\n", (long long)n, B, B, mb, reps);
    auto run = [&](const char* name, double bytes_mb, auto&& launch) -> int {
        for (int pass = 0; pass < 2; ++pass) {
            CK(hipEventRecord(e0, s));
            for (int it = 0; it < reps; ++it) {
                const int b = 1 + it % (nblk - 1);
                launch(cols_all + size_t(b - 1) * B, cols_all + size_t(b) * B);
            }
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (pass == 1) printf("%-10s %7.2f us per launch   %6.2f TB/s of column bytes\n", name, 1e3 * ms / reps, bytes_mb > 0 ? bytes_mb / (1e3 * ms / reps) : 0.0);
        }
        return 0;
    };
    if (run("empty", 0, [&](const int32_t*, const int32_t*) { hipLaunchKernelGGL(empty_kernel, dim3(nwg), dim3(512), 0, s, (int*)nullptr); })) return 1;
    if (run("loads", mb, [&](const int32_t* dc, const int32_t* c) { hipLaunchKernelGGL((floor_kernel<0>), dim3(nwg), dim3(512), 0, s, acc, n, w, r, dc, dlt, c, sink); })) return 1;
    if (run("decodeA", mb, [&](const int32_t* dc, const int32_t* c) { hipLaunchKernelGGL((floor_kernel<1>), dim3(nwg), dim3(512), 0, s, acc, n, w, r, dc, dlt, c, sink); })) return 1;
    if (run("decodeB", mb, [&](const int32_t* dc, const int32_t* c) { hipLaunchKernelGGL((floor_kernel<2>), dim3(nwg), dim3(512), 0, s, acc, n, w, r, dc, dlt, c, sink); })) return 1;
    if (run("old", mb, [&](const int32_t* dc, const int32_t* c) { step_launch<double, SnpAcc<double>, 16>(acc, n, w, r, dc, dlt, nz64, c, B, part, false, s); })) return 1;
    if (run("new", mb, [&](const int32_t* dc, const int32_t* c) { launch_panel_step_snp<double>(V, imp, w, r, dc, dlt, nz64, c, B, part, s, false); })) return 1;
    if (run("new A", mb / 2, [&](const int32_t* dc, const int32_t* c) { launch_panel_step_snp<double>(V, imp, w, r, dc, dlt, nz64, c, 0, part, s, false); })) return 1;
    if (run("new B", mb / 2, [&](const int32_t* dc, const int32_t* c) { launch_panel_step_snp<double>(V, imp, w, r, dc, dlt, nz0, c, B, part, s, false); })) return 1;
    return 0;
}
