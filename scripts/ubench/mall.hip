// mall.hip — what does the 256 MiB Infinity Cache do for the panel step's access pattern?
//
//  (1) re-read bandwidth of one buffer of S MB, streamed repeatedly, per load policy (plain / nt / sc1 / sc0 sc1):
//      capacity and bandwidth of the on-die levels as this access pattern sees them.
//  (2) the chain pattern of solver.hip::run_panel_passes: launch k reads block k-1 (phase A, second read of those
//      columns) and block k+1 (phase B, first read); blocks are W random columns of an n x p column-major f64 matrix,
//      row-sliced workgroups exactly like panel_step_kernel (wave wv takes columns wv, wv+4, ..; 16 B per lane).
//      Time per launch for every (policy A, policy B) pair and W in {64, 96, 128}, against the same bytes with no reuse.
//
// build: hipcc --offload-arch=gfx950 -O3 -o mall mall.hip ; run: ./mall [n p]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

enum { PLAIN = 0, NT = 1, SC1 = 2, SC0SC1 = 3, NTSC1 = 4 };
static const char* pol_name[] = {"plain", "nt", "sc1", "sc0sc1", "nt+sc1"};

template <int POL>
__device__ __forceinline__ void issue(d2& v, const d2* p) {
    if constexpr (POL == PLAIN) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p));
    else if constexpr (POL == NT) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p));
    else if constexpr (POL == SC1) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p));
    else if constexpr (POL == SC0SC1) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p));
    else asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(v) : "v"(p));
}
template <int U>
__device__ __forceinline__ void wait_all(d2 (&v)[U]) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < U; ++u) asm volatile("" : "+v"(v[u]));
}

// (1) contiguous stream, grid-stride, 8 loads of 16 B in flight per lane
template <int POL>
__global__ __launch_bounds__(256) void stream_kernel(const d2* __restrict__ buf, int64_t nvec, double* __restrict__ out) {
    constexpr int U = 8;
    double acc = 0;
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t i0 = int64_t(blockIdx.x) * 256 + threadIdx.x; i0 < nvec; i0 += stride * U) {
        d2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) issue<POL>(v[u], buf + min(i0 + u * stride, nvec - 1));
        wait_all<U>(v);
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y;
    }
    if (acc == 12345.678) out[0] = acc;
}

// (2) the panel step's pattern: workgroup = 128-row slice (256 threads: 4 waves, wave wv takes columns wv, wv+4, ...),
// lane = 2 consecutive rows (16 B); phase A over colsA (policy PA), phase B over colsB (policy PB)
template <int PA, int PB>
__global__ __launch_bounds__(256, 4) void step_kernel(const double* __restrict__ X, int64_t ld, int64_t n,
                                                      const int32_t* __restrict__ colsA, int na,
                                                      const int32_t* __restrict__ colsB, int nb, double* __restrict__ out) {
    constexpr int U = 16;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int64_t i = int64_t(blockIdx.x) * 128 + lane * 2;
    if (i + 2 > n) i = 0;
    double acc = 0;
    for (int m0 = wv; m0 < na; m0 += 4 * U) {
        d2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = colsA[min(m0 + 4 * u, na - 1)];
            issue<PA>(v[u], reinterpret_cast<const d2*>(X + int64_t(c) * ld + i));
        }
        wait_all<U>(v);
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y;
    }
    for (int m0 = wv; m0 < nb; m0 += 4 * U) {
        d2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = colsB[min(m0 + 4 * u, nb - 1)];
            issue<PB>(v[u], reinterpret_cast<const d2*>(X + int64_t(c) * ld + i));
        }
        wait_all<U>(v);
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x * 0.5 + v[u].y;
    }
    if (acc == 12345.678) out[0] = acc;
}

template <int PA, int PB>
static int run_chain(const double* X, int64_t ld, int64_t n, int32_t* d_cols, int nblocks, int W, bool reuse, double* out,
                     hipStream_t s, hipEvent_t a, hipEvent_t b) {
    const int ns = int((n + 127) / 128);
    const int launches = 2 * nblocks;
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a, s));
        for (int k = 0; k < launches; ++k) {
            // reuse: A = block k-1 (read as B two launches ago), B = block k+1
            // no reuse: A = a block that was last touched nblocks/2 launches ago
            const int ba = reuse ? (k + nblocks - 1) % nblocks : (k + nblocks / 2) % nblocks;
            const int bb = (k + 1) % nblocks;
            hipLaunchKernelGGL((step_kernel<PA, PB>), dim3(ns), dim3(256), 0, s, X, ld, n, d_cols + ba * 128, W,
                               d_cols + bb * 128, W, out);
        }
        CK(hipEventRecord(b, s));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        best = std::min(best, ms);
    }
    const double us = 1e3 * best / launches, bytes = 2.0 * W * n * 8;
    printf("  chain W=%3d A=%-7s B=%-7s %s: %6.1f us/launch  %5.2f TB/s\n", W, pol_name[PA], pol_name[PB],
           reuse ? "reuse   " : "no-reuse", us, bytes / (us * 1e-6) / 1e12);
    return 0;
}

template <int POL>
static int run_stream(const d2* buf, double mb, double* out, hipStream_t s, hipEvent_t a, hipEvent_t b) {
    const int64_t nvec = int64_t(mb * 1e6 / 16);
    const int reps = 30;
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((stream_kernel<POL>), dim3(2048), dim3(256), 0, s, buf, nvec, out);
    CK(hipEventRecord(a, s));
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((stream_kernel<POL>), dim3(2048), dim3(256), 0, s, buf, nvec, out);
    CK(hipEventRecord(b, s));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    printf("  stream %-7s %6.0f MB: %7.1f us  %5.2f TB/s\n", pol_name[POL], mb, 1e3 * ms / reps, mb * 1e6 / (ms / reps * 1e-3) / 1e12);
    return 0;
}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 100000, p = argc > 2 ? atoll(argv[2]) : 10000, ld = n;
    double* X;
    CK(hipMalloc(&X, ld * p * 8));
    CK(hipMemset(X, 0, ld * p * 8));
    double* out;
    CK(hipMalloc(&out, 64));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));

    printf("(1) repeated stream of one buffer\n");
    for (double mb : {16.0, 32.0, 64.0, 100.0, 150.0, 200.0, 240.0, 300.0, 400.0, 800.0, 2000.0}) {
        if (run_stream<PLAIN>(reinterpret_cast<const d2*>(X), mb, out, s, a, b)) return 1;
        if (run_stream<NT>(reinterpret_cast<const d2*>(X), mb, out, s, a, b)) return 1;
        if (run_stream<SC1>(reinterpret_cast<const d2*>(X), mb, out, s, a, b)) return 1;
        if (run_stream<SC0SC1>(reinterpret_cast<const d2*>(X), mb, out, s, a, b)) return 1;
    }

    printf("(2) chain pattern, n=%lld p=%lld\n", (long long)n, (long long)p);
    const int nblocks = 60;
    std::vector<int32_t> perm(p), h(nblocks * 128);
    for (int64_t j = 0; j < p; ++j) perm[j] = int32_t(j);
    std::mt19937 rng(1);
    std::shuffle(perm.begin(), perm.end(), rng);
    for (int k = 0; k < nblocks * 128; ++k) h[k] = perm[k % p];
    int32_t* d_cols;
    CK(hipMalloc(&d_cols, h.size() * 4));
    CK(hipMemcpy(d_cols, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (int W : {128, 96, 64, 32}) {
#define RC(PA, PB, RE) if (run_chain<PA, PB>(X, ld, n, d_cols, nblocks, W, RE, out, s, a, b)) return 1;
        RC(NT, NT, false)
        RC(NT, NT, true)
        RC(PLAIN, PLAIN, false)
        RC(PLAIN, PLAIN, true)
        RC(NT, PLAIN, true)
        RC(SC1, PLAIN, true)
        RC(NTSC1, PLAIN, true)
        RC(SC0SC1, PLAIN, true)
        RC(NT, SC1, true)
        RC(PLAIN, NT, true)
    }
    return 0;
}
