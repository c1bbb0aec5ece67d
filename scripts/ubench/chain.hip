// chain.hip — gate for a persistent, flag-synchronised form of the look-ahead chain.
//
// Two arms over the SAME device functions and the same access pattern as a headline pass (100k x 10k f64, blocks of 128
// random columns; the columns a stage reads for its gradients (phase B) are read again two stages later for the residual
// update (phase A)):
//   L  one launch per block: workgroup 0 = solve of block s (sums the slice partials of the previous launch, fetches the
//      diagonal and the cross block, 128 dependent visits in one wavefront), workgroups 1.. = step (A: r -= X[:, cols(s-1)] d,
//      B: partial gradients of block s+1) — the shape of panel_fused_kernel;
//   P  ONE launch for the whole pass: the step workgroups loop over the stages, each waiting only for the solve two stages back
//      (device flag), the solve workgroup waits for the counter of the step workgroups' partials; r stays in LDS.
// Prints microseconds per block for both arms; `reps` = row groups a step workgroup takes in turn (1: 196 workgroups,
// 2: 98, ...), to see how many CUs the chain needs to hold its bandwidth.
#include <hip/hip_runtime.h>
#include "../../adelie_amd/csrc/wavered.hpp"
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int PB = 128, RS = 128 /* rows per slice */, FS = 4;

struct Ctl { // device control block of the persistent arm
    int solve_done;  // blocks solved so far
    int pad0[31];
    int step_done;   // arrivals of step workgroups (monotone)
    int pad1[31];
    int abort_flag;
};

struct Args {
    const double* X; int64_t n, ld;
    const double* w; double* r;
    const int32_t* cols; // [S + 2][128] columns of block t
    double* part;        // [2][NWMAX * 128]
    int32_t* dcol;       // [2][128]
    double* dlt;         // [2][128]
    int32_t* nz;         // [2]
    const double* Dpool; // [S][128*128]
    const double* Cpool;
    Ctl* ctl;
    int S, reps, nwg; // stages, row groups per workgroup, step workgroups
};

template <bool COH, class T> __device__ __forceinline__ T ld_coh(const T* p) {
    if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
template <bool COH, class T> __device__ __forceinline__ void st_coh(T* p, T v) {
    if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
__device__ __forceinline__ bool spin_until(const int* p, int target, int* abort_flag) {
    int it = 0;
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (++it > (1 << 22)) { __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
        if ((it & 1023) == 0 && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
    }
    return true;
}

// one step stage of one 1024-thread workgroup over its row group `grp` (4 slices of 128 rows): phase A with the changes in
// (dcs, dls, nzs) — LDS copies —, phase B partial gradients of `cols`; r slice in `rl` (LDS, 512 values) when RLDS, else global
template <bool RLDS>
__device__ __forceinline__ void step_stage(const Args& a, int64_t grp, const int32_t* dcs, const double* dls, int nzs,
                                           const int32_t* cols, int nb, double* smem, double* rl, double* outp /* [128] LDS */) {
    const int sub = threadIdx.x >> 8, tid = threadIdx.x & 255, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    double* red = smem + size_t(sub) * 5 * RS; // [4][RS]
    double* wrs = red + 4 * RS;
    const int64_t slice = grp * FS + sub;
    const int64_t i = slice * RS + int64_t(lane) * 2;
    const bool ok = i + 2 <= a.n;
    const int64_t ii = ok ? i : 0;
    constexpr int UB = 8, U = 16;
    d2 xb[UB]; int jb[UB];
    if (nb > 0) {
#pragma unroll
        for (int u = 0; u < UB; ++u) jb[u] = cols[min(wv + 4 * u, nb - 1)];
#pragma unroll
        for (int u = 0; u < UB; ++u) xb[u] = __builtin_nontemporal_load(reinterpret_cast<const d2*>(a.X + int64_t(jb[u]) * a.ld + ii));
    }
    if (nzs > 0) {
        double acc0 = 0, acc1 = 0;
        for (int m0 = wv; m0 < nzs; m0 += 4 * U) {
            d2 xa[U]; int ja[U]; double cf[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int m = m0 + 4 * u;
                ja[u] = __builtin_amdgcn_readfirstlane(dcs[min(m, nzs - 1)]);
                const double c = dls[min(m, nzs - 1)];
                cf[u] = m < nzs ? c : 0.0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) xa[u] = __builtin_nontemporal_load(reinterpret_cast<const d2*>(a.X + int64_t(ja[u]) * a.ld + ii));
#pragma unroll
            for (int u = 0; u < U; ++u) { acc0 = fma(cf[u], xa[u][0], acc0); acc1 = fma(cf[u], xa[u][1], acc1); }
        }
        red[wv * RS + lane * 2] = acc0; red[wv * RS + lane * 2 + 1] = acc1;
        __syncthreads();
        if (wv == 0) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int q = lane * 2 + e;
                double wr = 0;
                if (ok) {
                    const double r0 = RLDS ? rl[sub * RS + q] : a.r[i + e];
                    const double rr = r0 - ((red[q] + red[RS + q]) + (red[2 * RS + q] + red[3 * RS + q]));
                    if (RLDS) rl[sub * RS + q] = rr; else a.r[i + e] = rr;
                    wr = a.w[i + e] * rr;
                }
                wrs[q] = wr;
            }
        }
    } else if (wv == 0) {
#pragma unroll
        for (int e = 0; e < 2; ++e) wrs[lane * 2 + e] = ok ? a.w[i + e] * (RLDS ? rl[sub * RS + lane * 2 + e] : a.r[i + e]) : 0.0;
    }
    __syncthreads();
    if (nb > 0) {
        const double wr0 = wrs[lane * 2], wr1 = wrs[lane * 2 + 1];
        for (int c0 = wv; c0 < nb; c0 += 4 * UB) {
            if (c0 != wv) {
#pragma unroll
                for (int u = 0; u < UB; ++u) jb[u] = cols[min(c0 + 4 * u, nb - 1)];
#pragma unroll
                for (int u = 0; u < UB; ++u) xb[u] = __builtin_nontemporal_load(reinterpret_cast<const d2*>(a.X + int64_t(jb[u]) * a.ld + ii));
            }
            double pu[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) pu[u] = ok ? fma(xb[u][0], wr0, xb[u][1] * wr1) : 0.0;
            const double tot = ahip::reduce8(pu, lane);
            if (lane < UB && c0 + 4 * lane < nb) red[c0 + 4 * lane] = tot; // (red[0][..] is free after phase A)
        }
    }
    __syncthreads();
    if (nb > 0 && threadIdx.x < nb) {
        const int c = threadIdx.x;
        outp[c] = (smem[c] + smem[size_t(5) * RS + c]) + (smem[size_t(10) * RS + c] + smem[size_t(15) * RS + c]);
    }
    __syncthreads();
}

// fake solve of block s: same loads as blk_solve_la_body (partials, D, C), 128 dependent visits in wave 0, all coordinates change
template <bool COH>
__device__ __forceinline__ void solve_stage(const Args& a, int s, char* smem_raw, double* dprev /* LDS [128] own previous changes */) {
    double* D = reinterpret_cast<double*>(smem_raw);
    double* corr = D + PB * PB; // [16][128]
    double* gsum = corr + 16 * PB; // [8][128]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const d2* dsrc = reinterpret_cast<const d2*>(a.Dpool + size_t(s) * PB * PB);
    const d2* csrc = reinterpret_cast<const d2*>(a.Cpool + size_t(s) * PB * PB);
    d2 dreg[8], creg[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) dreg[u] = dsrc[tid + u * 1024];
#pragma unroll
    for (int u = 0; u < 8; ++u) creg[u] = csrc[tid + u * 1024];
#pragma unroll
    for (int u = 0; u < 8; ++u) reinterpret_cast<d2*>(D)[tid + u * 1024] = dreg[u];
    {
        const int rowc = tid % 64, cg = tid / 64;
        double a0 = 0, a1 = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) { const double dl = dprev[cg + 16 * u]; a0 = fma(creg[u][0], dl, a0); a1 = fma(creg[u][1], dl, a1); }
        corr[cg * PB + rowc * 2] = a0; corr[cg * PB + rowc * 2 + 1] = a1;
    }
    {
        const int c = tid & 127, k0 = tid >> 7;
        const double* pc = a.part + size_t(s & 1) * (size_t(a.nwg) * PB) + c;
        double v[25];
#pragma unroll
        for (int u = 0; u < 25; ++u) v[u] = ld_coh<COH>(pc + int64_t(min(k0 + 8 * u, a.nwg - 1)) * PB);
        double ps = 0;
#pragma unroll
        for (int u = 0; u < 25; ++u) ps += (k0 + 8 * u < a.nwg) ? v[u] : 0.0;
        gsum[k0 * PB + c] = ps;
    }
    __syncthreads();
    if (wv == 0) {
        double g0 = 0, g1 = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) { g0 += gsum[q * PB + lane]; g1 += gsum[q * PB + lane + 64]; }
#pragma unroll
        for (int q = 0; q < 16; ++q) { g0 -= corr[q * PB + lane]; g1 -= corr[q * PB + lane + 64]; }
        double nb0 = 0, nb1 = 0;
        for (int i = 0; i < 128; ++i) {
            const double dc0 = D[i * PB + lane], dc1 = D[i * PB + lane + 64];
            const int il = i & 63;
            const double gsrc = i < 64 ? g0 : g1;
            const double gcur = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(gsrc), il), __builtin_amdgcn_readlane(__double2loint(gsrc), il));
            const double gk = fma(0.5, 1.0, gcur);
            const double v = fabs(gk) - 1e-300;
            double ak = 0;
            if (v > 0) { const double x = copysign(v, gk); const double q0 = x * 0.5; const double r = fma(-q0, 2.0, x); ak = fma(r, 0.5, q0); }
            if (ak != 0.125) {
                const double del = (ak - 0.125) * 1e-30;
                g0 = fma(-del, dc0, g0); g1 = fma(-del, dc1, g1);
                if (lane == il) { if (i < 64) nb0 = del; else nb1 = del; }
            }
        }
        dprev[lane] = nb0; dprev[lane + 64] = nb1;
        const int sl = s & 1;
        st_coh<COH>(a.dlt + sl * PB + lane, nb0 * 0.0); st_coh<COH>(a.dlt + sl * PB + lane + 64, nb1 * 0.0);
        st_coh<COH>(a.dcol + sl * PB + lane, a.cols[size_t(s) * PB + lane]);
        st_coh<COH>(a.dcol + sl * PB + lane + 64, a.cols[size_t(s) * PB + lane + 64]);
        if (lane == 0) st_coh<COH>(a.nz + sl, 128);
    }
    __syncthreads();
}

// ---- arm L: one launch per block -------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void stage_kernel(Args a, int s) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* smem = reinterpret_cast<double*>(smem_raw);
    if (blockIdx.x == 0) {
        if (s < 0) return;
        double* dprev = smem + (152 * 1024 / 8);
        if (threadIdx.x < 128) dprev[threadIdx.x] = 1e-30;
        __syncthreads();
        solve_stage<false>(a, s, smem_raw, dprev);
        return;
    }
    // step of launch s: applies block s-1's changes (slot (s-1)&1), prepares block s+1
    __shared__ int32_t dcs[128];
    __shared__ double dls[128];
    __shared__ double outp[128];
    const int sl = (s - 1) & 1;
    const int nzs = s >= 1 ? a.nz[sl] : 0;
    if (threadIdx.x < 128) { dcs[threadIdx.x] = a.dcol[sl * PB + threadIdx.x]; dls[threadIdx.x] = a.dlt[sl * PB + threadIdx.x]; }
    __syncthreads();
    for (int rep = 0; rep < a.reps; ++rep) {
        const int64_t grp = (int64_t(blockIdx.x) - 1) * a.reps + rep;
        if (grp * FS * RS >= a.n) break;
        step_stage<false>(a, grp, dcs, dls, nzs, a.cols + size_t(s + 1) * PB, (s + 1 < a.S) ? 128 : 0, smem, nullptr, outp);
        if (s + 1 < a.S && threadIdx.x < 128) a.part[size_t((s + 1) & 1) * (size_t(a.nwg) * PB) + size_t(grp) * PB + threadIdx.x] = outp[threadIdx.x];
    }
}

// ---- arm P: one launch for the whole pass ----------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void chain_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* smem = reinterpret_cast<double*>(smem_raw);
    __shared__ int s_ok;
    Ctl* ctl = a.ctl;
    if (blockIdx.x == 0) {
        double* dprev = smem + (152 * 1024 / 8);
        if (threadIdx.x < 128) dprev[threadIdx.x] = 1e-30;
        __syncthreads();
        for (int s = 0; s < a.S; ++s) {
            // partials of block s come from step stage s (stage 0 = opening)
            if (threadIdx.x == 0) s_ok = spin_until(&ctl->step_done, (s + 1) * (int(gridDim.x) - 1), &ctl->abort_flag) ? 1 : 0;
            __syncthreads();
            if (!s_ok) return;
            solve_stage<true>(a, s, smem_raw, dprev);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(&ctl->solve_done, s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    __shared__ int32_t dcs[128];
    __shared__ double dls[128];
    __shared__ double outp[128];
    __shared__ int s_nz;
    double* rl = smem + 20 * RS; // this workgroup's residual rows, resident for the whole pass (reps * 512 values)
    for (int rep = 0; rep < a.reps; ++rep) {
        const int64_t i = ((int64_t(blockIdx.x) - 1) * a.reps + rep) * FS * RS + threadIdx.x;
        if (threadIdx.x < FS * RS) rl[rep * FS * RS + threadIdx.x] = i < a.n ? a.r[i] : 0.0;
    }
    __syncthreads();
    // stage q = 0 .. S: stage q applies block q-2's changes and prepares block q (same as launch q-1 of arm L)
    for (int q = 0; q <= a.S; ++q) {
        const int nb = q < a.S ? 128 : 0;
        int nzs = 0;
        if (q >= 2) {
            if (threadIdx.x == 0) s_ok = spin_until(&ctl->solve_done, q - 1, &ctl->abort_flag) ? 1 : 0;
            __syncthreads();
            if (!s_ok) return;
            const int sl = (q - 2) & 1;
            if (threadIdx.x < 128) {
                dcs[threadIdx.x] = ld_coh<true>(a.dcol + sl * PB + threadIdx.x);
                dls[threadIdx.x] = ld_coh<true>(a.dlt + sl * PB + threadIdx.x);
            }
            if (threadIdx.x == 0) s_nz = ld_coh<true>(a.nz + sl);
            __syncthreads();
            nzs = s_nz;
        }
        for (int rep = 0; rep < a.reps; ++rep) {
            const int64_t grp = (int64_t(blockIdx.x) - 1) * a.reps + rep;
            if (grp * FS * RS >= a.n) break;
            step_stage<true>(a, grp, dcs, dls, nzs, a.cols + size_t(q) * PB, nb, smem, rl + rep * FS * RS, outp);
            if (nb > 0 && threadIdx.x < 128)
                st_coh<true>(a.part + size_t(q & 1) * (size_t(a.nwg) * PB) + size_t(grp) * PB + threadIdx.x, outp[threadIdx.x]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(&ctl->step_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int rep = 0; rep < a.reps; ++rep) {
        const int64_t i = ((int64_t(blockIdx.x) - 1) * a.reps + rep) * FS * RS + threadIdx.x;
        if (threadIdx.x < FS * RS && i < a.n) a.r[i] = rl[rep * FS * RS + threadIdx.x];
    }
}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 100000, p = argc > 2 ? atoll(argv[2]) : 10000;
    const int S = argc > 3 ? atoi(argv[3]) : 48;
    const int64_t ld = n;
    const int ngrp = int((n + FS * RS - 1) / (FS * RS));
    double* X; CK(hipMalloc(&X, size_t(ld) * p * 8)); CK(hipMemset(X, 0, size_t(ld) * p * 8));
    double *w, *r, *dlt, *part, *Dp, *Cp; int32_t *cols, *dcol, *nz; Ctl* ctl;
    CK(hipMalloc(&w, n * 8)); CK(hipMalloc(&r, n * 8)); CK(hipMemset(w, 0, n * 8)); CK(hipMemset(r, 0, n * 8));
    CK(hipMalloc(&dlt, 2 * PB * 8)); CK(hipMalloc(&dcol, 2 * PB * 4)); CK(hipMalloc(&nz, 8));
    CK(hipMemset(dlt, 0, 2 * PB * 8)); CK(hipMemset(dcol, 0, 2 * PB * 4)); CK(hipMemset(nz, 0, 8));
    CK(hipMalloc(&part, size_t(2) * ngrp * PB * 8)); CK(hipMemset(part, 0, size_t(2) * ngrp * PB * 8));
    CK(hipMalloc(&Dp, size_t(S) * PB * PB * 8)); CK(hipMalloc(&Cp, size_t(S) * PB * PB * 8));
    CK(hipMemset(Dp, 0, size_t(S) * PB * PB * 8)); CK(hipMemset(Cp, 0, size_t(S) * PB * PB * 8));
    CK(hipMalloc(&cols, size_t(S + 2) * PB * 4)); CK(hipMalloc(&ctl, sizeof(Ctl)));
    std::mt19937 rng(1);
    std::vector<int32_t> hc(size_t(S + 2) * PB);
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t lds = 154 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(stage_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    for (int reps : {1, 2}) {
        const int nwg = (ngrp + reps - 1) / reps;
        Args a{X, n, ld, w, r, cols, part, dcol, dlt, nz, Dp, Cp, ctl, S, reps, ngrp};
        for (int round = 0; round < 3; ++round) {
            for (auto& c : hc) c = int32_t(rng() % p);
            CK(hipMemcpy(cols, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
            // arm L: opening launch (s = -1) + S fused launches
            CK(hipMemset(nz, 0, 8));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, st));
            for (int s = -1; s < S; ++s) hipLaunchKernelGGL(stage_kernel, dim3(nwg + 1), dim3(1024), lds, st, a, s);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float msL; CK(hipEventElapsedTime(&msL, e0, e1));
            // arm P
            CK(hipMemset(ctl, 0, sizeof(Ctl))); CK(hipMemset(nz, 0, 8));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(chain_kernel, dim3(nwg + 1), dim3(1024), lds, st, a);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float msP; CK(hipEventElapsedTime(&msP, e0, e1));
            Ctl hctl; CK(hipMemcpy(&hctl, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
            const double bytes = 2.0 * PB * n * 8;
            printf("reps=%d step_wgs=%d S=%d | launch-per-block %.2f us/block (%.2f TB/s) | persistent %.2f us/block (%.2f TB/s) abort=%d solve_done=%d\n",
                   reps, nwg, S, 1e3 * msL / (S + 1), bytes / (msL / (S + 1) * 1e-3) / 1e12, 1e3 * msP / (S + 1),
                   bytes / (msP / (S + 1) * 1e-3) / 1e12, hctl.abort_flag, hctl.solve_done);
            fflush(stdout);
        }
    }
    return 0;
}
