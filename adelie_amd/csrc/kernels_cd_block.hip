// kernels_cd_block.hip — block Gauss–Seidel form of the lasso coordinate-descent pass, spread over the whole chip.
//
// Same iteration as cd_lasso_kernel / cd_kernel (reference solver_gaussian_pin_naive.hpp:16-168, visiting order
// preserved), reorganised so that a single CU's 62 GB/s HBM stream is no longer what a pass waits for.  The visiting
// list of a pass (activation order, or screen order) is cut into blocks of B = 128 consecutive visits.  For block j
//
//   blk_solve_kernel   (ONE workgroup)  loads the B x B diagonal Gram block D_j (pre-gathered, contiguous) and the
//                      block's g / beta / A / penalty into LDS, and one wavefront runs the B coordinate updates
//                      strictly in order, keeping g_B current with  g_B -= D_j[:,i] * del_i  (LDS only, no barrier);
//                      it emits the block's non-zero changes (index, del).
//   blk_update_kernel  (nv/64 workgroups) applies those changes to every screen value,
//                      g_r -= sum_m C[r, idx_m] * del_m,  streaming the touched Gram columns at full-chip bandwidth
//                      (deterministic: fixed column order per row, 4-wave LDS reduction), and gathers D_{j+1}.
//
// Kernel boundaries on the stream are the only synchronisation (cheaper than any grid barrier on this chip, see
// guides/MI355X_MICROARCH.md "boundary" vs "barrier-xcd").  The result is the same Gauss–Seidel sequence: inside a block
// every coordinate sees all earlier changes of the block through D_j, across blocks through the update kernel.
#include <cstdlib>
#include "kernels.hpp"
#include "blk_solve_body.hpp"

namespace ahip {

namespace {

// D_j[i + m*BLK] = C[idx_i + idx_m*ldc] for the block starting at list position j*BLK
template <class T>
__device__ __forceinline__ void gather_block(const CdBlkParams<T>& p, int j, int gtid, int gthreads) {
    const int base = j * p.bsz;
    const int nb = min(p.bsz, p.count - base);
    if (nb <= 0) return;
    T* D = p.Dbuf + size_t(j & 1) * BLK * BLK;
    for (int e = gtid; e < nb * nb; e += gthreads) {
        const int i = e % nb, m = e / nb;
        const int64_t ri = blk_index(p, base + i), cm = blk_index(p, base + m);
        D[i + m * BLK] = p.C[ri + cm * p.ldc];
    }
}

template <class T>
__global__ void blk_gather_kernel(CdBlkParams<T> p, int j) {
    gather_block(p, j, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

template <class T, bool NAIVE>
__global__ __launch_bounds__(256) void blk_solve_kernel(CdBlkParams<T> p, int j) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    blk_solve_body<T, NAIVE>(p, j, smem_raw, threadIdx.x);
}

// the panel solve with one-coefficient box constraints (CdBlkParams::clo / chi / cmu)
template <class T>
__global__ __launch_bounds__(256) void blk_solve_cons_kernel(CdBlkParams<T> p, int j) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    blk_solve_body<T, true, true>(p, j, smem_raw, threadIdx.x);
}

template <class T>
__global__ __launch_bounds__(256) void blk_update_kernel(CdBlkParams<T> p, int j) {
    __shared__ T red[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nz = p.st->nz;
    const int r = blockIdx.x * 64 + lane;
    if (nz > 0) {
        T acc = T(0);
        if (r < p.nv) {
            // 8 independent column reads in flight per lane (the fixed summation order keeps the result deterministic)
            int m = wv;
            for (; m + 28 < nz; m += 32) {
                T c[8], d[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    c[u] = p.C[r + int64_t(p.didx[m + 4 * u]) * p.ldc];
                    d[u] = p.dlt[m + 4 * u];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = fma(c[u], d[u], acc);
            }
            for (; m < nz; m += 4) acc = fma(p.C[r + int64_t(p.didx[m]) * p.ldc], p.dlt[m], acc);
        }
        red[wv][lane] = acc;
        __syncthreads();
        if (wv == 0 && r < p.nv) p.g[r] -= ((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]));
    }
    // prefetch-gather of the next block's diagonal block while the chip is otherwise idle
    gather_block(p, j + 1, blockIdx.x * 256 + tid, gridDim.x * 256);
}

// ordered compaction of beta - beta0 into the (design column, delta) list of the residual update
template <class T>
__global__ __launch_bounds__(1024) void cd_compact_kernel(const T* __restrict__ beta, const T* __restrict__ beta0,
                                                          const int32_t* __restrict__ vcol, int nv,
                                                          int32_t* __restrict__ dcols, T* __restrict__ dvals,
                                                          int32_t* __restrict__ n_delta) {
    constexpr int NT = 1024;
    __shared__ int cnt[NT + 1];
    const int tid = threadIdx.x;
    const int chunk = (nv + NT - 1) / NT;
    const int a0 = tid * chunk, a1 = min(nv, a0 + chunk);
    int c = 0;
    for (int a = a0; a < a1; ++a) c += (beta[a] != beta0[a]) ? 1 : 0;
    cnt[tid + 1] = c;
    if (tid == 0) cnt[0] = 0;
    __syncthreads();
    if (tid == 0)
        for (int t = 1; t <= NT; ++t) cnt[t] += cnt[t - 1];
    __syncthreads();
    int o = cnt[tid];
    for (int a = a0; a < a1; ++a) {
        const T b1 = beta[a], b0 = beta0[a];
        if (b1 != b0) { dcols[o] = vcol[a]; dvals[o] = b1 - b0; ++o; }
    }
    if (tid == 0) n_delta[0] = cnt[NT];
}


// The panel solve of one block as its own launch, with the one-round-trip prologue of the fused launch (blk_solve_la_body: all
// 1024 threads fetch, one wavefront visits): the plain passes of the IRLS paths run it once per block (config 4: 54 k
// launches per path, 29 us each with the 256-thread body and its four to five dependent round trips)
template <class T>
__global__ __launch_bounds__(1024) void blk_solve_la_kernel(CdBlkParams<T> p, int j) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    blk_solve_la_body<T>(p, j, smem_raw, threadIdx.x);
}

} // namespace

int cd_block_size() { return BLK; }

template <class T>
void launch_cd_block_pass(const CdBlkParams<T>& p, hipStream_t s) {
    if (p.count <= 0) return;
    const int nblk = (p.count + p.bsz - 1) / p.bsz;
    const unsigned ug = unsigned((p.nv + 63) / 64);
    hipLaunchKernelGGL((blk_gather_kernel<T>), dim3(64), dim3(256), 0, s, p, 0);
    static bool attr_done = false;
    const size_t lds = blk_solve_lds<T>();
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(blk_solve_kernel<double, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, int(blk_solve_lds<double>()));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(blk_solve_kernel<float, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, int(blk_solve_lds<float>()));
        attr_done = true;
    }
    for (int j = 0; j < nblk; ++j) {
        hipLaunchKernelGGL((blk_solve_kernel<T, false>), dim3(1), dim3(256), lds, s, p, j);
        hipLaunchKernelGGL((blk_update_kernel<T>), dim3(ug), dim3(256), 0, s, p, j);
    }
}

template <class T>
void launch_cd_panel_solve(const CdBlkParams<T>& p, int j, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(blk_solve_kernel<double, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, int(blk_solve_lds<double>()));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(blk_solve_kernel<float, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, int(blk_solve_lds<float>()));
        attr_done = true;
    }
    if (p.clo != nullptr) {
        static bool cons_attr_done = false;
        if (!cons_attr_done) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(blk_solve_cons_kernel<double>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, int(blk_solve_lds<double>()));
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(blk_solve_cons_kernel<float>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, int(blk_solve_lds<float>()));
            cons_attr_done = true;
        }
        hipLaunchKernelGGL((blk_solve_cons_kernel<T>), dim3(1), dim3(256), blk_solve_lds<T>(), s, p, j);
        return;
    }
    {
        static bool la_attr_done = false;
        if (!la_attr_done) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(blk_solve_la_kernel<double>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, int(blk_solve_lds_la<double>()));
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(blk_solve_la_kernel<float>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, int(blk_solve_lds_la<float>()));
            la_attr_done = true;
        }
        hipLaunchKernelGGL((blk_solve_la_kernel<T>), dim3(1), dim3(1024), blk_solve_lds_la<T>(), s, p, j);
        return;
    }
    hipLaunchKernelGGL((blk_solve_kernel<T, true>), dim3(1), dim3(256), blk_solve_lds<T>(), s, p, j);
}

template <class T>
void launch_cd_compact(const T* beta, const T* beta0, const int32_t* vcol, int nv, int32_t* dcols, T* dvals,
                       int32_t* n_delta, hipStream_t s) {
    hipLaunchKernelGGL((cd_compact_kernel<T>), dim3(1), dim3(1024), 0, s, beta, beta0, vcol, nv, dcols, dvals, n_delta);
}

#define INST(T)                                                                                       \
    template void launch_cd_block_pass<T>(const CdBlkParams<T>&, hipStream_t);                        \
    template void launch_cd_panel_solve<T>(const CdBlkParams<T>&, int, hipStream_t);                  \
    template void launch_cd_compact<T>(const T*, const T*, const int32_t*, int, int32_t*, T*, int32_t*, hipStream_t);
INST(double)
INST(float)
#undef INST

} // namespace ahip
