// kernels_cd_block_group.hip — block Gauss–Seidel passes for problems with groups (q > 1), spread over the chip.
//
// Same structure as kernels_cd_block.hip (one solver workgroup per block, the rest of the chip applies the block's
// changes to every screen value and gathers the next diagonal block), with a block = consecutive GROUPS of the visiting
// list whose sizes add up to <= 128 values.  The group coordinate update is the reference's
// (solver_gaussian_pin_naive.hpp:109-164, bcd/unconstrained/newton.hpp:35-142, optimization/newton.hpp:28-65):
// rotate into the eigenbasis of the centred weighted Gram block, Newton root find for the norm, rotate back; one
// wavefront does it lane-parallel over the q coefficients with wave reductions.  Groups wider than 128 values fall back
// to the single-workgroup kernel (kernels_cd.hip).
#include "kernels.hpp"
#include "grp_solve_body.hpp"

namespace ahip {

namespace {

template <class T>
__global__ __launch_bounds__(256) void grp_gather_kernel(CdGrpBlkParams<T> p, int j) {
    __shared__ int32_t vmap[GBLK], goff[GBLK + 1], gq[GBLK], gss[GBLK], meta[2];
    if (j >= p.nblk) return;
    block_layout(p, j, vmap, goff, gq, gss, meta);
    gather_group_block(p, j, vmap, meta[1], blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}

template <class T, bool NAIVE>
__global__ __launch_bounds__(256) void grp_solve_kernel(CdGrpBlkParams<T> p, int j) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    grp_solve_body<T, NAIVE>(p, j, smem_raw);
}

template <class T>
__global__ __launch_bounds__(256) void grp_update_kernel(CdGrpBlkParams<T> p, int j) {
    __shared__ T red[4][64];
    __shared__ int32_t vmap[GBLK], goff[GBLK + 1], gq[GBLK], gss[GBLK], meta[2];
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
    if (j + 1 < p.nblk) { // gather the next block's diagonal block
        block_layout(p, j + 1, vmap, goff, gq, gss, meta);
        gather_group_block(p, j + 1, vmap, meta[1], blockIdx.x * 256 + tid, gridDim.x * 256);
    }
}


} // namespace

template <class T>
void launch_cd_group_block_pass(const CdGrpBlkParams<T>& p, hipStream_t s) {
    if (p.nblk <= 0) return;
    const unsigned ug = unsigned((p.nv + 63) / 64);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(grp_solve_kernel<double, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, int(grp_solve_lds_total<double>()));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(grp_solve_kernel<float, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, int(grp_solve_lds_total<float>()));
        attr_done = true;
    }
    const size_t lds = grp_solve_lds_total<T>();
    hipLaunchKernelGGL((grp_gather_kernel<T>), dim3(64), dim3(256), 0, s, p, 0);
    for (int j = 0; j < p.nblk; ++j) {
        hipLaunchKernelGGL((grp_solve_kernel<T, false>), dim3(1), dim3(256), lds, s, p, j);
        hipLaunchKernelGGL((grp_update_kernel<T>), dim3(ug), dim3(256), 0, s, p, j);
    }
}

// Blocks [j0, j1) of a pass (a pass whose list holds groups that are visited on the host is enqueued in pieces): the first piece
// of a pass gathers block 0's diagonal block itself, every update kernel gathers the one of the block behind it.
template <class T>
void launch_cd_group_block_range(const CdGrpBlkParams<T>& p, int j0, int j1, hipStream_t s) {
    if (j1 <= j0) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(grp_solve_kernel<T, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, int(grp_solve_lds_total<T>()));
    const unsigned ug = unsigned((p.nv + 63) / 64);
    const size_t lds = grp_solve_lds_total<T>();
    if (j0 == 0) hipLaunchKernelGGL((grp_gather_kernel<T>), dim3(64), dim3(256), 0, s, p, 0);
    for (int j = j0; j < j1; ++j) {
        hipLaunchKernelGGL((grp_solve_kernel<T, false>), dim3(1), dim3(256), lds, s, p, j);
        hipLaunchKernelGGL((grp_update_kernel<T>), dim3(ug), dim3(256), 0, s, p, j);
    }
}
// the gradient update (and the gather of block j + 1) behind a block whose changes the HOST left in p.didx / p.dlt / st->nz
template <class T>
void launch_cd_group_block_update(const CdGrpBlkParams<T>& p, int j, hipStream_t s) {
    const unsigned ug = unsigned((p.nv + 63) / 64);
    hipLaunchKernelGGL((grp_update_kernel<T>), dim3(ug), dim3(256), 0, s, p, j);
}

// Layout descriptors of all blocks of a pass (CdGrpBlkParams::desc, GDESC_*): what block_layout derives per solve, once per
// pass and for all blocks in parallel.  One 128-thread workgroup per block.
template <class T>
__global__ __launch_bounds__(GBLK) void grp_layout_kernel(CdGrpBlkParams<T> p, int32_t* __restrict__ desc) {
    __shared__ int32_t gb_[GBLK], gq_[GBLK], gss_[GBLK], goff_[GBLK + 1];
    const int j = blockIdx.x, tid = threadIdx.x;
    int32_t* d = desc + size_t(j) * GDESC_STRIDE;
    const int g0 = p.blk_g0[j], g1 = p.blk_g0[j + 1];
    const int ng = g1 - g0;
    int ss = 0, b = 0, q = 0;
    if (tid < ng) {
        ss = p.list ? p.list[g0 + tid] : g0 + tid;
        b = p.sbegin[ss];
        q = p.ssize[ss];
    }
    gss_[tid] = ss; gb_[tid] = b; gq_[tid] = q;
    __syncthreads();
    if (tid == 0) {
        int o = 0;
        for (int k = 0; k < ng; ++k) { goff_[k] = o; o += gq_[k]; }
        for (int k = ng; k <= GBLK; ++k) goff_[k] = o;
    }
    __syncthreads();
    const int nval = goff_[ng];
    d[GDESC_GOFF + tid] = goff_[tid];
    if (tid == 0) { d[GDESC_GOFF + GBLK] = goff_[GBLK]; d[GDESC_NG] = ng; d[GDESC_NVAL] = nval; }
    d[GDESC_GQ + tid] = gq_[tid];
    d[GDESC_GSS + tid] = gss_[tid];
    if (tid < ng) {
        const int o = goff_[tid];
        const int32_t vo = q > 1 ? int32_t(p.voff[ss]) : 0;
        for (int t = 0; t < q; ++t) {
            d[GDESC_VMAP + o + t] = b + t;
            d[GDESC_VGRP + o + t] = tid;
            d[GDESC_VSS + o + t] = ss;
            d[GDESC_VOFF + o + t] = vo;
            d[GDESC_VPOS + o + t] = t;
            d[GDESC_VQ + o + t] = q;
        }
    }
    if (tid >= nval) {
        d[GDESC_VMAP + tid] = 0; d[GDESC_VGRP + tid] = 0; d[GDESC_VSS + tid] = 0;
        d[GDESC_VOFF + tid] = 0; d[GDESC_VPOS + tid] = 0; d[GDESC_VQ + tid] = 1;
    }
}

template <class T>
void launch_grp_layout(const CdGrpBlkParams<T>& p, int nblk, int32_t* desc, hipStream_t s) {
    if (nblk <= 0) return;
    hipLaunchKernelGGL((grp_layout_kernel<T>), dim3(unsigned(nblk)), dim3(GBLK), 0, s, p, desc);
}
template void launch_grp_layout<double>(const CdGrpBlkParams<double>&, int, int32_t*, hipStream_t);
template void launch_grp_layout<float>(const CdGrpBlkParams<float>&, int, int32_t*, hipStream_t);

// D <- R^T D R, R = blockdiag(V_k).  ONE workgroup of 1024 threads per block.  Groups of at most 16 values (the usual case): the
// block is copied into LDS with 16-byte loads, eight in flight per thread (leading dimension 129: both passes below are then
// free of bank conflicts — pass 1 walks down a column across the lanes, pass 2 along a row), the eigenbases of its groups next
// to it, and both passes run IN PLACE — pass 1 a work item = (row l, group k): the q entries D[l, o:o+q] into registers, q
// outputs back; pass 2 a work item = (group k, column c) likewise on T[o:o+q, c] — before the result is written out once,
// coalesced.  No global memory on the dependent loops (a first version that kept T = D R in a global scratch took 110 us per
// block: one L2 round trip per inner iteration; the second, 256 threads with scalar copies and an LDS stride of 128, 81 us in
// the path).  Blocks with a wider group take the scratch path (128 * 128 elements private to the stream).
constexpr int GROT_LD = GBLK + 1, GROT_NT = 1024, GROT_QM = 16;
template <class T>
constexpr size_t grp_rotate_lds() { return (size_t(GBLK) * GROT_LD + size_t(GROT_QM) * GBLK) * sizeof(T); }

template <class T>
__global__ __launch_bounds__(GROT_NT) void grp_block_rotate_kernel(T* Dptr, const T* Dsrc, const T* __restrict__ V, GrpRotArgs a,
                                                                   T* __restrict__ scratch) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    typedef T vec2 __attribute__((ext_vector_type(2)));
    constexpr int LD = GROT_LD, QM = GROT_QM, NT = GROT_NT;
    T* D = reinterpret_cast<T*>(smem_raw);
    T* Vl = D + size_t(GBLK) * LD; // eigenbasis of group k at Vl + 16 * goff[k]  (q*q <= 16*q)
    __shared__ int32_t vgrp[GBLK];
    __shared__ int32_t wide;
    const int tid = threadIdx.x;
    const int nval = a.goff[a.ng];
    if (tid == 0) wide = 0;
    __syncthreads();
    for (int k = tid; k < a.ng; k += NT) {
        for (int i = a.goff[k]; i < a.goff[k + 1]; ++i) vgrp[i] = k;
        if (a.goff[k + 1] - a.goff[k] > QM) wide = 1;
    }
    {   // all rows of the first nval columns, two values per load
        const int tot2 = nval * (GBLK / 2);
        constexpr int U = 8;
        const vec2* src = reinterpret_cast<const vec2*>(Dsrc);
        for (int e0 = tid; e0 < tot2; e0 += NT * U) {
            vec2 tmp[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = e0 + u * NT;
                tmp[u] = e < tot2 ? src[e] : vec2{T(0), T(0)};
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = e0 + u * NT;
                if (e < tot2) {
                    const int c = e / (GBLK / 2), r = (e % (GBLK / 2)) * 2;
                    D[r + c * LD] = tmp[u].x;
                    D[r + 1 + c * LD] = tmp[u].y;
                }
            }
        }
    }
    __syncthreads();
    if (!wide) {
        for (int k = tid / 64; k < a.ng; k += NT / 64) {
            const int o = a.goff[k], q = a.goff[k + 1] - o;
            if (q == 1) continue;
            for (int i = tid % 64; i < q * q; i += 64) Vl[QM * o + i] = V[a.voff[k] + i];
        }
        __syncthreads();
        for (int it = tid; it < nval * a.ng; it += NT) { // T[l, o + t] = sum_u D[l, o + u] V[u, t]
            const int l = it % nval, k = it / nval;
            const int o = a.goff[k], q = a.goff[k + 1] - o;
            if (q == 1) continue;
            const T* Vk = Vl + QM * o;
            T x[QM], y[QM];
#pragma unroll
            for (int u = 0; u < QM; ++u) x[u] = u < q ? D[l + (o + u) * LD] : T(0);
#pragma unroll
            for (int t = 0; t < QM; ++t) {
                T sacc = T(0);
                if (t < q) {
#pragma unroll
                    for (int u = 0; u < QM; ++u)
                        if (u < q) sacc = fma(x[u], Vk[u + t * q], sacc);
                }
                y[t] = sacc;
            }
#pragma unroll
            for (int t = 0; t < QM; ++t)
                if (t < q) D[l + (o + t) * LD] = y[t];
        }
        __syncthreads();
        for (int it = tid; it < nval * a.ng; it += NT) { // D~[o + s, c] = sum_u V[u, s] T[o + u, c]
            const int c = it % nval, k = it / nval;
            const int o = a.goff[k], q = a.goff[k + 1] - o;
            if (q == 1) continue;
            const T* Vk = Vl + QM * o;
            T x[QM], y[QM];
#pragma unroll
            for (int u = 0; u < QM; ++u) x[u] = u < q ? D[(o + u) + c * LD] : T(0);
#pragma unroll
            for (int sI = 0; sI < QM; ++sI) {
                T sacc = T(0);
                if (sI < q) {
#pragma unroll
                    for (int u = 0; u < QM; ++u)
                        if (u < q) sacc = fma(Vk[u + sI * q], x[u], sacc);
                }
                y[sI] = sacc;
            }
#pragma unroll
            for (int sI = 0; sI < QM; ++sI)
                if (sI < q) D[(o + sI) + c * LD] = y[sI];
        }
        __syncthreads();
        // rows [0, nval) in pairs (a pair may reach row nval, which still holds the value loaded above)
        const int rp = (nval + 1) / 2, tot2 = nval * rp;
        vec2* dst = reinterpret_cast<vec2*>(Dptr);
        for (int e = tid; e < tot2; e += NT) {
            const int c = e / rp, r = (e % rp) * 2;
            dst[(r + c * GBLK) / 2] = vec2{D[r + c * LD], D[r + 1 + c * LD]};
        }
        return;
    }
    for (int e = tid; e < nval * nval; e += NT) { // T[l, c] = sum_u D[l, o + u] V[u, t],  c = o + t
        const int l = e % nval, c = e / nval;
        const int k = vgrp[c], o = a.goff[k], q = a.goff[k + 1] - o;
        T s;
        if (q == 1) {
            s = D[l + c * LD];
        } else {
            const T* Vt = V + a.voff[k] + int64_t(c - o) * q;
            s = T(0);
            for (int u = 0; u < q; ++u) s = fma(D[l + (o + u) * LD], Vt[u], s);
        }
        scratch[l + c * GBLK] = s;
    }
    __threadfence_block();
    __syncthreads();
    for (int e = tid; e < nval * nval; e += NT) { // D~[r, c] = sum_u V[u, s] T[o + u, c],  r = o + s
        const int r = e % nval, c = e / nval;
        const int k = vgrp[r], o = a.goff[k], q = a.goff[k + 1] - o;
        T s;
        if (q == 1) {
            s = scratch[r + c * GBLK];
        } else {
            const T* Vs = V + a.voff[k] + int64_t(r - o) * q;
            s = T(0);
            for (int u = 0; u < q; ++u) s = fma(Vs[u], scratch[(o + u) + c * GBLK], s);
        }
        Dptr[r + c * GBLK] = s;
    }
}

template <class T>
void launch_grp_block_rotate(T* Dptr, const T* Dsrc, const T* V, const GrpRotArgs& a, T* scratch, hipStream_t s) {
    if (a.ng <= 0) return;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(grp_block_rotate_kernel<double>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, int(grp_rotate_lds<double>()));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(grp_block_rotate_kernel<float>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, int(grp_rotate_lds<float>()));
        attr_done = true;
    }
    hipLaunchKernelGGL((grp_block_rotate_kernel<T>), dim3(1), dim3(GROT_NT), grp_rotate_lds<T>(), s, Dptr, Dsrc, V, a, scratch);
}
template void launch_grp_block_rotate<double>(double*, const double*, const double*, const GrpRotArgs&, double*, hipStream_t);
template void launch_grp_block_rotate<float>(float*, const float*, const float*, const GrpRotArgs&, float*, hipStream_t);

template <class T>
void launch_cd_group_panel_solve(const CdGrpBlkParams<T>& p, int j, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(grp_solve_kernel<double, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, int(grp_solve_lds_total<double>()));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(grp_solve_kernel<float, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, int(grp_solve_lds_total<float>()));
        attr_done = true;
    }
    hipLaunchKernelGGL((grp_solve_kernel<T, true>), dim3(1), dim3(256), grp_solve_lds_total<T>(), s, p, j);
}
template void launch_cd_group_panel_solve<double>(const CdGrpBlkParams<double>&, int, hipStream_t);
template void launch_cd_group_panel_solve<float>(const CdGrpBlkParams<float>&, int, hipStream_t);

template void launch_cd_group_block_pass<double>(const CdGrpBlkParams<double>&, hipStream_t);
template void launch_cd_group_block_pass<float>(const CdGrpBlkParams<float>&, hipStream_t);
template void launch_cd_group_block_range<double>(const CdGrpBlkParams<double>&, int, int, hipStream_t);
template void launch_cd_group_block_range<float>(const CdGrpBlkParams<float>&, int, int, hipStream_t);
template void launch_cd_group_block_update<double>(const CdGrpBlkParams<double>&, int, hipStream_t);
template void launch_cd_group_block_update<float>(const CdGrpBlkParams<float>&, int, hipStream_t);

} // namespace ahip
