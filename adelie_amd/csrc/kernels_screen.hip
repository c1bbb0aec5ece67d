// kernels_screen.hip — the screening step of a BASIL iteration on the device.
//
// Reference: solver_base.hpp:273-403 (screen: strong rule, pivot rule with its slack and its fall-back) and
// optimization/search_pivot.hpp:7-62 (two-segment least-squares pivot of the sorted scores).  The host routine this replaces
// (solver.hip::Solver::screen) sorts the G scores once per lambda (~90 us at G = 1e4, ~480 us at 5e4) between the invariance
// sweep and the next fit; here the same decisions are taken by four small kernels that run on a side stream next to the
// speculative first pass of the next lambda, and the host reads a header (KKT flag, count) and the list of new screen groups
// from host-mapped memory.
//
// The result is the host routine's, bit for bit:
//   * order of the groups = ascending (score, group index), the total order the host sorts by; computed as a rank by counting
//     (rank_i = #{j : (s_j, j) < (s_i, i)}), exact and independent of any scheduling;
//   * the four running sums of search_pivot are accumulated sequentially, one lane per sum, in the host's order, with
//     floating-point contraction off; the per-candidate formulas are evaluated in parallel with the host's association;
//   * the minimum takes the first index among equal values (the host's strict `<`), NaNs never win.
#include "kernels.hpp"

#pragma clang fp contract(off)

namespace ahip {
namespace {

template <class T> struct KeyBits;
template <> struct KeyBits<double> {
    static __device__ uint64_t map(double v) {
        const uint64_t u = uint64_t(__double_as_longlong(v)), sign = uint64_t(1) << 63;
        return (u & sign) ? ~u : (u | sign);
    }
};
template <> struct KeyBits<float> {
    static __device__ uint64_t map(float v) {
        const uint32_t u = __float_as_uint(v), sign = uint32_t(1) << 31;
        return uint64_t((u & sign) ? ~u : (u | sign));
    }
};

// scores (solver_base.hpp:311-318) and their order-preserving integer keys; ranks zeroed for the counting kernel
template <class T>
__global__ void screen_key_kernel(ScreenArgs<T> a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.G) return;
    const T cap = a.alpha * a.lmda, pen = a.penalty[i];
    T wt;
    if (pen <= T(0)) wt = cap;
    else {
        const T q = a.abs_grad[i] / pen;
        wt = (cap < q) ? cap : q; // std::min(q, cap)
    }
    a.wt[i] = wt;
    a.key[i] = KeyBits<T>::map(wt);
    a.rank[i] = 0;
    // KKT at the lambda of the invariance step (solver_base.hpp:408-433): any violation raises the flag the select kernel reads
    if (a.slot[i] < 0 && a.abs_grad[i] > a.lmda * a.alpha * pen) atomicOr(a.flags, 1);
}

// number of keys in [j0, j1) below `ki` (LE: or equal to it); the keys are wave-uniform, i.e. scalar loads, eight per step
template <bool LE>
__device__ __forceinline__ int count_below(const uint64_t* __restrict__ key, int j0, int j1, uint64_t ki) {
    int cnt = 0, j = j0;
    for (; j + 8 <= j1; j += 8) {
        uint64_t k[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) k[u] = key[j + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) cnt += LE ? (k[u] <= ki) : (k[u] < ki);
    }
    for (; j < j1; ++j) cnt += LE ? (key[j] <= ki) : (key[j] < ki);
    return cnt;
}

// rank by counting: thread i counts the keys below (key_i, i) among the j of its split
__global__ __launch_bounds__(256) void screen_rank_kernel(const uint64_t* __restrict__ key, int G, int chunk,
                                                          int32_t* __restrict__ rank) {
    const int b0 = blockIdx.x * 256, b1 = min(G, b0 + 256);
    const int i = b0 + threadIdx.x;
    const int jlo = blockIdx.y * chunk, jhi = min(G, jlo + chunk);
    if (jlo >= jhi) return;
    const uint64_t ki = i < G ? key[i] : 0;
    // j below this block's groups: ties go before i; inside the block: the full pair comparison; above: strictly smaller keys
    int cnt = count_below<true>(key, jlo, min(jhi, b0), ki);
    const int s2 = max(jlo, b0), e2 = min(jhi, b1);
    for (int j = s2; j < e2; ++j) {
        const uint64_t kj = key[j];
        cnt += (kj < ki) || (kj == ki && j < i);
    }
    cnt += count_below<false>(key, max(jlo, b1), jhi, ki);
    if (i < G && cnt) atomicAdd(rank + i, cnt);
}

// order[r] = the group of rank r, with the sign bit set when it is in the screen set already; sorted[r] = its score
template <class T>
__global__ void screen_order_kernel(ScreenArgs<T> a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.G) return;
    const int r = a.rank[i];
    a.order[r] = a.slot[i] < 0 ? i : int32_t(uint32_t(i) | 0x80000000u);
    a.sorted[r] = a.wt[i];
}

constexpr int SEL_T = 1024, SEL_W = SEL_T / 64, SEL_TILE = 4096;

// exclusive position of a flagged thread among the flagged threads of the workgroup; `total` = their number
__device__ __forceinline__ int block_excl(bool f, int* wsum, int& total) {
    const uint64_t b = __ballot(f);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int pre = __popcll(b & ((uint64_t(1) << lane) - 1));
    if (lane == 0) wsum[w] = __popcll(b);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < SEL_W; ++k) {
        const int v = wsum[k];
        off += (k < w) ? v : 0;
        tot += v;
    }
    __syncthreads();
    total = tot;
    return off + pre;
}

// append, in group-index order, every group outside the screen set whose abs_grad exceeds thr(i)
template <class T, class Thr>
__device__ int append_above(const ScreenArgs<T>& a, int n_out, int* wsum, Thr thr) {
    int32_t* list = a.out + kScreenHeader;
    for (int base = 0; base < a.G; base += SEL_T) {
        const int i = base + int(threadIdx.x);
        const bool f = i < a.G && a.slot[i] < 0 && a.abs_grad[i] > thr(i);
        int tot;
        const int pos = block_excl(f, wsum, tot);
        if (f) {
            list[n_out + pos] = i;
            if (a.dev_list) a.dev_list[n_out + pos] = i;
        }
        n_out += tot;
    }
    return n_out;
}

template <class T>
__global__ __launch_bounds__(SEL_T) void screen_select_kernel(ScreenArgs<T> a) {
    __shared__ int wsum[SEL_W];
    __shared__ T red_v[SEL_W];
    __shared__ int red_i[SEL_W];
    __shared__ int sh_pivot;
    __shared__ T ytile[SEL_TILE];
    const int tid = threadIdx.x, G = a.G;
    int32_t* list = a.out + kScreenHeader;

    const int kkt = (__builtin_nontemporal_load(a.flags) & 1) ? 0 : 1; // raised by the key kernel
    __syncthreads();
    if (tid == 0) *a.flags = 0; // for the next request (same stream)
    const T lnext = a.lmda_next[kkt];
    const int nna = a.n_new_active[kkt], take = a.take[kkt];
    int n_out = 0, pivot_idx = -1;

    if (a.rule == 0) { // strong rule (:296-305)
        const T strong = (2 * lnext - a.lmda) * a.alpha;
        n_out = append_above(a, n_out, wsum, [&](int i) { return strong * a.penalty[i]; });
    } else {
        if (nna) {
            const int m = a.subset, base = G - m;
            // running sums of search_pivot over x_i = i, y_i = the i-th score of the subset (ascending): lane 0 sum x, 1 sum x^2,
            // 2 sum y, 3 sum y x — each in the host's sequential order.  The scores are gathered into LDS a tile at a time by
            // the whole workgroup (the gather through `order` is two dependent global loads per element: far too slow inside
            // a sequential loop); the four lanes then run over the tile with eight independent LDS reads in flight.
            T acc = T(0);
            T* dst = a.pre + size_t(tid & 3) * m;
            for (int t0 = 0; t0 < m; t0 += SEL_TILE) {
                const int tn = min(SEL_TILE, m - t0);
                for (int e = tid; e < tn; e += SEL_T) ytile[e] = a.sorted[base + t0 + e];
                __syncthreads();
                if (tid < 4) {
                    int e = 0;
                    for (; e + 8 <= tn; e += 8) {
                        T term[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const T x = T(t0 + e + u), y = ytile[e + u];
                            term[u] = tid == 0 ? x : tid == 1 ? x * x : tid == 2 ? y : y * x;
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            acc = (t0 + e + u == 0) ? term[u] : acc + term[u];
                            dst[t0 + e + u] = acc;
                        }
                    }
                    for (; e < tn; ++e) {
                        const T x = T(t0 + e), y = ytile[e];
                        const T term = tid == 0 ? x : tid == 1 ? x * x : tid == 2 ? y : y * x;
                        acc = (t0 + e == 0) ? term : acc + term;
                        dst[t0 + e] = acc;
                    }
                }
                __syncthreads();
            }
            __threadfence_block();
            __syncthreads();
            T best = T(1) / T(0);
            int best_i = 0;
            if (m > 1) {
                const T* xs = a.pre;
                const T* xq = a.pre + m;
                const T* ys = a.pre + 2 * size_t(m);
                const T* yx = a.pre + 3 * size_t(m);
                const T mT = T(m), y_mean = ys[m - 1] / mT;
                for (int i = 1 + tid; i < m; i += SEL_T) {
                    const T x = T(i), ip1 = T(i + 1);
                    const T x_sum = xs[i], xsq_sum = xq[i], y_sum = ys[i], yx_sum = yx[i];
                    const T t_bar = (ip1 * x - x_sum) / mT;
                    const T var_t = ((ip1 * x) * x - (2 * x) * x_sum + xsq_sum) - (mT * t_bar) * t_bar;
                    const T cov_ty = x * (y_sum - ip1 * y_mean) - (yx_sum - y_mean * x_sum);
                    const T b1 = cov_ty / var_t;
                    const T mse = (-b1 * b1) * var_t;
                    if (mse < best) { best = mse; best_i = i; } // ascending i within a thread: first index on ties
                }
            }
            // minimum over the workgroup: smaller value, then smaller index
            for (int o = 32; o; o >>= 1) {
                const T ov = __shfl_xor(best, o);
                const int oi = __shfl_xor(best_i, o);
                if (ov < best || (ov == best && oi < best_i)) { best = ov; best_i = oi; }
            }
            if ((tid & 63) == 0) { red_v[tid >> 6] = best; red_i[tid >> 6] = best_i; }
            __syncthreads();
            if (tid == 0) {
                T bv = red_v[0];
                int bi = red_i[0];
                for (int k = 1; k < SEL_W; ++k)
                    if (red_v[k] < bv || (red_v[k] == bv && red_i[k] < bi)) { bv = red_v[k]; bi = red_i[k]; }
                sh_pivot = bi;
            }
            __syncthreads();
            pivot_idx = sh_pivot;
            const int fpi = base + pivot_idx;
            // everything from the top of the order down to the pivot (:340-347) ...
            for (int t0 = 0; t0 < G - fpi; t0 += SEL_T) {
                const int t = t0 + tid, ii = G - 1 - t;
                const int g = ii >= fpi ? a.order[ii] : -1;
                const bool f = g >= 0;
                int tot;
                const int pos = block_excl(f, wsum, tot);
                if (f) {
                    list[n_out + pos] = g;
                    if (a.dev_list) a.dev_list[n_out + pos] = g;
                }
                n_out += tot;
            }
            // ... and the next `take` groups below it that are not screened yet (the slack, :348-358)
            int below = 0;
            for (int t0 = 0; t0 < fpi && below < take; t0 += SEL_T) {
                const int ii = fpi - 1 - (t0 + tid);
                const int g = ii >= 0 ? a.order[ii] : -1;
                const bool f = g >= 0;
                int tot;
                const int pos = block_excl(f, wsum, tot);
                if (f && below + pos < take) {
                    list[n_out + pos] = g;
                    if (a.dev_list) a.dev_list[n_out + pos] = g;
                }
                const int used = min(tot, take - below);
                n_out += used;
                below += used;
            }
        }
        if (n_out == 0 && !kkt) // fall-back (:363-371)
            n_out = append_above(a, n_out, wsum, [&](int i) { return lnext * a.penalty[i] * a.alpha; });
        if (n_out == 0 && !kkt) // progress guard of the host routine: KKT's own association of the same product
            n_out = append_above(a, n_out, wsum, [&](int i) { return lnext * a.alpha * a.penalty[i]; });
    }
    __syncthreads();
    if (tid == 0) {
        a.out[1] = kkt;
        a.out[2] = n_out;
        a.out[3] = pivot_idx;
        __threadfence_system();
        a.out[0] = a.seq; // last: the host accepts the block when it finds its own sequence number
    }
}

} // namespace

template <class T>
void launch_screen(const ScreenArgs<T>& a, bool need_order, hipStream_t s) {
    const int G = a.G;
    if (G <= 0) return;
    const int gb = (G + 255) / 256;
    hipLaunchKernelGGL((screen_key_kernel<T>), dim3(gb), dim3(256), 0, s, a);
    if (need_order) {
        int js = std::max(1, std::min(gb, 2048 / gb));
        int chunk = ((G + js - 1) / js + 255) / 256 * 256;
        js = (G + chunk - 1) / chunk;
        hipLaunchKernelGGL(screen_rank_kernel, dim3(gb, js), dim3(256), 0, s, a.key, G, chunk, a.rank);
        hipLaunchKernelGGL((screen_order_kernel<T>), dim3(gb), dim3(256), 0, s, a);
    }
    hipLaunchKernelGGL((screen_select_kernel<T>), dim3(1), dim3(SEL_T), 0, s, a);
}

namespace {
template <class T>
__global__ void screen_append_kernel(const char* __restrict__ img, AppendDst<T> d) {
    const AppendImage<T> L(d.Ng, d.Nv, d.cons != 0);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < d.Ng) {
        const int32_t b = reinterpret_cast<const int32_t*>(img + L.begin)[t];
        d.spen[d.ns_old + t] = reinterpret_cast<const T*>(img + L.pen)[t];
        d.sbegin[d.ns_old + t] = b;
        d.ssize[d.ns_old + t] = reinterpret_cast<const int32_t*>(img + L.size)[t];
        d.isact[d.ns_old + t] = int8_t(reinterpret_cast<const int32_t*>(img + L.isact)[t]);
        d.slot[reinterpret_cast<const int32_t*>(img + L.group)[t]] = b;
    }
    if (t < d.Nv) {
        d.beta[d.nv_old + t] = reinterpret_cast<const T*>(img + L.beta)[t];
        d.vcol[d.nv_old + t] = reinterpret_cast<const int32_t*>(img + L.vcol)[t];
        if (d.cons) {
            d.clo[d.nv_old + t] = reinterpret_cast<const T*>(img + L.lo)[t];
            d.chi[d.nv_old + t] = reinterpret_cast<const T*>(img + L.hi)[t];
            d.cmu[d.nv_old + t] = reinterpret_cast<const T*>(img + L.mu)[t];
        }
    }
}
} // namespace

template <class T>
void launch_screen_append(const void* image_dev, const AppendDst<T>& d, hipStream_t s) {
    const int nt = std::max(d.Ng, d.Nv);
    if (nt <= 0) return;
    hipLaunchKernelGGL((screen_append_kernel<T>), dim3((nt + 255) / 256), dim3(256), 0, s, static_cast<const char*>(image_dev), d);
}
template void launch_screen_append<double>(const void*, const AppendDst<double>&, hipStream_t);
template void launch_screen_append<float>(const void*, const AppendDst<float>&, hipStream_t);
template void launch_screen<double>(const ScreenArgs<double>&, bool, hipStream_t);
template void launch_screen<float>(const ScreenArgs<float>&, bool, hipStream_t);

} // namespace ahip
