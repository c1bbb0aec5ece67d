// kernels_sweep.hip — HBM-bound streaming kernels over the resident design (gfx950 / CDNA4).
//
//   sweep      out[c] = X[:,col(c)] . v            replaces MatrixNaiveDense::mul / bmul / cmul / sq_mul
//                                                  (reference matrix_naive_dense.ipp:23-34,61-80,125-146,199-217,
//                                                   kernels matrix/utils.hpp:131-161,194-269) and the invariance
//                                                  sweep of solver_gaussian_naive.hpp:377-393 (incl. the
//                                                  `grad -= resid_sum * X_means` epilogue, fused).
//   axpy_cols  out += sign * sum_k coef_k X[:,col_k]  replaces ctmul / btmul (matrix_naive_dense.ipp:49-59,106-123)
//                                                  and applies the residual update of a whole CD fit at once.
//   sp_tmul    out[l,:] = sum_e V[l,e] X[:,idx_e]  replaces MatrixNaiveDense::sp_tmul (:219-256).
//
// Design notes (MI355X): one column is n*sizeof(T) contiguous bytes (800 KB at n=100k f64).  A sweep block owns a
// panel of CB columns and walks the rows with 16-byte loads per lane (1 KiB per wave-instruction); the vector
// v = w*r (<= 4 MB) is read through L2 once per panel instead of once per column, the CB column streams are
// independent loads in flight (no LDS staging needed: there is no reuse of X), lanes reduce with wave shuffles
// (64-wide), waves through a few LDS words.  Grid = panels x row-splits >> 256 CUs; partial sums of the row
// splits are combined by a second tiny kernel so results are deterministic (no float atomics).
#include "kernels.hpp"
#include "accessors.hpp"
#include <cstdlib>

namespace ahip {

namespace {

constexpr int kThreads = 256;

template <class T>
__device__ __forceinline__ T wave_sum(T x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    return x;
}

template <class T, int VEC>
__device__ __forceinline__ Pack<T, VEC> load_vec(const T* p) {
    Pack<T, VEC> r;
    if constexpr (VEC == 1) r.v[0] = p[0];
    else {
        using V = typename VecOf<T>::type;
        V x = *reinterpret_cast<const V*>(p);
#pragma unroll
        for (int e = 0; e < VEC; ++e) r.v[e] = x[e];
    }
    return r;
}

// ---- sweep ----------------------------------------------------------------------------------------
template <class T, class Acc, int CB, int VEC, bool SQ>
__global__ __launch_bounds__(kThreads) void sweep_kernel(Acc X, const T* __restrict__ v, T* __restrict__ out,
                                                         int64_t n, int64_t c0, int64_t ncols,
                                                         const int32_t* __restrict__ cols, int64_t rows_per_split,
                                                         int nsplit, const T* __restrict__ sub_scale,
                                                         const T* __restrict__ sub_vec) {
    const int tid = threadIdx.x;
    const int64_t cb = blockIdx.x;
    const int split = blockIdx.y;
    const int64_t r0 = int64_t(split) * rows_per_split;
    const int64_t r1 = min(n, r0 + rows_per_split);

    int64_t cj[CB];
    decltype(X.colptr(0)) cp[CB];
#pragma unroll
    for (int k = 0; k < CB; ++k) {
        int64_t c = cb * CB + k;
        if (c >= ncols) c = ncols - 1; // duplicate work on the tail panel, discarded below
        cj[k] = cols ? int64_t(cols[c]) : c0 + c;
        cp[k] = X.colptr(cj[k]);
    }
    T acc[CB];
#pragma unroll
    for (int k = 0; k < CB; ++k) acc[k] = T(0);

    const int64_t body_end = r0 + ((r1 - r0) / VEC) * VEC;
#pragma unroll 2
    for (int64_t i = r0 + int64_t(tid) * VEC; i < body_end; i += int64_t(kThreads) * VEC) {
        const Pack<T, VEC> vv = load_vec<T, VEC>(v + i);
        Pack<T, VEC> xx[CB];
#pragma unroll
        for (int k = 0; k < CB; ++k) xx[k] = X.template load<VEC>(cp[k], i, cj[k]);
#pragma unroll
        for (int k = 0; k < CB; ++k)
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const T x = SQ ? xx[k].v[e] * xx[k].v[e] : xx[k].v[e];
                acc[k] = fma(x, vv.v[e], acc[k]);
            }
    }
    // row tail (< VEC rows)
    for (int64_t i = body_end + tid; i < r1; i += kThreads) {
        const T vi = v[i];
#pragma unroll
        for (int k = 0; k < CB; ++k) {
            const T x0 = X.template load<1>(cp[k], i, cj[k]).v[0];
            const T x = SQ ? x0 * x0 : x0;
            acc[k] = fma(x, vi, acc[k]);
        }
    }

    __shared__ T red[kThreads / 64][CB];
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int k = 0; k < CB; ++k) {
        const T s = wave_sum(acc[k]);
        if (lane == 0) red[wv][k] = s;
    }
    __syncthreads();
    if (tid < CB) {
        const int64_t c = cb * CB + tid;
        if (c < ncols) {
            T s = T(0);
#pragma unroll
            for (int w = 0; w < kThreads / 64; ++w) s += red[w][tid];
            if (nsplit == 1) {
                if (sub_vec) s -= sub_scale[0] * sub_vec[cols ? int64_t(cols[c]) : c0 + c];
                out[c] = s;
            } else {
                out[int64_t(split) * ncols + c] = s; // partial
            }
        }
    }
}

template <class T>
__global__ void sweep_reduce_kernel(const T* __restrict__ part, T* __restrict__ out, int64_t ncols, int nsplit,
                                    int64_t c0, const int32_t* __restrict__ cols, const T* __restrict__ sub_scale,
                                    const T* __restrict__ sub_vec) {
    const int64_t c = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= ncols) return;
    T s = T(0);
    for (int r = 0; r < nsplit; ++r) s += part[int64_t(r) * ncols + c];
    if (sub_vec) {
        const int64_t j = cols ? int64_t(cols[c]) : c0 + c;
        s -= sub_scale[0] * sub_vec[j];
    }
    out[c] = s;
}

constexpr int kSweepCB = 4;
// columns per sweep block (8 halves the re-reads of v through L2 and measured 6.66 against 6.85 TB/s)
inline int sweep_cb() { return kSweepCB; }

inline void sweep_shape(int64_t n, int64_t ncols, int vec, int64_t& blocks_c, int& nsplit, int64_t& rows_per_split) {
    const int cbv = sweep_cb();
    blocks_c = (ncols + cbv - 1) / cbv;
    const int64_t unit = int64_t(kThreads) * vec;      // rows per block iteration
    const int64_t max_split = (n + unit * 4 - 1) / (unit * 4); // >= 4 iterations per split
    int64_t want = (1024 + blocks_c - 1) / blocks_c;
    int64_t ns = want < 1 ? 1 : want;
    if (ns > max_split) ns = max_split;
    if (ns < 1) ns = 1;
    if (ns > 65535) ns = 65535;
    rows_per_split = (n + ns - 1) / ns;
    rows_per_split = ((rows_per_split + unit - 1) / unit) * unit;
    ns = (n + rows_per_split - 1) / rows_per_split;
    if (ns < 1) ns = 1;
    nsplit = int(ns);
}

template <class T, class Acc, int VEC>
void sweep_dispatch(Acc acc, const T* v, T* out, int64_t n, int64_t c0, int64_t ncols, const int32_t* cols,
                    const T* sub_scale, const T* sub_vec, bool square, T* work, hipStream_t s) {
    if (ncols <= 0) return;
    int64_t blocks_c, rows_per_split;
    int nsplit;
    sweep_shape(n, ncols, VEC, blocks_c, nsplit, rows_per_split);
    dim3 grid((unsigned)blocks_c, (unsigned)nsplit);
    T* dst = nsplit == 1 ? out : work;
    if (square && sweep_cb() == 8)
        hipLaunchKernelGGL((sweep_kernel<T, Acc, 8, VEC, true>), grid, dim3(kThreads), 0, s, acc, v, dst, n, c0,
                           ncols, cols, rows_per_split, nsplit, sub_scale, sub_vec);
    else if (square)
        hipLaunchKernelGGL((sweep_kernel<T, Acc, kSweepCB, VEC, true>), grid, dim3(kThreads), 0, s, acc, v, dst, n, c0,
                           ncols, cols, rows_per_split, nsplit, sub_scale, sub_vec);
    else if (sweep_cb() == 8)
        hipLaunchKernelGGL((sweep_kernel<T, Acc, 8, VEC, false>), grid, dim3(kThreads), 0, s, acc, v, dst, n, c0,
                           ncols, cols, rows_per_split, nsplit, sub_scale, sub_vec);
    else
        hipLaunchKernelGGL((sweep_kernel<T, Acc, kSweepCB, VEC, false>), grid, dim3(kThreads), 0, s, acc, v, dst, n, c0,
                           ncols, cols, rows_per_split, nsplit, sub_scale, sub_vec);
    if (nsplit > 1) {
        const int bt = 256;
        hipLaunchKernelGGL((sweep_reduce_kernel<T>), dim3((unsigned)((ncols + bt - 1) / bt)), dim3(bt), 0, s, work, out,
                           ncols, nsplit, c0, cols, sub_scale, sub_vec);
    }
}

// ---- axpy_cols --------------------------------------------------------------------------------------
template <class T, class Acc, int VEC>
__global__ __launch_bounds__(kThreads) void axpy_cols_kernel(Acc X, int64_t n, const int32_t* __restrict__ cols,
                                                             const T* __restrict__ coef,
                                                             const int32_t* __restrict__ count_dev, int32_t count,
                                                             T sign, T* __restrict__ out) {
    const int K = count_dev ? count_dev[0] : count;
    if (K <= 0) return;
    const int64_t i = (int64_t(blockIdx.x) * kThreads + threadIdx.x) * VEC;
    if (i >= n) return;
    if (i + VEC <= n) {
        T acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = T(0);
        int k = 0;
        for (; k + 4 <= K; k += 4) {
            Pack<T, VEC> xx[4];
            T cf[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t j = cols[k + u];
                cf[u] = coef[k + u];
                xx[u] = X.template load<VEC>(X.colptr(j), i, j);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] = fma(cf[u], xx[u].v[e], acc[e]);
        }
        for (; k < K; ++k) {
            const int64_t j = cols[k];
            const T cf = coef[k];
            const Pack<T, VEC> xx = X.template load<VEC>(X.colptr(j), i, j);
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] = fma(cf, xx.v[e], acc[e]);
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) out[i + e] += sign * acc[e];
    } else {
        for (int64_t r = i; r < n; ++r) {
            T acc = T(0);
            for (int k = 0; k < K; ++k) {
                const int64_t j = cols[k];
                acc = fma(coef[k], X.template load<1>(X.colptr(j), r, j).v[0], acc);
            }
            out[r] += sign * acc;
        }
    }
}

template <class T, class Acc, int VEC>
__global__ __launch_bounds__(kThreads) void sp_tmul_kernel(Acc X, int64_t n, const int64_t* __restrict__ indptr,
                                                           const int64_t* __restrict__ indices,
                                                           const T* __restrict__ values, T* __restrict__ out) {
    const int64_t l = blockIdx.y;
    const int64_t e0 = indptr[l], e1 = indptr[l + 1];
    const int64_t i = (int64_t(blockIdx.x) * kThreads + threadIdx.x) * VEC;
    if (i >= n) return;
    T* o = out + l * n;
    if (i + VEC <= n) {
        T acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = T(0);
        int64_t k = e0;
        for (; k + 4 <= e1; k += 4) {
            Pack<T, VEC> xx[4];
            T cf[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t j = indices[k + u];
                cf[u] = values[k + u];
                xx[u] = X.template load<VEC>(X.colptr(j), i, j);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] = fma(cf[u], xx[u].v[e], acc[e]);
        }
        for (; k < e1; ++k) {
            const int64_t j = indices[k];
            const T cf = values[k];
            const Pack<T, VEC> xx = X.template load<VEC>(X.colptr(j), i, j);
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] = fma(cf, xx.v[e], acc[e]);
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[i + e] = acc[e];
    } else {
        for (int64_t r = i; r < n; ++r) {
            T acc = T(0);
            for (int64_t k = e0; k < e1; ++k) {
                const int64_t j = indices[k];
                acc = fma(values[k], X.template load<1>(X.colptr(j), r, j).v[0], acc);
            }
            o[r] = acc;
        }
    }
}

template <class T>
__global__ void vmul_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, int64_t n) {
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x)
        out[i] = a[i] * b[i];
}
template <class T>
__global__ void fill_kernel(T* __restrict__ out, T value, int64_t n) {
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x)
        out[i] = value;
}

template <class T>
__global__ void abs_grad_kernel(const T* __restrict__ grad, const int64_t* __restrict__ groups,
                                const int64_t* __restrict__ group_sizes, int64_t G, const int32_t* __restrict__ slot,
                                const T* __restrict__ screen_beta, const T* __restrict__ penalty, T oma_lmda,
                                T* __restrict__ abs_grad) {
    const int64_t g = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const int64_t k = groups[g], sz = group_sizes[g];
    const int32_t sl = slot[g];
    T acc = T(0);
    if (sl >= 0) {
        const T regul = oma_lmda * penalty[g];
        for (int64_t t = 0; t < sz; ++t) {
            const T e = grad[k + t] - regul * screen_beta[sl + t];
            acc += e * e;
        }
    } else {
        for (int64_t t = 0; t < sz; ++t) acc += grad[k + t] * grad[k + t];
    }
    abs_grad[g] = sqrt(acc);
}

// update_abs_grad (solver_base.hpp:20-110) when some groups of one coefficient carry a box constraint
// lo <= beta <= hi (lo <= 0 <= hi, +-inf: none): a screened coordinate subtracts its multiplier (the constraint's gradient,
// :69-75), any other constrained one takes the multiplier that best explains its gradient at beta = 0 (solve_zero,
// constraint_box.ipp:268-284: free in the directions whose bound is exactly zero, zero elsewhere).  mu_out: every group's
// multiplier (what sparsify_dual reads off the constraint objects, :158-222).
template <class T>
__global__ void abs_grad_cons_kernel(const T* __restrict__ grad, const int64_t* __restrict__ groups,
                                     const int64_t* __restrict__ group_sizes, int64_t G,
                                     const int32_t* __restrict__ slot, const T* __restrict__ screen_beta,
                                     const T* __restrict__ penalty, T oma_lmda, const T* __restrict__ clo,
                                     const T* __restrict__ chi, const T* __restrict__ cmu, T* __restrict__ abs_grad,
                                     T* __restrict__ mu_out) {
    const int64_t g = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const T INF = T(1) / T(0), M = T(1e100); // Configs::max_solver_value (inf in single precision, as in the reference)
    const int64_t sz = group_sizes[g];
    if (sz != 1) { // groups of several coefficients carry no constraint: abs_grad_kernel's norm
        const int64_t k = groups[g];
        const int32_t sl0 = slot[g];
        T acc = T(0);
        for (int64_t t = 0; t < sz; ++t) {
            const T e0 = sl0 >= 0 ? grad[k + t] - oma_lmda * penalty[g] * screen_beta[sl0 + t] : grad[k + t];
            acc += e0 * e0;
        }
        abs_grad[g] = sqrt(acc);
        mu_out[g] = T(0);
        return;
    }
    const T v = grad[groups[g]], lo = clo[g], hi = chi[g];
    const bool constrained = lo != -INF || hi != INF;
    const int32_t sl = slot[g];
    T mu = T(0), e;
    if (sl >= 0) {
        if (constrained) mu = cmu[sl];
        e = v - oma_lmda * penalty[g] * screen_beta[sl] - mu;
    } else {
        if (constrained) mu = fmin(fmax(v, lo >= T(0) ? -M : T(0)), hi <= T(0) ? M : T(0));
        e = v - mu;
    }
    abs_grad[g] = fabs(e);
    mu_out[g] = mu;
}

template <class T>
__global__ void copy2d_kernel(const T* __restrict__ src, int64_t lds_, T* __restrict__ dst, int64_t ldd, int64_t rows,
                              int64_t cols) {
    const int64_t j = blockIdx.y;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < rows; i += int64_t(gridDim.x) * blockDim.x)
        dst[i + j * ldd] = src[i + j * lds_];
    (void)cols;
}

template <class T>
__global__ void diag_vars_kernel(const T* __restrict__ C, int64_t ldc, int32_t pos0, int32_t cnt, T* __restrict__ vars) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= cnt) return;
    const T d = C[int64_t(pos0 + a) * (ldc + 1)];
    vars[pos0 + a] = d > T(0) ? d : T(0);
}

// row-major (n,p) -> column-major with leading dimension ld, through a 32x33 LDS tile
template <class T>
__global__ void transpose_kernel(const T* __restrict__ src, int64_t n, int64_t p, T* __restrict__ dst, int64_t ld) {
    __shared__ T tile[32][33];
    const int64_t i0 = int64_t(blockIdx.y) * 32, j0 = int64_t(blockIdx.x) * 32;
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int64_t i = i0 + r, j = j0 + threadIdx.x;
        if (i < n && j < p) tile[r][threadIdx.x] = src[i * p + j];
    }
    __syncthreads();
    for (int c = threadIdx.y; c < 32; c += blockDim.y) {
        const int64_t j = j0 + c, i = i0 + threadIdx.x;
        if (i < n && j < p) dst[i + j * ld] = tile[threadIdx.x][c];
    }
}

__global__ void pack_snp_kernel(const int8_t* __restrict__ calldata, int64_t n, int64_t p, uint8_t* __restrict__ bits,
                                int64_t ldb) {
    const int64_t j = blockIdx.y;
    const int64_t nb = (n + 3) / 4;
    for (int64_t b = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; b < ldb; b += int64_t(gridDim.x) * blockDim.x) {
        unsigned byte = 0;
        if (b < nb) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int64_t i = b * 4 + k;
                unsigned code = 0;
                if (i < n) {
                    const int8_t c = calldata[i + j * n];
                    code = c < 0 ? 3u : unsigned(c);
                }
                byte |= code << (2 * k);
            }
        }
        bits[j * ldb + b] = uint8_t(byte);
    }
    (void)p;
}

// PLINK 1 .bed record (SNP-major) -> device 2-bit codes.  .bed: 2 bits per sample, low bits first,
// 00 = two copies of A1, 10 = one, 11 = none, 01 = missing; device code = count of A1 (0/1/2), 3 = missing.
// Each byte is four samples in both layouts, so a byte maps to a byte; the tail samples of the last byte are zeroed.
__global__ void bed_transcode_kernel(const uint8_t* __restrict__ bed, int64_t n, int64_t stride_in,
                                     uint8_t* __restrict__ bits, int64_t ldb) {
    const int64_t j = blockIdx.y;
    const int64_t nb = (n + 3) / 4;
    for (int64_t b = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; b < ldb; b += int64_t(gridDim.x) * blockDim.x) {
        unsigned out = 0;
        if (b < nb) {
            const unsigned in = bed[j * stride_in + b];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned f = (in >> (2 * k)) & 3u;
                // 00 -> 2, 01 -> 3 (missing), 10 -> 1, 11 -> 0
                const unsigned code = (b * 4 + k < n) ? ((0x1Eu >> (2 * f)) & 3u) : 0u;
                out |= code << (2 * k);
            }
        }
        bits[j * ldb + b] = uint8_t(out);
    }
}

// impute[j] = mean of the non-missing calls of column j (reference io/utils.hpp:10-31); one workgroup per column
template <class T>
__global__ __launch_bounds__(256) void snp_impute_kernel(const uint8_t* __restrict__ bits, int64_t n, int64_t ldb,
                                                         T* __restrict__ impute) {
    __shared__ unsigned long long red[3][4];
    const int64_t j = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t nb = (n + 3) / 4;
    unsigned long long c1 = 0, c2 = 0, c3 = 0;
    for (int64_t b = tid; b < nb; b += 256) {
        const unsigned byte = bits[j * ldb + b];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned c = (byte >> (2 * k)) & 3u;
            c1 += c == 1u; c2 += c == 2u; c3 += c == 3u;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        c1 += __shfl_xor(c1, off, 64); c2 += __shfl_xor(c2, off, 64); c3 += __shfl_xor(c3, off, 64);
    }
    if (lane == 0) { red[0][wv] = c1; red[1][wv] = c2; red[2][wv] = c3; }
    __syncthreads();
    if (tid == 0) {
        const unsigned long long s1 = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        const unsigned long long s2 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        const unsigned long long s3 = red[2][0] + red[2][1] + red[2][2] + red[2][3];
        const unsigned long long valid = (unsigned long long)n - s3;
        impute[j] = T(double(s1 + 2 * s2) / double(valid > 0 ? valid : 1));
    }
}

inline unsigned grid1d(int64_t n, int bt, int64_t cap = 4096) {
    int64_t g = (n + bt - 1) / bt;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return unsigned(g);
}

} // namespace

// row splits of the nibble-table sweep of a 2-bit design (sweep_snp_lut_kernel): ~8000 workgroups of 256 columns (five are
// resident per compute unit: six or more rounds, so that the last, partial round costs little), at least eight 256-row tiles
// per split
constexpr int kLutTile = 256;
inline void lut_shape(int64_t n, int64_t ncols, int64_t& blocks_c, int64_t& ns, int64_t& rps) {
    blocks_c = (ncols + kThreads - 1) / kThreads;
    ns = std::max<int64_t>(1, (8192 + blocks_c - 1) / blocks_c);
    ns = std::min<int64_t>(std::min(ns, std::max<int64_t>(1, n / (int64_t(kLutTile) * 8))), 65535);
    rps = (n + ns - 1) / ns;
    rps = ((rps + 2 * kLutTile - 1) / (2 * kLutTile)) * (2 * kLutTile); // whole 128-byte lines of a column (two tiles)
    ns = (n + rps - 1) / rps;
}
int64_t sweep_work_elems(int64_t n, int64_t ncols) {
    if (ncols <= 0) return 0;
    int64_t blocks_c, rps;
    int ns;
    sweep_shape(n, ncols, 1, blocks_c, ns, rps); // VEC=1 gives the largest split count
    int64_t a = int64_t(ns) * ncols;
    sweep_shape(n, ncols, 4, blocks_c, ns, rps);
    int64_t b = int64_t(ns) * ncols;
    int64_t lns;
    lut_shape(n, ncols, blocks_c, lns, rps);
    return std::max(std::max(a, b), lns * ncols) + 16;
}

template <class T>
void launch_vmul(const T* a, const T* b, T* out, int64_t n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL((vmul_kernel<T>), dim3(grid1d(n, 256)), dim3(256), 0, s, a, b, out, n);
}
template <class T>
void launch_fill(T* out, T value, int64_t n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL((fill_kernel<T>), dim3(grid1d(n, 256)), dim3(256), 0, s, out, value, n);
}

template <class T>
static inline bool dense_vec_ok(const DenseView<T>& X) {
    constexpr int V = VecOf<T>::N;
    return (X.ld % V == 0) && ((reinterpret_cast<uintptr_t>(X.X) % 16) == 0);
}

template <class T>
void launch_sweep(const DenseView<T>& X, const T* v, T* out, int64_t c0, int64_t ncols, const int32_t* cols,
                  const T* sub_scale, const T* sub_vec, bool square, T* work, hipStream_t s) {
    DenseAcc<T> acc{X.X, X.ld};
    if (dense_vec_ok(X) && (reinterpret_cast<uintptr_t>(v) % 16) == 0)
        sweep_dispatch<T, DenseAcc<T>, VecOf<T>::N>(acc, v, out, X.n, c0, ncols, cols, sub_scale, sub_vec, square, work, s);
    else
        sweep_dispatch<T, DenseAcc<T>, 1>(acc, v, out, X.n, c0, ncols, cols, sub_scale, sub_vec, square, work, s);
}
// ---- 2-bit sweep through nibble tables ------------------------------------------------------------------------------
// out[c] = sum_i x_ic v_i with x in {0, 1, 2, impute_c}.  The decode-and-multiply form above spends five integer / convert
// instructions and a multiply-add per call and is bound by their issue (5.8 ms for the 500k x 50k design of config 4, 1.08 TB/s).
// Here a thread owns a COLUMN and a workgroup walks row tiles of TR = 256 rows: per tile it first builds, in LDS, for every
// pair of rows (i, i+1) the 16 possible contributions of a nibble (two calls),
//     T012[b] = c0' v_i + c1' v_{i+1}   (c' = code if code < 3 else 0),     T3[b] = [c0 = 3] v_i + [c1 = 3] v_{i+1},
// — the vector v is the same for all columns, so a table serves every column of the workgroup — and then every thread adds
// one T012 entry per nibble of its column and one entry per BYTE of a second table that holds, for four rows, the sum of v over
// any subset of them (indexed by the four missing-call flags of the byte: three mask operations per word bring them
// together) — three LDS reads per four calls where a T3 entry per nibble made it four, the kernel being bound by what LDS
// delivers; the imputed value enters once at the end, S012 + impute_c * S3.  A table of 16 entries of 8 bytes is one row of LDS banks:
// any mix of indices over the lanes is conflict-free.  Columns are 128-byte aligned and padded (SnpView::ldb): a line of a
// column is two tiles.  Fixed summation order; row splits leave partials for sweep_reduce_kernel as above.
constexpr int LUT_TR = 256;
static_assert(LUT_TR == 256, "lut_shape's tile");
// How the columns come in.  A thread owning a column and fetching it 16 bytes at a time makes EIGHT separate requests for every
// 128-byte line, each wave instruction touching 64 different lines; with ~20 resident waves per compute unit the lines leave
// the L1 and even the L2 (5 MB in flight per XCD against 4 MB) between two of them: 13.6-14.3 GB of counter traffic per sweep
// of the 6.25 GB design (2.2 x; profiles/r05_cfg4_*, and unchanged when a thread merely takes the whole line in one trip:
// profiles/r06_cfg4_lut_line_per_trip.txt).  So the workgroup fetches cooperatively: per trip of 512 rows (one line per
// column) eight adjacent lanes take the eight 16-byte pieces of one column's line -- every line is requested once, whole, by
// one instruction -- and the pieces go through a 32 KB staging array in LDS (rotated by the column index: conflict-free in
// both directions) from which every thread reads back its own column.  The next trip's pieces are in flight while the
// current one is looked up.
// SQ: the squared design (x^2: the weighted column variances of an IRLS iteration): the table holds c'^2, the end impute^2.
template <class T, bool SQ>
__global__ __launch_bounds__(kThreads) void sweep_snp_lut_kernel(const uint8_t* __restrict__ bits, int64_t ldb,
                                                                 const T* __restrict__ impute, const T* __restrict__ v,
                                                                 T* __restrict__ out, int64_t n, int64_t c0, int64_t ncols,
                                                                 const int32_t* __restrict__ cols, int64_t rows_per_split,
                                                                 int nsplit, const T* __restrict__ sub_scale,
                                                                 const T* __restrict__ sub_vec) {
    static_assert(kThreads == 256, "256 columns per workgroup, 8 lanes per line");
    constexpr int NP = LUT_TR / 2;              // row pairs per tile
    typedef unsigned u4_t __attribute__((ext_vector_type(4)));
    constexpr int NQ = LUT_TR / 4;              // row quads per tile
    __shared__ T t012[NP][16];
    __shared__ T t3q[NQ][16];                   // missing calls, four rows per look-up: entry = sum of v over the set bits
    __shared__ u4_t stage[kThreads * 8];        // [column][piece rotated by the column]
    __shared__ int64_t cofs[kThreads];          // byte offset of the workgroup's columns
    const int tid = threadIdx.x;
    const int split = blockIdx.y;
    const int64_t r0 = int64_t(split) * rows_per_split; // (a multiple of 2 * LUT_TR: whole lines)
    const int64_t r1 = min(n, r0 + rows_per_split);
    int64_t c = int64_t(blockIdx.x) * kThreads + tid;
    const bool live = c < ncols;
    if (!live) c = ncols - 1;
    const int64_t cj = cols ? int64_t(cols[c]) : c0 + c;
    cofs[tid] = cj * ldb;
    __syncthreads();
    // fetch role: piece tid & 7 of the lines of columns (tid >> 3) + 32 u
    const int fp = tid & 7, fc = tid >> 3;
    const uint8_t* fptr[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) fptr[u] = bits + cofs[fc + 32 * u] + fp * 16;
    // table role of this thread: pair tid / 2, entries 8 * (tid & 1) .. + 8 of T012; quad tid / 4, entries 4 * (tid & 3) .. + 4 of T3
    const int tp = tid >> 1, te0 = (tid & 1) * 8;
    const int tq = tid >> 2, tqe0 = (tid & 3) * 4;
    T a = T(0), b = T(0);
    u4_t w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) w[u] = __builtin_nontemporal_load(reinterpret_cast<const u4_t*>(fptr[u] + r0 / 4));
    for (int64_t base = r0; base < r1; base += 2 * LUT_TR) {
        __syncthreads(); // the previous trip's lookups are done: the staging array is free
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int col = fc + 32 * u;
            stage[col * 8 + ((fp + col) & 7)] = w[u];
        }
        const int64_t nbase = base + 2 * LUT_TR;
        if (nbase < r1) {
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = __builtin_nontemporal_load(reinterpret_cast<const u4_t*>(fptr[u] + nbase / 4));
        }
        T vv[4], vq[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t i0 = base + h * LUT_TR + 2 * tp;
            vv[2 * h] = i0 < r1 ? v[i0] : T(0);
            vv[2 * h + 1] = i0 + 1 < r1 ? v[i0 + 1] : T(0);
            const int64_t j0 = base + h * LUT_TR + 4 * tq;
#pragma unroll
            for (int k = 0; k < 4; ++k) vq[4 * h + k] = j0 + k < r1 ? v[j0 + k] : T(0);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const T v0 = vv[2 * h], v1 = vv[2 * h + 1];
            if (h == 1) __syncthreads(); // tile 0's lookups are done
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int idx = te0 + e, k0 = idx & 3, k1 = idx >> 2;
                t012[tp][idx] = (k0 < 3 ? T(SQ ? k0 * k0 : k0) : T(0)) * v0 + (k1 < 3 ? T(SQ ? k1 * k1 : k1) : T(0)) * v1;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { // the missing calls of FOUR rows per entry (fixed order of the additions: rows ascending)
                const int idx = tqe0 + e;
                t3q[tq][idx] = (((idx & 1) ? vq[4 * h] : T(0)) + ((idx & 2) ? vq[4 * h + 1] : T(0))) +
                               (((idx & 4) ? vq[4 * h + 2] : T(0)) + ((idx & 8) ? vq[4 * h + 3] : T(0)));
            }
            __syncthreads(); // (h == 0: also makes the staged pieces visible)
            u4_t wd[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) wd[u] = stage[tid * 8 + ((4 * h + u + tid) & 7)];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int hh = 0; hh < 4; ++hh) {
                    const unsigned wq = wd[u][hh];
                    // bit 2k of `ms`: call k of the word is missing (code 3); then the four flags of every byte in its low nibble
                    unsigned ms = wq & (wq >> 1) & 0x55555555u;
                    ms = (ms | (ms >> 1)) & 0x33333333u;
                    ms = (ms | (ms >> 2)) & 0x0F0F0F0Fu;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int pr = (u * 4 + hh) * 8 + k; // row pair of this nibble (compile-time)
                        a += t012[pr][(wq >> (4 * k)) & 15u];
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int qr = (u * 4 + hh) * 4 + k; // row quad of this byte
                        b += t3q[qr][(ms >> (8 * k)) & 15u];
                    }
                }
            }
        }
    }
    if (!live) return;
    const T imp = impute[cj];
    T sres = fma(SQ ? imp * imp : imp, b, a);
    if (nsplit == 1) {
        if (sub_vec) sres -= sub_scale[0] * sub_vec[cj];
        out[c] = sres;
    } else {
        out[int64_t(split) * ncols + c] = sres; // partial
    }
}

template <class T>
void launch_sweep_snp(const SnpView& X, const T* impute, const T* v, T* out, int64_t c0, int64_t ncols,
                      const int32_t* cols, const T* sub_scale, const T* sub_vec, bool square, T* work, hipStream_t s) {
    if (ncols >= 512 && X.n >= 4096 && X.ldb % 128 == 0 && (reinterpret_cast<uintptr_t>(X.bits) % 128) == 0) {
        // the same partial layout and reduce as sweep_dispatch (`work`: sweep_work_elems covers lut_shape's splits)
        int64_t blocks_c, ns, rps;
        lut_shape(X.n, ncols, blocks_c, ns, rps);
        T* dst = ns == 1 ? out : work;
        if (square)
            hipLaunchKernelGGL((sweep_snp_lut_kernel<T, true>), dim3((unsigned)blocks_c, (unsigned)ns), dim3(kThreads), 0, s,
                               X.bits, X.ldb, impute, v, dst, X.n, c0, ncols, cols, rps, int(ns), sub_scale, sub_vec);
        else
            hipLaunchKernelGGL((sweep_snp_lut_kernel<T, false>), dim3((unsigned)blocks_c, (unsigned)ns), dim3(kThreads), 0, s,
                               X.bits, X.ldb, impute, v, dst, X.n, c0, ncols, cols, rps, int(ns), sub_scale, sub_vec);
        if (ns > 1) {
            const int bt = 256;
            hipLaunchKernelGGL((sweep_reduce_kernel<T>), dim3((unsigned)((ncols + bt - 1) / bt)), dim3(bt), 0, s, work, out,
                               ncols, int(ns), c0, cols, sub_scale, sub_vec);
        }
        return;
    }
    SnpAcc<T> acc{X.bits, X.ldb, impute};
    sweep_dispatch<T, SnpAcc<T>, VecOf<T>::N>(acc, v, out, X.n, c0, ncols, cols, sub_scale, sub_vec, square, work, s);
}

template <class T>
void launch_axpy_cols(const DenseView<T>& X, const int32_t* cols, const T* coef, const int32_t* count_dev,
                      int32_t count, T sign, T* out, hipStream_t s) {
    if (X.n <= 0 || (!count_dev && count <= 0)) return;
    DenseAcc<T> acc{X.X, X.ld};
    constexpr int V = VecOf<T>::N;
    if (dense_vec_ok(X)) {
        const unsigned g = unsigned((X.n + int64_t(kThreads) * V - 1) / (int64_t(kThreads) * V));
        hipLaunchKernelGGL((axpy_cols_kernel<T, DenseAcc<T>, V>), dim3(g), dim3(kThreads), 0, s, acc, X.n, cols, coef,
                           count_dev, count, sign, out);
    } else {
        const unsigned g = unsigned((X.n + kThreads - 1) / kThreads);
        hipLaunchKernelGGL((axpy_cols_kernel<T, DenseAcc<T>, 1>), dim3(g), dim3(kThreads), 0, s, acc, X.n, cols, coef,
                           count_dev, count, sign, out);
    }
}
template <class T>
void launch_axpy_cols_snp(const SnpView& X, const T* impute, const int32_t* cols, const T* coef,
                          const int32_t* count_dev, int32_t count, T sign, T* out, hipStream_t s) {
    if (X.n <= 0 || (!count_dev && count <= 0)) return;
    SnpAcc<T> acc{X.bits, X.ldb, impute};
    constexpr int V = VecOf<T>::N;
    const unsigned g = unsigned((X.n + int64_t(kThreads) * V - 1) / (int64_t(kThreads) * V));
    hipLaunchKernelGGL((axpy_cols_kernel<T, SnpAcc<T>, V>), dim3(g), dim3(kThreads), 0, s, acc, X.n, cols, coef,
                       count_dev, count, sign, out);
}

template <class T>
void launch_sp_tmul(const DenseView<T>& X, int64_t L, const int64_t* indptr, const int64_t* indices, const T* values,
                    T* out, hipStream_t s) {
    if (L <= 0 || X.n <= 0) return;
    DenseAcc<T> acc{X.X, X.ld};
    constexpr int V = VecOf<T>::N;
    const bool vec = dense_vec_ok(X) && (X.n % V == 0);
    for (int64_t l0 = 0; l0 < L; l0 += 65535) {
        const unsigned ly = unsigned(L - l0 < 65535 ? L - l0 : 65535);
        if (vec) {
            const unsigned g = unsigned((X.n + int64_t(kThreads) * V - 1) / (int64_t(kThreads) * V));
            hipLaunchKernelGGL((sp_tmul_kernel<T, DenseAcc<T>, V>), dim3(g, ly), dim3(kThreads), 0, s, acc, X.n,
                               indptr + l0, indices, values, out + l0 * X.n);
        } else {
            const unsigned g = unsigned((X.n + kThreads - 1) / kThreads);
            hipLaunchKernelGGL((sp_tmul_kernel<T, DenseAcc<T>, 1>), dim3(g, ly), dim3(kThreads), 0, s, acc, X.n,
                               indptr + l0, indices, values, out + l0 * X.n);
        }
    }
}
template <class T>
void launch_sp_tmul_snp(const SnpView& X, const T* impute, int64_t L, const int64_t* indptr, const int64_t* indices,
                        const T* values, T* out, hipStream_t s) {
    if (L <= 0 || X.n <= 0) return;
    SnpAcc<T> acc{X.bits, X.ldb, impute};
    constexpr int V = VecOf<T>::N;
    const bool vec = (X.n % V == 0);
    for (int64_t l0 = 0; l0 < L; l0 += 65535) {
        const unsigned ly = unsigned(L - l0 < 65535 ? L - l0 : 65535);
        if (vec) {
            const unsigned g = unsigned((X.n + int64_t(kThreads) * V - 1) / (int64_t(kThreads) * V));
            hipLaunchKernelGGL((sp_tmul_kernel<T, SnpAcc<T>, V>), dim3(g, ly), dim3(kThreads), 0, s, acc, X.n,
                               indptr + l0, indices, values, out + l0 * X.n);
        } else {
            const unsigned g = unsigned((X.n + kThreads - 1) / kThreads);
            hipLaunchKernelGGL((sp_tmul_kernel<T, SnpAcc<T>, 1>), dim3(g, ly), dim3(kThreads), 0, s, acc, X.n,
                               indptr + l0, indices, values, out + l0 * X.n);
        }
    }
}

template <class T>
void launch_abs_grad(const T* grad, const int64_t* groups, const int64_t* group_sizes, int64_t G, const int32_t* slot,
                     const T* screen_beta, const T* penalty, T oma_lmda, T* abs_grad, hipStream_t s) {
    if (G <= 0) return;
    hipLaunchKernelGGL((abs_grad_kernel<T>), dim3(unsigned((G + 255) / 256)), dim3(256), 0, s, grad, groups,
                       group_sizes, G, slot, screen_beta, penalty, oma_lmda, abs_grad);
}

template <class T>
void launch_abs_grad_cons(const T* grad, const int64_t* groups, const int64_t* group_sizes, int64_t G, const int32_t* slot,
                          const T* screen_beta, const T* penalty, T oma_lmda, const T* clo, const T* chi, const T* cmu,
                          T* abs_grad, T* mu_out, hipStream_t s) {
    if (G <= 0) return;
    hipLaunchKernelGGL((abs_grad_cons_kernel<T>), dim3(unsigned((G + 255) / 256)), dim3(256), 0, s, grad, groups, group_sizes,
                       G, slot, screen_beta, penalty, oma_lmda, clo, chi, cmu, abs_grad, mu_out);
}

template <class T>
void launch_copy2d(const T* src, int64_t lds_, T* dst, int64_t ldd, int64_t rows, int64_t cols, hipStream_t s) {
    if (rows <= 0 || cols <= 0) return;
    for (int64_t j0 = 0; j0 < cols; j0 += 65535) {
        const unsigned cy = unsigned(cols - j0 < 65535 ? cols - j0 : 65535);
        hipLaunchKernelGGL((copy2d_kernel<T>), dim3(grid1d(rows, 256, 64), cy), dim3(256), 0, s, src + j0 * lds_, lds_,
                           dst + j0 * ldd, ldd, rows, cols);
    }
}
// vars[pos] = max(D_y[i, i], 0) for the blocks of a build batch (block y at D0 + sb.dst[y], leading dimension ldb): pos = the
// screen position of the block's i-th visit = list ? list[base + sb.off[y] + i] : base + sb.off[y] + i
namespace {
template <class T>
__global__ __launch_bounds__(128) void block_diag_vars_kernel(const T* __restrict__ D0, SyrkBatch sb, int ldb, int32_t base,
                                                              const int32_t* __restrict__ list, T* __restrict__ vars) {
    const int y = blockIdx.x, i = threadIdx.x;
    if (i >= sb.nb[y]) return;
    const int32_t k = base + sb.off[y] + i;
    const T v = D0[sb.dst[y] + i + int64_t(i) * ldb];
    vars[list ? list[k] : k] = v > T(0) ? v : T(0);
}
} // namespace
template <class T>
void launch_block_diag_vars(const T* D0, const SyrkBatch& sb, int ldb, int32_t base, const int32_t* list, T* vars, hipStream_t s) {
    if (sb.count <= 0) return;
    hipLaunchKernelGGL((block_diag_vars_kernel<T>), dim3(unsigned(sb.count)), dim3(128), 0, s, D0, sb, ldb, base, list, vars);
}
template <class T>
void launch_diag_vars(const T* C, int64_t ldc, int32_t pos0, int32_t cnt, T* vars, hipStream_t s) {
    if (cnt <= 0) return;
    hipLaunchKernelGGL((diag_vars_kernel<T>), dim3((cnt + 255) / 256), dim3(256), 0, s, C, ldc, pos0, cnt, vars);
}
// dst[i + c*ldd] = (X[rows ? rows[i] : i, cols ? cols[c] : c] - (centers ? centers[col] : 0)) / (scales ? scales[col] : 1):
// a new dense design from a resident one (standardize / subset, reference matrix_naive_standardize.ipp,
// matrix_naive_subset.ipp -- views there; materialised here, HBM is not the scarce resource)
template <class T, class Acc>
__global__ void derive_dense_kernel(Acc X, int64_t nout, int64_t pout, const int64_t* __restrict__ rows,
                                    const int64_t* __restrict__ cols, const T* __restrict__ centers,
                                    const T* __restrict__ scales, T* __restrict__ dst, int64_t ldd) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t c = blockIdx.y;
    if (i >= nout || c >= pout) return;
    const int64_t j = cols ? cols[c] : c, r = rows ? rows[i] : i;
    const T x = X.template load<1>(X.colptr(j), r, j).v[0];
    const T ce = centers ? centers[c] : T(0), sc = scales ? scales[c] : T(1);
    dst[i + c * ldd] = (x - ce) / sc;
}
template <class T>
void launch_derive_dense(const DenseView<T>& X, int64_t nout, int64_t pout, const int64_t* rows, const int64_t* cols,
                         const T* centers, const T* scales, T* dst, int64_t ldd, hipStream_t s) {
    DenseAcc<T> acc{X.X, X.ld};
    for (int64_t c0 = 0; c0 < pout; c0 += 65535) {
        const int64_t pc = std::min<int64_t>(65535, pout - c0);
        hipLaunchKernelGGL((derive_dense_kernel<T, DenseAcc<T>>), dim3((unsigned)((nout + 255) / 256), (unsigned)pc), dim3(256), 0,
                           s, acc, nout, pc, rows, cols ? cols + c0 : nullptr, centers ? centers + c0 : nullptr,
                           scales ? scales + c0 : nullptr, dst + c0 * ldd, ldd);
        if (!cols) acc.X += 65535 * X.ld;
    }
}
template <class T>
void launch_derive_dense_snp(const SnpView& X, const T* impute, int64_t nout, int64_t pout, const int64_t* rows,
                             const int64_t* cols, const T* centers, const T* scales, T* dst, int64_t ldd, hipStream_t s) {
    SnpAcc<T> acc{X.bits, X.ldb, impute};
    for (int64_t c0 = 0; c0 < pout; c0 += 65535) {
        const int64_t pc = std::min<int64_t>(65535, pout - c0);
        hipLaunchKernelGGL((derive_dense_kernel<T, SnpAcc<T>>), dim3((unsigned)((nout + 255) / 256), (unsigned)pc), dim3(256), 0,
                           s, acc, nout, pc, rows, cols ? cols + c0 : nullptr, centers ? centers + c0 : nullptr,
                           scales ? scales + c0 : nullptr, dst + c0 * ldd, ldd);
        if (!cols) { acc.bits += 65535 * X.ldb; acc.impute += 65535; }
    }
}

// Rows / columns of a 2-bit design as another 2-bit design (matrix.subset on an SNP design: reference matrix_naive_subset.ipp
// wraps lazily; here the selected calls are re-packed, 2 bits per call, instead of being decoded into a dense copy).
// One thread per output byte: four calls gathered from the source column.
__global__ __launch_bounds__(256) void snp_subset_kernel(const uint8_t* __restrict__ src, int64_t ldb_src, int64_t nout, int64_t pout,
                                                          const int64_t* __restrict__ rows, const int64_t* __restrict__ cols,
                                                          uint8_t* __restrict__ dst, int64_t ldb_dst) {
    const int64_t bo = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const int64_t c = blockIdx.y;
    if (c >= pout || bo >= ldb_dst) return;
    const uint8_t* col = src + (cols ? cols[c] : c) * ldb_src;
    unsigned out = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t i = bo * 4 + k;
        if (i < nout) {
            const int64_t r = rows ? rows[i] : i;
            out |= ((unsigned(col[r >> 2]) >> (2 * (r & 3))) & 3u) << (2 * k);
        }
    }
    dst[c * ldb_dst + bo] = uint8_t(out); // (padding bytes and padding calls are zeros)
}
void launch_snp_subset(const SnpView& X, int64_t nout, int64_t pout, const int64_t* rows, const int64_t* cols, uint8_t* dst,
                       int64_t ldb_dst, hipStream_t s) {
    for (int64_t c0 = 0; c0 < pout; c0 += 65535) {
        const int64_t pc = std::min<int64_t>(65535, pout - c0);
        hipLaunchKernelGGL(snp_subset_kernel, dim3((unsigned)((ldb_dst + 255) / 256), (unsigned)pc), dim3(256), 0, s,
                           X.bits + (cols ? 0 : c0 * X.ldb), X.ldb, nout, pc, rows, cols ? cols + c0 : nullptr, dst + c0 * ldb_dst,
                           ldb_dst);
    }
}
// out[c] = src[cols ? cols[c] : c]
template <class T>
__global__ void gather_cols_kernel(const T* __restrict__ src, const int64_t* __restrict__ cols, int64_t pout, T* __restrict__ out) {
    const int64_t c = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c < pout) out[c] = src[cols ? cols[c] : c];
}
template <class T>
void launch_gather_cols(const T* src, const int64_t* cols, int64_t pout, T* out, hipStream_t s) {
    hipLaunchKernelGGL((gather_cols_kernel<T>), dim3((unsigned)((pout + 255) / 256)), dim3(256), 0, s, src, cols, pout, out);
}

// CSC -> resident dense columns (matrix.sparse; reference matrix_naive_sparse.ipp keeps the CSC arrays and walks them per
// operation -- with 288 GB of HBM the design is expanded once and every operation is the dense streaming kernel).
// One workgroup per column; duplicate (row, col) entries add up, as they do in the reference's sparse dot products.
template <class T>
__global__ void csc_scatter_kernel(const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                   const T* __restrict__ values, int64_t n, T* __restrict__ dst, int64_t ld) {
    const int64_t c = blockIdx.x;
    const int64_t b = indptr[c], e = indptr[c + 1];
    T* col = dst + c * ld;
    for (int64_t k = b + threadIdx.x; k < e; k += blockDim.x) {
        const int64_t i = indices[k];
        if (i >= 0 && i < n) atomicAdd(col + i, values[k]);
    }
}
template <class T>
void launch_csc_scatter(const int64_t* indptr, const int32_t* indices, const T* values, int64_t n, int64_t p, T* dst,
                        int64_t ld, hipStream_t s) {
    for (int64_t c0 = 0; c0 < p; c0 += 1 << 30) {
        const int64_t pc = std::min<int64_t>(int64_t(1) << 30, p - c0);
        hipLaunchKernelGGL((csc_scatter_kernel<T>), dim3((unsigned)pc), dim3(256), 0, s, indptr + c0, indices, values, n,
                           dst + c0 * ld, ld);
    }
}

template <class T>
void launch_transpose(const T* src, int64_t n, int64_t p, T* dst, int64_t ld, hipStream_t s) {
    if (n <= 0 || p <= 0) return;
    const int64_t gy_total = (n + 31) / 32;
    for (int64_t y0 = 0; y0 < gy_total; y0 += 65535) {
        const unsigned gy = unsigned(gy_total - y0 < 65535 ? gy_total - y0 : 65535);
        hipLaunchKernelGGL((transpose_kernel<T>), dim3(unsigned((p + 31) / 32), gy), dim3(32, 8), 0, s,
                           src + y0 * 32 * p, n - y0 * 32, p, dst + y0 * 32, ld);
    }
}
void launch_pack_snp(const int8_t* calldata, int64_t n, int64_t p, uint8_t* bits, int64_t ldb, hipStream_t s) {
    if (n <= 0 || p <= 0) return;
    for (int64_t j0 = 0; j0 < p; j0 += 65535) {
        const unsigned cy = unsigned(p - j0 < 65535 ? p - j0 : 65535);
        hipLaunchKernelGGL(pack_snp_kernel, dim3(grid1d(ldb, 256, 256), cy), dim3(256), 0, s, calldata + j0 * n, n, p,
                           bits + j0 * ldb, ldb);
    }
}

void launch_bed_transcode(const uint8_t* bed, int64_t n, int64_t p, int64_t stride_in, uint8_t* bits, int64_t ldb,
                          hipStream_t s) {
    if (n <= 0 || p <= 0) return;
    for (int64_t j0 = 0; j0 < p; j0 += 65535) {
        const unsigned cy = unsigned(p - j0 < 65535 ? p - j0 : 65535);
        hipLaunchKernelGGL(bed_transcode_kernel, dim3(grid1d(ldb, 256, 256), cy), dim3(256), 0, s, bed + j0 * stride_in, n,
                           stride_in, bits + j0 * ldb, ldb);
    }
}
template <class T>
void launch_snp_impute(const uint8_t* bits, int64_t n, int64_t p, int64_t ldb, T* impute, hipStream_t s) {
    if (p <= 0) return;
    hipLaunchKernelGGL((snp_impute_kernel<T>), dim3((unsigned)p), dim3(256), 0, s, bits, n, ldb, impute);
}
template void launch_snp_impute<double>(const uint8_t*, int64_t, int64_t, int64_t, double*, hipStream_t);
template void launch_snp_impute<float>(const uint8_t*, int64_t, int64_t, int64_t, float*, hipStream_t);

#define INST(T)                                                                                                        \
    template void launch_vmul<T>(const T*, const T*, T*, int64_t, hipStream_t);                                        \
    template void launch_fill<T>(T*, T, int64_t, hipStream_t);                                                         \
    template void launch_sweep<T>(const DenseView<T>&, const T*, T*, int64_t, int64_t, const int32_t*, const T*,       \
                                  const T*, bool, T*, hipStream_t);                                                    \
    template void launch_sweep_snp<T>(const SnpView&, const T*, const T*, T*, int64_t, int64_t, const int32_t*,        \
                                      const T*, const T*, bool, T*, hipStream_t);                                      \
    template void launch_axpy_cols<T>(const DenseView<T>&, const int32_t*, const T*, const int32_t*, int32_t, T, T*,   \
                                      hipStream_t);                                                                    \
    template void launch_axpy_cols_snp<T>(const SnpView&, const T*, const int32_t*, const T*, const int32_t*, int32_t, \
                                          T, T*, hipStream_t);                                                         \
    template void launch_sp_tmul<T>(const DenseView<T>&, int64_t, const int64_t*, const int64_t*, const T*, T*,        \
                                    hipStream_t);                                                                      \
    template void launch_sp_tmul_snp<T>(const SnpView&, const T*, int64_t, const int64_t*, const int64_t*, const T*,   \
                                        T*, hipStream_t);                                                              \
    template void launch_abs_grad<T>(const T*, const int64_t*, const int64_t*, int64_t, const int32_t*, const T*,      \
                                     const T*, T, T*, hipStream_t);                                                    \
    template void launch_abs_grad_cons<T>(const T*, const int64_t*, const int64_t*, int64_t, const int32_t*, const T*, \
                                          const T*, T, const T*, const T*, const T*, T*, T*, hipStream_t);             \
    template void launch_copy2d<T>(const T*, int64_t, T*, int64_t, int64_t, int64_t, hipStream_t);                     \
    template void launch_diag_vars<T>(const T*, int64_t, int32_t, int32_t, T*, hipStream_t);                           \
    template void launch_block_diag_vars<T>(const T*, const SyrkBatch&, int, int32_t, const int32_t*, T*, hipStream_t); \
    template void launch_csc_scatter<T>(const int64_t*, const int32_t*, const T*, int64_t, int64_t, T*, int64_t, hipStream_t);\
    template void launch_transpose<T>(const T*, int64_t, int64_t, T*, int64_t, hipStream_t);                            \
    template void launch_derive_dense<T>(const DenseView<T>&, int64_t, int64_t, const int64_t*, const int64_t*, const T*, \
                                         const T*, T*, int64_t, hipStream_t);                                          \
    template void launch_derive_dense_snp<T>(const SnpView&, const T*, int64_t, int64_t, const int64_t*, const int64_t*, \
                                             const T*, const T*, T*, int64_t, hipStream_t);                              \
    template void launch_gather_cols<T>(const T*, const int64_t*, int64_t, T*, hipStream_t);
INST(double)
INST(float)
#undef INST

} // namespace ahip
