// kernels_strip.hip — the NEW ROWS of a panel block's Gram blocks, straight from HBM into the matrix cores (gfx950).
//
// Both visiting lists of the panel engine (screen order, active order) only grow by appending, so when block j gains the
// members [have, nb) the entries of its diagonal block D_j = X_j^T W X_j and of its cross block C_j = X_j^T W X_{j-1} that
// exist stay valid; what is missing is the strip
//
//      S = V^T W [ X_{j-1} | X_j ]      V = the m = nb - have new columns,  at most 16 MT x 256 entries,
//
// i.e. rows [have, nb) of C_j, rows [have, nb) of D_j and (by symmetry) its columns [have, nb).  The reference computes the
// same numbers one group at a time with MatrixNaiveDense::cov (matrix_naive_dense.ipp:162-197) inside update_screen_derived
// (solver_gaussian_naive.hpp:41-125); the blocks are what lets a panel solve replace 128 cmul / ctmul pairs (pin_naive:84-108).
//
// A strip is short and wide: 2 n m 256 flop over (256 + m) n s bytes, 32 flop per byte at m = 16 — HBM-bound on a part with
// ~10 f64 flop per byte, where the staged kernels of kernels_gram.hip (whole 128 x 128 tiles through LDS, two barriers per 32
// rows, 128 KB of partial tile per K-split) are bound by their own staging whatever the row count.  So no LDS and no barriers
// here: a wavefront owns 16 MT rows x 64 columns (MT x 4 MFMA tiles) over its workgroup's K-split and feeds
// v_mfma_{f64,f32}_16x16x4 from registers it loaded itself.  The instruction wants A[i = lane & 15][k = lane >> 4] and
// B[k = lane >> 4][j = lane & 15]; which four rows of X make up the "k" of one instruction is free as long as both operands
// agree, so lane (i, q) takes the KC = 4 CONSECUTIVE rows 4 q .. 4 q + 3 of a 16-row chunk of its column (one or two 16-byte
// loads, 128 contiguous bytes per column and chunk) and instruction e of the chunk multiplies rows {4 q + e}.  Chunk t + 1 is
// in flight while the MFMAs of chunk t issue (two register sets).  The four waves of a workgroup take the four 64-column
// quarters of the strip over the same rows, so the V fragments they all need come out of the cache after the first.
// K-splits write partial strips [row][256]; a second kernel sums them in a fixed order, centres, and scatters the rows into
// the cross block and (mirrored, the new x new square from its lower triangle only) the diagonal block.
#include "gram_common.hpp"
#include "common.hpp"
#include <type_traits>

namespace ahip {

namespace {

constexpr int SGT = 256;      // threads per workgroup (4 waves = 4 column quarters)
constexpr int SKC = 4;        // consecutive rows per lane and chunk
constexpr int SCH = 4 * SKC;  // rows per chunk
constexpr int STG = 4;        // column tiles per wave
constexpr int SW = 256;       // columns of a strip (previous block | own block)

// One wavefront's share of a K-split: 16 MT rows x 16 TGL columns.  The main loop is branch-free: rows / columns beyond the
// strip's are CLAMPED to its last one (their products land in entries the reduce kernel never reads), whole 16-row chunks only;
// the ragged tail of the last K-split goes through guarded loads once.
template <class T, class Acc, bool VECOK, int MT, int TGL>
__device__ __forceinline__ void strip_wave(const Acc& X, const T* __restrict__ w, const int32_t* __restrict__ cols_base,
                                           const StripBatch& b, int y, int wv, int lane, int64_t k0, int64_t k1,
                                           T* __restrict__ P) {
    const int fr = lane & 15, fq = lane >> 4;
    const int m = b.m[y], c0n = b.c0n[y], ncol = c0n + b.c1n[y];
    const T* pa[MT];
    const T* pb[TGL];
    int64_t ja[MT], jb[TGL];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        ja[i] = int64_t(cols_base[b.voff[y] + min(i * 16 + fr, m - 1)]);
        pa[i] = X.colptr(ja[i]);
    }
#pragma unroll
    for (int t = 0; t < TGL; ++t) {
        const int c = min(wv * 64 + t * 16 + fr, ncol - 1);
        jb[t] = int64_t(c < c0n ? cols_base[b.c0off[y] + c] : cols_base[b.c1off[y] + c - c0n]);
        pb[t] = X.colptr(jb[t]);
    }

    typename Mfma<T>::acc_t acc[MT][TGL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int t = 0; t < TGL; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][t][e] = T(0);

    // Rows of a chunk held by lane (., q): element e of vector u = row u * 4 V + q V + e  (V = elements per 16-byte load: the
    // four q-lanes of a column read one contiguous 64-byte run per load instruction)
    constexpr int V = VECOK ? VecOf<T>::N : 1;
    constexpr int NV = SKC / V;
    T ra[2][MT][SKC], rb[2][TGL][SKC], rw[2][SKC];
    auto fetch = [&](int s, int64_t k) { // a whole chunk [k, k + 16)
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int64_t kk = k + u * 4 * V + fq * V;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const Pack<T, V> x = X.template load<V>(pa[i], kk, ja[i]);
#pragma unroll
                for (int e = 0; e < V; ++e) ra[s][i][u * V + e] = x.v[e];
            }
#pragma unroll
            for (int t = 0; t < TGL; ++t) {
                const Pack<T, V> x = X.template load<V>(pb[t], kk, jb[t]);
#pragma unroll
                for (int e = 0; e < V; ++e) rb[s][t][u * V + e] = x.v[e];
            }
#pragma unroll
            for (int e = 0; e < V; ++e) rw[s][u * V + e] = w[kk + e];
        }
    };
    auto fetch_tail = [&](int s, int64_t k) { // rows [k, k1), fewer than 16
#pragma unroll
        for (int u = 0; u < NV; ++u)
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const int64_t kk = k + u * 4 * V + fq * V + e;
                const bool in = kk < k1;
#pragma unroll
                for (int i = 0; i < MT; ++i) ra[s][i][u * V + e] = in ? X.template load<1>(pa[i], kk, ja[i]).v[0] : T(0);
#pragma unroll
                for (int t = 0; t < TGL; ++t) rb[s][t][u * V + e] = in ? X.template load<1>(pb[t], kk, jb[t]).v[0] : T(0);
                rw[s][u * V + e] = in ? w[kk] : T(0);
            }
    };
    auto run = [&](int s) {
#pragma unroll
        for (int e = 0; e < SKC; ++e) {
            T a[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) a[i] = ra[s][i][e] * rw[s][e];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int t = 0; t < TGL; ++t) acc[i][t] = Mfma<T>::run(a[i], rb[s][t][e], acc[i][t]);
        }
    };
    const int64_t kfull = k0 + ((k1 - k0) / SCH) * SCH;
    int64_t k = k0;
    if (k + SCH <= kfull) {
        fetch(0, k);
        while (k + 2 * SCH <= kfull) { // set 0 holds chunk k; chunk k + 16 exists
            fetch(1, k + SCH);
            run(0);
            fetch(0, (k + 3 * SCH <= kfull) ? k + 2 * SCH : k); // (no chunk left: a harmless re-load)
            run(1);
            k += 2 * SCH;
        }
        if (k + SCH <= kfull) {
            run(0);
            k += SCH;
        }
    }
    if (k < k1) {
        fetch_tail(0, k);
        run(0);
    }

    // partial strip -> P[row][256]
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int t = 0; t < TGL; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = i * 16 + Mfma<T>::row(lane, e);
                const int col = wv * 64 + t * 16 + fr;
                P[int64_t(row) * SW + col] = acc[i][t][e];
            }
}

template <class T, class Acc, bool VECOK, int MT>
__global__ __launch_bounds__(SGT, (MT <= 2 ? 2 : 1)) void strip_kernel(Acc X, const T* __restrict__ w,
                                                                      const int32_t* __restrict__ cols_base, StripBatch b,
                                                                      int64_t n, int64_t kchunk, int nsplit,
                                                                      T* __restrict__ part) {
    const int y = blockIdx.y, sp = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
    const int ncol = b.c0n[y] + b.c1n[y];
    const int tg_live = min(STG, (ncol - wv * 64 + 15) >> 4); // column tiles of this wave that hold columns (<= 0: none)
    if (tg_live <= 0 || b.m[y] <= 0) return;
    const int64_t k0 = int64_t(sp) * kchunk, k1 = min(n, k0 + kchunk);
    if (k0 >= k1) return;
    T* P = part + (int64_t(y) * nsplit + sp) * (16 * MT) * SW;
    if (tg_live == 4) strip_wave<T, Acc, VECOK, MT, 4>(X, w, cols_base, b, y, wv, lane, k0, k1, P);
    else if (tg_live == 3) strip_wave<T, Acc, VECOK, MT, 3>(X, w, cols_base, b, y, wv, lane, k0, k1, P);
    else if (tg_live == 2) strip_wave<T, Acc, VECOK, MT, 2>(X, w, cols_base, b, y, wv, lane, k0, k1, P);
    else strip_wave<T, Acc, VECOK, MT, 1>(X, w, cols_base, b, y, wv, lane, k0, k1, P);
}

// ---- f64, 16-byte aligned columns: full cache lines per load, fragments through a wave-private LDS transpose -----------------
// In the direct form above a load instruction touches 16 columns x 64 bytes — half a cache line per column — and the texture
// path, not HBM, bounds the kernel (13-15 GB/s per CU whatever MT).  Here lane L of a load instruction takes bytes
// [16 (L & 7), +16) of the 128-byte run of column L >> 3 (8 columns x one whole line per instruction), parks them in the
// wavefront's own LDS region ([column][16 rows + 2 pad]) and reads its MFMA fragment back from there (lane (c, q): rows
// 2 q, 2 q + 1 and 8 + 2 q, 9 + 2 q of column c — conflict-free with the 144-byte column pitch).  No workgroup barrier: a
// wavefront's LDS operations execute in order, the fences below only keep the compiler from reordering them.
constexpr int SLP = 18; // doubles per column in LDS (16 rows + 2 pad)

template <class Acc, int MT, int TGL>
__device__ __forceinline__ void strip_wave_lt(const Acc& X, const double* __restrict__ w, const int32_t* __restrict__ cols_base,
                                              const StripBatch& b, int y, int wv, int lane, int64_t k0, int64_t k1,
                                              double* __restrict__ P, double* __restrict__ L) {
    using T = double;
    constexpr int NF = MT + TGL; // fragments: rows first, then this wave's column tiles
    const int fr = lane & 15, fq = lane >> 4;
    const int lc = lane >> 3, lp = lane & 7;
    const int m = b.m[y], c0n = b.c0n[y], ncol = c0n + b.c1n[y];
    // coalesced role: instruction h of fragment f brings column 8 h + lc of the fragment
    const T* pg[NF][2];
    int64_t jg[NF][2];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c16 = h * 8 + lc;
            int64_t j;
            if (f < MT) {
                j = int64_t(cols_base[b.voff[y] + min(f * 16 + c16, m - 1)]);
            } else {
                const int c = min(wv * 64 + (f - MT) * 16 + c16, ncol - 1);
                j = int64_t(c < c0n ? cols_base[b.c0off[y] + c] : cols_base[b.c1off[y] + c - c0n]);
            }
            jg[f][h] = j;
            pg[f][h] = X.colptr(j) + lp * 2;
        }
    // fragment role (ragged tail only): column fr of fragment f
    typename Mfma<T>::acc_t acc[MT][TGL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int t = 0; t < TGL; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][t][e] = T(0);

    T g[NF][2][2], fv[NF][SKC], rw[SKC], rwn[SKC];
    auto fetch = [&](int64_t k) {
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const Pack<T, 2> x = X.template load<2>(pg[f][h], k, jg[f][h]);
                g[f][h][0] = x.v[0];
                g[f][h][1] = x.v[1];
            }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 2; ++e) rwn[u * 2 + e] = w[k + u * 8 + fq * 2 + e];
    };
    typedef double d2_t __attribute__((ext_vector_type(2)));
    auto park = [&]() {
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                d2_t v = {g[f][h][0], g[f][h][1]};
                *reinterpret_cast<d2_t*>(L + (f * 16 + h * 8 + lc) * SLP + lp * 2) = v;
            }
#pragma unroll
        for (int e = 0; e < SKC; ++e) rw[e] = rwn[e];
    };
    auto frags = [&]() {
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const d2_t v = *reinterpret_cast<const d2_t*>(L + (f * 16 + fr) * SLP + u * 8 + fq * 2);
                fv[f][u * 2] = v[0];
                fv[f][u * 2 + 1] = v[1];
            }
    };
    auto run = [&]() {
#pragma unroll
        for (int e = 0; e < SKC; ++e) {
            T a[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) a[i] = fv[i][e] * rw[e];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int t = 0; t < TGL; ++t) acc[i][t] = Mfma<T>::run(a[i], fv[MT + t][e], acc[i][t]);
        }
    };
    const int64_t kfull = k0 + ((k1 - k0) / SCH) * SCH;
    int64_t k = k0;
    if (k < kfull) {
        fetch(k);
        while (k < kfull) {
            park(); // (waits for the loads of chunk k)
            fetch((k + 2 * SCH <= kfull) ? k + SCH : k); // chunk k + 16 in flight during the MFMAs (none left: a harmless re-load)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            frags();
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            run();
            k += SCH;
        }
    }
    if (k < k1) { // ragged tail (fewer than 16 rows): guarded loads straight into the fragment layout
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int64_t kk = k + u * 8 + fq * 2 + e;
                const bool in = kk < k1;
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    int64_t j;
                    if (f < MT) {
                        j = int64_t(cols_base[b.voff[y] + min(f * 16 + fr, m - 1)]);
                    } else {
                        const int c = min(wv * 64 + (f - MT) * 16 + fr, ncol - 1);
                        j = int64_t(c < c0n ? cols_base[b.c0off[y] + c] : cols_base[b.c1off[y] + c - c0n]);
                    }
                    fv[f][u * 2 + e] = in ? X.template load<1>(X.colptr(j), kk, j).v[0] : T(0);
                }
                rw[u * 2 + e] = in ? w[kk] : T(0);
            }
        run();
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int t = 0; t < TGL; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = i * 16 + Mfma<T>::row(lane, e);
                const int col = wv * 64 + t * 16 + fr;
                P[int64_t(row) * SW + col] = acc[i][t][e];
            }
}

template <class Acc, int MT>
__global__ __launch_bounds__(SGT, (MT <= 2 ? 2 : 1)) void strip_lt_kernel(Acc X, const double* __restrict__ w,
                                                                         const int32_t* __restrict__ cols_base, StripBatch b,
                                                                         int64_t n, int64_t kchunk, int nsplit,
                                                                         double* __restrict__ part) {
    __shared__ double lds[4][(MT + STG) * 16 * SLP];
    const int y = blockIdx.y, sp = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
    const int ncol = b.c0n[y] + b.c1n[y];
    const int tg_live = min(STG, (ncol - wv * 64 + 15) >> 4);
    if (tg_live <= 0 || b.m[y] <= 0) return;
    const int64_t k0 = int64_t(sp) * kchunk, k1 = min(n, k0 + kchunk);
    if (k0 >= k1) return;
    double* P = part + (int64_t(y) * nsplit + sp) * (16 * MT) * SW;
    double* L = &lds[wv][0];
    if (tg_live == 4) strip_wave_lt<Acc, MT, 4>(X, w, cols_base, b, y, wv, lane, k0, k1, P, L);
    else if (tg_live == 3) strip_wave_lt<Acc, MT, 3>(X, w, cols_base, b, y, wv, lane, k0, k1, P, L);
    else if (tg_live == 2) strip_wave_lt<Acc, MT, 2>(X, w, cols_base, b, y, wv, lane, k0, k1, P, L);
    else strip_wave_lt<Acc, MT, 1>(X, w, cols_base, b, y, wv, lane, k0, k1, P, L);
}

// grid (16, m_max, count), 256 threads = 16 groups of K-splits x 16 consecutive columns.  Entry (row, col) of strip y:
// sum of the K-split partials in a fixed order (each group its splits in ascending order, the 16 group sums as a fixed tree).
template <class T>
__global__ __launch_bounds__(256) void strip_reduce_kernel(const T* __restrict__ part, int nsplit, int rows_pad, StripBatch b,
                                                          const int32_t* __restrict__ cols_base, const T* __restrict__ xm,
                                                          int center, T* __restrict__ D_base, T* __restrict__ X_base,
                                                          int64_t ldc) {
    __shared__ T red[16][17];
    const int y = blockIdx.z, row = blockIdx.y;
    const int m = b.m[y], c0n = b.c0n[y], c1n = b.c1n[y];
    if (row >= m) return;
    const int tc = threadIdx.x & 15, tg = threadIdx.x >> 4;
    const int col = blockIdx.x * 16 + tc;
    const int have = b.row0[y];
    const int cc = col - c0n;                      // position inside the own block (col >= c0n)
    const bool live = col < c0n + c1n;
    // the new x new square of the diagonal block is taken from its lower triangle only (and mirrored): exactly symmetric
    const bool skip = !live || (cc >= 0 && cc > have + row);
    T s = T(0);
    if (!skip) {
        const int64_t stride = int64_t(rows_pad) * SW;
        const T* base = part + int64_t(y) * nsplit * stride + int64_t(row) * SW + col;
        int sp = tg;
        for (; sp + 48 < nsplit; sp += 64) {
            const T v0 = base[int64_t(sp) * stride], v1 = base[int64_t(sp + 16) * stride];
            const T v2 = base[int64_t(sp + 32) * stride], v3 = base[int64_t(sp + 48) * stride];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; sp < nsplit; sp += 16) s += base[int64_t(sp) * stride];
    }
    red[tg][tc] = s;
    __syncthreads();
    if (tg != 0 || skip) return;
    T q[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) q[u] = (red[4 * u][tc] + red[4 * u + 1][tc]) + (red[4 * u + 2][tc] + red[4 * u + 3][tc]);
    s = (q[0] + q[1]) + (q[2] + q[3]);
    const int32_t vcol = cols_base[b.voff[y] + row];
    if (cc < 0) {
        if (center) s -= xm[vcol] * xm[cols_base[b.c0off[y] + col]];
        X_base[b.dstX[y] + (have + row) + int64_t(col) * ldc] = s;
    } else {
        if (center) s -= xm[vcol] * xm[cols_base[b.c1off[y] + cc]];
        T* Dp = D_base + b.dstD[y];
        Dp[(have + row) + int64_t(cc) * ldc] = s;
        Dp[cc + int64_t(have + row) * ldc] = s;
    }
}

inline void strip_shape(int64_t n, int count, int wgs, int& nsplit, int64_t& kchunk) {
    int64_t want = (int64_t(wgs) + count - 1) / count;
    const int64_t max_split = (n + 8 * SCH - 1) / (8 * SCH);
    if (want > max_split) want = max_split;
    if (want < 1) want = 1;
    kchunk = (n + want - 1) / want;
    kchunk = ((kchunk + 2 * SCH - 1) / (2 * SCH)) * (2 * SCH); // whole pairs of chunks: 16-byte aligned row offsets
    const int64_t ns = (n + kchunk - 1) / kchunk;
    nsplit = int(ns < 1 ? 1 : ns);
}

// K-splits of a strip launch (per entry).  Each split writes a partial strip of 16 MT x 256 values that the reduce kernel reads
// back (512 splits: 53 MB written per launch of the headline's screen strips, PMC), and in the path the strips only get the
// ~60 CUs the fused launches leave, so more splits than that buy nothing: 192 instead of 512 is worth 1.5 ms per headline path
// and 4.7 ms on config 3 (the chain starts to wait for the strips below ~112).  Hook ADELIE_HIP_STRIP_WGS.
constexpr int kStripWgsDefault = 192;
thread_local int t_strip_wgs = kStripWgsDefault;
thread_local bool t_strip_lds = true;

} // namespace

void set_strip_workgroups(int wgs) { t_strip_wgs = wgs < 1 ? kStripWgsDefault : wgs; }
void set_strip_lds(bool on) { t_strip_lds = on; }

int strip_row_tiles(int m) { return m <= 16 ? 1 : (m <= 32 ? 2 : (m <= 48 ? 3 : (m <= 64 ? 4 : 0))); }

int64_t strip_work_elems(int64_t n, int count, int m_max) {
    int ns;
    int64_t kc;
    strip_shape(n, count, 1024, ns, kc); // (sized for the widest spread the hook allows)
    const int mt = strip_row_tiles(m_max);
    return int64_t(count) * ns * 16 * (mt ? mt : 4) * SW;
}

template <class T>
void launch_strip_batch(const DenseView<T>& Xv, const T* w, const int32_t* cols_base, const StripBatch& b, const T* xm_by_col,
                        bool center, T* D_base, T* X_base, int64_t ldc, T* work, hipStream_t s) {
    if (b.count <= 0) return;
    int mx = 0;
    for (int y = 0; y < b.count; ++y) mx = std::max(mx, int(b.m[y]));
    const int MTv = strip_row_tiles(mx);
    if (MTv == 0) throw make_core_error("internal: strip build of more than 64 rows.");
    DenseAcc<T> acc{Xv.X, Xv.ld};
    const bool vecok = (Xv.ld % VecOf<T>::N == 0) && ((reinterpret_cast<uintptr_t>(Xv.X) % 16) == 0);
    int nsplit;
    int64_t kchunk;
    strip_shape(Xv.n, b.count, std::min(t_strip_wgs, 1024), nsplit, kchunk);
    const dim3 grid((unsigned)nsplit, (unsigned)b.count);
#define AHIP_STRIP(VOK, MTV)                                                                                            \
    hipLaunchKernelGGL((strip_kernel<T, DenseAcc<T>, VOK, MTV>), grid, dim3(SGT), 0, s, acc, w, cols_base, b, Xv.n, kchunk, \
                       nsplit, work)
    bool done = false;
    if constexpr (std::is_same<T, double>::value) {
        if (vecok && t_strip_lds) {
#define AHIP_STRIP_LT(MTV)                                                                                              \
    hipLaunchKernelGGL((strip_lt_kernel<DenseAcc<double>, MTV>), grid, dim3(SGT), 0, s, acc, w, cols_base, b, Xv.n, kchunk, \
                       nsplit, work)
            if (MTv == 1) AHIP_STRIP_LT(1);
            else if (MTv == 2) AHIP_STRIP_LT(2);
            else if (MTv == 3) AHIP_STRIP_LT(3);
            else AHIP_STRIP_LT(4);
#undef AHIP_STRIP_LT
            done = true;
        }
    }
    if (!done) {
        if (MTv == 1) { if (vecok) AHIP_STRIP(true, 1); else AHIP_STRIP(false, 1); }
        else if (MTv == 2) { if (vecok) AHIP_STRIP(true, 2); else AHIP_STRIP(false, 2); }
        else if (MTv == 3) { if (vecok) AHIP_STRIP(true, 3); else AHIP_STRIP(false, 3); }
        else { if (vecok) AHIP_STRIP(true, 4); else AHIP_STRIP(false, 4); }
    }
#undef AHIP_STRIP
    hipLaunchKernelGGL((strip_reduce_kernel<T>), dim3(16, (unsigned)mx, (unsigned)b.count), dim3(256), 0, s, work, nsplit,
                       16 * MTv, b, cols_base, xm_by_col, center ? 1 : 0, D_base, X_base, ldc);
}

template void launch_strip_batch<double>(const DenseView<double>&, const double*, const int32_t*, const StripBatch&,
                                         const double*, bool, double*, double*, int64_t, double*, hipStream_t);
template void launch_strip_batch<float>(const DenseView<float>&, const float*, const int32_t*, const StripBatch&, const float*,
                                        bool, float*, float*, int64_t, float*, hipStream_t);

} // namespace ahip
