// kernels_gram.hip — weighted Gram panels  C[a,b] = sum_i w_i X[i,ma] X[i,nb]  on the matrix cores (gfx950).
//
// Replaces, for the screen set as a whole, what the reference does one group at a time with
// MatrixNaiveDense::cov (matrix_naive_dense.ipp:162-197) inside update_screen_derived
// (solver_gaussian_naive.hpp:41-125): the centred weighted Gram  X_S^T W X_S - xbar xbar^T.  Its diagonal
// blocks are exactly the reference's per-group `XiTXi`; the off-diagonal blocks are what lets the
// coordinate-descent kernel (kernels_cd.hip) replace every per-visit X.cmul / X.ctmul by O(|S|) work.
//
// This is the one true GEMM on the path (K = n = 1e5..5e5, M,N = screen values), so it runs on MFMA:
//   f64: v_mfma_f64_16x16x4_f64   (A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D col=l&15,row=(l>>4)+4*reg)
//   f32: v_mfma_f32_16x16x4_f32   (same A/B maps,                         D col=l&15,row=(l>>4)*4+reg)
// Tiling: a 256-thread block (4 waves) owns a 128 x {64,128} output tile and one K-split; each wave owns 64x64 (or
// 32x64) = 4x4 (2x4) MFMA tiles, so one pair of LDS fragment reads feeds 16 (8) MFMAs.  The N tile is as wide as the
// batch of new screen columns allows (<= 128) so that the M-side panel - the whole screen set, the bulk of the bytes
// - is read once per launch.  Column panels are K-contiguous in HBM (column-major X), so a stage is (128+BN) columns x
// 32 rows: every thread brings 16 (8) consecutive rows of one column (128 B runs), fragments are read back with
// ds_read_b64 from rows padded to 34 elements (conflict free for the 16x4 fragment shape), the weights of the stage
// sit in LDS and scale the B fragment.  Global loads for stage t+1 are issued before the MFMAs of stage t.
// K-splits write partial tiles; a second kernel sums them (deterministic), centres, and writes C symmetrically.
#include "gram_common.hpp"

namespace ahip {

namespace {

constexpr int BM = 128, KT = 32, LDK = KT + 2, GT = 256;

// Workgroups a SMALL build (one diagonal block / one cross block / a batch of diagonal blocks) is spread over.  512 = one
// full round of two resident workgroups per CU, the fastest for the build itself.  The panel engine lowers it for builds
// that run on the side stream next to the look-ahead launches of a Gaussian pass (set_small_gram_workgroups): those launches
// need ~200 whole CUs, and a build that occupies every CU makes each of them wait.  Per host thread: set right before the
// launches it is meant for.
thread_local int t_small_gram_wgs = 512;

// Block tile BM x BN = 128 x {64,128}; 4 waves: 2x2 of 64x64 (BN=128) or 4x1 of 32x64 (BN=64).
// WNX = 4 (BN = 128): the four waves side by side, each 128 rows x 32 columns (8 x 2 MFMA tiles) — the "row strip" form of
// the batched cross blocks: the row tiles at or beyond M are skipped (MFMAs, fragment reads and the partial-tile stores), so
// a strip of M new rows costs ceil(M / 16) / 8 of a full block.
template <class T, class Acc, bool VECOK, int BN, int WNX = BN / 64>
__device__ __forceinline__ void gram_body(const Acc& X, const T* __restrict__ w, const int32_t* __restrict__ mcols,
                                          int32_t M, const int32_t* __restrict__ ncols, int32_t N, int64_t n,
                                          int64_t kchunk, int32_t m_pos0, int32_t n_pos0, int symmetric,
                                          T* __restrict__ part, int64_t Mpad, int64_t Npad, int32_t ncol0,
                                          int32_t Mt, int32_t Nt, int32_t nsplit) {
    constexpr int WN = WNX;                // waves along N
    constexpr int WM = 4 / WN;             // waves along M
    constexpr int TM = BM / WM / 16;       // MFMA tiles per wave along M (4, 2 or 8)
    constexpr int CW = BN / WN;            // columns per wave along N (64 or 32)
    constexpr int TN = CW / 16;
    constexpr bool STRIP = (WNX == 4 && BN == 128);
    constexpr int RA = KT * BM / GT;       // rows of one A column staged per thread (16)
    constexpr int RB = KT * BN / GT;       // rows of one B column staged per thread (16 or 8)
    // XCD-aware block -> tile map: workgroup L runs on XCD L % 8 (guides/MI355X_MICROARCH.md).  The Nt tiles that share
    // one (M tile, K split) - i.e. the same A panel, the bulk of the bytes - get ids L, L+8, ..., so they run on the same
    // XCD at about the same time and the panel is fetched from HBM once and re-read from that XCD's L2.
    int bm, bn, sp;
    {
        const int64_t L = blockIdx.x;
        const int64_t npair = int64_t(Mt) * nsplit;          // (bm, sp) pairs
        const int64_t grp = L / (8 * int64_t(Nt));            // group of 8 pairs
        const int64_t rem = L % (8 * int64_t(Nt));
        const int64_t pair = grp * 8 + (rem % 8);
        bn = int(rem / 8);
        if (pair >= npair) return;
        bm = int(pair % Mt);
        sp = int(pair / Mt);
    }
    // symmetric call (same column list on both sides): tiles that lie inside the new x new square and entirely above
    // its diagonal are skipped; the reduce kernel mirrors them from the lower triangle
    if (symmetric && (m_pos0 + bm * BM >= n_pos0) && (m_pos0 + (bm + 1) * BM <= n_pos0 + ncol0 + bn * BN)) return;

    __shared__ T As[BM * LDK];
    __shared__ T Bs[BN * LDK];
    __shared__ T Ws[KT];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv / WN, wn = wv % WN;
    const int64_t k0 = int64_t(sp) * kchunk;
    const int64_t kend = min(n, k0 + kchunk);

    // staging roles
    const int sca = tid / (KT / RA), sra = (tid % (KT / RA)) * RA;
    const int scb = tid / (KT / RB), srb = (tid % (KT / RB)) * RB;
    const int am = bm * BM + sca, bnn = ncol0 + bn * BN + scb;
    const bool a_ok = am < M, b_ok = bnn < N;
    const int64_t ja = a_ok ? int64_t(mcols[am]) : 0;
    const int64_t jb = b_ok ? int64_t(ncols[bnn]) : 0;

    typename Mfma<T>::acc_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = T(0);

    T ra[RA], rb[RB], rw = T(0);
    auto fetch = [&](int64_t k) {
        load_rows<T, Acc, VECOK, RA>(X, ja, a_ok, k + sra, kend, ra);
        load_rows<T, Acc, VECOK, RB>(X, jb, b_ok, k + srb, kend, rb);
        if (tid < KT) rw = (k + tid < kend) ? w[k + tid] : T(0);
    };

    if (k0 < kend) fetch(k0);
    const int fr = (lane & 15), fk = (lane >> 4);
    // STRIP: row tiles of this wave that hold rows below M (wave-uniform)
    const int tm_live = STRIP ? min(TM, (M - bm * BM - wm * (BM / WM) + 15) / 16) : TM;
    for (int64_t k = k0; k < kend; k += KT) {
        __syncthreads(); // previous stage fully consumed
#pragma unroll
        for (int e = 0; e < RA; ++e) As[sca * LDK + sra + e] = ra[e];
#pragma unroll
        for (int e = 0; e < RB; ++e) Bs[scb * LDK + srb + e] = rb[e];
        if (tid < KT) Ws[tid] = rw;
        __syncthreads();
        if (k + KT < kend) fetch(k + KT); // in flight while the MFMAs run
#pragma unroll
        for (int kk = 0; kk < KT / 4; ++kk) {
            T a[TM] = {}, b[TN];
            const T wk = Ws[kk * 4 + fk];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                if (!STRIP || i < tm_live) a[i] = As[(wm * (BM / WM) + i * 16 + fr) * LDK + kk * 4 + fk];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[(wn * CW + j * 16 + fr) * LDK + kk * 4 + fk] * wk;
#pragma unroll
            for (int i = 0; i < TM; ++i)
                if (!STRIP || i < tm_live) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = Mfma<T>::run(a[i], b[j], acc[i][j]);
                }
        }
    }

    // partial tile -> part[sp][col][row]  (col = N index, row = M index)
    T* P = part + int64_t(sp) * Mpad * Npad;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        if (STRIP && i >= tm_live) continue; // (the reduce only reads rows below M)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = bm * BM + wm * (BM / WM) + i * 16 + Mfma<T>::row(lane, e);
                const int col = ncol0 + bn * BN + wn * CW + j * 16 + (lane & 15);
                P[int64_t(col) * Mpad + row] = acc[i][j][e];
            }
    }
}

template <class T, class Acc, bool VECOK, int BN>
__global__ __launch_bounds__(GT, 2) void gram_kernel(Acc X, const T* __restrict__ w, const int32_t* __restrict__ mcols,
                                                  int32_t M, const int32_t* __restrict__ ncols, int32_t N, int64_t n,
                                                  int64_t kchunk, int32_t m_pos0, int32_t n_pos0, int symmetric,
                                                  T* __restrict__ part, int64_t Mpad, int64_t Npad, int32_t ncol0,
                                                  int32_t Mt, int32_t Nt, int32_t nsplit) {
    gram_body<T, Acc, VECOK, BN>(X, w, mcols, M, ncols, N, n, kchunk, m_pos0, n_pos0, symmetric, part, Mpad, Npad, ncol0, Mt,
                                 Nt, nsplit);
}

// Several M x N blocks (M, N <= 128: one tile each) per launch — the cross blocks of the look-ahead passes: block y of the
// batch = rows cols_base[moff[y]...], columns cols_base[noff[y]...], partials at part + y * nsplit * 128 * 128.  The K-splits
// of all blocks together make one round over the chip, so a block's partial-tile traffic (128 KB written and re-read per
// split) shrinks with the batch: 512 splits = 134 MB for a block built alone, 17 MB in a batch of eight.
template <class T, class Acc, bool VECOK>
__global__ __launch_bounds__(GT, 2) void gram_batch_kernel(Acc X, const T* __restrict__ w, const int32_t* __restrict__ cols_base,
                                                        GramBatch b, int64_t n, int64_t kchunk, T* __restrict__ part,
                                                        int32_t nsplit) {
    const int y = blockIdx.y;
    gram_body<T, Acc, VECOK, 128, 4>(X, w, cols_base + b.moff[y], b.m[y], cols_base + b.noff[y], b.nn[y], n, kchunk, 0, 0, 0,
                                     part + int64_t(y) * nsplit * BM * 128, BM, 128, 0, 1, 1, nsplit);
}

template <class T>
__device__ __forceinline__ void gram_reduce_body(const T* __restrict__ part, int nsplit, int64_t Mpad, int64_t Npad, int32_t M,
                                                 int32_t N, const int32_t* __restrict__ mcols, const int32_t* __restrict__ ncols,
                                                 int32_t m_pos0, int32_t n_pos0, const T* __restrict__ xm, int center,
                                                 int symmetric, T* __restrict__ C, int64_t ldc) {
    // 64 consecutive M indices per block (coalesced partial reads) x 4 interleaved groups of K-splits; the four group
    // sums are combined in a fixed order, so the result does not depend on scheduling
    __shared__ T red[4][64];
    const int ta = threadIdx.x & 63, tg = threadIdx.x >> 6;
    const int a = blockIdx.x * 64 + ta;
    const int b = blockIdx.y;
    const bool live = a < M && b < N;
    const int64_t rp = int64_t(m_pos0) + a, cp = int64_t(n_pos0) + b;
    // inside the (new x new) square every unordered pair is taken from its lower-triangle entry only,
    // so that C is exactly symmetric
    const bool skip = !live || (symmetric && rp >= n_pos0 && rp < cp);
    T s = T(0);
    if (!skip) {
        const int64_t stride = Mpad * Npad;
        const T* base = part + int64_t(b) * Mpad + a;
        int sp = tg;
        for (; sp + 12 < nsplit; sp += 16) {
            const T v0 = base[int64_t(sp) * stride], v1 = base[int64_t(sp + 4) * stride];
            const T v2 = base[int64_t(sp + 8) * stride], v3 = base[int64_t(sp + 12) * stride];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; sp < nsplit; sp += 4) s += base[int64_t(sp) * stride];
    }
    red[tg][ta] = s;
    __syncthreads();
    if (tg != 0 || skip) return;
    s = (red[0][ta] + red[1][ta]) + (red[2][ta] + red[3][ta]);
    if (center) s -= xm[mcols[a]] * xm[ncols[b]];
    C[rp + cp * ldc] = s;
    if (symmetric) C[cp + rp * ldc] = s;
}

template <class T>
__global__ void gram_reduce_kernel(const T* __restrict__ part, int nsplit, int64_t Mpad, int64_t Npad, int32_t M,
                                   int32_t N, const int32_t* __restrict__ mcols, const int32_t* __restrict__ ncols,
                                   int32_t m_pos0, int32_t n_pos0, const T* __restrict__ xm, int center,
                                   int symmetric, T* __restrict__ C, int64_t ldc) {
    gram_reduce_body<T>(part, nsplit, Mpad, Npad, M, N, mcols, ncols, m_pos0, n_pos0, xm, center, symmetric, C, ldc);
}

// grid (2, 128, count): block z of the batch into C_base + dst[z]
template <class T>
__global__ void gram_batch_reduce_kernel(const T* __restrict__ part, int nsplit, GramBatch b,
                                         const int32_t* __restrict__ cols_base, const T* __restrict__ xm, int center,
                                         T* __restrict__ C_base, int64_t ldc) {
    const int z = blockIdx.z;
    gram_reduce_body<T>(part + int64_t(z) * nsplit * BM * 128, nsplit, BM, 128, b.m[z], b.nn[z], cols_base + b.moff[z],
                        cols_base + b.noff[z], 0, 0, xm, center, 0, C_base + b.dst[z], ldc);
}

// ---- symmetric diagonal block of <= 64 or <= 128 columns (the panel engine's blocks) --------------------------------------------
// D = X_B^T W X_B for ONE column list on both sides: a single SB x KT panel is staged per stage (it feeds the A and the B
// fragments), and only the lower-triangle 16x16 tiles of the tile grid are computed (10 of 16 for SB = 64, 36 of 64 for
// SB = 128; the reduce kernel mirrors them), spread evenly over the four waves in runs that share fragments.
// Grid = K-splits only.
template <int W, int SB> struct SyrkPlan;
// SB = 32: the three lower tiles of the 2x2 grid, one per wave (the fourth wave only helps staging)
template <> struct SyrkPlan<0, 32> { static constexpr int N = 1; static constexpr int R[1] = {0}; static constexpr int C[1] = {0}; };
template <> struct SyrkPlan<1, 32> { static constexpr int N = 1; static constexpr int R[1] = {1}; static constexpr int C[1] = {0}; };
template <> struct SyrkPlan<2, 32> { static constexpr int N = 1; static constexpr int R[1] = {1}; static constexpr int C[1] = {1}; };
template <> struct SyrkPlan<3, 32> { static constexpr int N = 0; static constexpr int R[1] = {0}; static constexpr int C[1] = {0}; };
// SB = 64: 3 / 3 / 2 / 2 tiles
template <> struct SyrkPlan<0, 64> { static constexpr int N = 3; static constexpr int R[3] = {0, 1, 1}; static constexpr int C[3] = {0, 0, 1}; };
template <> struct SyrkPlan<1, 64> { static constexpr int N = 3; static constexpr int R[3] = {2, 2, 2}; static constexpr int C[3] = {0, 1, 2}; };
template <> struct SyrkPlan<2, 64> { static constexpr int N = 2; static constexpr int R[2] = {3, 3}; static constexpr int C[2] = {0, 1}; };
template <> struct SyrkPlan<3, 64> { static constexpr int N = 2; static constexpr int R[2] = {3, 3}; static constexpr int C[2] = {2, 3}; };
// SB = 128: 9 tiles per wave (rows 0-2 + row 3 cols 0-2 | (3,3), row 4, row 5 cols 0-2 | row 5 cols 3-5, row 6 cols 0-5 | (6,6), row 7)
template <> struct SyrkPlan<0, 128> {
    static constexpr int N = 9;
    static constexpr int R[9] = {0, 1, 1, 2, 2, 2, 3, 3, 3};
    static constexpr int C[9] = {0, 0, 1, 0, 1, 2, 0, 1, 2};
};
template <> struct SyrkPlan<1, 128> {
    static constexpr int N = 9;
    static constexpr int R[9] = {3, 4, 4, 4, 4, 4, 5, 5, 5};
    static constexpr int C[9] = {3, 0, 1, 2, 3, 4, 0, 1, 2};
};
template <> struct SyrkPlan<2, 128> {
    static constexpr int N = 9;
    static constexpr int R[9] = {5, 5, 5, 6, 6, 6, 6, 6, 6};
    static constexpr int C[9] = {3, 4, 5, 0, 1, 2, 3, 4, 5};
};
template <> struct SyrkPlan<3, 128> {
    static constexpr int N = 9;
    static constexpr int R[9] = {6, 7, 7, 7, 7, 7, 7, 7, 7};
    static constexpr int C[9] = {6, 0, 1, 2, 3, 4, 5, 6, 7};
};

template <class T, class Plan, int NA>
__device__ __forceinline__ void syrk_stage(const T* __restrict__ As, const T* __restrict__ Ws, int fr, int fk,
                                           typename Mfma<T>::acc_t (&acc)[NA]) {
#pragma unroll
    for (int kk = 0; kk < KT / 4; ++kk) {
        const T wk = Ws[kk * 4 + fk];
        const int ko = kk * 4 + fk;
#pragma unroll
        for (int t = 0; t < Plan::N; ++t) {
            const T a = As[(Plan::R[t] * 16 + fr) * LDK + ko];
            const T b = As[(Plan::C[t] * 16 + fr) * LDK + ko] * wk;
            acc[t] = Mfma<T>::run(a, b, acc[t]);
        }
    }
}
template <class T, class Plan, int NA, int SB>
__device__ __forceinline__ void syrk_store(T* __restrict__ P, int lane, const typename Mfma<T>::acc_t (&acc)[NA]) {
#pragma unroll
    for (int t = 0; t < Plan::N; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            P[(Plan::C[t] * 16 + (lane & 15)) * SB + Plan::R[t] * 16 + Mfma<T>::row(lane, e)] = acc[t][e];
}

// one K-split [k0, kend) of one diagonal block: partial lower-triangle tiles into P (SB x SB)
template <class T, class Acc, bool VECOK, int SB>
__device__ __forceinline__ void syrk_body(const Acc& X, const T* __restrict__ w, const int32_t* __restrict__ cols, int32_t M,
                                          int64_t k0, int64_t kend, T* __restrict__ P) {
    constexpr int RA = KT * SB / GT; // rows of one column staged per thread (8 or 16)
    constexpr int NA = SB == 32 ? 1 : (SB == 64 ? 3 : 9);
    __shared__ T As[SB * LDK];
    __shared__ T Ws[KT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sc = tid / (KT / RA), sr = (tid % (KT / RA)) * RA;
    const bool c_ok = sc < M;
    const int64_t jc = c_ok ? int64_t(cols[sc]) : 0;

    typename Mfma<T>::acc_t acc[NA];
#pragma unroll
    for (int t = 0; t < NA; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][e] = T(0);

    T ra[RA], rw = T(0);
    auto fetch = [&](int64_t k) {
        load_rows<T, Acc, VECOK, RA>(X, jc, c_ok, k + sr, kend, ra);
        if (tid < KT) rw = (k + tid < kend) ? w[k + tid] : T(0);
    };
    if (k0 < kend) fetch(k0);
    const int fr = (lane & 15), fk = (lane >> 4);
    for (int64_t k = k0; k < kend; k += KT) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < RA; ++e) As[sc * LDK + sr + e] = ra[e];
        if (tid < KT) Ws[tid] = rw;
        __syncthreads();
        if (k + KT < kend) fetch(k + KT);
        if (wv == 0) syrk_stage<T, SyrkPlan<0, SB>, NA>(As, Ws, fr, fk, acc);
        else if (wv == 1) syrk_stage<T, SyrkPlan<1, SB>, NA>(As, Ws, fr, fk, acc);
        else if (wv == 2) syrk_stage<T, SyrkPlan<2, SB>, NA>(As, Ws, fr, fk, acc);
        else syrk_stage<T, SyrkPlan<3, SB>, NA>(As, Ws, fr, fk, acc);
    }
    // partial tiles -> P[col][row] (SB x SB, only the lower-triangle tiles are written / ever read)
    if (wv == 0) syrk_store<T, SyrkPlan<0, SB>, NA, SB>(P, lane, acc);
    else if (wv == 1) syrk_store<T, SyrkPlan<1, SB>, NA, SB>(P, lane, acc);
    else if (wv == 2) syrk_store<T, SyrkPlan<2, SB>, NA, SB>(P, lane, acc);
    else syrk_store<T, SyrkPlan<3, SB>, NA, SB>(P, lane, acc);
}

// The same for a 2-bit SNP design.  A stage of the generic body is 32 rows x SB columns = 8 * SB BYTES of calls: loaded a byte
// at a time one stage ahead, the MFMAs of a stage (0.6 - 1.3 us) are over long before its loads return from HBM, and the
// matrix pipes idle two thirds of the time.  Here a workgroup brings a SUPER-stage of 256 rows (64 bytes per column) with one
// 16-byte load per lane, a whole super-stage (8 stages) ahead, parks the raw bytes in LDS, and every stage decodes its 32 rows
// from there (LDS latency instead of HBM latency).  k0 must be a multiple of 256 rows (syrk shapes round kchunk up to it for
// SNP designs); columns are 64-byte aligned (ldb), so every 16-byte load is aligned and stays inside the column's padding.
template <class T, int SB>
__device__ __forceinline__ void syrk_body_snp(const SnpAcc<T>& X, const T* __restrict__ w, const int32_t* __restrict__ cols,
                                              int32_t M, int64_t k0, int64_t kend, T* __restrict__ P) {
    constexpr int RA = KT * SB / GT;       // rows of one column decoded per thread and stage (4, 8 or 16)
    constexpr int NA = SB == 32 ? 1 : (SB == 64 ? 3 : 9);
    constexpr int SS = 256;                // rows per super-stage
    constexpr int NCH = SB * 4;            // 16-byte chunks per super-stage (4 per column)
    constexpr int CPT = (NCH + GT - 1) / GT; // chunks per thread (1 or 2)
    __shared__ T As[SB * LDK];
    __shared__ T Ws[KT];
    __shared__ u4v_t raw[2][NCH];          // raw[.][c * 4 + q]: bytes [16 q, 16 q + 16) of column c's super-stage
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sc = tid / (KT / RA), sr = (tid % (KT / RA)) * RA;
    const bool c_ok = sc < M;
    const T imp = c_ok ? X.impute[cols[sc]] : T(0);

    typename Mfma<T>::acc_t acc[NA];
#pragma unroll
    for (int t = 0; t < NA; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][e] = T(0);

    // this thread's chunks of a super-stage
    const uint8_t* gp[CPT];
    bool g_ok[CPT];
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
        const int ch = tid + u * GT;
        const int c = ch >> 2, q = ch & 3;
        g_ok[u] = ch < NCH && c < M;
        gp[u] = g_ok[u] ? X.bits + int64_t(cols[c]) * X.ldb + q * 16 : X.bits;
    }
    u4v_t rg[CPT];
    auto fetch_super = [&](int64_t k) { // rows [k, k + 256)
#pragma unroll
        for (int u = 0; u < CPT; ++u)
            rg[u] = g_ok[u] ? __builtin_nontemporal_load(reinterpret_cast<const u4v_t*>(gp[u] + (k >> 2))) : u4v_t{0, 0, 0, 0};
    };
    T rw = T(0);
    auto fetch_w = [&](int64_t k) { if (tid < KT) rw = (k + tid < kend) ? w[k + tid] : T(0); };

    if (k0 < kend) { fetch_super(k0); fetch_w(k0); }
    const int fr = (lane & 15), fk = (lane >> 4);
    int buf = 0;
    for (int64_t ks = k0; ks < kend; ks += SS, buf ^= 1) {
        // park the super-stage (the buffer was last read two super-stages ago: the barriers of the stages in between order it)
#pragma unroll
        for (int u = 0; u < CPT; ++u)
            if (tid + u * GT < NCH) raw[buf][tid + u * GT] = rg[u];
        if (ks + SS < kend) fetch_super(ks + SS);
        const unsigned char* rb = reinterpret_cast<const unsigned char*>(&raw[buf][0]) + sc * 64;
#pragma unroll 1
        for (int st = 0; st < SS / KT; ++st) {
            const int64_t k = ks + int64_t(st) * KT;
            if (k >= kend) break;
            __syncthreads(); // raw[buf] visible (st == 0); As / Ws of the previous stage consumed
            // decode this thread's RA rows of column sc: RA / 4 bytes at byte (32 st + sr) / 4 of the column's 64
            unsigned bits = 0;
            const int bo = (st * KT + sr) >> 2;
            if constexpr (RA == 4) bits = rb[bo];
            else if constexpr (RA == 8) bits = *reinterpret_cast<const unsigned short*>(rb + bo);
            else bits = *reinterpret_cast<const unsigned*>(rb + bo);
#pragma unroll
            for (int e = 0; e < RA; ++e) {
                const unsigned c = (bits >> (2 * e)) & 3u;
                As[sc * LDK + sr + e] = c_ok ? (c == 3u ? imp : T(c)) : T(0);
            }
            if (tid < KT) Ws[tid] = rw;
            __syncthreads();
            fetch_w(k + KT);
            if (wv == 0) syrk_stage<T, SyrkPlan<0, SB>, NA>(As, Ws, fr, fk, acc);
            else if (wv == 1) syrk_stage<T, SyrkPlan<1, SB>, NA>(As, Ws, fr, fk, acc);
            else if (wv == 2) syrk_stage<T, SyrkPlan<2, SB>, NA>(As, Ws, fr, fk, acc);
            else syrk_stage<T, SyrkPlan<3, SB>, NA>(As, Ws, fr, fk, acc);
        }
    }
    if (wv == 0) syrk_store<T, SyrkPlan<0, SB>, NA, SB>(P, lane, acc);
    else if (wv == 1) syrk_store<T, SyrkPlan<1, SB>, NA, SB>(P, lane, acc);
    else if (wv == 2) syrk_store<T, SyrkPlan<2, SB>, NA, SB>(P, lane, acc);
    else syrk_store<T, SyrkPlan<3, SB>, NA, SB>(P, lane, acc);
}

// SB = 64 on a 2-bit design, fed from registers.  The staged body above decodes every call once per workgroup but pays two
// barriers and an LDS round trip per 32 rows: alone it reaches 0.44 of the f64 MFMA rate, in a config-4 path 0.28.  Here every
// WAVE owns a quarter of the workgroup's rows and all ten lower-triangle tiles: lane (fr, fk) loads the 32-bit word (16 calls)
// of column 16 t + fr for the four column tiles t, decodes call 4 s + fk of each for s = 0..3 — the element it has to supply to
// v_mfma_16x16x4 both as the row-tile operand and, times the row's weight, as the column-tile operand — and issues ten MFMAs
// per s: 16 decodes and 16 multiplies feed 40 MFMAs, nothing goes through LDS and nothing waits at a barrier until the four
// waves add up their tiles at the end.  Same partial layout as syrk_store.
// MEANS: the weighted column sums  sum_i w_i x_ic  of the block's columns ride along -- the column-tile operand b = x * w is
// what they are made of: four additions per sixteen decodes next to forty MFMAs --, reduced over the lanes and the four waves
// at the end and parked in the partial tile's UNUSED upper triangle (syrk_means_index): a block built for an IRLS iteration
// brings the means its centring needs, and nobody sweeps the screen columns for them (solver_screen.hpp::step_means_now).
__host__ __device__ constexpr int syrk_means_index(int c) { return (48 + (c & 15)) * 64 + (c >> 4); } // tile (rows 0-15, columns 48-63)
template <class T, bool MEANS = false>
__device__ __forceinline__ void syrk_body_snp_reg64(const SnpAcc<T>& X, const T* __restrict__ w, const int32_t* __restrict__ cols,
                                                    int32_t M, int64_t k0, int64_t kend, T* __restrict__ P) {
    constexpr int SB = 64, NT = 10;
    constexpr int TR[NT] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3};
    constexpr int TC[NT] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3};
    __shared__ T red[2][NT * 256];
    __shared__ T mred[MEANS ? 4 : 1][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fk = lane >> 4;
    T ms[4] = {T(0), T(0), T(0), T(0)};
    typename Mfma<T>::acc_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][e] = T(0);
    // this wave's rows: a quarter of [k0, kend) in whole 64-row units (k0 is a multiple of 256): one 16-byte load per column
    // tile and lane brings 64 rows, requested a whole unit (1.3 k MFMA cycles x 4) ahead of its use
    const int64_t len = kend > k0 ? kend - k0 : 0;
    const int64_t q64 = ((len + 255) / 256) * 64; // rows per wave
    const int64_t r0 = k0 + int64_t(wv) * q64, r1 = min(kend, r0 + q64);
    const u4v_t* cp[4];
    T imp[4];
    bool ok[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int c = 16 * t + fr;
        ok[t] = c < M;
        const int64_t col = cols[ok[t] ? c : 0];
        cp[t] = reinterpret_cast<const u4v_t*>(X.bits + col * X.ldb);
        imp[t] = X.impute[col];
    }
    u4v_t wd[4];
    T wr[16];
    auto fetch = [&](int64_t r) { // the 64 rows from r on
#pragma unroll
        for (int t = 0; t < 4; ++t) wd[t] = ok[t] ? __builtin_nontemporal_load(cp[t] + (r >> 6)) : u4v_t{0, 0, 0, 0};
#pragma unroll
        for (int s4 = 0; s4 < 16; ++s4) {
            const int64_t i = r + 4 * s4 + fk;
            wr[s4] = i < kend ? w[i] : T(0);
        }
    };
    if (r0 < r1) fetch(r0);
    for (int64_t r = r0; r < r1; r += 64) {
        u4v_t cw[4];
        T cwr[16];
#pragma unroll
        for (int t = 0; t < 4; ++t) cw[t] = wd[t];
#pragma unroll
        for (int s4 = 0; s4 < 16; ++s4) cwr[s4] = wr[s4];
        if (r + 64 < r1) fetch(r + 64);
#pragma unroll
        for (int g = 0; g < 4; ++g) {     // 16-row word g of the unit
            unsigned sh[4];               // the word with this lane's four calls at bits 0, 8, 16, 24
#pragma unroll
            for (int t = 0; t < 4; ++t) sh[t] = cw[t][g] >> (2 * fk);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                T a[4], b[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const unsigned c = (sh[t] >> (8 * s4)) & 3u; // (a column beyond M was fetched as zeros: code 0)
                    const T x = c == 3u ? imp[t] : T(c);
                    a[t] = x;
                    b[t] = x * cwr[4 * g + s4];
                    if constexpr (MEANS) ms[t] += b[t];
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = Mfma<T>::run(a[TR[t]], b[TC[t]], acc[t]);
            }
        }
    }
    // the four waves' tiles, added in a fixed order: (w0 + w2) + (w1 + w3), two 20 KB slots of LDS
    auto put = [&](int slot) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[slot][(t * 4 + e) * 64 + lane] = acc[t][e];
    };
    auto add = [&](int slot) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[t][e] += red[slot][(t * 4 + e) * 64 + lane];
    };
    if constexpr (MEANS) { // over the four row quarters of a lane group (fk), then the four waves through LDS; fixed order
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            ms[t] += __shfl_xor(ms[t], 16, 64);
            ms[t] += __shfl_xor(ms[t], 32, 64);
            if (fk == 0) mred[wv][16 * t + fr] = ms[t];
        }
    }
    if (wv >= 2) put(wv - 2);
    __syncthreads();
    if (wv < 2) add(wv);
    __syncthreads();
    if (wv == 1) put(0);
    __syncthreads();
    if (wv == 0) {
        add(0);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) P[(TC[t] * 16 + fr) * SB + TR[t] * 16 + Mfma<T>::row(lane, e)] = acc[t][e];
        if constexpr (MEANS) P[syrk_means_index(lane)] = (mred[0][lane] + mred[2][lane]) + (mred[1][lane] + mred[3][lane]);
    }
}

template <class A> struct IsSnpAcc { static constexpr bool value = false; };
template <class T> struct IsSnpAcc<SnpAcc<T>> { static constexpr bool value = true; };

template <class T, class Acc, bool VECOK, int SB, bool MEANS = false>
__device__ __forceinline__ void syrk_any(const Acc& X, const T* __restrict__ w, const int32_t* __restrict__ cols, int32_t M,
                                         int64_t k0, int64_t kend, T* __restrict__ P) {
    if constexpr (IsSnpAcc<Acc>::value && SB == 64) {
#ifdef AHIP_SYRK_STAGED
        syrk_body_snp<T, SB>(X, w, cols, M, k0, kend, P);
#else
        syrk_body_snp_reg64<T, MEANS>(X, w, cols, M, k0, kend, P);
#endif
    } else if constexpr (IsSnpAcc<Acc>::value) syrk_body_snp<T, SB>(X, w, cols, M, k0, kend, P);
    else syrk_body<T, Acc, VECOK, SB>(X, w, cols, M, k0, kend, P);
}

template <class T, class Acc, bool VECOK, int SB>
__global__ __launch_bounds__(GT, 2) void syrk_kernel(Acc X, const T* __restrict__ w, const int32_t* __restrict__ cols,
                                                     int32_t M, int64_t n, int64_t kchunk, T* __restrict__ part) {
    const int sp = blockIdx.x;
    const int64_t k0 = int64_t(sp) * kchunk;
    syrk_any<T, Acc, VECOK, SB>(X, w, cols, M, k0, min(n, k0 + kchunk), part + int64_t(sp) * SB * SB);
}

// Several diagonal blocks in ONE launch (blockIdx.y = block, blockIdx.x = K-split).  Under IRLS weights every block of a pass
// is stale at once: building them one launch (512 K-splits) at a time makes each workgroup stream ~1000 rows and then write
// 20 KB of partial tiles — more bytes than it read — which a second kernel has to read back; with the blocks of a pass batched
// the same 512-1024 workgroups cover `count` blocks with 512 / count splits each, i.e. count times fewer partials per block
// and count times longer K loops per workgroup.
template <class T, class Acc, bool VECOK, int SB, bool MEANS = false>
__global__ __launch_bounds__(GT, 2) void syrk_batch_kernel(Acc X, const T* __restrict__ w, const int32_t* __restrict__ cols_base,
                                                           SyrkBatch b, int64_t n, int64_t kchunk, int nsplit,
                                                           T* __restrict__ part) {
    const int sp = blockIdx.x, y = blockIdx.y;
    const int64_t k0 = int64_t(sp) * kchunk;
    syrk_any<T, Acc, VECOK, SB, MEANS>(X, w, cols_base + b.off[y], b.nb[y], k0, min(n, k0 + kchunk),
                                       part + (int64_t(y) * nsplit + sp) * SB * SB);
}

// the weighted column sums the MEANS builds parked in their partial tiles: summed over the K-splits in a fixed order and
// scattered by design column (one workgroup per block of the batch, one thread per column)
template <class T>
__global__ __launch_bounds__(64) void syrk_batch_means_kernel(const T* __restrict__ part, int nsplit, SyrkBatch b,
                                                              const int32_t* __restrict__ cols_base, T* __restrict__ xm_by_col) {
    const int y = blockIdx.x, c = threadIdx.x;
    if (c >= b.nb[y]) return;
    const T* base = part + int64_t(y) * nsplit * 64 * 64 + syrk_means_index(c);
    T s = T(0);
    int sp = 0;
    for (; sp + 3 < nsplit; sp += 4) {
        const T v0 = base[int64_t(sp) * 4096], v1 = base[int64_t(sp + 1) * 4096], v2 = base[int64_t(sp + 2) * 4096],
                v3 = base[int64_t(sp + 3) * 4096];
        s += v0; s += v1; s += v2; s += v3;
    }
    for (; sp < nsplit; ++sp) s += base[int64_t(sp) * 4096];
    xm_by_col[cols_base[b.off[y] + c]] = s;
}

// deterministic sum of the K-split partials of every block of a batch, centring, symmetric write into the block's slot
template <class T>
__global__ void syrk_batch_reduce_kernel(const T* __restrict__ part, int nsplit, int SB, SyrkBatch b,
                                         const int32_t* __restrict__ cols_base, const T* __restrict__ xm, int center,
                                         T* __restrict__ C_base, int64_t ldc) {
    __shared__ T red[4][64];
    const int ta = threadIdx.x & 63, tg = threadIdx.x >> 6;
    const int y = blockIdx.z;
    const int M = b.nb[y];
    const int a = blockIdx.x * 64 + ta, bb = blockIdx.y;
    const bool skip = !(a < M && bb < M) || a < bb; // lower triangle only; mirrored below
    T s = T(0);
    if (!skip) {
        const int64_t stride = int64_t(SB) * SB;
        const T* base = part + int64_t(y) * nsplit * stride + int64_t(bb) * SB + a;
        int sp = tg;
        for (; sp + 12 < nsplit; sp += 16) {
            const T v0 = base[int64_t(sp) * stride], v1 = base[int64_t(sp + 4) * stride];
            const T v2 = base[int64_t(sp + 8) * stride], v3 = base[int64_t(sp + 12) * stride];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; sp < nsplit; sp += 4) s += base[int64_t(sp) * stride];
    }
    red[tg][ta] = s;
    __syncthreads();
    if (tg != 0 || skip) return;
    s = (red[0][ta] + red[1][ta]) + (red[2][ta] + red[3][ta]);
    const int32_t* cols = cols_base + b.off[y];
    if (center) s -= xm[cols[a]] * xm[cols[bb]];
    T* C = C_base + b.dst[y];
    C[a + int64_t(bb) * ldc] = s;
    C[bb + int64_t(a) * ldc] = s;
}

inline void syrk_shape(int64_t n, int& nsplit, int64_t& kchunk) {
    int64_t want = 512; // one full round of 2 resident blocks per CU
    const int64_t max_split = (n + KT * 8 - 1) / (KT * 8);
    if (want > max_split) want = max_split;
    if (want < 1) want = 1;
    kchunk = (n + want - 1) / want;
    kchunk = ((kchunk + 255) / 256) * 256; // whole super-stages of the SNP body (a multiple of KT as well)
    const int64_t ns = (n + kchunk - 1) / kchunk;
    nsplit = int(ns < 1 ? 1 : ns);
}

template <class T, class Acc>
void syrk_launch(Acc acc, bool vecok, const T* w, const int32_t* cols, int32_t M, int64_t n, const T* xm, bool center,
                 T* C, int64_t ldc, T* work, hipStream_t s) {
    if (M <= 0) return;
    int nsplit;
    int64_t kchunk;
    syrk_shape(n, nsplit, kchunk);
    const int64_t SB = M <= 32 ? 32 : (M <= 64 ? 64 : 128);
#define AHIP_SYRK(VOK, SBV) \
    hipLaunchKernelGGL((syrk_kernel<T, Acc, VOK, SBV>), dim3((unsigned)nsplit), dim3(GT), 0, s, acc, w, cols, M, n, kchunk, work)
    if (M <= 32) { if (vecok) AHIP_SYRK(true, 32); else AHIP_SYRK(false, 32); }
    else if (M <= 64) { if (vecok) AHIP_SYRK(true, 64); else AHIP_SYRK(false, 64); }
    else { if (vecok) AHIP_SYRK(true, 128); else AHIP_SYRK(false, 128); }
#undef AHIP_SYRK
    hipLaunchKernelGGL((gram_reduce_kernel<T>), dim3((unsigned)((M + 63) / 64), (unsigned)M), dim3(256), 0, s, work, nsplit, SB,
                       SB, M, M, cols, cols, 0, 0, xm, center ? 1 : 0, 1, C, ldc);
}

inline void syrk_batch_shape(int64_t n, int count, int& nsplit, int64_t& kchunk) {
    // about one full round of 2 resident workgroups per CU over all the blocks of the batch (count = 1: the single-block shape)
    int64_t want = (int64_t(t_small_gram_wgs) + count - 1) / count;
    const int64_t max_split = (n + KT * 8 - 1) / (KT * 8);
    if (want > max_split) want = max_split;
    if (want < 1) want = 1;
    kchunk = (n + want - 1) / want;
    kchunk = ((kchunk + 255) / 256) * 256; // whole super-stages of the SNP body (a multiple of KT as well)
    const int64_t ns = (n + kchunk - 1) / kchunk;
    nsplit = int(ns < 1 ? 1 : ns);
}

template <class T, class Acc>
void syrk_batch_launch(Acc acc, bool vecok, const T* w, const int32_t* cols_base, const SyrkBatch& b, int64_t n, const T* xm,
                       bool center, T* C_base, int64_t ldc, T* work, hipStream_t s, T* xm_build = nullptr) {
    if (b.count <= 0) return;
    int nsplit;
    int64_t kchunk;
    syrk_batch_shape(n, b.count, nsplit, kchunk);
    int mx = 0;
    for (int y = 0; y < b.count; ++y) mx = std::max(mx, int(b.nb[y]));
    const int SB = mx <= 32 ? 32 : (mx <= 64 ? 64 : 128);
    const dim3 grid((unsigned)nsplit, (unsigned)b.count);
#define AHIP_SYRKB(VOK, SBV) \
    hipLaunchKernelGGL((syrk_batch_kernel<T, Acc, VOK, SBV>), grid, dim3(GT), 0, s, acc, w, cols_base, b, n, kchunk, nsplit, work)
    bool built_means = false;
    if constexpr (IsSnpAcc<Acc>::value) {
#ifndef AHIP_SYRK_STAGED
        if (xm_build != nullptr && SB == 64) { // the builds bring the means of their own columns (syrk_body_snp_reg64<MEANS>)
            hipLaunchKernelGGL((syrk_batch_kernel<T, Acc, true, 64, true>), grid, dim3(GT), 0, s, acc, w, cols_base, b, n, kchunk,
                               nsplit, work);
            hipLaunchKernelGGL((syrk_batch_means_kernel<T>), dim3((unsigned)b.count), dim3(64), 0, s, work, nsplit, b, cols_base,
                               xm_build);
            xm = xm_build;
            built_means = true;
        }
#endif
    }
    if (!built_means) {
        if (SB == 32) { if (vecok) AHIP_SYRKB(true, 32); else AHIP_SYRKB(false, 32); }
        else if (SB == 64) { if (vecok) AHIP_SYRKB(true, 64); else AHIP_SYRKB(false, 64); }
        else { if (vecok) AHIP_SYRKB(true, 128); else AHIP_SYRKB(false, 128); }
    }
#undef AHIP_SYRKB
    hipLaunchKernelGGL((syrk_batch_reduce_kernel<T>), dim3((unsigned)((mx + 63) / 64), (unsigned)mx, (unsigned)b.count), dim3(256),
                       0, s, work, nsplit, SB, b, cols_base, xm, center ? 1 : 0, C_base, ldc);
}

// N tiling: full 128-wide tiles, then the remainder as one 64-wide tile when it fits (less padding than a 128 tile)
struct GramShape {
    int64_t Mt, n128, n64, Npad, kchunk;
    int nsplit;
};
inline GramShape gram_shape(int64_t n, int64_t M, int64_t N) {
    GramShape g;
    g.Mt = (M + BM - 1) / BM;
    g.n128 = N / 128;
    const int64_t rem = N - g.n128 * 128;
    g.n64 = 0;
    if (rem > 64) ++g.n128;
    else if (rem > 0) g.n64 = 1;
    g.Npad = g.n128 * 128 + g.n64 * 64;
    const int64_t tiles = g.Mt * (g.n128 + g.n64);
    // Many tiles: ~3072 blocks so that the last partial round over the 256 CUs x 2 resident blocks costs little.  A single
    // diagonal block of the panel engine: one full round (512 blocks) - more K-splits only add partial-tile traffic
    // (128 KB written and re-read per split; measured -18 % at n = 500k).
    const int64_t target = tiles <= 4 ? int64_t(t_small_gram_wgs) : 3072;
    int64_t want = (target + tiles - 1) / tiles;
    const int64_t max_split = (n + KT * 8 - 1) / (KT * 8);
    if (want > max_split) want = max_split;
    const int64_t cap = (int64_t(1) << 28) / (g.Mt * BM * g.Npad); // partial buffer <= 2^28 elements
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    g.kchunk = (n + want - 1) / want;
    g.kchunk = ((g.kchunk + KT - 1) / KT) * KT;
    int64_t ns = (n + g.kchunk - 1) / g.kchunk;
    g.nsplit = int(ns < 1 ? 1 : ns);
    return g;
}

template <class T, class Acc>
void gram_launch(Acc acc, bool vecok, const T* w, const int32_t* mcols, int32_t M, int32_t m_pos0, const int32_t* ncols,
                 int32_t N, int32_t n_pos0, int64_t n, const T* xm, bool center, T* C, int64_t ldc, T* work,
                 hipStream_t s) {
    if (M <= 0 || N <= 0) return;
    const GramShape g = gram_shape(n, M, N);
    const int64_t Mpad = g.Mt * BM, Npad = g.Npad;
    // "symmetric": the N list is the tail (or all) of the M list at the same positions -> skip tiles that lie
    // entirely above the diagonal of the (new x new) square; the reduce kernel mirrors them.
    const int symmetric = (mcols + (n_pos0 - m_pos0) == ncols && m_pos0 + M == n_pos0 + N && n_pos0 >= m_pos0) ? 1 : 0;
    auto blocks = [&](int64_t nt) {
        const int64_t npair = g.Mt * g.nsplit;
        return unsigned(((npair + 7) / 8) * 8 * nt);
    };
#define AHIP_GRAM_LAUNCH(VOK, BNV, NT_, COL0)                                                                           \
    hipLaunchKernelGGL((gram_kernel<T, Acc, VOK, BNV>), dim3(blocks(NT_)), dim3(GT), 0, s, acc, w, mcols, M, ncols, N, n, \
                       g.kchunk, m_pos0, n_pos0, symmetric, work, Mpad, Npad, int32_t(COL0), int32_t(g.Mt), int32_t(NT_), \
                       int32_t(g.nsplit))
    if (g.n128 > 0) {
        if (vecok) AHIP_GRAM_LAUNCH(true, 128, g.n128, 0); else AHIP_GRAM_LAUNCH(false, 128, g.n128, 0);
    }
    if (g.n64 > 0) {
        if (vecok) AHIP_GRAM_LAUNCH(true, 64, 1, g.n128 * 128); else AHIP_GRAM_LAUNCH(false, 64, 1, g.n128 * 128);
    }
#undef AHIP_GRAM_LAUNCH
    hipLaunchKernelGGL((gram_reduce_kernel<T>), dim3((unsigned)((M + 63) / 64), (unsigned)N), dim3(256), 0, s, work,
                       g.nsplit, Mpad, Npad, M, N, mcols, ncols, m_pos0, n_pos0, xm, center ? 1 : 0, symmetric, C, ldc);
}

} // namespace

void set_small_gram_workgroups(int wgs) { t_small_gram_wgs = wgs < 1 ? 512 : wgs; }

int64_t gram_batch_work_elems(int64_t n, int count) {
    const int keep = t_small_gram_wgs;
    t_small_gram_wgs = 512; // the buffer is sized for the widest spread
    int ns;
    int64_t kc;
    syrk_batch_shape(n, count, ns, kc);
    t_small_gram_wgs = keep;
    return int64_t(count) * ns * BM * 128;
}

template <class T, class Acc>
void gram_batch_launch(Acc acc, bool vecok, const T* w, const int32_t* cols_base, const GramBatch& b, int64_t n, const T* xm,
                       bool center, T* C_base, int64_t ldc, T* work, hipStream_t s) {
    if (b.count <= 0) return;
    int nsplit;
    int64_t kchunk;
    syrk_batch_shape(n, b.count, nsplit, kchunk);
    const dim3 grid(unsigned(((nsplit + 7) / 8) * 8), unsigned(b.count));
    if (vecok)
        hipLaunchKernelGGL((gram_batch_kernel<T, Acc, true>), grid, dim3(GT), 0, s, acc, w, cols_base, b, n, kchunk, work,
                           int32_t(nsplit));
    else
        hipLaunchKernelGGL((gram_batch_kernel<T, Acc, false>), grid, dim3(GT), 0, s, acc, w, cols_base, b, n, kchunk, work,
                           int32_t(nsplit));
    hipLaunchKernelGGL((gram_batch_reduce_kernel<T>), dim3(2, 128, unsigned(b.count)), dim3(256), 0, s, work, nsplit, b,
                       cols_base, xm, center ? 1 : 0, C_base, ldc);
}

template <class T>
void launch_gram_batch(const DenseView<T>& X, const T* w, const int32_t* cols_base, const GramBatch& b, const T* xm_by_col,
                       bool center, T* C_base, int64_t ldc, T* work, hipStream_t s) {
    DenseAcc<T> acc{X.X, X.ld};
    const bool vecok = (X.ld % VecOf<T>::N == 0) && ((reinterpret_cast<uintptr_t>(X.X) % 16) == 0);
    gram_batch_launch<T, DenseAcc<T>>(acc, vecok, w, cols_base, b, X.n, xm_by_col, center, C_base, ldc, work, s);
}
template <class T>
void launch_gram_batch_snp(const SnpView& X, const T* impute, const T* w, const int32_t* cols_base, const GramBatch& b,
                           const T* xm_by_col, bool center, T* C_base, int64_t ldc, T* work, hipStream_t s) {
    SnpAcc<T> acc{X.bits, X.ldb, impute};
    gram_batch_launch<T, SnpAcc<T>>(acc, true, w, cols_base, b, X.n, xm_by_col, center, C_base, ldc, work, s);
}

int64_t syrk_batch_work_elems(int64_t n, int count) {
    int nsplit;
    int64_t kchunk;
    const int keep = t_small_gram_wgs;
    t_small_gram_wgs = 512; // the buffer is sized for the widest spread
    syrk_batch_shape(n, count, nsplit, kchunk);
    t_small_gram_wgs = keep;
    return int64_t(nsplit) * count * 128 * 128;
}
template <class T>
void launch_syrk_batch(const DenseView<T>& X, const T* w, const int32_t* cols_base, const SyrkBatch& b, const T* xm_by_col,
                       bool center, T* C_base, int64_t ldc, T* work, hipStream_t s) {
    DenseAcc<T> acc{X.X, X.ld};
    constexpr int V = VecOf<T>::N;
    const bool vecok = (X.ld % V == 0) && ((reinterpret_cast<uintptr_t>(X.X) % 16) == 0);
    syrk_batch_launch<T, DenseAcc<T>>(acc, vecok, w, cols_base, b, X.n, xm_by_col, center, C_base, ldc, work, s);
}
template <class T>
void launch_syrk_batch_snp(const SnpView& X, const T* impute, const T* w, const int32_t* cols_base, const SyrkBatch& b,
                           const T* xm_by_col, bool center, T* C_base, int64_t ldc, T* work, hipStream_t s, T* xm_build) {
    SnpAcc<T> acc{X.bits, X.ldb, impute};
    syrk_batch_launch<T, SnpAcc<T>>(acc, true, w, cols_base, b, X.n, xm_by_col, center, C_base, ldc, work, s, xm_build);
}
bool syrk_batch_snp_brings_means(const SyrkBatch& b) { // (what launch_syrk_batch_snp does with a non-null xm_build)
#ifdef AHIP_SYRK_STAGED
    (void)b;
    return false;
#else
    int mx = 0;
    for (int y = 0; y < b.count; ++y) mx = std::max(mx, int(b.nb[y]));
    return b.count > 0 && mx > 32 && mx <= 64;
#endif
}

int64_t syrk_work_elems(int64_t n, int64_t M) {
    int nsplit;
    int64_t kchunk;
    syrk_shape(n, nsplit, kchunk);
    const int64_t SB = M <= 32 ? 32 : (M <= 64 ? 64 : 128);
    return int64_t(nsplit) * SB * SB;
}
template <class T>
void launch_syrk(const DenseView<T>& X, const T* w, const int32_t* cols, int32_t M, const T* xm_by_col, bool center, T* C,
                   int64_t ldc, T* work, hipStream_t s) {
    DenseAcc<T> acc{X.X, X.ld};
    constexpr int V = VecOf<T>::N;
    const bool vecok = (X.ld % V == 0) && ((reinterpret_cast<uintptr_t>(X.X) % 16) == 0);
    syrk_launch<T, DenseAcc<T>>(acc, vecok, w, cols, M, X.n, xm_by_col, center, C, ldc, work, s);
}
template <class T>
void launch_syrk_snp(const SnpView& X, const T* impute, const T* w, const int32_t* cols, int32_t M, const T* xm_by_col,
                       bool center, T* C, int64_t ldc, T* work, hipStream_t s) {
    SnpAcc<T> acc{X.bits, X.ldb, impute};
    syrk_launch<T, SnpAcc<T>>(acc, true, w, cols, M, X.n, xm_by_col, center, C, ldc, work, s);
}

template <class T>
void launch_syrk_multi(const MultiView<T>& X, const T* w, const int32_t* ucols, int32_t M, T* C, int64_t ldc, T* work,
                       hipStream_t s) {
    if (X.bits) { // 2-bit base: as launch_syrk_snp
        SnpOnesAcc<T> sacc{X.bits, X.ldb, X.impute, int64_t(X.icpt)};
        syrk_launch<T, SnpOnesAcc<T>>(sacc, true, w, ucols, M, X.nb, nullptr, false, C, ldc, work, s);
        return;
    }
    DenseOnesAcc<T> acc{X.X, X.ld, X.ones, int64_t(X.icpt)};
    constexpr int V = VecOf<T>::N;
    const bool vecok = (X.ld % V == 0) && ((reinterpret_cast<uintptr_t>(X.X) % 16) == 0) &&
                       ((reinterpret_cast<uintptr_t>(X.ones) % 16) == 0);
    syrk_launch<T, DenseOnesAcc<T>>(acc, vecok, w, ucols, M, X.nb, nullptr, false, C, ldc, work, s);
}

// general M x N Gram over extended features of the multi-response view: C[a + b*ldc] = sum_i w_i x_{mcols[a]} x_{ncols[b]}
template <class T>
void launch_gram_multi(const MultiView<T>& X, const T* w, const int32_t* mcols, int32_t M, const int32_t* ncols, int32_t N,
                       T* C, int64_t ldc, T* work, hipStream_t s) {
    if (X.bits) {
        SnpOnesAcc<T> sacc{X.bits, X.ldb, X.impute, int64_t(X.icpt)};
        gram_launch<T, SnpOnesAcc<T>>(sacc, true, w, mcols, M, 0, ncols, N, 0, X.nb, nullptr, false, C, ldc, work, s);
        return;
    }
    DenseOnesAcc<T> acc{X.X, X.ld, X.ones, int64_t(X.icpt)};
    constexpr int V = VecOf<T>::N;
    const bool vecok = (X.ld % V == 0) && ((reinterpret_cast<uintptr_t>(X.X) % 16) == 0) &&
                       ((reinterpret_cast<uintptr_t>(X.ones) % 16) == 0);
    gram_launch<T, DenseOnesAcc<T>>(acc, vecok, w, mcols, M, 0, ncols, N, 0, X.nb, nullptr, false, C, ldc, work, s);
}

int64_t gram_work_elems(int64_t n, int64_t M, int64_t N) {
    if (M <= 0 || N <= 0) return 0;
    const int keep = t_small_gram_wgs;
    t_small_gram_wgs = 512; // the buffer is sized for the widest spread
    const GramShape g = gram_shape(n, M, N);
    t_small_gram_wgs = keep;
    return int64_t(g.nsplit) * g.Mt * BM * g.Npad;
}

template <class T>
void launch_gram(const DenseView<T>& X, const T* w, const int32_t* mcols, int32_t M, int32_t m_pos0,
                 const int32_t* ncols, int32_t N, int32_t n_pos0, const T* xm_by_col, bool center, T* C, int64_t ldc,
                 T* work, hipStream_t s) {
    DenseAcc<T> acc{X.X, X.ld};
    constexpr int V = VecOf<T>::N;
    const bool vecok = (X.ld % V == 0) && ((reinterpret_cast<uintptr_t>(X.X) % 16) == 0);
    gram_launch<T, DenseAcc<T>>(acc, vecok, w, mcols, M, m_pos0, ncols, N, n_pos0, X.n, xm_by_col, center, C, ldc, work, s);
}
template <class T>
void launch_gram_snp(const SnpView& X, const T* impute, const T* w, const int32_t* mcols, int32_t M, int32_t m_pos0,
                     const int32_t* ncols, int32_t N, int32_t n_pos0, const T* xm_by_col, bool center, T* C,
                     int64_t ldc, T* work, hipStream_t s) {
    SnpAcc<T> acc{X.bits, X.ldb, impute};
    gram_launch<T, SnpAcc<T>>(acc, true, w, mcols, M, m_pos0, ncols, N, n_pos0, X.n, xm_by_col, center, C, ldc, work, s);
}

#define INST(T)                                                                                                        \
    template void launch_gram<T>(const DenseView<T>&, const T*, const int32_t*, int32_t, int32_t, const int32_t*,      \
                                 int32_t, int32_t, const T*, bool, T*, int64_t, T*, hipStream_t);                      \
    template void launch_gram_snp<T>(const SnpView&, const T*, const T*, const int32_t*, int32_t, int32_t,             \
                                     const int32_t*, int32_t, int32_t, const T*, bool, T*, int64_t, T*, hipStream_t);
INST(double)
INST(float)
#undef INST
#define INST2(T)                                                                                                       \
    template void launch_gram_batch<T>(const DenseView<T>&, const T*, const int32_t*, const GramBatch&, const T*, bool, T*, \
                                       int64_t, T*, hipStream_t);                                                      \
    template void launch_gram_batch_snp<T>(const SnpView&, const T*, const T*, const int32_t*, const GramBatch&, const T*, \
                                           bool, T*, int64_t, T*, hipStream_t);                                        \
    template void launch_syrk_batch<T>(const DenseView<T>&, const T*, const int32_t*, const SyrkBatch&, const T*, bool, T*, \
                                       int64_t, T*, hipStream_t);                                                      \
    template void launch_syrk_batch_snp<T>(const SnpView&, const T*, const T*, const int32_t*, const SyrkBatch&, const T*, \
                                           bool, T*, int64_t, T*, hipStream_t, T*);                                    \
    template void launch_syrk<T>(const DenseView<T>&, const T*, const int32_t*, int32_t, const T*, bool, T*, int64_t, \
                                   T*, hipStream_t);                                                                   \
    template void launch_syrk_snp<T>(const SnpView&, const T*, const T*, const int32_t*, int32_t, const T*, bool, T*, \
                                       int64_t, T*, hipStream_t);
INST2(double)
INST2(float)
template void launch_gram_multi<double>(const MultiView<double>&, const double*, const int32_t*, int32_t, const int32_t*,
                                        int32_t, double*, int64_t, double*, hipStream_t);
template void launch_gram_multi<float>(const MultiView<float>&, const float*, const int32_t*, int32_t, const int32_t*, int32_t,
                                       float*, int64_t, float*, hipStream_t);
template void launch_syrk_multi<double>(const MultiView<double>&, const double*, const int32_t*, int32_t, double*, int64_t,
                                        double*, hipStream_t);
template void launch_syrk_multi<float>(const MultiView<float>&, const float*, const int32_t*, int32_t, float*, int64_t,
                                       float*, hipStream_t);
#undef INST2

} // namespace ahip
