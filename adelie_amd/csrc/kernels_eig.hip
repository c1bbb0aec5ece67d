// kernels_eig.hip — eigen-decomposition of the Gram blocks of newly screened groups, on the device.
//
// The reference takes X_g^T W X_g (centred) of every new screen group, runs Eigen::SelfAdjointEigenSolver on it and keeps
// the eigenvectors as the group's transform and the clamped eigenvalues as its variances (solver_gaussian_naive.hpp:99-125).
// Here the blocks already sit on the device (sub-blocks of the panel engine's diagonal blocks, or of the screen Gram
// matrix), so one wavefront per group runs the cyclic Jacobi iteration on its block in LDS: same rotation order and the
// same formulas as the host routine it replaces (solver.hip::jacobi_eigh, the same cyclic scheme the CPU checker uses), arithmetic in double
// for either design dtype.  Nothing comes back to the host: the solver enqueues this behind the block builds and goes on.
#include "kernels.hpp"

namespace ahip {
namespace {

template <class T>
__global__ __launch_bounds__(64) void grp_eig_kernel(const T* __restrict__ src_base, const EigDesc* __restrict__ desc,
                                                     T* __restrict__ vars, T* __restrict__ V) {
#pragma clang fp contract(off)
    extern __shared__ double eig_sm[];
    const EigDesc d = desc[blockIdx.x];
    const int q = d.q, lane = threadIdx.x;
    const T* S = src_base + d.src;
    if (q == 1) { // solver_gaussian_naive.hpp:99-111
        if (lane == 0) {
            const T v = S[0];
            vars[d.vars_pos] = v > T(0) ? v : T(0);
        }
        return;
    }
    double* A = eig_sm;
    double* Vm = eig_sm + q * q;
    for (int e = lane; e < q * q; e += 64) {
        const int r = e % q, c = e / q;
        A[e] = double(S[r + int64_t(c) * d.ld]);
        Vm[e] = (r == c) ? 1.0 : 0.0;
    }
    __syncthreads();
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0, dg = 0;
        for (int i = 0; i < q; ++i) { // every lane, same order as the host routine
            dg += A[i + i * q] * A[i + i * q];
            for (int j = i + 1; j < q; ++j) off += A[i + j * q] * A[i + j * q];
        }
        if (off == 0 || off <= 1e-32 * (dg + off)) break;
        for (int r = 0; r < q - 1; ++r)
            for (int c = r + 1; c < q; ++c) {
                const double arc = A[r + c * q];
                if (arc == 0.0) continue;
                const double theta = (A[c + c * q] - A[r + r * q]) / (2.0 * arc);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
                __syncthreads(); // every lane has read the pivot entries
                for (int k = lane; k < q; k += 64) {
                    const double x = A[k + r * q], y = A[k + c * q];
                    A[k + r * q] = cs * x - sn * y;
                    A[k + c * q] = sn * x + cs * y;
                }
                __syncthreads();
                for (int k = lane; k < q; k += 64) {
                    const double x = A[r + k * q], y = A[c + k * q];
                    A[r + k * q] = cs * x - sn * y;
                    A[c + k * q] = sn * x + cs * y;
                }
                for (int k = lane; k < q; k += 64) {
                    const double x = Vm[k + r * q], y = Vm[k + c * q];
                    Vm[k + r * q] = cs * x - sn * y;
                    Vm[k + c * q] = sn * x + cs * y;
                }
                __syncthreads();
            }
    }
    __syncthreads();
    // eigenvalues ascending (ties keep their index order), eigenvectors in the columns of V
    for (int k = lane; k < q; k += 64) {
        const double ev = A[k + k * q];
        int rank = 0;
        for (int j = 0; j < q; ++j) {
            const double o = A[j + j * q];
            rank += (o < ev || (o == ev && j < k)) ? 1 : 0;
        }
        vars[d.vars_pos + rank] = T(ev >= 0 ? ev : 0.0);
        for (int i = 0; i < q; ++i) V[d.v_off + i + int64_t(rank) * q] = T(Vm[i + k * q]);
    }
}

} // namespace

template <class T>
void launch_grp_eig(const T* src_base, const EigDesc* desc_dev, int count, int max_q, T* vars, T* V, hipStream_t s) {
    if (count <= 0) return;
    const size_t lds = size_t(2) * size_t(max_q) * size_t(max_q) * sizeof(double);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(grp_eig_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  int(size_t(2) * kEigMaxQ * kEigMaxQ * sizeof(double)));
        attr_done = true;
    }
    hipLaunchKernelGGL((grp_eig_kernel<T>), dim3(unsigned(count)), dim3(64), lds, s, src_base, desc_dev, vars, V);
}
template void launch_grp_eig<double>(const double*, const EigDesc*, int, int, double*, double*, hipStream_t);
template void launch_grp_eig<float>(const float*, const EigDesc*, int, int, float*, float*, hipStream_t);

} // namespace ahip
