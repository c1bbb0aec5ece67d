// fetch_calib.hip — known-byte-count kernels to calibrate rocprofv3's FETCH_SIZE on gfx950 for access patterns other than
// the "16 B per lane, wide coalesced streaming" one the guide's x2 correction was measured on.
//   coalesced16     lane l of a wavefront reads 16 B at base + 16 l (1 KB per wave-load), whole buffer once
//   tile64_strided  thread t owns a "column" of `ldb` bytes and walks it in 64-byte tiles (4 x 16 B loads per tile); adjacent
//                   threads are `ldb` apart — the pattern of sweep_snp_lut_kernel on a 2-bit design (256-row tile = 64 B)
//   tile128_strided the same with 128-byte tiles (a whole line per thread and tile)
//   tile32_strided  32-byte tiles
// Every kernel reads every byte of the buffer exactly once: known bytes = buffer size.  Run under
//   rocprofv3 --pmc FETCH_SIZE -d out -o f -- scripts/ubench/fetch_calib
// and compare the counter (KB) with the printed byte counts (scripts/prof_summary.py prints the PMC table).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__global__ void coalesced16(const u4* __restrict__ src, int64_t n16, unsigned* out) {
    unsigned acc = 0;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += int64_t(gridDim.x) * blockDim.x) {
        const u4 v = __builtin_nontemporal_load(src + i);
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345u) out[0] = acc;
}
template <int TILE>
__global__ void tile_strided(const unsigned char* __restrict__ src, int64_t ldb, int64_t ncols, unsigned* out) {
    const int64_t c = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= ncols) return;
    const unsigned char* col = src + c * ldb;
    unsigned acc = 0;
    for (int64_t t = 0; t < ldb; t += TILE) {
#pragma unroll
        for (int e = 0; e < TILE / 16; ++e) {
            const u4 v = *reinterpret_cast<const u4*>(col + t + 16 * e);
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x12345u) out[0] = acc;
}
int main() {
    const int64_t ldb = 125056, ncols = 16384; // 2.05 GB: columns of a 500k-row 2-bit design (padded to 64 B)
    const int64_t bytes = ldb * ncols;
    unsigned char* buf; unsigned* out;
    CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 1, bytes)); CK(hipMalloc(&out, 64));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto run = [&](const char* name, auto launch) -> int {
        float best = 1e9f;
        for (int it = 0; it < 3; ++it) {
            CK(hipEventRecord(a, 0));
            launch();
            CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best;
        }
        printf("%-16s known bytes %lld (%.1f KB)  %.3f ms  %.2f TB/s\n", name, (long long)bytes, bytes / 1024.0, best, bytes / (best * 1e-3) / 1e12);
        return 0;
    };
    if (run("coalesced16", [&] { hipLaunchKernelGGL(coalesced16, dim3(4096), dim3(256), 0, 0, reinterpret_cast<const u4*>(buf), bytes / 16, out); })) return 1;
    if (run("tile128_strided", [&] { hipLaunchKernelGGL((tile_strided<128>), dim3((ncols + 255) / 256), dim3(256), 0, 0, buf, ldb, ncols, out); })) return 1;
    if (run("tile64_strided", [&] { hipLaunchKernelGGL((tile_strided<64>), dim3((ncols + 255) / 256), dim3(256), 0, 0, buf, ldb, ncols, out); })) return 1;
    if (run("tile32_strided", [&] { hipLaunchKernelGGL((tile_strided<32>), dim3((ncols + 255) / 256), dim3(256), 0, 0, buf, ldb, ncols, out); })) return 1;
    return 0;
}
