// Ceiling of the column-pass access pattern: a tile = 8 columns x 512 rows of 16-byte elements (128-byte pieces at a
// 32 KB stride), 512 threads x 8 elements, load -> barrier -> store, with the LDS footprint of the FFT kernels (64 KB:
// two workgroups per CU) or without (8 workgroups per CU).  hipcc --offload-arch=gfx950 -O3 strided_copy.hip -o strided_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int N1 = 512, N2 = 2048, T = 512, P = 8, C = 8;

// mode 0: strided read (work[k1][j2] pieces) -> contiguous write;  1: contiguous read -> strided write;  2: strided both
template <int MODE>
__global__ __launch_bounds__(T) void k_copy(const double2 * __restrict__ src, double2 * __restrict__ dst, int use_lds) {
    extern __shared__ double2 sm[];
    const int tid = threadIdx.x;
    const int64_t m = (int64_t)N1 * N2;
    unsigned bx = blockIdx.x;
    bx = (bx & 7u) * (gridDim.x >> 3) + (bx >> 3);
    const double2 * s = src + (int64_t)blockIdx.y * m;
    double2 * d = dst + (int64_t)blockIdx.y * m;
    double2 v[P];
#pragma unroll
    for (int k = 0; k < P; ++k) {
        const int e = tid + k * T;
        const int64_t strided = ((int64_t)(e >> 3) * N2) + (int64_t)bx * C + (e & 7);
        const int64_t contig = ((int64_t)bx << 12) + e;
        v[k] = s[(MODE == 1) ? contig : strided];
    }
    if (use_lds) {
#pragma unroll
        for (int k = 0; k < P; ++k) sm[tid + k * T] = v[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < P; ++k) v[k] = sm[(tid + k * T) ^ 1];
    }
#pragma unroll
    for (int k = 0; k < P; ++k) {
        const int e = tid + k * T;
        const int64_t strided = ((int64_t)(e >> 3) * N2) + (int64_t)bx * C + (e & 7);
        const int64_t contig = ((int64_t)bx << 12) + e;
        d[(MODE == 0) ? contig : strided] = v[k];
    }
}

// the same copy, software pipelined: a persistent workgroup issues the loads of its NEXT tile before it passes the
// current one through LDS and stores it (the loads then do not queue behind the stores, and two tiles per workgroup
// are in flight)
template <int MODE>
__global__ __launch_bounds__(T) void k_copy_pipelined(const double2 * __restrict__ src, double2 * __restrict__ dst,
                                                      int n_tiles, int nb) {
    extern __shared__ double2 sm[];
    const int tid = threadIdx.x;
    const int64_t m = (int64_t)N1 * N2;
    const int total = n_tiles * nb;
    auto addr = [&](int t, int e, bool strided) -> int64_t {
        const int b = t / n_tiles;
        unsigned bx = t % n_tiles;
        bx = (bx & 7u) * (n_tiles >> 3) + (bx >> 3);
        return (int64_t)b * m + (strided ? ((int64_t)(e >> 3) * N2) + (int64_t)bx * C + (e & 7) : ((int64_t)bx << 12) + e);
    };
    int t = blockIdx.x;
    if (t >= total) return;
    double2 v[P], vn[P];
#pragma unroll
    for (int k = 0; k < P; ++k) v[k] = src[addr(t, tid + k * T, MODE != 1)];
    while (true) {
        const int tn = t + gridDim.x;
        if (tn < total) {
#pragma unroll
            for (int k = 0; k < P; ++k) vn[k] = src[addr(tn, tid + k * T, MODE != 1)];
        }
#pragma unroll
        for (int k = 0; k < P; ++k) sm[tid + k * T] = v[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < P; ++k) v[k] = sm[(tid + k * T) ^ 1];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < P; ++k) dst[addr(t, tid + k * T, MODE != 0)] = v[k];
        if (tn >= total) break;
#pragma unroll
        for (int k = 0; k < P; ++k) v[k] = vn[k];
        t = tn;
    }
}

int main() {
    const int nb = 256;
    const size_t bytes = (size_t)nb * N1 * N2 * sizeof(double2);
    double2 *a, *b;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 0, bytes));
    CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void *)k_copy<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void *)k_copy<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void *)k_copy<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    const dim3 grid(N2 / C, nb), block(T);
    for (int mode = 0; mode < 3; ++mode) {
        for (int lds = 0; lds < 2; ++lds) {
            float best = 1e9f;
            for (int it = 0; it < 5; ++it) {
                CK(hipEventRecord(e0));
                const size_t sh = lds ? 65536 : 0;
                if (mode == 0) hipLaunchKernelGGL(k_copy<0>, grid, block, sh, 0, a, b, lds);
                if (mode == 1) hipLaunchKernelGGL(k_copy<1>, grid, block, sh, 0, a, b, lds);
                if (mode == 2) hipLaunchKernelGGL(k_copy<2>, grid, block, sh, 0, a, b, lds);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf("mode %d (%s)  %s  %.3f ms  %.2f TB/s (read + write)\n", mode,
                   mode == 0 ? "strided read, contiguous write" : mode == 1 ? "contiguous read, strided write" : "strided both",
                   lds ? "64 KB LDS (2 WG/CU)" : "no LDS (8 WG/CU)  ", best, 2.0 * bytes / best / 1e9);
        }
    }
    CK(hipFuncSetAttribute((const void *)k_copy_pipelined<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void *)k_copy_pipelined<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void *)k_copy_pipelined<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    for (int mode = 0; mode < 3; ++mode) {
        for (int wg = 512; wg <= 1024; wg += 512) {      // 2 per CU (the LDS limit), and an over-subscribed grid
            float best = 1e9f;
            for (int it = 0; it < 5; ++it) {
                CK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k_copy_pipelined<0>, dim3(wg), block, 65536, 0, a, b, N2 / C, nb);
                if (mode == 1) hipLaunchKernelGGL(k_copy_pipelined<1>, dim3(wg), block, 65536, 0, a, b, N2 / C, nb);
                if (mode == 2) hipLaunchKernelGGL(k_copy_pipelined<2>, dim3(wg), block, 65536, 0, a, b, N2 / C, nb);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf("mode %d pipelined (next tile's loads before this tile's stores), %4d persistent workgroups  %.3f ms  %.2f TB/s\n",
                   mode, wg, best, 2.0 * bytes / best / 1e9);
        }
    }
    return 0;
}
