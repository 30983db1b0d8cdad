// How many workgroups of 512 threads does a CU of gfx950 really hold at a given LDS footprint?  Every workgroup spins for
// a fixed wall-clock time; the launch time of G workgroups then gives the number that ran concurrently.  Next to it the
// runtime's own answer (hipOccupancyMaxActiveBlocksPerMultiprocessor).
//   hipcc --offload-arch=gfx950 -O3 lds_occupancy.hip -o lds_occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int T>
__global__ __launch_bounds__(T) void k_spin(long long ticks, double * sink) {
    extern __shared__ double sm[];
    sm[threadIdx.x] = (double)threadIdx.x;
    __syncthreads();
    const long long t0 = wall_clock64();
    double acc = sm[(threadIdx.x + 1) % T];
    while (wall_clock64() - t0 < ticks) acc = acc * 1.0000001 + 1.0e-9;
    if (acc == 12345.678) sink[0] = acc;
}

template <int T>
void sweep(int n_cu, double * d_sink) {
    const int G = 16384;
    const long long ticks = 2000;     // 20 us at 100 MHz
    const int sizes[] = {8 << 10, 32 << 10, 40 << 10, 48 << 10, 53 << 10, 54 << 10, 64 << 10, 65536 + 7152, 80 << 10, 81 << 10,
                         96 << 10, 128 << 10, 160 << 10};
    for (int lds : sizes) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_spin<T>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        int occ = 0;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_spin<T>, T, lds));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_spin<T>, dim3(G), dim3(T), lds, 0, ticks, d_sink);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_spin<T>, dim3(G), dim3(T), lds, 0, ticks, d_sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double conc = (double)G * 20.0e-3 / ms;
        printf("threads %3d  LDS %6d B   API says %d per CU   measured %.2f ms -> %.0f concurrent = %.2f per CU\n", T, lds, occ, ms,
               conc, conc / n_cu);
    }
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("%s: %d CUs, sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu\n", prop.name, prop.multiProcessorCount,
           prop.sharedMemPerBlock, prop.maxSharedMemoryPerMultiProcessor);
    double * d_sink;
    CK(hipMalloc(&d_sink, 64));
    sweep<512>(prop.multiProcessorCount, d_sink);
    sweep<256>(prop.multiProcessorCount, d_sink);
    return 0;
}
