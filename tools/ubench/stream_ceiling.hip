// What a plain stream reaches on this box, independent of the library's launch shape: read-only, write-only, copy,
// in-place scale (8 and 16 bytes per lane) and the byte mix of scan_map (40 B read + 8 B written per element) over
// cfg-3 sized buffers (5.9 GB each).  The ceilings the projection kernels are measured against (profiles/r03_a).
//   hipcc --offload-arch=gfx950 -O3 stream_ceiling.hip -o stream_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <functional>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int T = 256;

__global__ __launch_bounds__(T) void k_read16(const double2 * __restrict__ a, int64_t n2, double * __restrict__ out) {
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * T + threadIdx.x; i < n2; i += (int64_t)gridDim.x * T) {
        const double2 v = a[i];
        acc += v.x + v.y;
    }
    if (acc == 1.2345e300) out[0] = acc;
}
__global__ __launch_bounds__(T) void k_read8(const double * __restrict__ a, int64_t n, double * __restrict__ out) {
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * T + threadIdx.x; i < n; i += (int64_t)gridDim.x * T) acc += a[i];
    if (acc == 1.2345e300) out[0] = acc;
}
__global__ __launch_bounds__(T) void k_write16(double2 * __restrict__ a, int64_t n2) {
    for (int64_t i = (int64_t)blockIdx.x * T + threadIdx.x; i < n2; i += (int64_t)gridDim.x * T) a[i] = make_double2(1.0, 2.0);
}
__global__ __launch_bounds__(T) void k_write8(double * __restrict__ a, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * T + threadIdx.x; i < n; i += (int64_t)gridDim.x * T) a[i] = 1.0;
}
// every workgroup writes contiguous strips of `strip` double2
__global__ __launch_bounds__(T) void k_write16_strips(double2 * __restrict__ a, int64_t n2, int strip) {
    const int64_t n_strip = n2 / strip;
    for (int64_t s = blockIdx.x; s < n_strip; s += gridDim.x) {
        double2 * p = a + s * strip;
        for (int i = threadIdx.x; i < strip; i += T) p[i] = make_double2(1.0, 2.0);
    }
}
__global__ __launch_bounds__(T) void k_copy16(const double2 * __restrict__ a, double2 * __restrict__ b, int64_t n2) {
    for (int64_t i = (int64_t)blockIdx.x * T + threadIdx.x; i < n2; i += (int64_t)gridDim.x * T) b[i] = a[i];
}
__global__ __launch_bounds__(T) void k_scale16(double2 * __restrict__ a, int64_t n2) {
    for (int64_t i = (int64_t)blockIdx.x * T + threadIdx.x; i < n2; i += (int64_t)gridDim.x * T) {
        double2 v = a[i];
        v.x *= 1.0000001;
        v.y *= 1.0000001;
        a[i] = v;
    }
}
__global__ __launch_bounds__(T) void k_scale8(double * __restrict__ a, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * T + threadIdx.x; i < n; i += (int64_t)gridDim.x * T) a[i] *= 1.0000001;
}
// the byte mix of scan_map without its gather: pixels (8) + weights (24) + tod (8) read, tod (8) written, two
// elements per lane
__global__ __launch_bounds__(T) void k_mix_scan(const longlong2 * __restrict__ pix, const double2 * __restrict__ w,
                                                double2 * __restrict__ tod, int64_t n2) {
    for (int64_t i = (int64_t)blockIdx.x * T + threadIdx.x; i < n2; i += (int64_t)gridDim.x * T) {
        const longlong2 p = pix[i];
        const double2 w0 = w[3 * i], w1 = w[3 * i + 1], w2 = w[3 * i + 2];
        double2 d = tod[i];
        d.x -= (double)p.x * (w0.x + w0.y + w1.x);
        d.y -= (double)p.y * (w1.y + w2.x + w2.y);
        tod[i] = d;
    }
}
// the byte mix of build_noise_weighted without its scatter: 40 B read + 1 B flag per element
__global__ __launch_bounds__(T) void k_mix_bnw(const longlong2 * __restrict__ pix, const double2 * __restrict__ w,
                                               const double2 * __restrict__ tod, const uint16_t * __restrict__ fl, int64_t n2,
                                               double * __restrict__ out) {
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * T + threadIdx.x; i < n2; i += (int64_t)gridDim.x * T) {
        const longlong2 p = pix[i];
        const double2 w0 = w[3 * i], w1 = w[3 * i + 1], w2 = w[3 * i + 2];
        const double2 d = tod[i];
        const uint16_t f = fl[i];
        acc += (f ? 0.0 : d.x * (double)p.x * (w0.x + w0.y + w1.x) + d.y * (double)p.y * (w1.y + w2.x + w2.y));
    }
    if (acc == 1.2345e300) out[0] = acc;
}

int main(int argc, char ** argv) {
    const int64_t n = (argc > 1) ? atoll(argv[1]) : (int64_t)1024 * 720000;   // elements (doubles) per array
    const int64_t n2 = n / 2;
    double *tod, *tod_b, *w, *out;
    int64_t * pix;
    uint8_t * fl;
    CK(hipMalloc(&pix, n * 8));
    CK(hipMalloc(&w, n * 24));
    CK(hipMalloc(&tod, n * 8));
    CK(hipMalloc(&tod_b, n * 8));
    CK(hipMalloc(&fl, n));
    CK(hipMalloc(&out, 8));
    CK(hipMemset(pix, 0, n * 8));
    CK(hipMemset(w, 0, n * 24));
    CK(hipMemset(tod, 0, n * 8));
    CK(hipMemset(tod_b, 0, n * 8));
    CK(hipMemset(fl, 0, n));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto run = [&](const char * name, double bytes, std::function<void()> fn) {
        float best = 1e9f, sum = 0.f;
        fn();
        for (int it = 0; it < 5; ++it) {
            CK(hipEventRecord(e0));
            fn();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
            sum += ms;
        }
        printf("%-64s best %7.3f ms  mean %7.3f ms  %6.3f TB/s (best)\n", name, best, sum / 5, bytes / best / 1e9);
    };
    const int grids[] = {2048, 8192, 32768, 1 << 20};
    char nm[128];
    for (int g : grids) {
        snprintf(nm, sizeof nm, "read-only 16 B/lane, grid %d", g);
        run(nm, 8.0 * n, [&] { hipLaunchKernelGGL(k_read16, dim3(g), dim3(T), 0, 0, (const double2 *)tod, n2, out); });
    }
    run("read-only 8 B/lane, grid 8192", 8.0 * n, [&] { hipLaunchKernelGGL(k_read8, dim3(8192), dim3(T), 0, 0, tod, n, out); });
    for (int g : grids) {
        snprintf(nm, sizeof nm, "write-only 16 B/lane, grid %d", g);
        run(nm, 8.0 * n, [&] { hipLaunchKernelGGL(k_write16, dim3(g), dim3(T), 0, 0, (double2 *)tod_b, n2); });
    }
    run("write-only 8 B/lane, grid 8192", 8.0 * n, [&] { hipLaunchKernelGGL(k_write8, dim3(8192), dim3(T), 0, 0, tod_b, n); });
    for (int strip : {512, 4096, 65536}) {
        snprintf(nm, sizeof nm, "write-only 16 B/lane, contiguous strips of %d KB per workgroup", strip * 16 / 1024);
        run(nm, 8.0 * n, [&] { hipLaunchKernelGGL(k_write16_strips, dim3(8192), dim3(T), 0, 0, (double2 *)tod_b, n2, strip); });
    }
    run("hipMemsetAsync", 8.0 * n, [&] { CK(hipMemsetAsync(tod_b, 0, n * 8, 0)); });
    for (int g : grids) {
        snprintf(nm, sizeof nm, "copy 16 B/lane (read + write), grid %d", g);
        run(nm, 16.0 * n, [&] { hipLaunchKernelGGL(k_copy16, dim3(g), dim3(T), 0, 0, (const double2 *)tod, (double2 *)tod_b, n2); });
    }
    run("hipMemcpyAsync device to device (read + write)", 16.0 * n,
        [&] { CK(hipMemcpyAsync(tod_b, tod, n * 8, hipMemcpyDeviceToDevice, 0)); });
    for (int g : grids) {
        snprintf(nm, sizeof nm, "in-place scale 16 B/lane (read + write), grid %d", g);
        run(nm, 16.0 * n, [&] { hipLaunchKernelGGL(k_scale16, dim3(g), dim3(T), 0, 0, (double2 *)tod, n2); });
    }
    run("in-place scale 8 B/lane (read + write), grid 8192", 16.0 * n,
        [&] { hipLaunchKernelGGL(k_scale8, dim3(8192), dim3(T), 0, 0, tod, n); });
    for (int g : grids) {
        snprintf(nm, sizeof nm, "scan_map byte mix (40 B read + 8 B write), grid %d", g);
        run(nm, 48.0 * n, [&] {
            hipLaunchKernelGGL(k_mix_scan, dim3(g), dim3(T), 0, 0, (const longlong2 *)pix, (const double2 *)w, (double2 *)tod, n2);
        });
    }
    for (int g : grids) {
        snprintf(nm, sizeof nm, "build_noise_weighted byte mix (41 B read), grid %d", g);
        run(nm, 41.0 * n, [&] {
            hipLaunchKernelGGL(k_mix_bnw, dim3(g), dim3(T), 0, 0, (const longlong2 *)pix, (const double2 *)w,
                               (const double2 *)tod, (const uint16_t *)fl, n2, out);
        });
    }
    return 0;
}
