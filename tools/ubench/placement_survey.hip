// What does a box hand out?  Streams (1024 rows in flight, the timestream kernels' pattern) over N fresh 5.9 GB
// allocations, all alive, with the time hipMalloc took for each -- first on the device as found, then right after a
// large allocate / touch / free cycle (what a test suite or a previous process leaves behind), then again after a pause.
//   hipcc --offload-arch=gfx950 -O3 placement_survey.hip -o placement_survey ; ./placement_survey [n_candidates] [churn GB]
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_probe(double * __restrict__ p, int64_t row_len, double one) {
    double * row = p + (int64_t)blockIdx.x * row_len;
    for (int64_t c0 = (int64_t)blockIdx.y * 1024; c0 < row_len; c0 += (int64_t)gridDim.y * 1024) {
        for (int i = threadIdx.x; i < 1024 && c0 + i < row_len; i += 256) row[c0 + i] = row[c0 + i] * one;
    }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static double probe(double * p, int64_t row_len) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_probe, dim3(1024, 704), dim3(256), 0, 0, p, row_len, 1.0);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

static void survey(const char * label, int n) {
    const int64_t row_len = 720000;
    const size_t bytes = (size_t)1024 * row_len * 8;
    std::vector<double *> blocks;
    printf("%s\n  TB/s (hipMalloc ms):", label);
    int fast = 0;
    const double t0 = now();
    for (int i = 0; i < n; ++i) {
        double * p = nullptr;
        const double a = now();
        if (hipMalloc(&p, bytes) != hipSuccess) break;
        const double b = now();
        const double ms = probe(p, row_len);
        const double tbs = 2.0 * bytes / ms / 1e9;
        if (tbs >= 5.65) ++fast;
        printf(" %.2f (%.0f)", tbs, (b - a) * 1e3);
        blocks.push_back(p);
    }
    printf("\n  %d of %zu fast, %.2f s\n", fast, blocks.size(), now() - t0);
    const double f0 = now();
    for (double * p : blocks) CK(hipFree(p));
    printf("  freeing them took %.2f s\n", now() - f0);
}

int main(int argc, char ** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 16;
    const size_t churn_gb = argc > 2 ? (size_t)atol(argv[2]) : 150;
    survey("== fresh process, device as found", n);
    {
        const double t0 = now();
        std::vector<char *> big;
        for (size_t g = 0; g < churn_gb; g += 10) {
            char * p = nullptr;
            if (hipMalloc(&p, (size_t)10 << 30) != hipSuccess) break;
            CK(hipMemset(p, 1, (size_t)10 << 30));
            big.push_back(p);
        }
        CK(hipDeviceSynchronize());
        for (char * p : big) CK(hipFree(p));
        printf("== churn: %zu x 10 GB allocated, set, freed in %.2f s\n", big.size(), now() - t0);
    }
    survey("== right after the churn", n);
    survey("== once more", n);
    sleep(10);
    survey("== after a 10 s pause", n);
    return 0;
}
