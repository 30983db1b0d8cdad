// Latency of dependent cross-lane chains on one wave (gfx950): which primitive should carry a
// serial recurrence?  hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_chain tools/ubench_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ double rl(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                            __builtin_amdgcn_readlane(__double2loint(v), lane));
}
template <int CTRL>
__device__ __forceinline__ double dpp(double x) {
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, true),
                            __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ double bperm(double x, int src_lane) {
    return __hiloint2double(__builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(x)),
                            __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(x)));
}

template <int MODE>
__global__ void k(double * out, long long * cycles, int iters, double c) {
    const int lane = threadIdx.x;
    double p = 1.0 + lane * 1e-3;
    double q = c;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {            // fma -> fma
            p = __builtin_fma(p, q, c);
        } else if (MODE == 1) {     // readlane -> fma
            p = __builtin_fma(rl(p, 0), q, c);
        } else if (MODE == 2) {     // wave_shl dpp -> fma
            p = __builtin_fma(dpp<0x130>(p), q, c);
        } else if (MODE == 3) {     // row_shr:1 dpp -> fma
            p = __builtin_fma(dpp<0x111>(p), q, c);
        } else if (MODE == 4) {     // bpermute -> fma
            p = __builtin_fma(bperm(p, (lane + 1) & 63), q, c);
        } else if (MODE == 5) {     // readlane -> fma -> fma -> wave_shl   (the solver's step)
            const double y = __builtin_fma(-rl(p, 0), q, c);
            p = __builtin_fma(q, y, p);
            p = dpp<0x130>(p);
        } else if (MODE == 6) {     // readlane -> fma -> fma, fixed lanes (circular window)
            const double y = __builtin_fma(-rl(p, i & 63), q, c);
            p = __builtin_fma(q, y, p);
        } else if (MODE == 7) {     // row_bcast15 chain
            p = __builtin_fma(dpp<0x142>(p), q, c);
        } else if (MODE == 8) {     // v_add_f64 dependent
            p = p + q;
        } else if (MODE == 9) {     // readfirstlane -> fma
            const double f = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(p)),
                                              __builtin_amdgcn_readfirstlane(__double2loint(p)));
            p = __builtin_fma(f, q, c);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[lane] = p;
    if (lane == 0) cycles[0] = t1 - t0;
}

template <int MODE>
void run(const char * name, double * out, long long * cyc) {
    const int iters = 100000;
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, out, cyc, iters, 0.999);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, out, cyc, iters, 0.999);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    long long h = 0;
    hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-44s %7.1f ns/iter   %7.1f counter ticks/iter\n", name, 1e6 * ms / iters, (double)h / iters);
}

int main() {
    double * out;
    long long * cyc;
    hipMalloc(&out, 64 * sizeof(double));
    hipMalloc(&cyc, sizeof(long long));
    run<0>("fma -> fma", out, cyc);
    run<8>("add -> add", out, cyc);
    run<1>("readlane(0) -> fma", out, cyc);
    run<9>("readfirstlane -> fma", out, cyc);
    run<2>("dpp wave_shl:1 -> fma", out, cyc);
    run<3>("dpp row_shr:1 -> fma", out, cyc);
    run<7>("dpp row_bcast15 -> fma", out, cyc);
    run<4>("ds_bpermute -> fma", out, cyc);
    run<5>("readlane -> fma -> fma -> wave_shl (solver)", out, cyc);
    run<6>("readlane(i) -> fma -> fma (circular)", out, cyc);
    return 0;
}
