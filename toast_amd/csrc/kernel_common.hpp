// kernel_common.hpp -- device and launch helpers shared by the kernel translation units
// (kernels.hip, otf_kernels.hip).  Everything here has internal linkage.
#pragma once

#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

#include "hpix_math.hpp"
#include "runtime.hpp"

namespace {

using namespace toast_hip;

constexpr int kThreads = 256;

struct alignas(16) Quat {
    double x, y, z, w;
};

__device__ __forceinline__ Quat load_quat(const double * p) {
    const double2 a = *reinterpret_cast<const double2 *>(p);
    const double2 b = *reinterpret_cast<const double2 *>(p + 2);
    return Quat{a.x, a.y, b.x, b.y};
}

__device__ __forceinline__ void store_quat(double * p, const double * r) {
    *reinterpret_cast<double2 *>(p) = make_double2(r[0], r[1]);
    *reinterpret_cast<double2 *>(p + 2) = make_double2(r[2], r[3]);
}

// ------------------------------------------------------------------------------------
// Segmented sum over runs of equal, *adjacent* keys: after the call the LAST lane of each run
// (return value true) holds the run total in v[].
//
// A segmented inclusive scan built only from DPP moves on the VALU (no LDS crossbar traffic as
// with ds_bpermute / __shfl_up): four row_shr steps inside the 16-lane rows, then row_bcast15
// (lane 15 / 47 -> rows 1 / 3) and row_bcast31 (lane 31 -> rows 2, 3) carry the partial sums of
// runs that cross a row boundary.  Every step is skipped when no run in the wave needs it
// (wave-uniform tests on the ballot mask): a run of n samples costs ~log2(n) steps.
// ------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp_f64(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}

template <int CTRL>
__device__ __forceinline__ int64_t dpp_i64(int64_t x) {
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(x & 0xffffffffll), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(x >> 32), CTRL, 0xf, 0xf, false);
    return ((int64_t)hi << 32) | (int64_t)(unsigned int)lo;
}

constexpr int kDppRowShr = 0x110;      // + n: lane i reads lane i - n of its 16-lane row
constexpr int kDppWaveShr1 = 0x138;    // lane i reads lane i - 1 of the wave
constexpr int kDppRowBcast15 = 0x142;  // lane 15 of each row -> every lane of the next row
constexpr int kDppRowBcast31 = 0x143;  // lane 31 -> every lane of rows 2 and 3

template <int NV, int D>
__device__ __forceinline__ void run_reduce_step(int lane, int lo, double (&v)[NV]) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const double o = dpp_f64<kDppRowShr + D>(v[k]);
        if (lane - D >= lo) v[k] += o;
    }
}

template <int NV>
__device__ __forceinline__ bool wave_run_reduce(int64_t key, double (&v)[NV]) {
    const int lane = threadIdx.x & 63;
    const int64_t prev = dpp_i64<kDppWaveShr1>(key);
    const bool head = (lane == 0) || (prev != key);
    const unsigned long long heads = __ballot(head);
    // run start = highest head bit at or below my lane
    const unsigned long long below = heads & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
    const int start = 63 - __clzll(below);
    const int row0 = lane & ~15;
    const int lo = (start > row0) ? start : row0;   // first lane summed by the in-row steps
    // covered = lanes whose (row-limited) start lies within the distance already summed
    unsigned long long covered = heads | 0x0001000100010001ull;
    if (~covered != 0ull) {
        run_reduce_step<NV, 1>(lane, lo, v);
        covered |= covered << 1;
        if (~covered != 0ull) {
            run_reduce_step<NV, 2>(lane, lo, v);
            covered |= covered << 2;
            if (~covered != 0ull) {
                run_reduce_step<NV, 4>(lane, lo, v);
                covered |= covered << 4;
                if (~covered != 0ull) run_reduce_step<NV, 8>(lane, lo, v);
            }
        }
    }
    // runs continuing across lanes 15|16 or 47|48
    if ((heads & 0x0001000000010000ull) != 0x0001000000010000ull) {
        const bool take = ((lane & 16) != 0) && (start < row0);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const double o = dpp_f64<kDppRowBcast15, 0xa>(v[k]);
            if (take) v[k] += o;
        }
    }
    // runs continuing across lanes 31|32
    if ((heads & 0x0000000100000000ull) == 0ull) {
        const bool take = (lane >= 32) && (start < 32);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const double o = dpp_f64<kDppRowBcast31, 0xc>(v[k]);
            if (take) v[k] += o;
        }
    }
    return (lane == 63) || ((heads >> (lane + 1)) & 1ull);
}

// Scatter the per-lane contributions of E detectors (E = 1, 2) into the map: one run reduction
// and one set of atomics when the wave's two keys agree lane by lane (or one is invalid), two
// passes otherwise.
template <int NNZ, int E>
__device__ __forceinline__ void scatter_runs(int64_t (&key)[E], double (&v)[E][NNZ], double * __restrict__ zmap) {
    if constexpr (E == 2) {
        const bool mergeable = (key[0] == key[1]) | (key[0] < 0) | (key[1] < 0);
        if (__all(mergeable)) {
            const int64_t km = (key[0] >= 0) ? key[0] : key[1];
            double vm[NNZ];
#pragma unroll
            for (int k = 0; k < NNZ; ++k) vm[k] = v[0][k] + v[1][k];
            const bool tail = wave_run_reduce<NNZ>(km, vm);
            if (tail && km >= 0) {
                double * z = zmap + NNZ * km;
#pragma unroll
                for (int k = 0; k < NNZ; ++k) unsafeAtomicAdd(z + k, vm[k]);
            }
            return;
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const bool tail = wave_run_reduce<NNZ>(key[e], v[e]);
        if (tail && key[e] >= 0) {
            double * z = zmap + NNZ * key[e];
#pragma unroll
            for (int k = 0; k < NNZ; ++k) unsafeAtomicAdd(z + k, v[e][k]);
        }
    }
}

// Two CONSECUTIVE samples per lane (A before B): a wave then reduces runs over 128 samples with one segmented scan.
// A joins B inside the lane when their keys agree; otherwise A ends the run of the previous lanes and is handed to the
// previous lane (DPP wave_shl:1) when that lane's B carries the same key, or -- a run of one sample, or lane 0 --
// goes to the map by itself.  The scan then runs over the B keys.
constexpr int kDppWaveShl1 = 0x130;    // lane i reads lane i + 1 of the wave

template <int NNZ>
__device__ __forceinline__ void scatter_runs2(int64_t ka, double (&va)[NNZ], int64_t kb, double (&vb)[NNZ],
                                              double * __restrict__ zmap) {
    const int lane = threadIdx.x & 63;
    const bool same = ka == kb;
    const bool apart = !same & (ka >= 0);
    if (same) {
#pragma unroll
        for (int k = 0; k < NNZ; ++k) vb[k] += va[k];
    }
    if (__any(apart)) {
        const int64_t prev_b = dpp_i64<kDppWaveShr1>(kb);
        const bool give = apart & (lane > 0) & (prev_b == ka);
#pragma unroll
        for (int k = 0; k < NNZ; ++k) vb[k] += dpp_f64<kDppWaveShl1>(give ? va[k] : 0.0);
        if (apart & !give) {
            double * z = zmap + NNZ * ka;
#pragma unroll
            for (int k = 0; k < NNZ; ++k) unsafeAtomicAdd(z + k, va[k]);
        }
    }
    const bool tail = wave_run_reduce<NNZ>(kb, vb);
    if (tail && kb >= 0) {
        double * z = zmap + NNZ * kb;
#pragma unroll
        for (int k = 0; k < NNZ; ++k) unsafeAtomicAdd(z + k, vb[k]);
    }
}

// ------------------------------------------------------------------------------------
// host helpers
// ------------------------------------------------------------------------------------
inline hipStream_t as_stream(void * s) { return static_cast<hipStream_t>(s); }

inline void check_launch() { TH_HIP(hipGetLastError()); }

inline void need_aligned(const void * p, const char * what) {
    if ((reinterpret_cast<uintptr_t>(p) & 15) != 0) {
        fail_arg(std::string(what) + " must be 16-byte aligned");
    }
}

inline dim3 chunk_grid(int64_t n_det, size_t n_chunks) {
    const unsigned gy = (unsigned)((n_chunks < 65535) ? n_chunks : 65535);
    return dim3((unsigned)n_det, gy ? gy : 1, 1);
}

inline int log2_exact(int64_t nside) {
    if (nside <= 0 || (nside & (nside - 1)) != 0) fail_arg("nside must be a positive power of two");
    int f = 0;
    while ((int64_t(1) << f) != nside) ++f;
    return f;
}

inline dim3 flat_grid(int64_t n) {
    int64_t b = (n + kThreads - 1) / kThreads;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return dim3((unsigned)b);
}

struct OffsetViews {
    std::vector<int64_t> first;
    std::vector<int64_t> aoff;
};

inline OffsetViews offset_views(const toast_hip_interval * ivl, const int64_t * n_amp_views, int64_t n_view) {
    OffsetViews v;
    int64_t run = 0;  // template_offset.cpp:57-63
    for (int64_t i = 0; i < n_view; ++i) {
        v.first.push_back(ivl[i].first);
        v.aoff.push_back(run);
        run += n_amp_views[i];
    }
    return v;
}

}  // namespace
