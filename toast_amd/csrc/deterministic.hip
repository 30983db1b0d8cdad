// deterministic.hip -- order-deterministic A^T scatter (debug mode, TOAST_HIP_DETERMINISTIC=1).
//
// The production accumulate kernels (kernels.hip) add run-reduced partial sums with hardware
// fp64 atomics: the result depends on the order in which workgroups retire, i.e. zmap differs
// from run to run at the 1e-16 level.  The reference's host path is order-deterministic: every
// OpenMP thread owns a range of sub-pixels and walks detectors, intervals and samples in order
// (src/toast/_libtoast/ops_mapmaker_utils.cpp:294-378; the same for the inverse covariance,
// src/libtoast/src/toast_map_cov.cpp:96-153), so each map value is the left-to-right sum
//     zmap[p] = (((zmap[p] + c_1) + c_2) + ...),   c_i in (detector, interval, sample) order.
// This file reproduces exactly that sum on the GPU with a sorted segmented reduction:
//   k_det_keys      key = local map index of every det-sample of a detector group (sentinel
//                   when flagged / outside the map), value = (detector, sample), written in
//                   (detector, interval, sample) order
//   rocPRIM         stable LSD radix sort of the pairs by key (order inside a pixel preserved)
//   k_det_reduce    one thread per distinct pixel walks its run in order and accumulates with the
//                   reference's operation order: (data * det_scale) * w_k added to the running
//                   value that starts from the current map content
// Detector groups are processed one after the other (bounded scratch); because the order is
// detector-major this does not change any sum.  Result: bit-identical from run to run AND
// bit-identical to the reference's host path / the oracle.  Cost: ~20x the atomic kernels.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/rocprim.hpp>

#include <cstdlib>

#include "kernel_common.hpp"
#include "runtime.hpp"

namespace toast_hip {

namespace {
int g_deterministic = -1;
}

bool deterministic_mode() {
    if (g_deterministic < 0) {
        const char * e = std::getenv("TOAST_HIP_DETERMINISTIC");
        g_deterministic = (e != nullptr && e[0] != '\0' && e[0] != '0') ? 1 : 0;
    }
    return g_deterministic == 1;
}

void set_deterministic_mode(int on) { g_deterministic = on ? 1 : 0; }

namespace {
int g_stokes_nan = -1;
}

bool stokes_reference_nan() {
    if (g_stokes_nan < 0) {
        // default: the reference's result, NaN included; TOAST_HIP_STOKES_REFERENCE_NAN=0 selects the finite form
        const char * e = std::getenv("TOAST_HIP_STOKES_REFERENCE_NAN");
        g_stokes_nan = (e != nullptr && e[0] == '0') ? 0 : 1;
    }
    return g_stokes_nan == 1;
}

namespace {

constexpr int kDetThreads = 256;

struct DetArgs {
    const Chunk * chunks;
    const int64_t * chunk_pos;      // position of the first sample of chunk c inside one detector's entries
    int n_chunks;
    const int32_t * p_idx;
    const int32_t * w_idx;
    const int32_t * d_idx;
    const int32_t * f_idx;
    const double * det_scale;
    const int64_t * g2l;
    const int64_t * pixels;
    const double * weights;
    const double * tod;
    const uint8_t * dflags;
    const uint8_t * sflags;
    uint8_t dmask, smask;
    int use_dflags, use_sflags;
    FastDiv nps_div;
    int64_t n_samp;
    int64_t n_in_views;             // samples inside the intervals (entries per detector)
    int det0;                       // first detector of this group
};

__global__ __launch_bounds__(kDetThreads) void k_det_keys(const DetArgs a, uint64_t sentinel,
                                                          uint64_t * __restrict__ keys,
                                                          uint32_t * __restrict__ vals) {
    const int det_local = blockIdx.x;
    const int det = a.det0 + det_local;
    const int64_t * prow = a.pixels + (int64_t)a.p_idx[det] * a.n_samp;
    const uint8_t * frow = a.use_dflags ? a.dflags + (int64_t)a.f_idx[det] * a.n_samp : nullptr;
    const int64_t nps = a.nps_div.d;
    for (int ci = blockIdx.y; ci < a.n_chunks; ci += gridDim.y) {
        const Chunk c = a.chunks[ci];
        const int64_t base = (int64_t)det_local * a.n_in_views + a.chunk_pos[ci];
        for (int i = threadIdx.x; i < c.count; i += kDetThreads) {
            const int64_t s = c.first + i;
            const int64_t p = prow[s];
            const uint8_t fd = a.use_dflags ? frow[s] : (uint8_t)0;
            const uint8_t fs = a.use_sflags ? a.sflags[s] : (uint8_t)0;
            uint64_t key = sentinel;
            if ((p >= 0) && ((fd & a.dmask) == 0) && ((fs & a.smask) == 0)) {
                const int64_t gsm = fastdiv(p, a.nps_div);
                const int64_t lsm = a.g2l[gsm];
                if (lsm >= 0) key = (uint64_t)(lsm * nps + (p - gsm * nps));   // non-local submap: sentinel
            }
            keys[base + i] = key;
            vals[base + i] = (uint32_t)((int64_t)det_local * a.n_samp + s);
        }
    }
}

// MODE 0: zmap (NV = nnz values, term (tod * scale) * w_k);  MODE 1: packed inverse covariance
// (NV = nnz (nnz + 1) / 2, term w_k * (w_j * scale)), both in the reference's operation order.
template <int MODE>
__global__ __launch_bounds__(kDetThreads) void k_det_reduce(const DetArgs a, int nnz, uint64_t sentinel,
                                                            int64_t n, const uint64_t * __restrict__ keys,
                                                            const uint32_t * __restrict__ vals,
                                                            double * __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kDetThreads + threadIdx.x;
    if (i >= n) return;
    const uint64_t key = keys[i];
    if (key == sentinel) return;
    if (i > 0 && keys[i - 1] == key) return;       // not the head of a run
    const int nv = (MODE == 0) ? nnz : nnz * (nnz + 1) / 2;
    double acc[8];
    double * z = out + (int64_t)key * nv;
    for (int k = 0; k < nv; ++k) acc[k] = z[k];
    for (int64_t j = i; j < n && keys[j] == key; ++j) {
        const uint32_t v = vals[j];
        const int det_local = (int)(v / (uint32_t)a.n_samp);
        const int64_t s = (int64_t)(v - (uint32_t)det_local * (uint32_t)a.n_samp);
        const int det = a.det0 + det_local;
        const double * w = a.weights + ((int64_t)a.w_idx[det] * a.n_samp + s) * nnz;
        if (MODE == 0) {
            const double sd = a.tod[(int64_t)a.d_idx[det] * a.n_samp + s] * a.det_scale[det];
            for (int k = 0; k < nnz; ++k) acc[k] += sd * w[k];
        } else {
            const double ds = a.det_scale[det];
            int off = 0;
            for (int jj = 0; jj < nnz; ++jj) {
                const double sw = w[jj] * ds;
                for (int k = jj; k < nnz; ++k, ++off) acc[off] += w[k] * sw;
            }
        }
    }
    for (int k = 0; k < nv; ++k) z[k] = acc[k];
}

}  // namespace

// Shared driver: mode 0 = build_noise_weighted, 1 = inverse covariance.  Index / scale arrays are
// host pointers (as in the *_dev entry points), everything large is a device pointer.
void deterministic_scatter(int mode, const int64_t * d_g2l, double * d_out, int64_t n_pix_submap, int64_t nnz,
                           const int32_t * pixel_index, const int64_t * d_pixels,
                           const int32_t * weight_index, const double * d_weights, const int32_t * data_index,
                           const double * d_det_data, const int32_t * flag_index, const uint8_t * d_det_flags,
                           int use_d, const double * det_scale, uint8_t det_flag_mask, int64_t n_det,
                           int64_t n_samp, const toast_hip_interval * intervals, int64_t n_view,
                           const uint8_t * d_shared_flags, int use_s, uint8_t shared_flag_mask, hipStream_t st) {
    const int nv = (mode == 0) ? (int)nnz : (int)(nnz * (nnz + 1) / 2);
    if (nv > 8) fail_arg("deterministic mode supports at most 8 values per pixel");
    if (n_samp >= (int64_t(1) << 32)) fail_arg("deterministic mode: n_samp must be below 2^32");
    const auto chunks = make_chunks(intervals, n_view, n_samp);
    if (chunks.empty() || n_det <= 0) return;
    std::vector<int64_t> chunk_pos(chunks.size());
    int64_t n_in = 0;
    for (size_t c = 0; c < chunks.size(); ++c) {
        chunk_pos[c] = n_in;
        n_in += chunks[c].count;
    }
    std::vector<int32_t> fidx(n_det, 0), didx(n_det, 0);
    if (use_d) std::memcpy(fidx.data(), flag_index, sizeof(int32_t) * n_det);
    if (data_index) std::memcpy(didx.data(), data_index, sizeof(int32_t) * n_det);
    ParamBlock pb;
    const size_t o_ch = pb.push_vec(chunks);
    const size_t o_cp = pb.push_vec(chunk_pos);
    const size_t o_pi = pb.push(pixel_index, sizeof(int32_t) * n_det);
    const size_t o_wi = pb.push(weight_index, sizeof(int32_t) * n_det);
    const size_t o_di = pb.push_vec(didx);
    const size_t o_fi = pb.push_vec(fidx);
    const size_t o_ds = pb.push(det_scale, sizeof(double) * n_det);
    const char * d = pb.commit(st);
    DetArgs a;
    a.chunks = (const Chunk *)(d + o_ch);
    a.chunk_pos = (const int64_t *)(d + o_cp);
    a.n_chunks = (int)chunks.size();
    a.p_idx = (const int32_t *)(d + o_pi);
    a.w_idx = (const int32_t *)(d + o_wi);
    a.d_idx = (const int32_t *)(d + o_di);
    a.f_idx = (const int32_t *)(d + o_fi);
    a.det_scale = (const double *)(d + o_ds);
    a.g2l = d_g2l;
    a.pixels = d_pixels;
    a.weights = d_weights;
    a.tod = d_det_data;
    a.dflags = d_det_flags;
    a.sflags = d_shared_flags;
    a.dmask = det_flag_mask;
    a.smask = shared_flag_mask;
    a.use_dflags = use_d;
    a.use_sflags = use_s;
    a.nps_div = make_fastdiv(n_pix_submap);
    a.n_samp = n_samp;
    a.n_in_views = n_in;
    // key range: local map indices (a map of 2^44 pixels does not fit any memory); the sentinel sorts last
    const int bits = 44;
    const uint64_t sentinel = uint64_t(1) << bits;
    // detector groups: <= 2^27 entries and (detector, sample) must fit 32 bits
    int64_t group = (int64_t(1) << 27) / (n_in > 0 ? n_in : 1);
    const int64_t cap32 = ((int64_t(1) << 32) - 1) / n_samp;
    if (group > cap32) group = cap32;
    if (group < 1) group = 1;
    if (group > n_det) group = n_det;
    const size_t n_max = (size_t)(group * n_in);
    size_t temp_bytes = 0;
    TH_HIP(rocprim::radix_sort_pairs(nullptr, temp_bytes, (uint64_t *)nullptr, (uint64_t *)nullptr,
                                     (uint32_t *)nullptr, (uint32_t *)nullptr, n_max, 0u, (unsigned)(bits + 1), st));
    const size_t kb = (n_max * sizeof(uint64_t) + 255) & ~size_t(255);
    const size_t vb = (n_max * sizeof(uint32_t) + 255) & ~size_t(255);
    char * scratch = (char *)Manager::get().scratch(Manager::kScratchSort, 2 * kb + 2 * vb + temp_bytes + 256);
    uint64_t * keys_in = (uint64_t *)scratch;
    uint64_t * keys_out = (uint64_t *)(scratch + kb);
    uint32_t * vals_in = (uint32_t *)(scratch + 2 * kb);
    uint32_t * vals_out = (uint32_t *)(scratch + 2 * kb + vb);
    void * temp = scratch + 2 * kb + 2 * vb;
    for (int64_t det0 = 0; det0 < n_det; det0 += group) {
        const int64_t nd = (n_det - det0 < group) ? (n_det - det0) : group;
        const int64_t n = nd * n_in;
        a.det0 = (int)det0;
        unsigned gy = (unsigned)chunks.size();
        if (gy > 1024) gy = 1024;
        hipLaunchKernelGGL(k_det_keys, dim3((unsigned)nd, gy), dim3(kDetThreads), 0, st, a, sentinel, keys_in, vals_in);
        size_t tb = temp_bytes;
        TH_HIP(rocprim::radix_sort_pairs(temp, tb, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u,
                                         (unsigned)(bits + 1), st));
        const dim3 grid((unsigned)((n + kDetThreads - 1) / kDetThreads));
        if (mode == 0) {
            hipLaunchKernelGGL(k_det_reduce<0>, grid, dim3(kDetThreads), 0, st, a, (int)nnz, sentinel, n, keys_out,
                               vals_out, d_out);
        } else {
            hipLaunchKernelGGL(k_det_reduce<1>, grid, dim3(kDetThreads), 0, st, a, (int)nnz, sentinel, n, keys_out,
                               vals_out, d_out);
        }
        TH_HIP(hipGetLastError());
    }
}

}  // namespace toast_hip

extern "C" {

int toast_hip_set_deterministic(int on) {
    toast_hip::set_deterministic_mode(on);
    return TOAST_HIP_OK;
}

int toast_hip_get_deterministic(void) { return toast_hip::deterministic_mode() ? 1 : 0; }

int toast_hip_set_stokes_reference_nan(int on) {
    toast_hip::g_stokes_nan = on ? 1 : 0;
    return TOAST_HIP_OK;
}

}  // extern "C"
