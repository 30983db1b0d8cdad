// ground_filter.hip -- the template-regression kernels behind toast.ops.GroundFilter on the device.
//
// Reference: src/libtoast/src/toast_tod_filter.cpp (`legendre` :269-331, `bin_proj` :160-177,
// `bin_invcov` :179-215, `add_templates` :333-355; bindings src/toast/_libtoast/tod_filter.cpp:100,
// 213, 254, 291), driven per detector by src/toast/ops/groundfilter.py:334-393.  There the
// templates [n_template][n_samp] are shared by all detectors of an observation and every detector
// makes three passes over them on the host (projection, Gram matrix, subtraction).
//
// Here the templates are resident in HBM once and all detectors go through each kernel in one
// launch.  A K-tile of the templates is staged in LDS and reused by every detector of a block:
//
//   k_legendre            templates from the abscissa, same recurrence and rounding as the reference
//   k_template_gram       G[r][c] = sum_i T[r][i] T[c][i] good[i]            (samples good for ALL detectors)
//   k_template_project    proj[d][r] = sum_i T[r][i] signal[d][i] good[d][i]  (bin_proj for all detectors)
//   k_template_gram_flagged  D[d][r][c] = sum over the samples only detector d flags: invcov_d = G - D_d
//                         (bin_invcov: a detector's Gram matrix differs from the common one only by its
//                         own flagged samples -- ~1 % of them -- instead of n_template^2/2 FMAs for
//                         every det-sample)
//   k_template_subtract   signal[d][i] -= sum_r coeff[d][r] T[r][i]           (add_templates + `ref -= fit`)
//
// All of them are HBM / LDS bound: 9 B (signal + flag) per det-sample for the projection, 16 B for
// the subtraction; no MFMA (n_template is 10-40, the reduction dimension is the long one).

#include "kernel_common.hpp"

namespace {

constexpr int kTile = 256;        // samples per LDS tile
constexpr int kTmplGroup = 16;    // template rows per tile (LDS: 16 x 256 x 8 B = 32 KB)

// toast_tod_filter.cpp:269-331.  One thread per sample; the recurrence runs in registers.
__global__ __launch_bounds__(kThreads) void k_legendre(const double * __restrict__ x, int64_t n_samp, int64_t start_order,
                                                       int64_t stop_order, double * __restrict__ templates) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n_samp) return;
    const double xi = x[i];
    if (start_order == 0 && stop_order > 0) templates[i] = 1. / f_sqrt(2.);
    if (start_order <= 1 && stop_order > 1) templates[(1 - start_order) * n_samp + i] = (1. / f_sqrt(2. / 3.)) * xi;
    double val = xi, prev = 1.0;
    for (int64_t order = 2; order < stop_order; ++order) {
        const double orderinv = 1. / (double)order;
        const double next = ((double)(2 * order - 1) * xi * val - (double)(order - 1) * prev) * orderinv;
        prev = val;
        val = next;
        if (order >= start_order) {
            const double norm = 1. / f_sqrt(2. / (2. * (double)order + 1.));
            templates[(order - start_order) * n_samp + i] = val * norm;
        }
    }
}

// Split and binned templates (groundfilter.py:208-257): out = keep ? src : 0 with keep = (key ==
// value) or (key != value); src == nullptr stands for a row of ones.
__global__ __launch_bounds__(kThreads) void k_template_select(const double * __restrict__ src, const int32_t * __restrict__ key,
                                                              int32_t value, int keep_equal, int64_t n_samp,
                                                              double * __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n_samp) return;
    const bool keep = ((key[i] == value) == (keep_equal != 0));
    out[i] = keep ? (src != nullptr ? src[i] : 1.0) : 0.0;
}

__device__ __forceinline__ double block_sum(double v, double * scratch) {
    // wave totals by DPP, then the (up to 4) waves of the block through LDS
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[wave] = v;
    __syncthreads();
    double total = 0.0;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) total += scratch[k];
    return total;
}

// G[r][c] over the samples that are good for every detector (shared flags only).  One block per
// (pair of rows, slice of samples); tiny next to the per-detector kernels.
__global__ __launch_bounds__(kThreads) void k_template_gram(
    const double * __restrict__ templates, int64_t n_template, int64_t n_samp, const uint8_t * __restrict__ shared_flags,
    uint8_t shared_mask, int64_t slice, double * __restrict__ gram) {
    __shared__ double scratch[4];
    const int64_t pair = blockIdx.x;   // (pairs can exceed the 65535 limit of grid.y with binned templates)
    // pair -> (r, c), r <= c
    int64_t r = 0, rem = pair;
    while (rem >= n_template - r) {
        rem -= n_template - r;
        ++r;
    }
    const int64_t c = r + rem;
    const int64_t i0 = (int64_t)blockIdx.y * slice;
    const int64_t i1 = (i0 + slice < n_samp) ? i0 + slice : n_samp;
    const double * __restrict__ tr = templates + r * n_samp;
    const double * __restrict__ tc = templates + c * n_samp;
    double acc = 0.0;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += kThreads) {
        const bool good = (shared_flags == nullptr) || ((shared_flags[i] & shared_mask) == 0);
        if (good) acc += tr[i] * tc[i];
    }
    const double total = block_sum(acc, scratch);
    if (threadIdx.x == 0 && total != 0.0) {
        atomicAdd(&gram[r * n_template + c], total);
        if (c != r) atomicAdd(&gram[c * n_template + r], total);
    }
}

// proj[d][r] = sum_i T[r][i] signal[d][i] good[d][i].  One WAVE per (slice of samples, DPW
// detectors), no LDS and no barriers: a step loads 64 samples of the NTG template rows (coalesced,
// shared through L2 by the waves working on other detectors) and of the DPW signals -- NTG + 2 DPW
// independent loads in flight per lane -- and each lane keeps NTG x DPW accumulators across the
// whole slice, so the cross-lane reduction happens once per (detector, row, slice).
template <int NTG, int DPW>
__global__ __launch_bounds__(kThreads) void k_template_project(
    const double * __restrict__ templates, int64_t n_template, int64_t t0, int64_t n_samp, const int32_t * __restrict__ sig_index,
    const double * __restrict__ signal, const int32_t * __restrict__ flag_index, const uint8_t * __restrict__ det_flags,
    uint8_t det_mask, const uint8_t * __restrict__ shared_flags, uint8_t shared_mask, int64_t n_det, int64_t slice,
    double * __restrict__ proj) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nt = (n_template - t0 < NTG) ? (int)(n_template - t0) : NTG;
    // detector groups on the fast grid index: blocks that run together share a slice, i.e. the
    // same template rows (measured HBM traffic 2.94 -> 2.49 GB, profiles/r01_g_ground_filter_pmc.txt)
    const int64_t i0 = (int64_t)blockIdx.y * slice;
    const int64_t i1 = (i0 + slice < n_samp) ? i0 + slice : n_samp;
    const int64_t d0 = ((int64_t)blockIdx.x * 4 + wave) * DPW;
    if (d0 >= n_det) return;
    const double * sig[DPW];
    const uint8_t * df[DPW];
#pragma unroll
    for (int dd = 0; dd < DPW; ++dd) {
        const int64_t d = (d0 + dd < n_det) ? d0 + dd : n_det - 1;   // clamped: the duplicate is not written
        sig[dd] = signal + (int64_t)sig_index[d] * n_samp;
        df[dd] = (det_flags != nullptr) ? det_flags + (int64_t)flag_index[d] * n_samp : nullptr;
    }
    double acc[DPW][NTG];
#pragma unroll
    for (int dd = 0; dd < DPW; ++dd) {
#pragma unroll
        for (int r = 0; r < NTG; ++r) acc[dd][r] = 0.0;
    }
    const double * __restrict__ trow = templates + t0 * n_samp;
    for (int64_t i = i0 + lane; i < i1; i += 64) {
        double t[NTG];
#pragma unroll
        for (int r = 0; r < NTG; ++r) t[r] = (r < nt) ? trow[r * n_samp + i] : 0.0;
        const bool cgood = (shared_flags == nullptr) || ((shared_flags[i] & shared_mask) == 0);
#pragma unroll
        for (int dd = 0; dd < DPW; ++dd) {
            double sv = sig[dd][i];
            const bool bad = !cgood || (df[dd] != nullptr && (df[dd][i] & det_mask) != 0);
            if (bad) sv = 0.0;
#pragma unroll
            for (int r = 0; r < NTG; ++r) acc[dd][r] += t[r] * sv;
        }
    }
#pragma unroll
    for (int dd = 0; dd < DPW; ++dd) {
        if (d0 + dd >= n_det) continue;
#pragma unroll
        for (int r = 0; r < NTG; ++r) {
            if (r >= nt) continue;
            double v = acc[dd][r];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
            if (lane == 0 && v != 0.0) atomicAdd(&proj[(d0 + dd) * n_template + t0 + r], v);
        }
    }
}

// D[d][r][c] over the samples that are good for everybody but flagged by detector d.  One block
// per (detector, slice): the flagged samples of the slice are collected in an LDS queue, then the
// threads -- one per (r, c) pair -- walk the queue.
__global__ __launch_bounds__(kThreads) void k_template_gram_flagged(
    const double * __restrict__ templates, int64_t n_template, int64_t n_samp, const int32_t * __restrict__ flag_index,
    const uint8_t * __restrict__ det_flags, uint8_t det_mask, const uint8_t * __restrict__ shared_flags, uint8_t shared_mask,
    int64_t slice, double * __restrict__ dgram, int64_t * __restrict__ n_flagged) {
    constexpr int kQueue = 1024;
    __shared__ int queue[kQueue];
    __shared__ int n_queued;
    const int64_t d = blockIdx.y;
    const int64_t i0 = (int64_t)blockIdx.x * slice;
    const int64_t i1 = (i0 + slice < n_samp) ? i0 + slice : n_samp;
    const uint8_t * __restrict__ df = det_flags + (int64_t)flag_index[d] * n_samp;
    const int64_t n_pair = n_template * (n_template + 1) / 2;
    double * __restrict__ out = dgram + d * n_template * n_template;
    // With at most 256 pairs (n_template <= 22) every thread owns one pair for the whole slice and
    // issues its atomics once; otherwise once per 1024-sample sub-block.
    const bool one_pair = n_pair <= kThreads;
    int64_t my_r = 0, my_c = 0;
    if (one_pair && threadIdx.x < n_pair) {
        int64_t rem = threadIdx.x;
        while (rem >= n_template - my_r) {
            rem -= n_template - my_r;
            ++my_r;
        }
        my_c = my_r + rem;
    }
    double slice_acc = 0.0;
    for (int64_t base = i0; base < i1; base += kQueue) {
        if (threadIdx.x == 0) n_queued = 0;
        __syncthreads();
        for (int k = threadIdx.x; k < kQueue; k += kThreads) {
            const int64_t i = base + k;
            if (i < i1 && (df[i] & det_mask) != 0 &&
                ((shared_flags == nullptr) || ((shared_flags[i] & shared_mask) == 0))) {
                queue[atomicAdd(&n_queued, 1)] = k;
            }
        }
        __syncthreads();
        const int nq = n_queued;
        if (nq > 0) {
            if (threadIdx.x == 0) atomicAdd(reinterpret_cast<unsigned long long *>(&n_flagged[d]), (unsigned long long)nq);
            if (one_pair) {
                if (threadIdx.x < n_pair) {
                    const double * __restrict__ tr = templates + my_r * n_samp + base;
                    const double * __restrict__ tc = templates + my_c * n_samp + base;
                    for (int q = 0; q < nq; ++q) slice_acc += tr[queue[q]] * tc[queue[q]];
                }
            } else {
                for (int64_t pair = threadIdx.x; pair < n_pair; pair += kThreads) {
                    int64_t r = 0, rem = pair;
                    while (rem >= n_template - r) {
                        rem -= n_template - r;
                        ++r;
                    }
                    const int64_t c = r + rem;
                    const double * __restrict__ tr = templates + r * n_samp + base;
                    const double * __restrict__ tc = templates + c * n_samp + base;
                    double acc = 0.0;
                    for (int q = 0; q < nq; ++q) acc += tr[queue[q]] * tc[queue[q]];
                    if (acc != 0.0) {
                        atomicAdd(&out[r * n_template + c], acc);
                        if (c != r) atomicAdd(&out[c * n_template + r], acc);
                    }
                }
            }
        }
        __syncthreads();
    }
    if (one_pair && threadIdx.x < n_pair && slice_acc != 0.0) {
        atomicAdd(&out[my_r * n_template + my_c], slice_acc);
        if (my_c != my_r) atomicAdd(&out[my_c * n_template + my_r], slice_acc);
    }
}

// signal[d][i] -= fit, fit = sum_r coeff[d][r] T[r][i] accumulated from zero in template order, like
// add_templates into a zeroed buffer followed by `ref -= fit` (groundfilter.py:384-393).  Block =
// (tile of samples, group of detectors); the template tile is staged in LDS once per block.
template <int NTG>
__global__ __launch_bounds__(kThreads) void k_template_subtract(
    const double * __restrict__ templates, int64_t n_template, int64_t first_template, int64_t n_samp,
    const int32_t * __restrict__ sig_index, double * __restrict__ signal, const double * __restrict__ coeff, int64_t n_det,
    int dets_per_block) {
    __shared__ double tile[NTG * kTile];
    const int64_t base = (int64_t)blockIdx.x * kTile;
    const int64_t d0 = (int64_t)blockIdx.y * dets_per_block;
    const int64_t i = base + threadIdx.x;
    double fit[16];   // detectors of this block handled per pass
    for (int64_t dpass = 0; dpass < dets_per_block; dpass += 16) {
#pragma unroll
        for (int dd = 0; dd < 16; ++dd) fit[dd] = 0.0;
        for (int64_t t0 = first_template; t0 < n_template; t0 += NTG) {
            const int nt = (n_template - t0 < NTG) ? (int)(n_template - t0) : NTG;
            __syncthreads();
            for (int q = threadIdx.x; q < nt * kTile; q += kThreads) {
                const int r = q / kTile, k = q - r * kTile;
                tile[q] = (base + k < n_samp) ? templates[(t0 + r) * n_samp + base + k] : 0.0;
            }
            __syncthreads();
#pragma unroll
            for (int dd = 0; dd < 16; ++dd) {
                const int64_t d = d0 + dpass + dd;
                if (d >= n_det || dpass + dd >= dets_per_block) continue;
                const double * __restrict__ cf = coeff + d * n_template + t0;   // uniform: scalar loads
                double f = fit[dd];
                for (int r = 0; r < nt; ++r) f += cf[r] * tile[r * kTile + threadIdx.x];
                fit[dd] = f;
            }
        }
        if (i < n_samp) {
#pragma unroll
            for (int dd = 0; dd < 16; ++dd) {
                const int64_t d = d0 + dpass + dd;
                if (d >= n_det || dpass + dd >= dets_per_block) continue;
                double * __restrict__ sig = signal + (int64_t)sig_index[d] * n_samp;
                sig[i] -= fit[dd];
            }
        }
    }
}

}  // namespace

extern "C" {

int toast_hip_legendre_templates_dev(const double * d_x, int64_t n_samp, int64_t start_order, int64_t stop_order,
                                     double * d_templates, void * stream) {
    return guarded([&] {
        if (n_samp <= 0 || stop_order <= start_order) return;
        if (start_order < 0) fail_arg("legendre_templates: start_order must be >= 0");
        hipLaunchKernelGGL(k_legendre, flat_grid(n_samp), dim3(kThreads), 0, as_stream(stream), d_x, n_samp, start_order,
                           stop_order, d_templates);
        check_launch();
    });
}

int toast_hip_template_select_dev(const double * d_src, const int32_t * d_key, int32_t value, int keep_equal,
                                  int64_t n_samp, double * d_out, void * stream) {
    return guarded([&] {
        if (n_samp <= 0) return;
        hipLaunchKernelGGL(k_template_select, flat_grid(n_samp), dim3(kThreads), 0, as_stream(stream), d_src, d_key, value,
                           keep_equal, n_samp, d_out);
        check_launch();
    });
}

int toast_hip_template_fit_dev(const double * d_templates, int64_t n_template, int64_t n_samp,
                               const int32_t * signal_index, const double * d_signal,
                               const int32_t * flag_index, const uint8_t * d_det_flags, uint8_t det_flag_mask,
                               const uint8_t * d_shared_flags, uint8_t shared_flag_mask, int64_t n_det,
                               double * d_proj, double * d_gram_common, double * d_gram_flagged,
                               int64_t * d_n_flagged, void * stream) {
    return guarded([&] {
        if (n_template <= 0 || n_samp <= 0 || n_det <= 0) return;
        hipStream_t st = as_stream(stream);
        TH_HIP(hipMemsetAsync(d_proj, 0, sizeof(double) * n_det * n_template, st));
        TH_HIP(hipMemsetAsync(d_gram_common, 0, sizeof(double) * n_template * n_template, st));
        TH_HIP(hipMemsetAsync(d_gram_flagged, 0, sizeof(double) * n_det * n_template * n_template, st));
        TH_HIP(hipMemsetAsync(d_n_flagged, 0, sizeof(int64_t) * n_det, st));
        ParamBlock pb;
        const size_t o_si = pb.push(signal_index, sizeof(int32_t) * n_det);
        std::vector<int32_t> no_flags(n_det, 0);
        const size_t o_fi = pb.push(d_det_flags != nullptr ? flag_index : no_flags.data(), sizeof(int32_t) * n_det);
        const char * dparam = pb.commit(st);
        const int32_t * sidx = (const int32_t *)(dparam + o_si);
        const int32_t * fidx = (const int32_t *)(dparam + o_fi);
        const int64_t n_pair = n_template * (n_template + 1) / 2;
        const int64_t gslice = 65536;
        hipLaunchKernelGGL(k_template_gram, dim3((unsigned)n_pair, (unsigned)((n_samp + gslice - 1) / gslice)), dim3(kThreads),
                           0, st, d_templates, n_template, n_samp, d_shared_flags, shared_flag_mask, gslice, d_gram_common);
        check_launch();
        // template rows per pass x detectors per wave: 64 accumulators per lane either way
        const int64_t pslice = 4096;
        auto project = [&](auto kernel, int ntg, int dpw) {
            const dim3 pgrid((unsigned)((n_det + 4 * dpw - 1) / (4 * dpw)), (unsigned)((n_samp + pslice - 1) / pslice));
            for (int64_t t0 = 0; t0 < n_template; t0 += ntg) {
                hipLaunchKernelGGL(kernel, pgrid, dim3(kThreads), 0, st, d_templates, n_template, t0, n_samp, sidx,
                                   d_signal, fidx, d_det_flags, det_flag_mask, d_shared_flags, shared_flag_mask, n_det,
                                   pslice, d_proj);
                check_launch();
            }
        };
        if (n_template <= 8) {
            project(k_template_project<8, 8>, 8, 8);
        } else if (n_template <= 16) {
            project(k_template_project<16, 4>, 16, 4);
        } else if (n_template <= 24 || (n_template > 32 && n_template <= 48)) {
            project(k_template_project<24, 3>, 24, 3);
        } else {
            project(k_template_project<32, 2>, 32, 2);
        }
        if (d_det_flags != nullptr) {
            const int64_t fslice = 65536;
            hipLaunchKernelGGL(k_template_gram_flagged, dim3((unsigned)((n_samp + fslice - 1) / fslice), (unsigned)n_det),
                               dim3(kThreads), 0, st, d_templates, n_template, n_samp, fidx, d_det_flags, det_flag_mask,
                               d_shared_flags, shared_flag_mask, fslice, d_gram_flagged, d_n_flagged);
            check_launch();
        }
    });
}

int toast_hip_template_subtract_dev(const double * d_templates, int64_t n_template, int64_t first_template,
                                    int64_t n_samp, const int32_t * signal_index, double * d_signal,
                                    const double * d_coeff, int64_t n_det, void * stream) {
    return guarded([&] {
        if (n_samp <= 0 || n_det <= 0 || first_template >= n_template) return;
        if (first_template < 0) fail_arg("template_subtract: first_template must be >= 0");
        hipStream_t st = as_stream(stream);
        ParamBlock pb;
        const size_t o_si = pb.push(signal_index, sizeof(int32_t) * n_det);
        const int32_t * sidx = (const int32_t *)(pb.commit(st) + o_si);
        const int dets_per_block = 32;
        const dim3 grid((unsigned)((n_samp + kTile - 1) / kTile), (unsigned)((n_det + dets_per_block - 1) / dets_per_block));
        hipLaunchKernelGGL(k_template_subtract<kTmplGroup>, grid, dim3(kThreads), 0, st, d_templates, n_template,
                           first_template, n_samp, sidx, d_signal, d_coeff, n_det, dets_per_block);
        check_launch();
    });
}

}  // extern "C"
