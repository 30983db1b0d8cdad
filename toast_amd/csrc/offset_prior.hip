// offset_prior.hip -- the Offset template's amplitude-domain noise prior and its preconditioners
// on the device.
//
// The reference applies these on the host only (templates/offset/offset.py:884-960 `_add_prior`,
// :963-1005 `_apply_precond`, both raise NotImplementedError under use_accel): one
// scipy.signal.convolve / scipy.linalg.cho_solve_banded per detector and view, which forces the
// amplitude vectors back to the host twice per PCG iteration.  Here the amplitudes stay in HBM:
//
//   k_offset_convolve     "same"-mode correlation of every (detector, observation, view) segment
//                         with its truncated filter (the prior, and the Toeplitz preconditioner of
//                         precond_width <= 1).  One thread per amplitude.
//   k_offset_banded_solve forward / backward substitution with the banded Cholesky factor of each
//                         segment (precond_width > 1).  The recurrence along a segment is
//                         sequential, the band is not: one 64-lane wave per segment, lane k holds
//                         band k, the history of the last w-1 solutions lives in the lanes'
//                         registers and moves by one lane per step (DPP wave_shr), the band sum is
//                         a DPP reduction; the factor rows stream through a wave-private LDS
//                         tile with coalesced loads.  No workgroup barriers; ~1000 segments run
//                         concurrently.

#include "kernel_common.hpp"

namespace {

__device__ __forceinline__ int64_t find_segment(const int64_t * __restrict__ seg_start, int64_t n_seg, int64_t i) {
    int64_t lo = 0, hi = n_seg;  // seg_start[lo] <= i < seg_start[hi]
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (seg_start[mid] <= i) {
            lo = mid;
        } else {
            hi = mid;
        }
    }
    return lo;
}

// out[i] (+)= sum_t filt[t] * in[j + c - t], c = (L - 1) / 2, j = i - segment start, terms outside
// the segment dropped: scipy.signal.convolve(in, filt, mode="same") per segment; flagged
// amplitudes are zeroed afterwards (offset.py:918-943, 985-1001).
template <bool ACCUMULATE>
__global__ __launch_bounds__(kThreads) void k_offset_convolve(
    int64_t n_amp, int64_t n_seg, const int64_t * __restrict__ seg_start, const int64_t * __restrict__ filt_start,
    const int64_t * __restrict__ filt_len, const double * __restrict__ filters, const double * __restrict__ in,
    const uint8_t * __restrict__ flags, double * __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n_amp; i += (int64_t)gridDim.x * kThreads) {
        if (flags[i] != 0) {
            out[i] = 0.0;
            continue;
        }
        const int64_t s = find_segment(seg_start, n_seg, i);
        const int64_t first = seg_start[s];
        const int64_t n = seg_start[s + 1] - first;
        const int64_t len = filt_len[s];
        const double * __restrict__ f = filters + filt_start[s];
        const double * __restrict__ x = in + first;
        const int64_t jc = (i - first) + ((len - 1) >> 1);
        const int64_t t0 = (jc - (n - 1) > 0) ? jc - (n - 1) : 0;
        const int64_t t1 = (jc < len - 1) ? jc : len - 1;
        double acc = 0.0;
        for (int64_t t = t0; t <= t1; ++t) acc += f[t] * x[jc - t];
        out[i] = ACCUMULATE ? out[i] + acc : acc;
    }
}

__device__ __forceinline__ double wave_total(double v) {
    v += dpp_f64<kDppRowShr + 1>(v);
    v += dpp_f64<kDppRowShr + 2>(v);
    v += dpp_f64<kDppRowShr + 4>(v);
    v += dpp_f64<kDppRowShr + 8>(v);
    v += dpp_f64<kDppRowBcast15, 0xa>(v);
    v += dpp_f64<kDppRowBcast31, 0xc>(v);
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double lane_value(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// One direction of the banded solve for one segment.  `coef` rows hold, for step i, the reciprocal
// of the diagonal in slot 0 and the w-1 couplings to the previous solutions of this direction in
// slots 1..w-1 (zero where the band leaves the segment).  NR registers per lane: band k = lane +
// 64 r.  The rows of R consecutive steps are contiguous in memory: they are staged through LDS
// with fully coalesced loads, and the row of the next step is read from LDS while the current
// step reduces.
template <int NR, int R, bool BACKWARD>
__device__ __forceinline__ void banded_sweep(int64_t n, int w, const double * __restrict__ coef,
                                             const double * __restrict__ rhs, double * __restrict__ out,
                                             const uint8_t * __restrict__ flags, double * __restrict__ tile) {
    const int lane = threadIdx.x & 63;
    double h[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) h[r] = 0.0;
    for (int64_t blk = 0; blk < n; blk += R) {
        const int steps = (n - blk < R) ? (int)(n - blk) : R;
        // rows [blk, blk + steps) of this direction: a contiguous span of the factor
        const int64_t row_lo = BACKWARD ? n - blk - steps : blk;
        const double * __restrict__ src = coef + row_lo * w;
        __builtin_amdgcn_wave_barrier();
        for (int q = lane; q < steps * w; q += 64) tile[q] = src[q];
        // right-hand sides of this block, one per lane, in sweep order
        const int64_t mine = BACKWARD ? n - 1 - (blk + lane) : blk + lane;
        const bool live = lane < steps;
        const double b = live ? rhs[mine] : 0.0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double y_keep = 0.0;
        double c_next[NR];
        {
            const int row = BACKWARD ? steps - 1 : 0;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int k = lane + 64 * r;
                c_next[r] = (k < w) ? tile[row * w + k] : 0.0;
            }
        }
        for (int sidx = 0; sidx < steps; ++sidx) {
            double c[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) c[r] = c_next[r];
            if (sidx + 1 < steps) {
                const int row = BACKWARD ? steps - 2 - sidx : sidx + 1;
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int k = lane + 64 * r;
                    c_next[r] = (k < w) ? tile[row * w + k] : 0.0;
                }
            }
            double p = (lane == 0) ? 0.0 : c[0] * h[0];
#pragma unroll
            for (int r = 1; r < NR; ++r) p += c[r] * h[r];
            const double s = wave_total(p);
            const double rdiag = lane_value(c[0], 0);
            const double y = (lane_value(b, sidx) - s) * rdiag;
            if (lane == sidx) y_keep = y;
            // shift the history by one band: lane k takes lane k-1, lane 0 of register r takes lane
            // 63 of register r-1, the new solution enters at band 1
#pragma unroll
            for (int r = NR - 1; r >= 0; --r) {
                double v = h[r];
                if (r == 0 && lane == 0) v = y;
                double moved = dpp_f64<kDppWaveShr1>(v);
                if (r > 0) {
                    const double carry = lane_value(h[r - 1], 63);
                    if (lane == 0) moved = carry;
                }
                h[r] = moved;
            }
        }
        if (live) {
            double v = y_keep;
            if (BACKWARD && flags[mine] != 0) v = 0.0;
            out[mine] = v;
        }
    }
}

// L y = b, then L^T x = y (scipy.linalg.cho_solve_banded with a lower factor, offset.py:990-999);
// flagged amplitudes are zeroed in the result.
template <int NR>
__global__ __launch_bounds__(64) void k_offset_banded_solve(
    const int64_t * __restrict__ seg_start, const int32_t * __restrict__ band_width,
    const int64_t * __restrict__ band_start, const double * __restrict__ fwd, const double * __restrict__ bwd,
    const double * __restrict__ in, const uint8_t * __restrict__ flags, double * __restrict__ out) {
    const int64_t s = blockIdx.x;
    const int64_t first = seg_start[s];
    const int64_t n = seg_start[s + 1] - first;
    const int w = band_width[s];
    if (n <= 0) return;
    const double * __restrict__ cf = fwd + band_start[s];
    const double * __restrict__ cb = bwd + band_start[s];
    constexpr int R = (NR == 1) ? 64 : 16;   // rows per LDS tile: R * 64 * NR doubles = 32 KB
    __shared__ double tile[R * 64 * NR];
    banded_sweep<NR, R, false>(n, w, cf, in + first, out + first, flags + first, tile);
    __threadfence_block();
    banded_sweep<NR, R, true>(n, w, cb, out + first, out + first, flags + first, tile);
}

}  // namespace

extern "C" {

int toast_hip_template_offset_convolve_dev(int64_t n_amp, int64_t n_seg, const int64_t * d_seg_start,
                                           const int64_t * d_filt_start, const int64_t * d_filt_len,
                                           const double * d_filters, const double * d_amp_in,
                                           const uint8_t * d_amplitude_flags, double * d_amp_out, int accumulate,
                                           void * stream) {
    return guarded([&] {
        if (n_amp <= 0 || n_seg <= 0) return;
        if (d_amp_in == d_amp_out) fail_arg("offset convolve: input and output amplitudes must differ");
        if (accumulate) {
            hipLaunchKernelGGL(k_offset_convolve<true>, flat_grid(n_amp), dim3(kThreads), 0, as_stream(stream), n_amp,
                               n_seg, d_seg_start, d_filt_start, d_filt_len, d_filters, d_amp_in, d_amplitude_flags,
                               d_amp_out);
        } else {
            hipLaunchKernelGGL(k_offset_convolve<false>, flat_grid(n_amp), dim3(kThreads), 0, as_stream(stream), n_amp,
                               n_seg, d_seg_start, d_filt_start, d_filt_len, d_filters, d_amp_in, d_amplitude_flags,
                               d_amp_out);
        }
        check_launch();
    });
}

int toast_hip_template_offset_banded_solve_dev(int64_t n_seg, const int64_t * d_seg_start,
                                               const int32_t * d_band_width, int32_t max_band_width,
                                               const int64_t * d_band_start, const double * d_forward,
                                               const double * d_backward, const double * d_amp_in,
                                               const uint8_t * d_amplitude_flags, double * d_amp_out,
                                               void * stream) {
    return guarded([&] {
        if (n_seg <= 0) return;
        if (max_band_width < 1 || max_band_width > 256) fail_arg("offset banded solve: band width must be 1..256");
        const dim3 grid((unsigned)n_seg);
        if (max_band_width <= 64) {
            hipLaunchKernelGGL(k_offset_banded_solve<1>, grid, dim3(64), 0, as_stream(stream), d_seg_start,
                               d_band_width, d_band_start, d_forward, d_backward, d_amp_in, d_amplitude_flags,
                               d_amp_out);
        } else if (max_band_width <= 128) {
            hipLaunchKernelGGL(k_offset_banded_solve<2>, grid, dim3(64), 0, as_stream(stream), d_seg_start,
                               d_band_width, d_band_start, d_forward, d_backward, d_amp_in, d_amplitude_flags,
                               d_amp_out);
        } else {
            hipLaunchKernelGGL(k_offset_banded_solve<4>, grid, dim3(64), 0, as_stream(stream), d_seg_start,
                               d_band_width, d_band_start, d_forward, d_backward, d_amp_in, d_amplitude_flags,
                               d_amp_out);
        }
        check_launch();
    });
}

}  // extern "C"
