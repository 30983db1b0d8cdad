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
//                         sequential, the band is not: one 64-lane wave per segment in the
//                         column-oriented (axpy) form -- lane m holds the pending sum of the
//                         unknown m steps ahead; a step broadcasts the finished unknown from lane 0
//                         (v_readlane), every lane adds its coupling times that unknown with one
//                         FMA and the window moves by one lane (DPP wave_shl).  No reduction on the
//                         critical path, ~10 VALU instructions per step.  The factor rows stream
//                         through a wave-private LDS tile that is prefetched one tile ahead; no
//                         workgroup barriers; ~1000 segments run concurrently.

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

__device__ __forceinline__ double lane_value(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

template <int CTRL>
__device__ __forceinline__ double dpp_f64_zero_fill(double x) {   // lanes without a source read 0
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

constexpr int kDppWaveShl1 = 0x130;  // lane i reads lane i + 1 of the wave

// One direction of the banded solve for one segment.  Row i of `coef` holds the reciprocal of the
// diagonal in slot 0 and, in slots 1..w-1, the couplings of unknown i to the w-1 unknowns that
// FOLLOW it in sweep order (zero where the band leaves the segment): column i of L going forward,
// row i of L going backward.  NR registers per lane: pending unknown m = lane + 64 r steps ahead.
// The rows of R consecutive steps are contiguous in memory: they are staged through LDS with fully
// coalesced loads, and the rows of the next steps are read from LDS while the current steps run.
// `side` is 2 R doubles of LDS: rhs / diagonal per step, and the solved unknowns of the block.
template <int NR, int R, int WMAX, bool BACKWARD>
__device__ __forceinline__ void banded_sweep(int64_t n, int w, const double * __restrict__ coef,
                                             const double * rhs, double * out,
                                             const uint8_t * __restrict__ flags, double * __restrict__ tile,
                                             double * __restrict__ side) {
    constexpr int kPre = R * WMAX / 64;  // doubles per lane of one tile: R rows x WMAX bands / 64 lanes
    const int lane = threadIdx.x & 63;
    double pend[NR];  // sum of coupling x solved unknown, per pending unknown of the window
#pragma unroll
    for (int r = 0; r < NR; ++r) pend[r] = 0.0;
    // The tile and the right-hand sides of the NEXT block are fetched into registers while the
    // current block runs its recurrence, so the HBM latency of the factor (the only large
    // operand, 2 w doubles per amplitude) is off the critical path.
    double pre[kPre];
    double b_pre = 0.0;
    auto fetch = [&](int64_t blk) {
        const int steps = (n - blk < R) ? (int)(n - blk) : R;
        const int64_t row_lo = BACKWARD ? n - blk - steps : blk;
        const double * __restrict__ src = coef + row_lo * w;
        const int total = steps * w;
#pragma unroll
        for (int u = 0; u < kPre; ++u) {
            const int q = u * 64 + lane;
            pre[u] = (q < total) ? src[q] : 0.0;
        }
        const int64_t mine = BACKWARD ? n - 1 - (blk + lane) : blk + lane;
        b_pre = (lane < steps) ? rhs[mine] : 0.0;
    };
    fetch(0);
    for (int64_t blk = 0; blk < n; blk += R) {
        const int steps = (n - blk < R) ? (int)(n - blk) : R;
        const int total = steps * w;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < kPre; ++u) {
            const int q = u * 64 + lane;
            if (q < total) tile[q] = pre[u];
        }
        // right-hand sides of this block, one per lane, in sweep order
        const int64_t mine = BACKWARD ? n - 1 - (blk + lane) : blk + lane;
        const bool live = lane < steps;
        const double b = b_pre;
        if (blk + R < n) fetch(blk + R);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // rhs / diagonal of every step of the block, read back below as a broadcast
        {
            const int my_row = BACKWARD ? steps - 1 - lane : lane;
            if (lane < R) side[lane] = live ? b * tile[my_row * w] : 0.0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // The rows are read from LDS a group of G steps ahead (the step itself is shorter than the
        // LDS latency); rows past the end of the block read as zero, which makes those steps no-ops.
        constexpr int G = (NR >= 4) ? 1 : 4 / NR;
        double c_next[G][NR], rd_next[G], t_next[G];
        auto read_rows = [&](int first_step) {
#pragma unroll
            for (int u = 0; u < G; ++u) {
                const int st = first_step + u;
                const bool valid = st < steps;
                const int row = BACKWARD ? steps - 1 - st : st;
                rd_next[u] = valid ? tile[row * w] : 0.0;   // same address in every lane: a broadcast
                t_next[u] = valid ? side[st] : 0.0;
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int k = lane + 64 * r;
                    c_next[u][r] = (valid && k < w) ? tile[row * w + k] : 0.0;
                }
            }
        };
        read_rows(0);
        for (int g = 0; g < steps; g += G) {
            double c[G][NR], rd[G], t[G];
#pragma unroll
            for (int u = 0; u < G; ++u) {
                rd[u] = rd_next[u];
                t[u] = t_next[u];
#pragma unroll
                for (int r = 0; r < NR; ++r) c[u][r] = c_next[u][r];
            }
            if (g + G < steps) read_rows(g + G);
#pragma unroll
            for (int u = 0; u < G; ++u) {
                // y = (b - sum) / diag, as b / diag - sum / diag in one FMA
                const double y = __builtin_fma(-lane_value(pend[0], 0), rd[u], t[u]);
                if (lane == 0 && g + u < steps) side[R + g + u] = y;
                // every pending sum takes its term of this unknown (lane 0 of register 0 is leaving
                // the window: what it accumulates is discarded), then the window moves on by one lane
#pragma unroll
                for (int r = 0; r < NR; ++r) pend[r] = __builtin_fma(c[u][r], y, pend[r]);
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    double moved = dpp_f64_zero_fill<kDppWaveShl1>(pend[r]);
                    if (r + 1 < NR) {
                        const double carry = lane_value(pend[r + 1], 0);
                        if (lane == 63) moved = carry;
                    }
                    pend[r] = moved;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const double y_keep = live ? side[R + lane] : 0.0;
        if (live) {
            double v = y_keep;
            if (BACKWARD && flags[mine] != 0) v = 0.0;
            out[mine] = v;
        }
    }
}

// L y = b, then L^T x = y (scipy.linalg.cho_solve_banded with a lower factor, offset.py:990-999);
// flagged amplitudes are zeroed in the result.
template <int NR, int WMAX>
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
    constexpr int R = (NR == 1) ? 64 : 16;   // rows per LDS tile: at most R * WMAX doubles = 32 KB
    __shared__ double tile[R * WMAX];
    __shared__ double side[2 * R];
    banded_sweep<NR, R, WMAX, false>(n, w, cf, in + first, out + first, flags + first, tile, side);
    __threadfence_block();
    banded_sweep<NR, R, WMAX, true>(n, w, cb, out + first, out + first, flags + first, tile, side);
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
        auto launch = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, grid, dim3(64), 0, as_stream(stream), d_seg_start, d_band_width, d_band_start,
                               d_forward, d_backward, d_amp_in, d_amplitude_flags, d_amp_out);
        };
        if (max_band_width <= 16) {
            launch(k_offset_banded_solve<1, 16>);
        } else if (max_band_width <= 32) {
            launch(k_offset_banded_solve<1, 32>);
        } else if (max_band_width <= 64) {
            launch(k_offset_banded_solve<1, 64>);
        } else if (max_band_width <= 128) {
            launch(k_offset_banded_solve<2, 128>);
        } else {
            launch(k_offset_banded_solve<4, 256>);
        }
        check_launch();
    });
}

}  // extern "C"
