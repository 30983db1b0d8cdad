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

// The same convolution for filters of more than 32 taps (the prior: 87 taps; the Toeplitz
// preconditioner of precond_width <= 1 is not truncated much: 8191 taps for 3600 baselines).  One block per (segment, tile of 1024 outputs);
// each thread owns 4 consecutive outputs and slides a 4-value window of the input through its
// registers, so a tap costs one conflict-free LDS read (the window is stored with one pad slot per
// 32 values) and one scalar load of the tap for 4 FMAs.
constexpr int kConvOut = 4 * kThreads;    // outputs per block
constexpr int kConvTaps = 256;            // taps per LDS window
__device__ __forceinline__ int conv_pad(int k) { return k + (k >> 5); }

template <bool ACCUMULATE>
__global__ __launch_bounds__(kThreads) void k_offset_convolve_tiled(
    const int64_t * __restrict__ seg_start, const int64_t * __restrict__ filt_start, const int64_t * __restrict__ filt_len,
    const double * __restrict__ filters, const double * __restrict__ in, const uint8_t * __restrict__ flags,
    double * __restrict__ out) {
    constexpr int kWindow = kConvOut + kConvTaps - 1;
    __shared__ double win[kWindow + kWindow / 32 + 1];
    const int64_t s = blockIdx.x;      // segments on grid.x: detectors x observations can exceed 65535
    const int64_t first = seg_start[s];
    const int64_t n = seg_start[s + 1] - first;
    const int64_t j0 = (int64_t)blockIdx.y * kConvOut;
    if (j0 >= n) return;
    const int64_t len = filt_len[s];
    const int64_t c = (len - 1) >> 1;
    const double * __restrict__ f = filters + filt_start[s];
    const double * __restrict__ x = in + first;
    const int64_t j_hi = (j0 + kConvOut < n) ? j0 + kConvOut : n;   // one past the last output of the tile
    // taps that reach any input of the segment from any output of the tile
    int64_t t_lo = j0 + c - (n - 1);
    if (t_lo < 0) t_lo = 0;
    int64_t t_hi = j_hi - 1 + c;
    if (t_hi > len - 1) t_hi = len - 1;
    const int tid = threadIdx.x;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t t0 = t_lo; t0 <= t_hi; t0 += kConvTaps) {
        // window: x[wbase + k], k = 0 .. kWindow-1, zero outside the segment
        const int64_t wbase = j0 + c - (t0 + kConvTaps - 1);
        __syncthreads();
        for (int k = tid; k < kWindow; k += kThreads) {
            const int64_t i = wbase + k;
            win[conv_pad(k)] = (i >= 0 && i < n) ? x[i] : 0.0;
        }
        __syncthreads();
        // outputs j0 + 4 tid + r at tap t0 + u read window slot 4 tid + r + (kConvTaps - 1) - u
        const int top = 4 * tid + kConvTaps - 1;
        double w0 = win[conv_pad(top)], w1 = win[conv_pad(top + 1)], w2 = win[conv_pad(top + 2)],
               w3 = win[conv_pad(top + 3)];
        const int64_t taps = (t_hi - t0 + 1 < kConvTaps) ? t_hi - t0 + 1 : kConvTaps;
        const double * __restrict__ ft = f + t0;
        int u = 0;
        for (; u + 4 <= taps; u += 4) {
            const double f0 = ft[u], f1 = ft[u + 1], f2 = ft[u + 2], f3 = ft[u + 3];
            acc[0] = __builtin_fma(f0, w0, acc[0]);
            acc[1] = __builtin_fma(f0, w1, acc[1]);
            acc[2] = __builtin_fma(f0, w2, acc[2]);
            acc[3] = __builtin_fma(f0, w3, acc[3]);
            w3 = win[conv_pad(top - u - 1)];
            acc[0] = __builtin_fma(f1, w3, acc[0]);
            acc[1] = __builtin_fma(f1, w0, acc[1]);
            acc[2] = __builtin_fma(f1, w1, acc[2]);
            acc[3] = __builtin_fma(f1, w2, acc[3]);
            w2 = win[conv_pad(top - u - 2)];
            acc[0] = __builtin_fma(f2, w2, acc[0]);
            acc[1] = __builtin_fma(f2, w3, acc[1]);
            acc[2] = __builtin_fma(f2, w0, acc[2]);
            acc[3] = __builtin_fma(f2, w1, acc[3]);
            w1 = win[conv_pad(top - u - 3)];
            acc[0] = __builtin_fma(f3, w1, acc[0]);
            acc[1] = __builtin_fma(f3, w2, acc[1]);
            acc[2] = __builtin_fma(f3, w3, acc[2]);
            acc[3] = __builtin_fma(f3, w0, acc[3]);
            w0 = win[conv_pad(top - u - 4 >= 0 ? top - u - 4 : 0)];
            // after four taps the registers are back in order: (w0 .. w3) = slots top-u-4 .. top-u-1
        }
        for (; u < taps; ++u) {
            const double fu = ft[u];
            acc[0] = __builtin_fma(fu, w0, acc[0]);
            acc[1] = __builtin_fma(fu, w1, acc[1]);
            acc[2] = __builtin_fma(fu, w2, acc[2]);
            acc[3] = __builtin_fma(fu, w3, acc[3]);
            w3 = w2;
            w2 = w1;
            w1 = w0;
            w0 = win[conv_pad(top - u - 1 >= 0 ? top - u - 1 : 0)];
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t j = j0 + 4 * tid + r;
        if (j < n) {
            const int64_t i = first + j;
            out[i] = (flags[i] != 0) ? 0.0 : (ACCUMULATE ? out[i] + acc[r] : acc[r]);
        }
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


__device__ __forceinline__ void lds_sync() {   // LDS writes of this wave visible to its later reads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Geometry of one solve: NR registers per lane (window of 64 NR pending unknowns), tiles of R steps.
template <int NR, int WMAX>
struct Banded {
    static constexpr int kWin = 64 * NR;                 // LDS row stride; slots >= w stay zero
    static constexpr int R = (NR == 1) ? 64 : 16;        // steps per tile
    static constexpr int G = (NR >= 4) ? 1 : 4 / NR;     // steps whose operands are read together
    static constexpr int kRows = R + 2 * G;              // + zero rows read by the pipelined tail
    static constexpr int kPre = R * WMAX / 64;           // tile doubles per lane (global prefetch)
    static constexpr int kSide = 2 * kRows;              // rhs / diag per step, solved unknowns
};

// One direction of the banded solve for one segment.  Row i of `coef` holds the reciprocal of the
// diagonal in slot 0 and, in slots 1..w-1, the couplings of unknown i to the w-1 unknowns that
// FOLLOW it in sweep order (zero where the band leaves the segment): column i of L going forward,
// row i of L going backward.  Lane m (+ 64 r) holds the pending sum of the unknown m steps ahead.
//
// A single wave issues roughly one instruction per 9 cycles on this recurrence
// (tools/ubench_chain.hip), so the step is written for instruction count: the rows of R steps
// are staged in LDS in sweep order with a stride of 64 NR doubles and zero padding, so that a step
// needs three unconditional LDS reads (its couplings, and 1 / diag and rhs / diag as broadcasts),
// two v_readlane, two FMAs, one DPP move and one LDS write; operands are read two groups ahead
// in alternating register sets.
template <int NR, int WMAX, bool BACKWARD>
__device__ __forceinline__ void banded_sweep(int64_t n, int w, const double * __restrict__ coef,
                                             const double * rhs, double * out,
                                             const uint8_t * __restrict__ flags, double * __restrict__ tile,
                                             double * __restrict__ side) {
    using B = Banded<NR, WMAX>;
    constexpr int R = B::R, G = B::G, kWin = B::kWin, kPre = B::kPre;
    const int lane = threadIdx.x & 63;
    double * __restrict__ solved = side + B::kRows;
    const unsigned long long inv_w = (1ull << 32) / (unsigned)w + 1ull;   // q / w == (q * inv_w) >> 32, q < 2^16
    double pend[NR];
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
    struct Operands {
        double c[G][NR], rd[G], t[G];
    };
    auto read_rows = [&](Operands & o, int first_step) {
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const double * __restrict__ row = tile + (first_step + u) * kWin;
            o.rd[u] = row[0];                   // same address in every lane: a broadcast
            o.t[u] = side[first_step + u];
#pragma unroll
            for (int r = 0; r < NR; ++r) o.c[u][r] = row[lane + 64 * r];
        }
    };
    auto run_steps = [&](const Operands & o, int first_step) {
#pragma unroll
        for (int u = 0; u < G; ++u) {
            // y = (b - sum) / diag, as b / diag - sum / diag in one FMA
            const double y = __builtin_fma(-lane_value(pend[0], 0), o.rd[u], o.t[u]);
            if (lane == 0) solved[first_step + u] = y;
            // every pending sum takes its term of this unknown (lane 0 of register 0 is leaving the
            // window: what it accumulates is discarded), then the window moves on by one lane
#pragma unroll
            for (int r = 0; r < NR; ++r) pend[r] = __builtin_fma(o.c[u][r], y, pend[r]);
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
    };
    fetch(0);
    for (int64_t blk = 0; blk < n; blk += R) {
        const int steps = (n - blk < R) ? (int)(n - blk) : R;
        const int total = steps * w;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < kPre; ++u) {
            const int q = u * 64 + lane;
            if (q < total) {
                const int row = (int)(((unsigned long long)q * inv_w) >> 32);
                const int k = q - row * w;
                const int st = BACKWARD ? steps - 1 - row : row;   // sweep order
                tile[st * kWin + k] = pre[u];
            }
        }
        if (steps < R) {   // last block: the rows behind it must read as zero (no-op steps)
            for (int q = lane; q < (R - steps) * kWin; q += 64) tile[steps * kWin + q] = 0.0;
        }
        // right-hand sides of this block, one per lane, in sweep order
        const int64_t mine = BACKWARD ? n - 1 - (blk + lane) : blk + lane;
        const bool live = lane < steps;
        const double b = b_pre;
        if (blk + R < n) fetch(blk + R);
        lds_sync();
        if (lane < R) side[lane] = live ? b * tile[lane * kWin] : 0.0;   // rhs / diagonal
        lds_sync();
        Operands oa, ob;
        read_rows(oa, 0);
        for (int g = 0; g < steps; g += 2 * G) {
            read_rows(ob, g + G);
            run_steps(oa, g);
            read_rows(oa, g + 2 * G);
            run_steps(ob, g + G);
        }
        lds_sync();
        if (live) {
            double v = solved[lane];
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
    using B = Banded<NR, WMAX>;
    const int64_t s = blockIdx.x;
    const int64_t first = seg_start[s];
    const int64_t n = seg_start[s + 1] - first;
    const int w = band_width[s];
    if (n <= 0 || w < 1) return;
    const double * __restrict__ cf = fwd + band_start[s];
    const double * __restrict__ cb = bwd + band_start[s];
    __shared__ double tile[B::kRows * B::kWin];
    __shared__ double side[B::kSide];
    for (int q = threadIdx.x; q < B::kRows * B::kWin; q += 64) tile[q] = 0.0;
    for (int q = threadIdx.x; q < B::kSide; q += 64) side[q] = 0.0;
    banded_sweep<NR, WMAX, false>(n, w, cf, in + first, out + first, flags + first, tile, side);
    __threadfence_block();
    banded_sweep<NR, WMAX, true>(n, w, cb, out + first, out + first, flags + first, tile, side);
}

// Banded Cholesky factorisation of every segment's preconditioner matrix
//   M = diag(diag_scale / offset_var) + Toeplitz(band)         (offset.py:522-545)
// written straight into the two tables k_offset_banded_solve reads.  The reference factorises on
// the host with scipy.linalg.cholesky_banded, one call per detector and view; at cfg3 that is
// 1024 LAPACK calls (0.8 - 4 s) plus a 1.2 GB upload of the factors.  Here: one wave per segment,
// left-looking, column j from the previous w-1 columns kept in an LDS ring:
//   L[j+m][j] = (M[j+m][j] - sum_k L[j+m][j-k] L[j][j-k]) / L[j][j],   lane m, k = 1 .. w-1-m.
// status[s] != 0: the matrix of segment s is not positive definite at this band width (the
// caller widens the band as the reference does, offset.py:546-566).
template <int WMAX>
__global__ __launch_bounds__(64) void k_offset_banded_cholesky(
    const int64_t * __restrict__ seg_start, const int32_t * __restrict__ band_width,
    const int64_t * __restrict__ band_start, const int64_t * __restrict__ toep_start,
    const int32_t * __restrict__ toep_len, const double * __restrict__ toeplitz,
    const double * __restrict__ diag_scale, const double * __restrict__ offset_var, double * __restrict__ fwd,
    double * __restrict__ bwd, int32_t * __restrict__ status) {
    const int64_t s = blockIdx.x;
    const int64_t first = seg_start[s];
    const int64_t n = seg_start[s + 1] - first;
    const int w = band_width[s];
    if (n <= 0 || w < 1) return;
    const int lane = threadIdx.x & 63;
    __shared__ double hist[WMAX * WMAX];   // ring of the last WMAX columns, WMAX slots each
    for (int q = lane; q < WMAX * WMAX; q += 64) hist[q] = 0.0;
    const double band = (lane < toep_len[s]) ? toeplitz[toep_start[s] + lane] : 0.0;
    const double dscale = diag_scale[s];
    const double * __restrict__ var = offset_var + first;
    double * __restrict__ f = fwd + band_start[s];
    double * __restrict__ b = bwd + band_start[s];
    lds_sync();
    bool ok = true;
    for (int64_t j = 0; j < n; ++j) {
        double acc = band;
        if (lane == 0) {
            const double v = var[j];
            // a flagged amplitude (variance 0) keeps the Toeplitz part only: see
            // templates/offset_prior.py -- the reference divides by zero there
            if (v > 0.0) acc += dscale / v;
        }
        const int kmax = (j < w - 1) ? (int)j : w - 1;
        for (int k = 1; k <= kmax; ++k) {
            const double * __restrict__ col = hist + (int)((j - k) & (WMAX - 1)) * WMAX;
            const double mine = (lane + k < w) ? col[lane + k] : 0.0;
            acc = __builtin_fma(-mine, col[k], acc);
        }
        const double d = lane_value(acc, 0);
        if (!(d > 0.0)) {
            ok = false;
            break;
        }
        const double ljj = sqrt(d);
        double l = (lane == 0) ? ljj : acc / ljj;
        if (lane >= w || j + lane >= n) l = 0.0;
        __builtin_amdgcn_wave_barrier();
        if (lane < WMAX) hist[(int)(j & (WMAX - 1)) * WMAX + lane] = l;
        if (lane < w) {
            const double outv = (lane == 0) ? 1.0 / ljj : l;
            f[j * w + lane] = outv;                               // L[j + m][j]: column j
            if (j + lane < n) b[(j + lane) * w + lane] = outv;    // L[i][i - m] with i = j + m
        }
        lds_sync();
    }
    if (lane == 0) status[s] = ok ? 0 : 1;
}

}  // namespace

extern "C" {

int toast_hip_template_offset_convolve_dev(int64_t n_amp, int64_t n_seg, const int64_t * d_seg_start,
                                           int64_t max_segment_len, const int64_t * d_filt_start,
                                           const int64_t * d_filt_len, int64_t max_filter_len,
                                           const double * d_filters, const double * d_amp_in,
                                           const uint8_t * d_amplitude_flags, double * d_amp_out, int accumulate,
                                           void * stream) {
    return guarded([&] {
        if (n_amp <= 0 || n_seg <= 0) return;
        if (d_amp_in == d_amp_out) fail_arg("offset convolve: input and output amplitudes must differ");
        if (max_filter_len > 32 && max_segment_len > 0) {
            // all but tiny filters: LDS-tiled, register-blocked kernel (87 taps: 0.037 vs 0.32 ms at cfg3)
            const dim3 grid((unsigned)n_seg, (unsigned)((max_segment_len + kConvOut - 1) / kConvOut));
            auto kern = accumulate ? k_offset_convolve_tiled<true> : k_offset_convolve_tiled<false>;
            hipLaunchKernelGGL(kern, grid, dim3(kThreads), 0, as_stream(stream), d_seg_start, d_filt_start, d_filt_len,
                               d_filters, d_amp_in, d_amplitude_flags, d_amp_out);
        } else if (accumulate) {
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

int toast_hip_template_offset_banded_cholesky_dev(int64_t n_seg, const int64_t * d_seg_start,
                                                  const int32_t * d_band_width, int32_t max_band_width,
                                                  const int64_t * d_band_start, const int64_t * d_toeplitz_start,
                                                  const int32_t * d_toeplitz_len, const double * d_toeplitz,
                                                  const double * d_diag_scale, const double * d_offset_var,
                                                  double * d_forward, double * d_backward, int32_t * d_status,
                                                  void * stream) {
    return guarded([&] {
        if (n_seg <= 0) return;
        if (max_band_width < 1 || max_band_width > 64) fail_arg("offset banded cholesky: band width must be 1..64");
        const dim3 grid((unsigned)n_seg);
        auto launch = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, grid, dim3(64), 0, as_stream(stream), d_seg_start, d_band_width, d_band_start,
                               d_toeplitz_start, d_toeplitz_len, d_toeplitz, d_diag_scale, d_offset_var, d_forward,
                               d_backward, d_status);
        };
        if (max_band_width <= 32) {
            launch(k_offset_banded_cholesky<32>);
        } else {
            launch(k_offset_banded_cholesky<64>);
        }
        check_launch();
    });
}

}  // extern "C"
