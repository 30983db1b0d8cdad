// fft_fused.hip -- FFT noise weighting in three HBM passes (hand-written, gfx950).
//
// Same mathematics as the rocFFT pipeline of fft_filter.hip (reference: toast.fft.convolve with
// algorithm="numpy", src/toast/fft.py:163-212 padding / apodisation, :190-212 kernel
// interpolation, :296-350 rfft -> multiply -> irfft -> crop), organised so that a timestream
// crosses HBM three times instead of fourteen (profiles/r02_a_fft_rocfft_baseline_rocprofv3.txt):
//
//   The padded real series x[0 .. n_fft) is transformed as ONE complex FFT of length M = n_fft / 2
//   on z[j] = x[2j] + i x[2j+1], factored M = N1 x N2 (four-step), with the real <-> complex
//   packing, the kernel multiplication and the inverse packing done where the data already sits in
//   LDS:
//
//   pass 1  k_fft_cols<FWD>   for every group of C adjacent columns j2: read the TIMESTREAM directly
//                             (mirror + apodisation of set_rfft_input evaluated on the fly: no
//                             fill pass), FFT of length N1 down the columns, twiddle w_M^(k1 j2),
//                             write work[k1][j2]
//   pass 2  k_fft_rows        for every row pair (k1, N1 - k1): FFT of length N2 along both rows
//                             -> Z[k1 + N1 k2]; bins k and M - k now sit in the same tile:
//                             X[k] = E + w_N^k O (real-FFT unpacking), Y = K(f) X (or X / K),
//                             DC removed, Nyquist made real, repacked to Z'[k]; inverse row FFT;
//                             write work in place
//   pass 3  k_fft_cols<INV>   twiddle, inverse FFT of length N1 down the columns, crop and scale
//                             straight into the timestream (no crop pass)
//
// Every FFT inside a pass is a Stockham autosort transform on a tile of 4096 complex doubles held
// by 256 threads x 16 points: radix-16 / 8 / 4 / 2 butterflies in registers, tile exchanges
// through 64 KB of LDS (XOR-swizzled against bank conflicts), the first and last radix-16 stage
// go straight from / to global memory.  The inverse transforms reuse the forward code on data with
// real and imaginary parts swapped (ifft(z) = swap(fft(swap(z)))).
//
// Algorithmic HBM bytes per timestream sample: pass 1 reads 8 (+ mirror
// re-reads served by L2) and writes 8 n_fft / n_samp, pass 2 reads and writes 8 n_fft / n_samp,
// pass 3 reads 8 n_fft / n_samp and writes 8: 16 + 32 n_fft / n_samp = 109 B at cfg-3.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <array>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "runtime.hpp"
#include "runtime.hpp"
#include "fft_device.hpp"

namespace toast_hip {
namespace fused_fft {

// fft_reg.hip
bool rows_reg_supported(const Params & p);
void launch_rows_reg(const Params & p, unsigned n_det, bool tab_lds, size_t tab_bytes, hipStream_t st);
void launch_rows_reg16(const Params & p, unsigned n_det, bool tab_lds, size_t tab_bytes, hipStream_t st);
void pack_tables(const Params & p, char * blob, int64_t n_kern, hipStream_t st);
bool cols_reg_supported(const Params & p, int min_log_n1);
int cols_reg_log_c(int log_n1);
int cols_reg_twiddles(int log_n1, int * out);
void launch_cols_reg(const Params & p, bool inv, unsigned n_det, hipStream_t st);
// column passes: 1 (default) = tile in registers where fft_reg.hip has a kernel for the length (n_fft 2^21 .. 2^23,
// aligned timestreams), 0 = tile in LDS (k_fft_cols).  TOAST_HIP_FFT_COLS=reg|lds
namespace {
int g_cols_mode = -1;      // 0: LDS tiles, 1: register tiles for N1 >= 1024 (default), 2: register tiles for N1 >= 512 too
}
static int cols_mode() {
    if (g_cols_mode < 0) {
        const char * e = std::getenv("TOAST_HIP_FFT_COLS");
        const std::string v = (e != nullptr) ? std::string(e) : std::string();
        g_cols_mode = (v == "lds") ? 0 : (v == "reg9") ? 2 : 1;
    }
    return g_cols_mode;
}
static bool cols_reg_for(const Params & p) { return cols_mode() != 0 && cols_reg_supported(p, cols_mode() == 2 ? 9 : 10); }
// pass 3 at N1 = 512 (n_fft 2^21) runs in registers unless the LDS tile is asked for at every length (mode 0): there the
// forward LDS-tile kernel and the register kernel are a tie (2.75-3.07 against 2.94-2.97 ms from box to box), the inverse
// register kernel with 16 points per lane is 10 % ahead (same tile geometry: 8 columns)
static bool cols_inv_reg_for(const Params & p) { return cols_mode() != 0 && cols_reg_supported(p, 9); }
void set_cols_reg(int on) { g_cols_mode = (on == 2) ? 2 : (on ? 1 : 0); }

// pass 1 (INV = false) and pass 3 (INV = true): transforms of length N1 down the columns
template <int LT, int P, bool INV>
__global__ __launch_bounds__((1 << LT) / P, (P == 16 ? 2 : 4)) void k_fft_cols(const Params p) {
    constexpr int T = (1 << LT) / P;
    extern __shared__ double2 sm[];
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int log_c = LT - p.log_n1;                  // columns per tile
    const int64_t n2 = int64_t(1) << p.log_n2;
    // XCD-aware tile order: workgroups go round-robin to the 8 XCDs, so with tile = blockIdx.x every XCD's L2 would
    // hold every 8th 128-byte piece of a row of the work array; this way XCD x owns a contiguous eighth of the columns
    unsigned bx = blockIdx.x;
    if (!INV && p.tile_order != nullptr) bx = (unsigned)p.tile_order[bx];
    else if (p.xcd_order && (gridDim.x & 7u) == 0u) bx = (bx & 7u) * (gridDim.x >> 3) + (bx >> 3);
    const int64_t c0 = (int64_t)bx << log_c;
    const int64_t m = int64_t(1) << (p.log_n1 + p.log_n2);
    double2 * __restrict__ work = p.work + (int64_t)b * m;
    double * __restrict__ row = p.tod + (int64_t)p.d_idx[p.det0 + b] * p.n_samp;
    double2 v[P];
    if (!INV) {
        if (p.aligned) {
            // n_buffer, n_samp, n_reflect even and 16-byte aligned rows: (x[2j], x[2j+1]) is one aligned
            // pair of the timestream (reversed in the mirrored parts), one 16-byte load per point
#pragma unroll
            for (int k = 0; k < P; ++k) {
                const int e = tid + k * T;
                const int64_t j = ((int64_t)(e >> log_c) << p.log_n2) + c0 + (e & ((1 << log_c) - 1));
                v[k] = padded_pair(row, p.apod, 2 * j - p.n_buffer, p.n_samp, p.n_reflect, (p.stream_hint & 2) != 0);
            }
        } else {
#pragma unroll
            for (int k = 0; k < P; ++k) {
                const int e = tid + k * T;
                const int64_t j = ((int64_t)(e >> log_c) << p.log_n2) + c0 + (e & ((1 << log_c) - 1));
                v[k].x = padded(row, p.apod, 2 * j, p.n_samp, p.n_buffer, p.n_reflect);
                v[k].y = padded(row, p.apod, 2 * j + 1, p.n_samp, p.n_buffer, p.n_reflect);
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const int e = tid + k * T;
            const int64_t k1 = e >> log_c;
            const int64_t j2 = c0 + (e & ((1 << log_c) - 1));
            v[k] = (p.stream_hint & 4) ? load_nt(work + (k1 << p.log_n2) + j2) : work[(k1 << p.log_n2) + j2];
        }
        col_twiddles<LT, P>(v, p, tid, log_c, c0, col_twiddles_prepare<LT, P>(p, tid, log_c, c0));
    }
    PHASE_DECL;
    PHASE_WAIT_LOADS;
    PHASE_MARK(INV ? 10 : 5);
    tile_fft_t<LT, P>(v, sm, tid, p.log_n1, p.tb.wtile, tid, tid);
    PHASE_MARK(INV ? 11 : 6);
    if (!INV) {
        col_twiddles<LT, P>(v, p, tid, log_c, c0, col_twiddles_prepare<LT, P>(p, tid, log_c, c0));
        const int tid_tail = opaque_vgpr(tid);
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const int e = tid_tail + k * T;
            const int64_t k1 = e >> log_c;
            const int64_t j2 = c0 + (e & ((1 << log_c) - 1));
            if (p.stream_hint & 1) store_nt(work + (k1 << p.log_n2) + j2, v[k]);
            else work[(k1 << p.log_n2) + j2] = v[k];
        }
    } else {
        // the transform ran on swapped data: Re z' = v.y, Im z' = v.x; crop + scale (fft.py:341-350)
        const int log_n2_tail = opaque_sgpr(p.log_n2);
        const int tid_tail = opaque_vgpr(tid);
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const int e = tid_tail + k * T;
            const int64_t j = ((int64_t)(e >> log_c) << log_n2_tail) + c0 + (e & ((1 << log_c) - 1));
            const int64_t s = 2 * j - p.n_buffer;
            if (p.aligned) {
                // s and n_samp even, row 16-byte aligned: both samples of the pair are inside or outside together
                if (s >= 0 && s < p.n_samp) {
                    const double2 o = make_double2(v[k].y * p.scale, v[k].x * p.scale);
                    if (p.stream_hint & 8) store_nt(reinterpret_cast<double2 *>(row + s), o);
                    else *reinterpret_cast<double2 *>(row + s) = o;
                }
            } else {
                if (s >= 0 && s < p.n_samp) row[s] = v[k].y * p.scale;
                if (s + 1 >= 0 && s + 1 < p.n_samp) row[s + 1] = v[k].x * p.scale;
            }
        }
    }
    PHASE_MARK(INV ? 12 : 7);
    PHASE_WAIT_LOADS;
    PHASE_MARK(INV ? 13 : 8);
    (void)n2;
}


// pass 2: rows (k1, N1 - k1) [block 0: rows 0 and N1 / 2]
// LT = log2 of the tile = 2 N2: 12 (N2 = 2048, 64 KB, two workgroups per CU) or 11 (N2 = 1024, 32 KB, four per CU)
template <int LT, int P, bool TLDS>
__global__ __launch_bounds__((1 << LT) / P, (P == 16 ? 2 : 4)) void k_fft_rows(const Params p) {
    constexpr int kRowTile = 1 << LT;
    constexpr int T = kRowTile / P;
    extern __shared__ double2 sm[];
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int g = blockIdx.x;
    const int64_t n1 = int64_t(1) << p.log_n1;
    const int n2 = 1 << p.log_n2;            // = tile / 2
    const int64_t m = n1 << p.log_n2;
    double2 * __restrict__ work = p.work + (int64_t)b * m;
    const int64_t r0 = (g == 0) ? 0 : g;
    const int64_t r1 = (g == 0) ? (n1 >> 1) : (n1 - g);
    PHASE_ENTRY;
    double2 v[P];
    // tile element e = 2 k2 + r
#pragma unroll
    for (int k = 0; k < P; ++k) {
        const int e = tid + k * T;
        const int64_t rr = (e & 1) ? r1 : r0;
        v[k] = (p.stream_hint & 16) ? load_nt(work + (rr << p.log_n2) + (e >> 1)) : work[(rr << p.log_n2) + (e >> 1)];
    }
    const auto kt = KTabSel<TLDS>::make(p, p.per_det ? (int64_t)(p.det0 + b) : 0, reinterpret_cast<char *>(sm + kRowTile),
                                        tid, T);
    PHASE_WAIT_LOADS;
    PHASE_SINCE_ENTRY(14);      // kernel entry -> tile and kernel tables landed
    PHASE_DECL;
    PHASE_MARK(0);
    tile_fft_t<LT, P>(v, sm, tid, p.log_n2, p.tb.wtile, tid, tid);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < P; ++k) sm[sw(tid + k * T)] = v[k];
    __syncthreads();
    PHASE_MARK(1);

    // bins k and M - k: real-FFT unpacking, kernel, repacking (all factors 1/2 are in p.scale)
    if (g != 0) {
        // every workgroup but the first: pair q = (row k1, element q) with (row N1 - k1, element N2 - 1 - q), bin
        // k = g + N1 q.  A thread's pairs q = tid + T i have bins N1 T = N / P apart, so their unpacking twiddles are
        // w_N^k0 times the CONSTANTS w_P^i: one table look-up per thread, and the loop unrolls completely (the
        // kernel-interval look-ups of all pairs are in flight together).
        constexpr int NP = kRowTile / 2 / T;
        constexpr double c16[8] = {1.0, 0.92387953251128673848, 0.70710678118654752440, 0.38268343236508977173,
                                   0.0, -0.38268343236508977173, -0.70710678118654752440, -0.92387953251128673848};
        constexpr double s16[8] = {0.0, 0.38268343236508977173, 0.70710678118654752440, 0.92387953251128673848,
                                   1.0, 0.92387953251128673848, 0.70710678118654752440, 0.38268343236508977173};
        const int k0 = g + ((int)n1) * tid;
        const int dk = ((int)n1) * T;
        // w_N^k0 = w_N^g w_N^(N1 tid) = w_N^g w_(2 N2)^tid: one factor that is the same for the whole workgroup and one
        // entry of the tile table at a lane-contiguous index, instead of three gathers from the three-level table
        const double2 w0 = cmul(tw_big(p.tb, g), p.tb.wtile[tid << (kLT - LT)]);
        int lo_a[NP], lo_b[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            lo_a[i] = kernel_interval(kt, k0 + dk * i);
            lo_b[i] = kernel_interval(kt, (int)m - (k0 + dk * i));
        }
#if defined(TOAST_FFT_PHASE_CLOCK)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::"v"(w0.x), "v"(lo_a[NP - 1]), "v"(lo_b[NP - 1]) : "memory");
        PHASE_MARK(9);      // twiddle + interval look-ups
#endif
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int q = tid + T * i;
            const int k = k0 + dk * i;
            const double2 wk = (i == 0) ? w0 : cmul(w0, make_double2(c16[i * (16 / P)], -s16[i * (16 / P)]));
            pair_update(sm, 2 * q, 2 * (n2 - 1 - q) + 1, wk, kernel_eval(kt, lo_a[i], k),
                        kernel_eval(kt, lo_b[i], (int)m - k), p.deconvolve);
        }
    } else {
        for (int i = 0; i < n2 / T; ++i) {
            const int q = tid + T * i;
            int ea, eb;
            int64_t k;
            if (q < n2 / 2) {
                ea = 2 * q + 1;
                eb = 2 * (n2 - 1 - q) + 1;
                k = (n1 >> 1) + n1 * q;
            } else {
                const int qq = q - n2 / 2;
                if (qq == 0) {
                    // DC and Nyquist share element 0: Z[0] = a + i b, X[0] = a + b, X[M] = a - b
                    const double2 z0 = sm[sw(0)];
                    const double2 km = kernel_at(kt, m);
                    double2 ym = apply_kernel(make_double2(2.0 * (z0.x - z0.y), 0.0), km, p.deconvolve);
                    ym.y = 0.0;                                  // Nyquist bin of a real transform is real
                    // Y[0] = 0 (DC removed): Z'[0] = (Y[M], -Y[M]); stored swapped
                    sm[sw(0)] = make_double2(-ym.x, ym.x);
                    ea = eb = n2;                                // bin M / 2 pairs with itself (row 0, k2 = N2 / 2)
                    k = m >> 1;
                } else {
                    ea = 2 * qq;
                    eb = 2 * (n2 - qq);
                    k = n1 * qq;
                }
            }
            pair_update(sm, ea, eb, tw_big(p.tb, k), kernel_at(kt, k), kernel_at(kt, m - k), p.deconvolve);
        }
    }
#if defined(TOAST_FFT_PHASE_CLOCK)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    PHASE_MARK(10);         // evaluation + pair updates (this thread's)
#endif
    __syncthreads();
    PHASE_MARK(2);
#pragma unroll
    for (int k = 0; k < P; ++k) v[k] = sm[sw(tid + k * T)];
    tile_fft_t<LT, P>(v, sm, tid, p.log_n2, p.tb.wtile, tid, tid);
    PHASE_MARK(3);
    const int log_n2_tail = opaque_sgpr(p.log_n2);
    const int tid_tail = opaque_vgpr(tid);
#pragma unroll
    for (int k = 0; k < P; ++k) {
        const int e = tid_tail + k * T;
        const int64_t rr = (e & 1) ? r1 : r0;
        if (p.stream_hint & 32) store_nt(work + (rr << log_n2_tail) + (e >> 1), v[k]);
        else work[(rr << log_n2_tail) + (e >> 1)] = v[k];
    }
    PHASE_WAIT_LOADS;
    PHASE_MARK(4);
    PHASE_SINCE_ENTRY(15);      // the workgroup's whole life
}

// pass 2, row pairs (k1, N1 - k1) with 0 < k1 < N1 / 2: ONE row (N2 = 2048 elements, 32 KB of LDS) in the tile at
// a time, so that four or five workgroups share a CU instead of two (the passes are latency bound:
// profiles/r02_c_pmc_hot_kernels.txt).  Bin k of row k1 pairs with bin M - k = element N2 - 1 - q of row N1 - k1.  The
// last radix-P butterfly of the second row's transform is computed by the MIRRORED thread (u = T - 1 - tid), which
// leaves both members of each of the thread's P pairs in its own registers: unpacking, kernel and repacking need no
// LDS round trip, and the first butterfly of that row's inverse transform starts from the same registers.
template <int P, int WPE, bool TLDS>
__global__ __launch_bounds__((kTile / 2) / P, WPE) void k_fft_rows_split(const Params p) {
    constexpr int LT = kLT - 1;
    constexpr int T = (1 << LT) / P;
    extern __shared__ double2 sm[];
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int g = blockIdx.x + 1;
    const int64_t n1 = int64_t(1) << p.log_n1;
    const int64_t m = n1 << LT;
    double2 * __restrict__ row_a = p.work + (int64_t)b * m + ((int64_t)g << LT);
    double2 * __restrict__ row_b = p.work + (int64_t)b * m + ((n1 - g) << LT);
    double2 va[P], vb[P];
#pragma unroll
    for (int k = 0; k < P; ++k) va[k] = row_a[tid + k * T];
#pragma unroll
    for (int k = 0; k < P; ++k) vb[k] = row_b[tid + k * T];
    const int um = T - 1 - tid;
    const auto kt = KTabSel<TLDS>::make(p, p.per_det ? (int64_t)(p.det0 + b) : 0,
                                        reinterpret_cast<char *>(sm + (kTile / 2)), tid, T);
    tile_fft_t<LT, P>(va, sm, tid, LT, p.tb.wtile, tid, tid);      // va[j] = Z[g + N1 (tid + j T)]
    tile_fft_t<LT, P>(vb, sm, tid, LT, p.tb.wtile, tid, um);       // vb[j] = Z[M - (g + N1 (tid + (P-1-j) T))]

    {
        // a thread's bins k = k0 + N1 T j are N / (2 P) apart: unpacking twiddles w_N^k0 times the constants w_2P^j
        constexpr double c16[8] = {1.0, 0.92387953251128673848, 0.70710678118654752440, 0.38268343236508977173,
                                   0.0, -0.38268343236508977173, -0.70710678118654752440, -0.92387953251128673848};
        constexpr double s16[8] = {0.0, 0.38268343236508977173, 0.70710678118654752440, 0.92387953251128673848,
                                   1.0, 0.92387953251128673848, 0.70710678118654752440, 0.38268343236508977173};
        static_assert(P == 8 || P == 4, "k_fft_rows_split: 4 or 8 points per thread");
        const int k0 = g + ((int)n1) * tid;
        const int dk = ((int)n1) * T;
        const double2 w0 = tw_big(p.tb, k0);
        int lo_a[P], lo_b[P];
#pragma unroll
        for (int j = 0; j < P; ++j) {
            lo_a[j] = kernel_interval(kt, k0 + dk * j);
            lo_b[j] = kernel_interval(kt, (int)m - (k0 + dk * j));
        }
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int k = k0 + dk * j;
            const double2 wk = (j == 0) ? w0 : cmul(w0, make_double2(c16[j * (8 / P)], -s16[j * (8 / P)]));
            pair_update_reg(va[j], vb[P - 1 - j], false, wk, kernel_eval(kt, lo_a[j], k),
                            kernel_eval(kt, lo_b[j], (int)m - k), p.deconvolve);
        }
    }
    tile_fft_t<LT, P>(vb, sm, tid, LT, p.tb.wtile, um, tid);
#pragma unroll
    for (int k = 0; k < P; ++k) row_b[tid + k * T] = vb[k];
    tile_fft_t<LT, P>(va, sm, tid, LT, p.tb.wtile, tid, tid);
#pragma unroll
    for (int k = 0; k < P; ++k) row_a[tid + k * T] = va[k];
}

// interval of the kernel's piecewise cubics at bin q N1 (frequency q N1 fstep), q = 0 .. N2 + 1
// (threads n_block .. n_block + N1 - 1: the interval at every bin of the first block, hint0)
__global__ void k_knot_hint(const double * __restrict__ knots, int n_knot, double fstep, int log_n1, int64_t n_block,
                            int32_t * __restrict__ hint, uint16_t * __restrict__ hint16, int32_t * __restrict__ hint0) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n1 = int64_t(1) << log_n1;
    if (i >= n_block + n1) return;
    const bool first = i >= n_block;
    const double x = (double)(first ? (i - n_block) : (i << log_n1)) * fstep;
    int lo = 0, hi = n_knot - 2;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (knots[mid] <= x) {
            lo = mid;
        } else {
            hi = mid - 1;
        }
    }
    if (first) {
        hint0[i - n_block] = lo;
        return;
    }
    hint[i] = lo;
    hint16[i] = (uint16_t)(lo < 65535 ? lo : 65535);
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct Plan {
    double2 * tables = nullptr;   // wtile | t0 | t1 | t2
    int64_t n2_entries = 0;
    // pass 1 tile orders per (n_samp, n_buffer, n_reflect, tiles, columns per tile): device table or nullptr (none found)
    std::map<std::array<int64_t, 5>, int32_t *> orders;
    std::map<std::array<int64_t, 5>, int32_t *> chains;    // pass 1 of fft_reg.hip: chain_tile_order
};

std::mutex g_mutex;
std::map<std::pair<int, int64_t>, Plan> g_plans;

static void fill_twiddle(std::vector<double2> & out, size_t at, int64_t e, int64_t n) {
    // w_n^e = exp(-2 pi i e / n) in extended precision, rounded once
    const long double a = 2.0L * 3.14159265358979323846264338327950288L * (long double)e / (long double)n;
    out[at] = make_double2((double)cosl(a), (double)(-sinl(a)));
}

static Plan & get_plan(int64_t n_fft, hipStream_t st) {
    int dev = 0;
    TH_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_mutex);
    auto key = std::make_pair(dev, n_fft);
    auto it = g_plans.find(key);
    if (it != g_plans.end()) return it->second;
    Plan pl;
    pl.n2_entries = (n_fft >> 14) > 0 ? (n_fft >> 14) : 1;
    std::vector<double2> h((size_t)kTile + 256 + (size_t)pl.n2_entries + 144 + 144);
    for (int64_t e = 0; e < kTile; ++e) fill_twiddle(h, (size_t)e, e, kTile);
    for (int64_t e = 0; e < 128; ++e) fill_twiddle(h, (size_t)kTile + e, e, n_fft);
    for (int64_t e = 0; e < 128; ++e) fill_twiddle(h, (size_t)kTile + 128 + e, (e << 7) % n_fft, n_fft);
    for (int64_t e = 0; e < pl.n2_entries; ++e) fill_twiddle(h, (size_t)kTile + 256 + e, (e << 14) % n_fft, n_fft);
    // fft_reg.hip, row transform 2048 = 16 x 8 x 16: w_2048^u (u < 128) after the first stage, w_2048^(16 h) (h < 16) after
    // the second -- the entries of the tile table those stages would gather, packed for a copy into LDS
    for (int64_t e = 0; e < 128; ++e) h[(size_t)kTile + 256 + (size_t)pl.n2_entries + e] = h[(size_t)(2 * e)];
    for (int64_t e = 0; e < 16; ++e) h[(size_t)kTile + 256 + (size_t)pl.n2_entries + 128 + e] = h[(size_t)(32 * e)];
    // fft_reg.hip, column transform of length N1 = M / 2048 when that is 512, 1024 or 2048
    {
        int log_m = 0;
        while ((int64_t(1) << log_m) < n_fft / 2) ++log_m;
        const int log_n1 = log_m - (kLT - 1);
        if (log_n1 >= 9 && log_n1 <= 11) {
            int ex[144];
            const int n = cols_reg_twiddles(log_n1, ex);
            for (int i = 0; i < n; ++i) h[(size_t)kTile + 256 + (size_t)pl.n2_entries + 144 + i] = h[(size_t)ex[i] << (kLT - log_n1)];
        }
    }
    void * d = nullptr;
    TH_HIP(hipMalloc(&d, h.size() * sizeof(double2)));
    copy_to_device(d, h.data(), h.size() * sizeof(double2), st);
    pl.tables = static_cast<double2 *>(d);
    static bool attr_set = false;
    if (!attr_set) {
        const int lds = kTile * sizeof(double2);
        const int lds_tab = lds + kTabLdsMax;     // the row pass keeps the kernel tables behind the tile
        auto set = [](const void * fn, int bytes) {
            TH_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        };
        set(reinterpret_cast<const void *>(&k_fft_cols<kLT, 16, false>), lds);
        set(reinterpret_cast<const void *>(&k_fft_cols<kLT, 16, true>), lds);
        set(reinterpret_cast<const void *>(&k_fft_cols<kLT, 8, false>), lds);
        set(reinterpret_cast<const void *>(&k_fft_cols<kLT, 8, true>), lds);
        set(reinterpret_cast<const void *>(&k_fft_rows<kLT, 16, false>), lds);
        set(reinterpret_cast<const void *>(&k_fft_rows<kLT, 8, false>), lds);
        set(reinterpret_cast<const void *>(&k_fft_rows<kLT, 16, true>), lds_tab);
        set(reinterpret_cast<const void *>(&k_fft_rows<kLT, 8, true>), lds_tab);
        set(reinterpret_cast<const void *>(&k_fft_rows<kLT - 1, 8, false>), lds / 2);
        set(reinterpret_cast<const void *>(&k_fft_rows<kLT - 1, 8, true>), lds / 2 + kTabLdsHalf);
        attr_set = true;
    }
    return g_plans.emplace(key, pl).first->second;
}


// Pass 1 reads every timestream line up to three times: directly, and as the left and the right mirror image of the
// padded series (set_rfft_input, src/toast/fft.py:163-188) -- by three DIFFERENT column tiles.  Which tiles share lines
// is fixed by n_buffer and n_samp modulo the row length: the tiles fall into classes (cfg-3: 8 classes of 32 of the 256
// tiles).  If the classes can be dealt to the 8 XCDs evenly, every class runs on one XCD (workgroup -> XCD is round robin
// over the linear workgroup index) and the second and third read of a line can hit that XCD's L2 -- provided the tile's
// own output does not sweep it out first (non-temporal stores, Params::stream_hint bit 0).
static std::vector<int32_t> mirror_tile_order(int64_t n_samp, int64_t n_buffer, int64_t n_reflect, int64_t n_tiles,
                                              int64_t cols_per_tile) {
    std::vector<int32_t> none;
    if (n_reflect <= 0 || n_tiles < 8 || (n_tiles % 8) != 0) return none;
    const int64_t piece = 2 * cols_per_tile;             // reals of one row in a tile
    const int64_t row = piece * n_tiles;                 // reals per row of the padded series
    if (row % 16 != 0) return none;
    const int64_t n_line = row / 16;                     // 128-byte line columns
    auto fdiv = [](int64_t a, int64_t b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); };
    auto col = [&](int64_t src) { return ((fdiv(src, 16) % n_line) + n_line) % n_line; };
    // union-find over tiles (0 .. n_tiles) and line columns (n_tiles .. n_tiles + n_line)
    std::vector<int32_t> parent((size_t)(n_tiles + n_line));
    for (size_t i = 0; i < parent.size(); ++i) parent[i] = (int32_t)i;
    std::function<int32_t(int32_t)> find = [&](int32_t x) {
        while (parent[x] != x) {
            parent[x] = parent[parent[x]];
            x = parent[x];
        }
        return x;
    };
    auto unite = [&](int64_t a, int64_t b) { parent[find((int32_t)a)] = find((int32_t)b); };
    std::vector<std::vector<int32_t>> lines_of((size_t)n_tiles);
    for (int64_t c = 0; c < n_tiles; ++c) {
        for (int64_t x : {c * piece, c * piece + piece - 1}) {          // first and last real of the piece
            const int64_t s = x - n_buffer;                               // (row offsets are multiples of `row`: same column)
            for (int64_t src : {s, -1 - s, 2 * n_samp - 1 - s}) {
                const int64_t l = col(src);
                unite(c, n_tiles + l);
                lines_of[(size_t)c].push_back((int32_t)l);
            }
        }
    }
    std::map<int32_t, std::vector<int32_t>> classes;
    for (int64_t c = 0; c < n_tiles; ++c) classes[find((int32_t)c)].push_back((int32_t)c);
    if (classes.size() < 8) return none;
    // largest class first into the emptiest of the 8 bins; every bin must end with n_tiles / 8 tiles
    std::vector<std::vector<int32_t>> cls;
    for (auto & kv : classes) cls.push_back(kv.second);
    std::stable_sort(cls.begin(), cls.end(), [](const std::vector<int32_t> & a, const std::vector<int32_t> & b) { return a.size() > b.size(); });
    std::vector<std::vector<int32_t>> bins(8);
    for (auto & c : cls) {
        size_t best = 0;
        for (size_t b = 1; b < 8; ++b) {
            if (bins[b].size() < bins[best].size()) best = b;
        }
        // inside a class: tiles that share a line next to each other (sorted by their smallest line column)
        std::stable_sort(c.begin(), c.end(), [&](int32_t a, int32_t b) {
            return *std::min_element(lines_of[(size_t)a].begin(), lines_of[(size_t)a].end()) <
                   *std::min_element(lines_of[(size_t)b].begin(), lines_of[(size_t)b].end());
        });
        bins[best].insert(bins[best].end(), c.begin(), c.end());
    }
    for (auto & b : bins) {
        if ((int64_t)b.size() != n_tiles / 8) return none;
    }
    std::vector<int32_t> order((size_t)n_tiles);
    for (int64_t b = 0; b < n_tiles; ++b) order[(size_t)b] = bins[(size_t)(b & 7)][(size_t)(b >> 3)];
    return order;
}

// Pass 1 of the register kernels: the order in which an XCD takes its column tiles.  Every 128-byte line of the
// timestream is read by three tiles (directly and as the two mirror images of the padded series) and the window table by
// the same tile of every detector; lines written by a tile narrower than 128 bytes are shared with its neighbours.  The
// tiles are grouped into "supers" (the tiles of one 128-byte line column), the supers are chained so that neighbours in
// the chain read the same timestream lines, and the chain is cut into 8 pieces, one per XCD (convolve() then runs each
// piece in segments of a few tiles x a few detectors: the segment's timestream lines and window entries stay in that
// XCD's L2 while they are re-read).  Empty when the geometry does not divide.
static std::vector<int32_t> chain_tile_order(int64_t n_samp, int64_t n_buffer, int64_t n_reflect, int64_t n_tiles,
                                             int64_t cols_per_tile) {
    std::vector<int32_t> none;
    const int64_t piece = 2 * cols_per_tile;             // reals of one row in a tile
    const int64_t tps = (piece >= 16) ? 1 : 16 / piece;  // tiles per super
    if (n_reflect <= 0 || n_tiles % (8 * tps) != 0) return none;
    const int64_t n_super = n_tiles / tps;
    const int64_t span = piece * tps;                    // reals of one row in a super
    const int64_t row = span * n_super;
    if (row % 16 != 0) return none;
    const int64_t n_line = row / 16;
    auto fdiv = [](int64_t a, int64_t b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); };
    auto col = [&](int64_t src) { return ((fdiv(src, 16) % n_line) + n_line) % n_line; };
    std::vector<std::vector<int32_t>> cols_of((size_t)n_super);
    std::vector<std::vector<int32_t>> supers_of((size_t)n_line);
    for (int64_t s = 0; s < n_super; ++s) {
        for (int64_t x : {s * span, s * span + span - 1}) {
            const int64_t rel = x - n_buffer;
            for (int64_t src : {rel, -1 - rel, 2 * n_samp - 1 - rel}) {
                const int32_t c = (int32_t)col(src);
                if (std::find(cols_of[(size_t)s].begin(), cols_of[(size_t)s].end(), c) == cols_of[(size_t)s].end()) {
                    cols_of[(size_t)s].push_back(c);
                    supers_of[(size_t)c].push_back((int32_t)s);
                }
            }
        }
    }
    std::vector<char> seen((size_t)n_super, 0);
    std::vector<int32_t> chain;
    chain.reserve((size_t)n_super);
    int64_t next_free = 0;
    auto neighbour = [&](int32_t s) -> int32_t {
        for (int32_t c : cols_of[(size_t)s]) {
            for (int32_t t : supers_of[(size_t)c]) {
                if (!seen[(size_t)t]) return t;
            }
        }
        return -1;
    };
    while ((int64_t)chain.size() < n_super) {
        int32_t cur = -1;
        for (size_t back = 0; back < 3 && back < chain.size() && cur < 0; ++back) cur = neighbour(chain[chain.size() - 1 - back]);
        if (cur < 0) {
            while (seen[(size_t)next_free]) ++next_free;
            cur = (int32_t)next_free;
        }
        seen[(size_t)cur] = 1;
        chain.push_back(cur);
    }
    std::vector<int32_t> order;
    order.reserve((size_t)n_tiles);
    for (int32_t s : chain) {
        for (int64_t t = 0; t < tps; ++t) order.push_back((int32_t)(s * tps + t));
    }
    return order;
}

int chain_tile_order_host(int64_t n_samp, int64_t n_buffer, int64_t n_reflect, int64_t n_tiles, int64_t cols_per_tile,
                          int32_t * order) {
    const std::vector<int32_t> o = chain_tile_order(n_samp, n_buffer, n_reflect, n_tiles, cols_per_tile);
    if (order != nullptr) {
        for (size_t i = 0; i < o.size(); ++i) order[i] = o[i];
    }
    return (int)o.size();
}

int mirror_tile_order_host(int64_t n_samp, int64_t n_buffer, int64_t n_reflect, int64_t n_tiles, int64_t cols_per_tile,
                           int32_t * order) {
    const std::vector<int32_t> o = mirror_tile_order(n_samp, n_buffer, n_reflect, n_tiles, cols_per_tile);
    if (order != nullptr) {
        for (size_t i = 0; i < o.size(); ++i) order[i] = o[i];
    }
    return (int)o.size();
}

static const int32_t * tile_order_for(Plan & pl, int64_t n_samp, int64_t n_buffer, int64_t n_reflect, int64_t n_tiles,
                                      int64_t cols_per_tile, hipStream_t st) {
    std::lock_guard<std::mutex> lock(g_mutex);
    const std::array<int64_t, 5> key = {n_samp, n_buffer, n_reflect, n_tiles, cols_per_tile};
    auto it = pl.orders.find(key);
    if (it != pl.orders.end()) return it->second;
    const std::vector<int32_t> order = mirror_tile_order(n_samp, n_buffer, n_reflect, n_tiles, cols_per_tile);
    int32_t * d = nullptr;
    if (!order.empty()) {
        TH_HIP(hipMalloc(reinterpret_cast<void **>(&d), order.size() * sizeof(int32_t)));
        copy_to_device(d, order.data(), order.size() * sizeof(int32_t), st);
    }
    pl.orders[key] = d;
    return d;
}

static const int32_t * chain_order_for(Plan & pl, int64_t n_samp, int64_t n_buffer, int64_t n_reflect, int64_t n_tiles,
                                       int64_t cols_per_tile, hipStream_t st) {
    std::lock_guard<std::mutex> lock(g_mutex);
    const std::array<int64_t, 5> key = {n_samp, n_buffer, n_reflect, n_tiles, cols_per_tile};
    auto it = pl.chains.find(key);
    if (it != pl.chains.end()) return it->second;
    const std::vector<int32_t> order = chain_tile_order(n_samp, n_buffer, n_reflect, n_tiles, cols_per_tile);
    int32_t * d = nullptr;
    if (!order.empty()) {
        TH_HIP(hipMalloc(reinterpret_cast<void **>(&d), order.size() * sizeof(int32_t)));
        copy_to_device(d, order.data(), order.size() * sizeof(int32_t), st);
    }
    pl.chains[key] = d;
    return d;
}

namespace {
// points per thread of (row pass, forward column pass, inverse column pass); measured best at cfg-3:
// 8 / 8 / 8 (profiles/r02_f_fft_points.txt; the inverse column pass moved from 16 to 8 with the XCD-aware tile order)
int g_points[3] = {0, 0, 0};
void read_points() {
    if (g_points[0] != 0) return;
    g_points[0] = 8;
    g_points[1] = 8;
    g_points[2] = 8;
    const char * e = std::getenv("TOAST_HIP_FFT_POINTS");
    if (e != nullptr) {
        int a = 0, b = 0, c = 0;
        if (std::sscanf(e, "%d,%d,%d", &a, &b, &c) == 3) {
            g_points[0] = (a == 16) ? 16 : 8;
            g_points[1] = (b == 16) ? 16 : 8;
            g_points[2] = (c == 16) ? 16 : 8;
        }
    }
}
}  // namespace
int points_of(int pass) {
    read_points();
    return g_points[pass];
}
// row pass: 2 (default) = tile in registers, 16 points per lane: a row in two waves, three workgroups per CU
// (k_fft_rows_reg16, fft_reg.hip); 3 = 32 points per lane, one row per wave (k_fft_rows_reg); 0 = row pair per 64 KB LDS
// tile (k_fft_rows); 1 = one row per 32 KB tile (k_fft_rows_split: experiment; needs more registers than four
// workgroups per CU leave).  TOAST_HIP_FFT_ROWS=reg|reg32|lds|split / toast_hip_fft_rows_split(mode)
namespace {
int g_rows_split = -1;
}
int rows_split() {
    if (g_rows_split < 0) {
        const char * e = std::getenv("TOAST_HIP_FFT_ROWS");
        const std::string v = (e != nullptr) ? std::string(e) : std::string();
        g_rows_split = (v == "split") ? 1 : (v == "lds") ? 0 : (v == "reg32") ? 3 : 2;
    }
    return g_rows_split;
}
void set_rows_split(int split) { g_rows_split = (split >= 1 && split <= 3) ? split : 0; }
namespace {
int g_rows_n2 = -1;
}
int rows_n2() {
    if (g_rows_n2 < 0) {
        const char * e = std::getenv("TOAST_HIP_FFT_N2");
        g_rows_n2 = (e != nullptr && std::atoi(e) == 1024) ? 1024 : 2048;
    }
    return g_rows_n2;
}
void set_rows_n2(int n2) { g_rows_n2 = (n2 == 1024) ? 1024 : 2048; }
void set_points(int rows, int cols_fwd, int cols_inv) {
    g_points[0] = 0;
    read_points();       // defaults
    if (rows == 8 || rows == 16) g_points[0] = rows;
    if (cols_fwd == 8 || cols_fwd == 16) g_points[1] = cols_fwd;
    if (cols_inv == 8 || cols_inv == 16) g_points[2] = cols_inv;
}

bool supported(int64_t n_fft) {
    // M = n_fft / 2 = N1 N2 with N2 = kTile / 2 and N1 >= 2
    return n_fft >= 2 * kTile && n_fft <= (int64_t(1) << 24);
}

// Bytes the three passes move per timestream sample (bench / DESIGN accounting).
double pipeline_bytes_per_sample(int64_t n_samp, int64_t n_fft) {
    return 16.0 + 32.0 * (double)n_fft / (double)n_samp;
}

void convolve(double * d_tod, const int32_t * d_idx, int64_t n_det, int64_t n_samp, int64_t n_fft,
              int64_t n_buffer, int64_t n_reflect, double fstep, const double * d_knots, int64_t n_knot,
              const double * d_mag, const double * d_ang, int per_det, int deconvolve, const double * d_apod,
              int64_t max_batch, hipStream_t st) {
    Plan & pl = get_plan(n_fft, st);
    const int64_t m = n_fft / 2;
    int log_m = 0;
    while ((int64_t(1) << log_m) < m) ++log_m;
    Params p;
    p.tod = d_tod;
    p.d_idx = d_idx;
    p.apod = d_apod;
    p.n_samp = n_samp;
    p.n_fft = n_fft;
    p.n_buffer = n_buffer;
    p.n_reflect = n_reflect;
    // Rows of N2 = 2048 (64 KB pair tiles) or, for M <= 2^20, of N2 = 1024 (32 KB pair tiles: four workgroups per CU in
    // the row pass; the column passes then move 64-byte instead of 128-byte pieces).  TOAST_HIP_FFT_N2=1024|2048.
    p.log_n2 = kLT - 1;
    if (rows_n2() == 1024 && log_m >= 11 && log_m <= 20) p.log_n2 = kLT - 2;
    p.log_n1 = log_m - p.log_n2;
    p.tb.wtile = pl.tables;
    p.tb.t0 = pl.tables + kTile;
    p.tb.t1 = pl.tables + kTile + 128;
    p.tb.t2 = pl.tables + kTile + 256;
    p.wrow = pl.tables + kTile + 256 + pl.n2_entries;
    p.wcol = p.wrow + 144;
    p.knots = d_knots;
    p.n_knot = (int)n_knot;
    p.mag_coef = d_mag;
    p.ang_coef = d_ang;
    p.per_det = per_det;
    p.deconvolve = deconvolve;
    p.fstep = fstep;
    p.aligned = ((n_buffer % 2) == 0 && (n_samp % 2) == 0 && (n_reflect % 2) == 0 &&
                 (reinterpret_cast<uintptr_t>(d_tod) % 16) == 0 && (reinterpret_cast<uintptr_t>(d_apod) % 16) == 0)
                    ? 1 : 0;
    {
        static int xo = -1;
        if (xo < 0) {
            const char * e = std::getenv("TOAST_HIP_FFT_XCD");
            // 0: tiles in launch order; 1: a contiguous eighth of the columns per XCD; 2 (default): pass 1 with the mirror
            // partners of a tile on one XCD where the geometry allows it (else as 1), passes 3 as 1
            xo = (e != nullptr && e[0] == '0') ? 0 : (e != nullptr && e[0] == '1') ? 1 : 2;
        }
        p.xcd_order = xo;
        static int sh = -1;
        if (sh < 0) {
            const char * e = std::getenv("TOAST_HIP_FFT_STREAM_HINT");
            // default 1: pass 1 writes the work array with non-temporal stores (profiles/r04_c section 5: the window table
            // and the mirrored re-reads of the timestream then stay in L2; every other bit measured neutral or worse)
            sh = (e != nullptr && e[0] != '\0') ? std::atoi(e) : 1;
        }
        p.stream_hint = sh;
        // ... but not with two columns per tile (N1 = 2048, n_fft 2^23): a non-temporal store sends every 32-byte piece to
        // memory on its own, 61 -> 90 ms per call; with four columns (64-byte pieces, 2^22) it still gains 1.5 %
        // (profiles/r04_c section 6)
        static const bool forced = std::getenv("TOAST_HIP_FFT_STREAM_HINT") != nullptr;
        // (pass 1 of the register kernels: plain stores at every length -- 49.4 against 51.6 ms per call at 2^23, 44.5
        // against 47.2 at 2^22, profiles/r05_a)
        if (!forced && (cols_reg_for(p) || (kLT - p.log_n1) < 2)) p.stream_hint &= ~1;
    }
    // unnormalised inverse of length M on un-halved packing factors: 1 / (4 M), a power of two
    p.scale = 1.0 / (4.0 * (double)m);

    int64_t batch = (max_batch > 0) ? max_batch : 512;
    const int64_t cap = (int64_t)((size_t(8) << 30) / ((size_t)m * sizeof(double2)));
    if (batch > cap) batch = cap > 0 ? cap : 1;
    if (batch > n_det) batch = n_det;
    const int64_t n_hint = (int64_t(1) << p.log_n2) + 2;       // bins q N1, q = 0 .. N2 + 1
    const int64_t n_hint0 = int64_t(1) << p.log_n1;             // every bin of the first block
    const size_t hint_bytes = ((size_t)n_hint * (sizeof(int32_t) + sizeof(uint16_t)) + (size_t)n_hint0 * sizeof(int32_t) +
                               255 + 8) & ~size_t(255);
    // row pass: knots, this detector's cubics and 16-bit hints in LDS when they fit (layout: KTabSel<true>::make)
    const size_t tab_bytes = (((size_t)n_knot + (size_t)4 * (n_knot - 1) * (d_ang ? 2 : 1)) * sizeof(double) +
                              (size_t)n_hint * sizeof(uint16_t) + 15) & ~size_t(15);
    // ... which the register row pass copies from one packed block per kernel (k_pack_tables)
    const int64_t n_kern = per_det ? n_det : 1;
    const size_t blob_bytes = (tab_bytes <= (size_t)kTabLdsMax) ? (((size_t)n_kern * tab_bytes + 255) & ~size_t(255)) : 0;
    char * scratch = (char *)Manager::get().scratch(Manager::kScratchFftWork,
                                                    hint_bytes + blob_bytes + (size_t)batch * m * sizeof(double2), st);
    int32_t * d_hint = (int32_t *)scratch;
    int32_t * d_hint0 = d_hint + n_hint;
    uint16_t * d_hint16 = (uint16_t *)(d_hint0 + n_hint0);
    p.knot_hint = d_hint;
    p.knot_hint16 = d_hint16;
    p.knot_hint0 = d_hint0;
    p.tab_blob = blob_bytes ? scratch + hint_bytes : nullptr;
    p.tab_bytes = blob_bytes ? (int)tab_bytes : 0;
    p.tab_copy_bytes = blob_bytes ? (int)((((size_t)n_knot + (size_t)4 * (n_knot - 1) * (d_ang ? 2 : 1)) * sizeof(double) + 15) & ~size_t(15)) : 0;
    p.work = (double2 *)(scratch + hint_bytes + blob_bytes);
    hipLaunchKernelGGL(k_knot_hint, dim3((unsigned)((n_hint + n_hint0 + 255) / 256)), dim3(256), 0, st, d_knots,
                       (int)n_knot, fstep, p.log_n1, n_hint, d_hint, d_hint16, d_hint0);
    if (blob_bytes && rows_split() >= 2) pack_tables(p, scratch + hint_bytes, n_kern, st);
    static int tab_lds_env = -1;
    if (tab_lds_env < 0) {
        const char * e = std::getenv("TOAST_HIP_FFT_TABLES");      // "global": experiment switch
        tab_lds_env = (e != nullptr && std::string(e) == "global") ? 0 : 1;
    }
    const bool half_rows = p.log_n2 == kLT - 2;
    const bool tab_lds = tab_lds_env && n_knot < 65535 && tab_bytes <= (size_t)(half_rows ? kTabLdsHalf : kTabLdsMax);
    const size_t lds = kTile * sizeof(double2);
    const size_t lds_rows = lds + (tab_lds ? tab_bytes : 0);
    const size_t lds_split = lds / 2 + (tab_lds ? tab_bytes : 0);
    const bool cols_reg = cols_reg_for(p);
    const int log_c = cols_reg ? cols_reg_log_c(p.log_n1) : (kLT - p.log_n1);               // columns per tile
    const unsigned n_col_tiles = (unsigned)(int64_t(1) << (p.log_n2 - log_c));               // N2 / C
    p.tile_order = nullptr;
    if (p.xcd_order == 2 && p.aligned) {
        p.tile_order = tile_order_for(pl, n_samp, n_buffer, n_reflect, (int64_t)n_col_tiles, int64_t(1) << log_c, st);
    }
    // pass 1 of the register kernels: segments of the tile chain x groups of detectors per XCD (TOAST_HIP_FFT_FWD_SEG="k,g",
    // "0": the two-dimensional grid with tile_order)
    p.fwd_seq = nullptr;
    p.fwd_k = p.fwd_g = p.fwd_tiles = p.fwd_dets = 0;
    if (cols_reg) {
        static int seg_k = -1, seg_g = -1;
        if (seg_k < 0) {
            seg_k = 8;          // eight tiles (four 128-byte line columns at N1 = 2048) x all detectors of the batch: the tiles
            seg_g = 1 << 20;    // that run side by side on an XCD read the same window entries and share written lines
                                // (4 before pass 1 was freed of its scratch traffic; now 46.1-46.4 against 46.9-47.2 ms at 2^23)
            const char * e = std::getenv("TOAST_HIP_FFT_FWD_SEG");
            if (e != nullptr) {
                int a = 0, b = 0;
                const int n = std::sscanf(e, "%d,%d", &a, &b);
                if (n == 2 && a > 0 && b > 0) {
                    seg_k = a;
                    seg_g = b;
                } else if (n >= 1 && a == 0) {
                    seg_k = 0;
                }
            }
        }
        const int64_t tpx = (int64_t)n_col_tiles / 8;
        if (seg_k > 0 && (n_col_tiles % 8u) == 0u && tpx % seg_k == 0) {
            p.fwd_seq = chain_order_for(pl, n_samp, n_buffer, n_reflect, (int64_t)n_col_tiles, int64_t(1) << log_c, st);
            p.fwd_k = seg_k;
            p.fwd_g = seg_g;
            p.fwd_tiles = (int)n_col_tiles;
        }
    }
    const unsigned n_row_tiles = (unsigned)((int64_t(1) << p.log_n1) / 2);                 // N1 / 2
    for (int64_t det0 = 0; det0 < n_det; det0 += batch) {
        const int64_t nb = (n_det - det0 < batch) ? (n_det - det0) : batch;
        p.det0 = (int)det0;
        // points per thread of each pass: TOAST_HIP_FFT_POINTS="rows,cols_fwd,cols_inv" / toast_hip_fft_points
        if (cols_reg) {
            p.fwd_dets = (int)nb;
            launch_cols_reg(p, false, (unsigned)nb, st);
        } else if (points_of(1) == 8) {
            hipLaunchKernelGGL((k_fft_cols<kLT, 8, false>), dim3(n_col_tiles, (unsigned)nb), dim3(kTile / 8), lds, st, p);
        } else {
            hipLaunchKernelGGL((k_fft_cols<kLT, 16, false>), dim3(n_col_tiles, (unsigned)nb), dim3(kTile / 16), lds, st, p);
        }
        const bool reg_rows = rows_split() >= 2 && !half_rows && rows_reg_supported(p);
        const bool split = rows_split() == 1 && !half_rows;
        // split / registers: rows 0 and N1 / 2 only (self-paired)
        const dim3 g_pair((split || reg_rows) ? 1 : n_row_tiles, (unsigned)nb);
        if (half_rows) {
            const size_t lds_half = lds / 2 + (tab_lds ? tab_bytes : 0);
            if (tab_lds) {
                hipLaunchKernelGGL((k_fft_rows<kLT - 1, 8, true>), g_pair, dim3(kTile / 16), lds_half, st, p);
            } else {
                hipLaunchKernelGGL((k_fft_rows<kLT - 1, 8, false>), g_pair, dim3(kTile / 16), lds_half, st, p);
            }
        } else if (split || points_of(0) == 8) {
            if (tab_lds) {
                hipLaunchKernelGGL((k_fft_rows<kLT, 8, true>), g_pair, dim3(kTile / 8), lds_rows, st, p);
            } else {
                hipLaunchKernelGGL((k_fft_rows<kLT, 8, false>), g_pair, dim3(kTile / 8), lds_rows, st, p);
            }
        } else if (tab_lds) {
            hipLaunchKernelGGL((k_fft_rows<kLT, 16, true>), g_pair, dim3(kTile / 16), lds_rows, st, p);
        } else {
            hipLaunchKernelGGL((k_fft_rows<kLT, 16, false>), g_pair, dim3(kTile / 16), lds_rows, st, p);
        }
        if (reg_rows) {
            if (rows_split() == 2) launch_rows_reg16(p, (unsigned)nb, tab_lds, tab_bytes, st);
            else launch_rows_reg(p, (unsigned)nb, tab_lds, tab_bytes, st);
        }
        if (split && n_row_tiles > 1) {
            const dim3 gr(n_row_tiles - 1, (unsigned)nb), bl(kTile / 16);
            if (tab_lds) {
                hipLaunchKernelGGL((k_fft_rows_split<8, 2, true>), gr, bl, lds_split, st, p);
            } else {
                hipLaunchKernelGGL((k_fft_rows_split<8, 2, false>), gr, bl, lds_split, st, p);
            }
        }
        if (cols_reg || cols_inv_reg_for(p)) {
            launch_cols_reg(p, true, (unsigned)nb, st);
        } else if (points_of(2) == 8) {
            hipLaunchKernelGGL((k_fft_cols<kLT, 8, true>), dim3(n_col_tiles, (unsigned)nb), dim3(kTile / 8), lds, st, p);
        } else {
            hipLaunchKernelGGL((k_fft_cols<kLT, 16, true>), dim3(n_col_tiles, (unsigned)nb), dim3(kTile / 16), lds, st, p);
        }
        TH_HIP(hipGetLastError());
    }
}

#if defined(TOAST_FFT_PHASE_CLOCK)
void phase_ticks(unsigned long long * out, int reset) {
    TH_HIP(hipDeviceSynchronize());
    TH_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase_ticks), 16 * sizeof(unsigned long long)));
    if (reset) {
        unsigned long long z[16] = {0};
        TH_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_phase_ticks), z, sizeof(z)));
    }
}
#endif

}  // namespace fused_fft
}  // namespace toast_hip

#if defined(TOAST_FFT_PHASE_CLOCK)
extern "C" void toast_hip_fft_phase_ticks(unsigned long long * out, int reset) {
    toast_hip::fused_fft::phase_ticks(out, reset);
}
#endif
