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

namespace toast_hip {
namespace fused_fft {

constexpr int kLT = 12;             // log2 of the tile (4096 complex doubles = 64 KB of LDS)
constexpr int kTile = 1 << kLT;
// LDS budget of the row pass for the kernel tables (two 64 KB tiles + tables per CU: 2 x (64 + 15) KB <= 160 KB)
constexpr int kTabLdsMax = 15 * 1024;
// ... and with 32 KB tiles (N2 = 1024): four workgroups per CU, 4 x (32 + 7.5) KB <= 160 KB
constexpr int kTabLdsHalf = 7 * 1024 + 512;
// Every kernel is a template on P = points per thread (kTile / P threads per workgroup):
//   P = 16: 256 threads, radix-16 ends, one LDS round trip fewer per transform, ~250 VGPRs -> 2 waves / SIMD
//   P = 8:  512 threads, radix-8 stages, ~100 VGPRs -> 4 waves / SIMD (two workgroups per CU either way: LDS)

// ------------------------------------------------------------------------------------------
// complex helpers (explicit fma: the library is built with -ffp-contract=off)
// ------------------------------------------------------------------------------------------
typedef double nt_double2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 load_nt(const double2 * p) {
    const nt_double2 v = __builtin_nontemporal_load(reinterpret_cast<const nt_double2 *>(p));
    return make_double2(v.x, v.y);
}
__device__ __forceinline__ void store_nt(double2 * p, double2 v) {
    nt_double2 w;
    w.x = v.x;
    w.y = v.y;
    __builtin_nontemporal_store(w, reinterpret_cast<nt_double2 *>(p));
}

__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
    return make_double2(__builtin_fma(a.x, b.x, -(a.y * b.y)), __builtin_fma(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double2 cconj(double2 a) { return make_double2(a.x, -a.y); }
// a * (-i)
__device__ __forceinline__ double2 mul_mi(double2 a) { return make_double2(a.y, -a.x); }

// LDS position of tile element i: XOR swizzle of the low four bits with the next four -- the
// strided writes of a Stockham stage (stride 16 elements = 256 B) then spread over all banks.
__device__ __forceinline__ int sw(int i) { return i ^ ((i >> 4) & 15); }

// forward DFT of R points in registers, natural order in and out
template <int R>
struct DFT;
template <>
struct DFT<1> {
    static __device__ __forceinline__ void run(double2 *) {}
};
template <>
struct DFT<2> {
    static __device__ __forceinline__ void run(double2 * a) {
        const double2 t = a[0];
        a[0] = cadd(t, a[1]);
        a[1] = csub(t, a[1]);
    }
};
template <int R>
struct DFT {
    static __device__ __forceinline__ void run(double2 * a) {
        constexpr int H = R / 2;
        double2 e[H], o[H];
#pragma unroll
        for (int k = 0; k < H; ++k) {
            e[k] = a[2 * k];
            o[k] = a[2 * k + 1];
        }
        DFT<H>::run(e);
        DFT<H>::run(o);
        // w_R^k = (cos(2 pi k / R), -sin(2 pi k / R)), k < R / 2
        constexpr double c16[8] = {1.0, 0.92387953251128673848, 0.70710678118654752440, 0.38268343236508977173,
                                   0.0, -0.38268343236508977173, -0.70710678118654752440, -0.92387953251128673848};
        constexpr double s16[8] = {0.0, 0.38268343236508977173, 0.70710678118654752440, 0.92387953251128673848,
                                   1.0, 0.92387953251128673848, 0.70710678118654752440, 0.38268343236508977173};
#pragma unroll
        for (int k = 0; k < H; ++k) {
            constexpr int step = 16 / R;
            double2 t;
            if (k == 0) {
                t = o[0];
            } else if (2 * k == H) {
                t = mul_mi(o[k]);
            } else {
                t = cmul(o[k], make_double2(c16[k * step], -s16[k * step]));
            }
            a[k] = cadd(e[k], t);
            a[k + H] = csub(e[k], t);
        }
    }
};

template <int R>
struct Log2;
template <>
struct Log2<2> {
    static constexpr int v = 1;
};
template <>
struct Log2<4> {
    static constexpr int v = 2;
};
template <>
struct Log2<8> {
    static constexpr int v = 3;
};
template <>
struct Log2<16> {
    static constexpr int v = 4;
};

__device__ __forceinline__ int out_idx(int u, int j, int log_s, int log_r) {
    return (u & ((1 << log_s) - 1)) | ((u >> log_s) << (log_s + log_r)) | (j << log_s);
}

// out[j] *= w^(j) for j = 1 .. R-1 with w = base (powers by a product tree of depth <= 4)
template <int R>
__device__ __forceinline__ void apply_powers(double2 * a, double2 w1) {
    if (R >= 2) a[1] = cmul(a[1], w1);
    if (R >= 4) {
        const double2 w2 = cmul(w1, w1);
        const double2 w3 = cmul(w2, w1);
        a[2] = cmul(a[2], w2);
        a[3] = cmul(a[3], w3);
        if (R >= 8) {
            const double2 w4 = cmul(w2, w2);
            const double2 w5 = cmul(w4, w1);
            const double2 w6 = cmul(w3, w3);
            const double2 w7 = cmul(w4, w3);
            a[4] = cmul(a[4], w4);
            a[5] = cmul(a[5], w5);
            a[6] = cmul(a[6], w6);
            a[7] = cmul(a[7], w7);
            if (R >= 16) {
                const double2 w8 = cmul(w4, w4);
                a[8] = cmul(a[8], w8);
                a[9] = cmul(a[9], cmul(w8, w1));
                a[10] = cmul(a[10], cmul(w5, w5));
                a[11] = cmul(a[11], cmul(w8, w3));
                a[12] = cmul(a[12], cmul(w6, w6));
                a[13] = cmul(a[13], cmul(w8, w5));
                a[14] = cmul(a[14], cmul(w7, w7));
                a[15] = cmul(a[15], cmul(w8, w7));
            }
        }
    }
}

// One Stockham stage of radix R on the whole tile of 2^LT elements, LDS -> LDS.  Remaining transform length is
// (tile >> log_s); the stage twiddle w_n^(j p) = w_tile^((p << log_s) * j) is built from w_tile^(u & ~(s-1)), and
// w_tile^e = wtile[e << (kLT - LT)] (the table holds the 4096th roots).
template <int LT, int P, int R>
__device__ __forceinline__ void stage_lds(double2 * sm, int tid, int log_s, bool last,
                                          const double2 * __restrict__ wtile) {
    constexpr int T = (1 << LT) / P;
    constexpr int B = P / R;
    constexpr int Q = (1 << LT) / R;
    constexpr int LR = Log2<R>::v;
    double2 v[B][R];
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const int u = tid + T * b;
#pragma unroll
        for (int k = 0; k < R; ++k) v[b][k] = sm[sw(u + k * Q)];
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const int u = tid + T * b;
        DFT<R>::run(v[b]);
        if (!last) apply_powers<R>(v[b], wtile[((u >> log_s) << log_s) << (kLT - LT)]);
#pragma unroll
        for (int j = 0; j < R; ++j) sm[sw(out_idx(u, j, log_s, LR))] = v[b][j];
    }
    __syncthreads();
}

template <int LT, int P>
__device__ __forceinline__ void stage_lds_any(int r, double2 * sm, int tid, int log_s, bool last,
                                              const double2 * __restrict__ wtile) {
    if (P >= 16 && r == 16) {
        stage_lds<LT, P, (P >= 16 ? 16 : P)>(sm, tid, log_s, last, wtile);
    } else if (r == 8) {
        stage_lds<LT, P, 8>(sm, tid, log_s, last, wtile);
    } else if (r == 4) {
        stage_lds<LT, P, 4>(sm, tid, log_s, last, wtile);
    } else {
        stage_lds<LT, P, 2>(sm, tid, log_s, last, wtile);
    }
}

// Forward FFTs of length n = 2^log_n along the slow axis of the tile of 2^LT elements (tile / n interleaved
// transforms).  In: v[k] = tile element u_in + k * T, out: v[k] = element u_out + k * T, T = tile / P threads;
// u_in / u_out are any permutation of the thread index (u = tid: natural order; the row pass mirrors one of them).
template <int LT, int P>
__device__ __forceinline__ void tile_fft_t(double2 (&v)[P], double2 * sm, int tid, int log_n,
                                           const double2 * __restrict__ wtile, int u_in, int u_out) {
    constexpr int T = (1 << LT) / P;
    constexpr int LP = Log2<P>::v;
    const int log_s0 = LT - log_n;
    if (log_n >= 2 * LP) {
        // plan [P, middle stages, P]: the radix-P ends work straight on the registers
        DFT<P>::run(v);
        apply_powers<P>(v, wtile[((u_in >> log_s0) << log_s0) << (kLT - LT)]);
        __syncthreads();   // earlier readers of the tile are done
#pragma unroll
        for (int j = 0; j < P; ++j) sm[sw(out_idx(u_in, j, log_s0, LP))] = v[j];
        __syncthreads();
        int log_mid = log_n - 2 * LP, log_s = log_s0 + LP;
        while (log_mid > 0) {
            const int lr = log_mid >= LP ? LP : log_mid;
            stage_lds_any<LT, P>(1 << lr, sm, tid, log_s, false, wtile);
            log_s += lr;
            log_mid -= lr;
        }
#pragma unroll
        for (int k = 0; k < P; ++k) v[k] = sm[sw(u_out + k * T)];
        DFT<P>::run(v);
        return;
    }
    // short transforms (only small problems get here): every stage through LDS
    __syncthreads();
#pragma unroll
    for (int k = 0; k < P; ++k) sm[sw(u_in + k * T)] = v[k];
    __syncthreads();
    int log_rem = log_n, log_s = log_s0;
    while (log_rem > 0) {
        const int lr = log_rem >= LP ? LP : log_rem;
        stage_lds_any<LT, P>(1 << lr, sm, tid, log_s, log_rem == lr, wtile);
        log_s += lr;
        log_rem -= lr;
    }
#pragma unroll
    for (int k = 0; k < P; ++k) v[k] = sm[sw(u_out + k * T)];
}

// the 4096-element tile in natural order (column passes, self-paired rows)
template <int P>
__device__ __forceinline__ void tile_fft(double2 (&v)[P], double2 * sm, int tid, int log_n,
                                         const double2 * __restrict__ wtile) {
    tile_fft_t<kLT, P>(v, sm, tid, log_n, wtile, tid, tid);
}

// ------------------------------------------------------------------------------------------
// tables: wtile[e] = w_tile^e (e < kTile); three-level w_N^e = t2[e >> 14] t1[(e >> 7) & 127] t0[e & 127]
// ------------------------------------------------------------------------------------------
struct Tables {
    const double2 * wtile;
    const double2 * t0;
    const double2 * t1;
    const double2 * t2;
};

__device__ __forceinline__ double2 tw_big(const Tables & tb, int64_t e) {
    const double2 a = tb.t2[e >> 14];
    const double2 b = tb.t1[(e >> 7) & 127];
    const double2 c = tb.t0[e & 127];
    return cmul(cmul(a, b), c);
}

struct Params {
    double * tod;                 // [rows, n_samp]
    const int32_t * d_idx;        // row of detector b
    int det0;
    double2 * work;               // [batch, M]
    const double * apod;          // n_reflect
    int64_t n_samp, n_fft, n_buffer, n_reflect;
    int log_n1, log_n2;           // M = N1 N2
    Tables tb;
    // kernel K(f)
    const double * knots;
    int n_knot;
    const double * mag_coef;
    const double * ang_coef;      // nullptr: real kernel
    const int32_t * knot_hint;    // interval index at bin q N1, q = 0 .. N2 + 1
    const uint16_t * knot_hint16; // the same in 16 bits (n_knot < 65536), for the LDS copy
    const int32_t * knot_hint0;   // interval index at every bin j of the FIRST block, j = 0 .. N1 - 1
    int per_det, deconvolve;
    int aligned;                  // pass 1 may use padded_pair
    int xcd_order;                // column passes: contiguous column ranges per XCD
    const int32_t * tile_order;   // pass 1: column tile of workgroup blockIdx.x (mirror partners on one XCD), or nullptr
    int stream_hint;              // bit 0: pass 1 writes the work array with non-temporal stores, bit 1: reads the
                                  // timestream with non-temporal loads (so that the window table stays in L2), bit 2 / 3:
                                  // the same for pass 3's loads / stores, bit 4 / 5: the row pass' loads / stores
    double fstep, scale;
};

// set_rfft_input evaluated for one element of the padded series (src/toast/fft.py:163-188)
__device__ __forceinline__ double padded(const double * __restrict__ row, const double * __restrict__ apod,
                                         int64_t i, int64_t n_samp, int64_t n_buffer, int64_t n_reflect) {
    const int64_t s = i - n_buffer;
    if (s >= 0 && s < n_samp) return row[s];
    if (s < 0 && s >= -n_reflect) {
        const int64_t j = s + n_reflect;
        return row[n_reflect - 1 - j] * apod[j];
    }
    if (s >= n_samp && s < n_samp + n_reflect) {
        const int64_t j = s - n_samp;
        return row[n_samp - 1 - j] * apod[n_reflect - 1 - j];
    }
    return 0.0;
}

// v[k] *= w_M^(k1 j2) for the thread's P tile elements e = tid + T k.  When the tile has at most T columns
// all of them share the column j2 and their rows are k1 = k1_0 + k (T >> log_c): the factors are
// w^(e0) (w^d)^k -- two table look-ups and a product tree instead of P look-ups.
template <int P>
struct ColTw {
    double2 w0, wd;
};
// the table look-ups (independent of the data: issued before the transform whose barriers they could not cross)
template <int LT, int P>
__device__ __forceinline__ ColTw<P> col_twiddles_prepare(const Params & p, int tid, int log_c, int64_t c0) {
    constexpr int T = (1 << LT) / P;
    ColTw<P> tw;
    tw.w0 = make_double2(1.0, 0.0);
    tw.wd = tw.w0;
    if ((1 << log_c) <= T) {
        const int64_t j2 = c0 + (tid & ((1 << log_c) - 1));
        const int64_t k10 = tid >> log_c;
        const int64_t dk = T >> log_c;
        tw.w0 = tw_big(p.tb, 2 * k10 * j2);
        tw.wd = tw_big(p.tb, 2 * dk * j2);
    }
    return tw;
}
template <int LT, int P>
__device__ __forceinline__ void col_twiddles(double2 (&v)[P], const Params & p, int tid, int log_c, int64_t c0,
                                             const ColTw<P> & tw) {
    constexpr int T = (1 << LT) / P;
    if ((1 << log_c) <= T) {
        v[0] = cmul(v[0], tw.w0);
        apply_powers<P>(v, tw.wd);
#pragma unroll
        for (int k = 1; k < P; ++k) v[k] = cmul(v[k], tw.w0);
    } else {
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const int e = tid + k * T;
            const int64_t k1 = e >> log_c;
            const int64_t j2 = c0 + (e & ((1 << log_c) - 1));
            v[k] = cmul(v[k], tw_big(p.tb, 2 * k1 * j2));
        }
    }
}

// the same for the pair (x[s + n_buffer], x[s + n_buffer + 1]), s even, when n_samp and n_reflect are even
// and the row is 16-byte aligned: identical values (one product per element), half the loads.
// BRANCH-FREE: source and window positions are selected arithmetically and both 16-byte loads are issued for every
// point, so that all 2 P loads of a thread are in flight together.  (With one branch per region the compiler waited
// for each mirrored point's two loads before the next point: two thirds of the padded series are mirrored, i.e.
// five or six serialised memory round trips per thread -- the largest part of the forward column pass,
// profiles/r02_g_fft_phase_clocks.txt section 7.)
__device__ __forceinline__ double2 padded_pair(const double * __restrict__ row, const double * __restrict__ apod,
                                               int64_t s, int64_t n_samp, int64_t n_reflect, bool nt = false) {
    const bool direct = (s >= 0) & (s < n_samp);
    const bool left = (s < 0) & (s >= -n_reflect);
    const bool right = (s >= n_samp) & (s < n_samp + n_reflect);
    const int64_t jl = s + n_reflect;             // left mirror: window index j, source n_reflect - 2 - j
    const int64_t jr = s - n_samp;                // right mirror: window index n_reflect - 2 - j, source n_samp - 2 - j
    int64_t src = direct ? s : (left ? n_reflect - 2 - jl : n_samp - 2 - jr);
    int64_t win = left ? jl : n_reflect - 2 - jr;
    if (!(direct | left | right)) src = 0;
    if (!(left | right)) win = 0;
    const double2 r = nt ? load_nt(reinterpret_cast<const double2 *>(row + src))
                         : *reinterpret_cast<const double2 *>(row + src);
    const double2 a = *reinterpret_cast<const double2 *>(apod + win);
    const double2 m = left ? make_double2(r.y * a.x, r.x * a.y) : make_double2(r.y * a.y, r.x * a.x);
    double2 out = direct ? r : m;
    if (!(direct | left | right)) out = make_double2(0.0, 0.0);
    return out;
}

// Experimental build (TOAST_HIP_EXTRA_FLAGS=-DTOAST_FFT_PHASE_CLOCK python -m toast_amd.build --force; tools/exp_fft_phases.py):
// thread 0 of every workgroup adds the 100 MHz wall-clock ticks between phase boundaries to g_phase_ticks.
#if defined(TOAST_FFT_PHASE_CLOCK)
__device__ unsigned long long g_phase_ticks[16];
# define PHASE_ENTRY const unsigned long long ph_e = wall_clock64()
# define PHASE_DECL unsigned long long ph_t = wall_clock64()
# define PHASE_SINCE_ENTRY(i) if (threadIdx.x == 0) atomicAdd(&g_phase_ticks[i], wall_clock64() - ph_e)
# define PHASE_WAIT_LOADS asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
# define PHASE_MARK(i)                                                        \
    do {                                                                      \
        const unsigned long long ph_n = wall_clock64();                       \
        if (threadIdx.x == 0) atomicAdd(&g_phase_ticks[i], ph_n - ph_t);      \
        ph_t = ph_n;                                                          \
    } while (0)
#else
# define PHASE_ENTRY
# define PHASE_DECL
# define PHASE_SINCE_ENTRY(i)
# define PHASE_WAIT_LOADS
# define PHASE_MARK(i)
#endif

// The element offsets of a thread's loads are needed again for its stores at the end of the kernel.  Left alone, the
// compiler keeps the 64-bit offsets alive across the whole transform and, at the 128-VGPR budget of the P = 8 kernels,
// spills one or two of them to scratch.  Re-deriving them from an opaque copy of the shift costs a few integer
// instructions at the end and keeps the kernels free of scratch.
__device__ __forceinline__ int opaque_sgpr(int x) {
    asm volatile("" : "+s"(x));
    return x;
}
__device__ __forceinline__ int opaque_vgpr(int x) {
    asm volatile("" : "+v"(x));
    return x;
}

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

// K(f) at bin k (src/toast/fft.py:190-212): PCHIP piecewise cubics of |K| and arg K
__device__ __forceinline__ double ppoly_at(const double * __restrict__ knots, int n_knot,
                                           const double * __restrict__ coef, int lo, double x) {
    const double * c = coef + 4 * lo;
    const double dx = x - knots[lo];
    return ((c[0] * dx + c[1]) * dx + c[2]) * dx + c[3];
}

// The kernel tables as the row pass sees them: in global memory, or (TLDS) a copy behind the tile in the workgroup's
// LDS.  Measured with phase clocks (tools/exp_fft_phases.py): with the tables in global memory the unpack / multiply /
// repack phase took 42 % of the row pass -- three DEPENDENT cache round trips per bin (hint -> knot -> coefficients)
// at the latency of a memory system that is busy streaming the tiles.
template <typename H>
struct KTab {
    const H * hint;            // interval at bin q N1, q = 0 .. N2 + 1
    const int32_t * hint0;     // interval at every bin of the first block (global memory: one lane per workgroup asks)
    const double * knots;
    const double * mc;         // |K| cubics of this workgroup's detector
    const double * ac;         // arg K cubics, nullptr: real kernel
    int log_n1;
    double fstep;
};

// Interval of bin k in the knot vector.  hint[q] is the interval at bin q N1 (the first bin of element q of every
// row), so the answer lies in [hint[q], hint[q + 1]]: no search at all where no knot falls into the block (most of
// them: the noise kernels' frequencies are log spaced), a short walk otherwise.  The FIRST block holds most knots of a
// log-spaced vector (cfg-3: 53 of 77 below bin N1): its bins are looked up directly in hint0 -- the one lane per
// workgroup that owns such a bin used to walk ~45 knots, LDS round trip by round trip, while the other 511 threads
// waited for it at the barrier (profiles/r04_c: 5.5 of a workgroup's 25 us).
template <typename H>
__device__ __forceinline__ int kernel_interval(const KTab<H> & t, int k) {
    const double x = (double)k * t.fstep;
    const int q = k >> t.log_n1;
    if (q == 0) return t.hint0[k];
    const int h0 = (int)t.hint[q];
    const int h1 = (int)t.hint[q + 1];
    int lo = h0;
    if (h1 > h0) {
        if (t.knots[h0 + 1] <= x) {
            ++lo;
            while (lo < h1 && t.knots[lo + 1] <= x) ++lo;
        }
    }
    return lo;
}

template <typename H>
__device__ __forceinline__ double2 kernel_eval(const KTab<H> & t, int lo, int k) {
    const double x = (double)k * t.fstep;
    const double mag = ppoly_at(t.knots, 0, t.mc, lo, x);
    if (t.ac == nullptr) return make_double2(mag, 0.0);
    const double ang = ppoly_at(t.knots, 0, t.ac, lo, x);
    return make_double2(mag * cos(ang), mag * sin(ang));
}

template <typename H>
__device__ __forceinline__ double2 kernel_at(const KTab<H> & t, int64_t k) {
    return kernel_eval(t, kernel_interval(t, (int)k), (int)k);
}

// Tables of the row pass: global (TLDS = false, 32-bit hints) or copied into LDS at `tab` (16-bit hints; the host
// checks that they fit, Params::tab_lds_bytes).  The copy is visible after the next __syncthreads().
template <bool TLDS>
struct KTabSel {
    using H = int32_t;
    static __device__ __forceinline__ KTab<H> make(const Params & p, int64_t kern, char *, int, int) {
        KTab<H> t;
        t.hint = p.knot_hint;
        t.hint0 = p.knot_hint0;
        t.knots = p.knots;
        t.mc = p.mag_coef + kern * 4 * (p.n_knot - 1);
        t.ac = p.ang_coef ? p.ang_coef + kern * 4 * (p.n_knot - 1) : nullptr;
        t.log_n1 = p.log_n1;
        t.fstep = p.fstep;
        return t;
    }
};
template <>
struct KTabSel<true> {
    using H = uint16_t;
    static __device__ __forceinline__ KTab<H> make(const Params & p, int64_t kern, char * tab, int tid, int nthread) {
        const int n_hint = (1 << p.log_n2) + 2;
        const int n_coef = 4 * (p.n_knot - 1);
        double * s_knots = reinterpret_cast<double *>(tab);
        double * s_mc = s_knots + p.n_knot;
        double * s_ac = s_mc + n_coef;
        H * s_hint = reinterpret_cast<H *>(s_ac + (p.ang_coef ? n_coef : 0));
        const double * __restrict__ g_mc = p.mag_coef + kern * n_coef;
        for (int i = tid; i < p.n_knot; i += nthread) s_knots[i] = p.knots[i];
        for (int i = tid; i < n_coef; i += nthread) s_mc[i] = g_mc[i];
        if (p.ang_coef) {
            const double * __restrict__ g_ac = p.ang_coef + kern * n_coef;
            for (int i = tid; i < n_coef; i += nthread) s_ac[i] = g_ac[i];
        }
        for (int i = tid; i < n_hint; i += nthread) s_hint[i] = p.knot_hint16[i];
        KTab<H> t;
        t.hint = s_hint;
        t.hint0 = p.knot_hint0;
        t.knots = s_knots;
        t.mc = s_mc;
        t.ac = p.ang_coef ? s_ac : nullptr;
        t.log_n1 = p.log_n1;
        t.fstep = p.fstep;
        return t;
    }
};

__device__ __forceinline__ double2 apply_kernel(double2 v, double2 kk, int deconvolve) {
    if (deconvolve) {
        const double den = kk.x * kk.x + kk.y * kk.y;
        return make_double2((v.x * kk.x + v.y * kk.y) / den, (v.y * kk.x - v.x * kk.y) / den);
    }
    return make_double2(v.x * kk.x - v.y * kk.y, v.x * kk.y + v.y * kk.x);
}

// Bins k (tile element ea) and M - k (element eb) of the packed transform: real-FFT unpacking X = E + w^k O,
// Y = K X, repacking Z'[k] = Ye + i Yo, Z'[M - k] = conj(Ye) + i conj(Yo), stored with re / im swapped for the
// inverse transform.  ea == eb: the bin that pairs with itself (k = M / 2).
__device__ __forceinline__ void pair_update_reg(double2 & za, double2 & zb, bool same, double2 wk, double2 ka,
                                                double2 kb, int deconvolve) {
    const double2 cb = cconj(zb);
    const double2 ee = cadd(za, cb);
    const double2 oo = mul_mi(csub(za, cb));
    const double2 t = cmul(wk, oo);
    const double2 xa = cadd(ee, t);
    const double2 xb = cconj(csub(ee, t));
    const double2 ya = apply_kernel(xa, ka, deconvolve);
    const double2 yb = apply_kernel(xb, kb, deconvolve);
    const double2 cyb = cconj(yb);
    const double2 ye = cadd(ya, cyb);
    const double2 yo = cmul(cconj(wk), csub(ya, cyb));
    za = make_double2(ye.y + yo.x, ye.x - yo.y);
    if (!same) zb = make_double2(yo.x - ye.y, ye.x + yo.y);
}

__device__ __forceinline__ void pair_update(double2 * sm, int ea, int eb, double2 wk, double2 ka, double2 kb,
                                            int deconvolve) {
    double2 za = sm[sw(ea)];
    double2 zb = sm[sw(eb)];
    pair_update_reg(za, zb, ea == eb, wk, ka, kb, deconvolve);
    sm[sw(ea)] = za;
    if (eb != ea) sm[sw(eb)] = zb;
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
    std::vector<double2> h((size_t)kTile + 256 + (size_t)pl.n2_entries);
    for (int64_t e = 0; e < kTile; ++e) fill_twiddle(h, (size_t)e, e, kTile);
    for (int64_t e = 0; e < 128; ++e) fill_twiddle(h, (size_t)kTile + e, e, n_fft);
    for (int64_t e = 0; e < 128; ++e) fill_twiddle(h, (size_t)kTile + 128 + e, (e << 7) % n_fft, n_fft);
    for (int64_t e = 0; e < pl.n2_entries; ++e) fill_twiddle(h, (size_t)kTile + 256 + e, (e << 14) % n_fft, n_fft);
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
// row pass: 0 = row pair per 64 KB tile (k_fft_rows, default), 1 = one row per 32 KB tile (k_fft_rows_split:
// experiment; needs more registers than four workgroups per CU leave)
namespace {
int g_rows_split = -1;
}
int rows_split() {
    if (g_rows_split < 0) {
        const char * e = std::getenv("TOAST_HIP_FFT_ROWS");
        g_rows_split = (e != nullptr && std::string(e) == "split") ? 1 : 0;
    }
    return g_rows_split;
}
void set_rows_split(int split) { g_rows_split = split ? 1 : 0; }
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
        if (!forced && (kLT - p.log_n1) < 2) p.stream_hint &= ~1;
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
    char * scratch = (char *)Manager::get().scratch(Manager::kScratchFftWork,
                                                    hint_bytes + (size_t)batch * m * sizeof(double2), st);
    int32_t * d_hint = (int32_t *)scratch;
    int32_t * d_hint0 = d_hint + n_hint;
    uint16_t * d_hint16 = (uint16_t *)(d_hint0 + n_hint0);
    p.knot_hint = d_hint;
    p.knot_hint16 = d_hint16;
    p.knot_hint0 = d_hint0;
    p.work = (double2 *)(scratch + hint_bytes);
    hipLaunchKernelGGL(k_knot_hint, dim3((unsigned)((n_hint + n_hint0 + 255) / 256)), dim3(256), 0, st, d_knots,
                       (int)n_knot, fstep, p.log_n1, n_hint, d_hint, d_hint16, d_hint0);
    // row pass: knots, this detector's cubics and 16-bit hints in LDS when they fit (layout: KTabSel<true>::make)
    const size_t tab_bytes = (((size_t)n_knot + (size_t)4 * (n_knot - 1) * (d_ang ? 2 : 1)) * sizeof(double) +
                              (size_t)n_hint * sizeof(uint16_t) + 15) & ~size_t(15);
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
    const unsigned n_col_tiles = (unsigned)(int64_t(1) << (p.log_n2 - (kLT - p.log_n1)));   // N2 / C
    p.tile_order = nullptr;
    if (p.xcd_order == 2 && p.aligned) {
        p.tile_order = tile_order_for(pl, n_samp, n_buffer, n_reflect, (int64_t)n_col_tiles,
                                      int64_t(1) << (kLT - p.log_n1), st);
    }
    const unsigned n_row_tiles = (unsigned)((int64_t(1) << p.log_n1) / 2);                 // N1 / 2
    for (int64_t det0 = 0; det0 < n_det; det0 += batch) {
        const int64_t nb = (n_det - det0 < batch) ? (n_det - det0) : batch;
        p.det0 = (int)det0;
        // points per thread of each pass: TOAST_HIP_FFT_POINTS="rows,cols_fwd,cols_inv" / toast_hip_fft_points
        if (points_of(1) == 8) {
            hipLaunchKernelGGL((k_fft_cols<kLT, 8, false>), dim3(n_col_tiles, (unsigned)nb), dim3(kTile / 8), lds, st, p);
        } else {
            hipLaunchKernelGGL((k_fft_cols<kLT, 16, false>), dim3(n_col_tiles, (unsigned)nb), dim3(kTile / 16), lds, st, p);
        }
        const bool split = rows_split() != 0 && !half_rows;
        const dim3 g_pair(split ? 1 : n_row_tiles, (unsigned)nb);   // split: rows 0 and N1 / 2 only (self-paired)
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
        if (split && n_row_tiles > 1) {
            const dim3 gr(n_row_tiles - 1, (unsigned)nb), bl(kTile / 16);
            if (tab_lds) {
                hipLaunchKernelGGL((k_fft_rows_split<8, 2, true>), gr, bl, lds_split, st, p);
            } else {
                hipLaunchKernelGGL((k_fft_rows_split<8, 2, false>), gr, bl, lds_split, st, p);
            }
        }
        if (points_of(2) == 8) {
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
