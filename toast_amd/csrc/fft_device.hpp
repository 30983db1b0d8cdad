// fft_device.hpp -- device helpers shared by the FFT noise-weighting kernels (fft_fused.hip: tiles held in LDS;
// fft_reg.hip: tiles held in registers).  See fft_fused.hip for the pipeline and its reference citations.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

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
    const char * tab_blob;        // fft_reg.hip: [n_kern][tab_bytes] knots | mag | ang | 16-bit hints as the row pass keeps
    int tab_bytes;                // them in LDS (KTabSel<true> layout, a multiple of 16 bytes), or nullptr / 0
    int tab_copy_bytes;           // ... its part without the hints (knots | mag | ang, rounded up to 16 bytes)
    const double2 * wrow;         // fft_reg.hip: stage twiddles of the 2048-point row transform (128 + 16 entries)
    const double2 * wcol;         // fft_reg.hip: stage twiddles of the column transform (cols_reg_twiddles), or nullptr
    int per_det, deconvolve;
    int aligned;                  // pass 1 may use padded_pair
    int xcd_order;                // column passes: contiguous column ranges per XCD
    const int32_t * tile_order;   // pass 1: column tile of workgroup blockIdx.x (mirror partners on one XCD), or nullptr
    // pass 1 of fft_reg.hip with a one-dimensional grid (fwd_seq != nullptr): workgroup -> (tile, detector) so that an
    // XCD works through its share of the tile sequence in segments of fwd_k tiles x groups of fwd_g detectors
    const int32_t * fwd_seq;      // n_tiles column tiles: entries [x n_tiles / 8, (x + 1) n_tiles / 8) belong to XCD x
    int fwd_k, fwd_g, fwd_tiles, fwd_dets;
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

}  // namespace fused_fft
}  // namespace toast_hip
