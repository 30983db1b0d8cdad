// fft_reg.hip -- the passes of the fused FFT noise weighting with the tile held in REGISTERS (round 5).
//
// Same pipeline, same arithmetic per butterfly and the same Params as fft_fused.hip (reference: toast.fft.convolve,
// src/toast/fft.py:163-350); what differs is where a tile lives while it is transformed.  fft_fused.hip keeps the tile
// of 4096 complex doubles in 64 KB of LDS and gives every thread 8 of its points: two workgroups per CU, ~15 workgroup
// barriers per tile, and 84 % of a workgroup's life spent in chains of butterfly -> LDS exchange -> barrier with
// nothing saturated (profiles/r04_c section 1).  A CU has 512 KB of vector registers and 160 KB of LDS, so here
//
//   * a lane holds 32 points (128 VGPRs of data, 2 waves per SIMD); a wave of the row pass holds one whole row of
//     N2 = 2048 points, a pair of waves of a column pass one tile of N1 x C points;
//   * the LDS is only the exchange buffer between two Stockham stages, and real and imaginary parts cross it one
//     after the other: 8 bytes per point, 16 KB per wave;
//   * a row is transformed by ONE wave: 2048 = 16 x 8 x 16, two exchanges, no workgroup barrier at all (the LDS
//     executes a wave's accesses in order); the two rows (k1, N1 - k1) whose bins pair up in the real-FFT unpacking
//     belong to two waves of a workgroup, the second of which computes its last butterflies with mirrored lane
//     indices so that both members of a bin pair sit in the same lane -- the pairing costs two barriers and one
//     16 KB round trip per wave;
//   * twice the rows in flight per CU (8 instead of 4) and 32 independent points per lane between dependent steps.
//
// Rows 0 and N1 / 2 (the self-paired ones) stay with k_fft_rows of fft_fused.hip.
#include <hip/hip_runtime.h>

#undef TOAST_FFT_PHASE_CLOCK
#include "runtime.hpp"
#include "fft_device.hpp"

namespace toast_hip {
namespace fused_fft {

// Experimental build (TOAST_HIP_EXTRA_FLAGS=-DTOAST_FFT_REG_CLOCK): lane 0 of every wave adds the 100 MHz wall-clock
// ticks between phase boundaries to g_reg_ticks (read with toast_hip_fft_reg_ticks).
#if defined(TOAST_FFT_REG_CLOCK)
__device__ unsigned long long g_reg_ticks[16];
# define RCLK_DECL unsigned long long rc_t = wall_clock64()
# define RCLK_MARK(i)                                                              \
    do {                                                                           \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                \
        const unsigned long long rc_n = wall_clock64();                            \
        if ((threadIdx.x & 63) == 0) atomicAdd(&g_reg_ticks[i], rc_n - rc_t);      \
        rc_t = rc_n;                                                               \
    } while (0)
#else
# define RCLK_DECL
# define RCLK_MARK(i)
#endif

template <>
struct Log2<32> {
    static constexpr int v = 5;
};

// w_64^k = (kC64[k], -kS64[k]), k < 32
__device__ constexpr double kC64[32] = {
    1.0000000000000000000, 0.99518472667219688624, 0.98078528040323044913, 0.95694033573220886494,
    0.92387953251128675613, 0.88192126434835502971, 0.83146961230254523708, 0.77301045336273696081,
    0.70710678118654752440, 0.63439328416364549822, 0.55557023301960222474, 0.47139673682599764856,
    0.38268343236508977173, 0.29028467725446236764, 0.19509032201612826785, 0.098017140329560601994,
    0.0, -0.098017140329560601994, -0.19509032201612826785, -0.29028467725446236764,
    -0.38268343236508977173, -0.47139673682599764856, -0.55557023301960222474, -0.63439328416364549822,
    -0.70710678118654752440, -0.77301045336273696081, -0.83146961230254523708, -0.88192126434835502971,
    -0.92387953251128675613, -0.95694033573220886494, -0.98078528040323044913, -0.99518472667219688624};
__device__ constexpr double kS64[32] = {
    0.0, 0.098017140329560601994, 0.19509032201612826785, 0.29028467725446236764,
    0.38268343236508977173, 0.47139673682599764856, 0.55557023301960222474, 0.63439328416364549822,
    0.70710678118654752440, 0.77301045336273696081, 0.83146961230254523708, 0.88192126434835502971,
    0.92387953251128675613, 0.95694033573220886494, 0.98078528040323044913, 0.99518472667219688624,
    1.0000000000000000000, 0.99518472667219688624, 0.98078528040323044913, 0.95694033573220886494,
    0.92387953251128675613, 0.88192126434835502971, 0.83146961230254523708, 0.77301045336273696081,
    0.70710678118654752440, 0.63439328416364549822, 0.55557023301960222474, 0.47139673682599764856,
    0.38268343236508977173, 0.29028467725446236764, 0.19509032201612826785, 0.098017140329560601994};

// 32-point DFT in registers: two 16-point transforms of the even / odd points and the twiddles w_32^k
template <>
struct DFT<32> {
    static __device__ __forceinline__ void run(double2 * a) {
        double2 e[16], o[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            e[k] = a[2 * k];
            o[k] = a[2 * k + 1];
        }
        DFT<16>::run(e);
        DFT<16>::run(o);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            double2 t;
            if (k == 0) {
                t = o[0];
            } else if (k == 8) {
                t = mul_mi(o[k]);
            } else {
                t = cmul(o[k], make_double2(kC64[2 * k], -kS64[2 * k]));
            }
            a[k] = cadd(e[k], t);
            a[k + 16] = csub(e[k], t);
        }
    }
};

// a[j] *= w1^j, j = 1 .. R - 1.  The powers are formed one after the other (w^j = w^(j-1) w for odd j, (w^(j/2))^2 for
// even j up to 8, then a running product): few of them are alive at any time -- the data of a lane already fills
// half of its registers.
template <int R>
__device__ __forceinline__ void apply_powers_n(double2 * a, double2 w1) {
    double2 w = w1;
    a[1] = cmul(a[1], w);
#pragma unroll
    for (int j = 2; j < R; ++j) {
        w = cmul(w, w1);
        a[j] = cmul(a[j], w);
    }
}

// Exchange buffer position (in doubles) of tile element i: one double of padding per 16.  A stage writes with a stride
// of R elements between neighbouring lanes (16 u + j -> 17 u + j: ds_write_b64 is served in groups of 16 contiguous
// lanes over 32 banks = 16 doubles, MI355X_MICROARCH.md section LDS) and reads lane-contiguous; and the
// position of element u + k Q stays (a function of u) + (a constant per k), so that one address register and the
// instructions' immediate offsets address a butterfly's points.
__device__ __forceinline__ int swd(int i) { return i + (i >> 4); }
constexpr int kPadRow = 2048 + 128;         // doubles of one wave's exchange buffer (one row of 2048 points)

// Synchronisation between the writes and the reads of an exchange.  One wave (T == 64): the LDS executes a wave's
// instructions in order, so a compiler-level fence is all it takes; several waves: a workgroup barrier.
template <int T>
__device__ __forceinline__ void tile_sync() {
    if constexpr (T == 64) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

// A tile of 2^LT points spread over T = 2^LT / P threads, P points each.  A Stockham stage of radix R has Q = 2^LT / R
// butterflies; thread `tid` computes B = P / R of them, butterfly b on the tile elements u + k Q (k < R) with
// u = tid + T b (or Q - 1 - that: `mirror`), kept in v[b + B k].  The first stage takes the points as they are loaded
// (element tid + T i in v[i]: lane-contiguous), the last one leaves them that way.
template <int LT, int P, int R>
__device__ __forceinline__ int bfly_u(int tid, int b, bool mirror) {
    constexpr int T = (1 << LT) / P;
    constexpr int Q = (1 << LT) / R;
    const int u = tid + T * b;
    return mirror ? (Q - 1 - u) : u;
}

// butterflies of one stage in registers.  tw != nullptr: output j is multiplied by w^j with w = tw[u >> log_s], the
// stage's own packed table (w_n^((u >> log_s) 2^(log_s - log_s0)): n = the transform length, 2^log_s0 transforms
// interleaved in the tile) -- in LDS: a gather from global memory in the middle of a transform is a memory round trip
// that nothing hides at two waves per SIMD (the first version of this kernel spent most of its life in them)
template <int LT, int P, int R>
__device__ __forceinline__ void reg_butterflies(double2 (&v)[P], int tid, bool mirror, int log_s,
                                                const double2 * __restrict__ tw) {
    constexpr int B = P / R;
#pragma unroll
    for (int b = 0; b < B; ++b) {
        double2 t[R];
#pragma unroll
        for (int k = 0; k < R; ++k) t[k] = v[b + B * k];
        DFT<R>::run(t);
        if (tw != nullptr) {
            const int u = bfly_u<LT, P, R>(tid, b, mirror);
            apply_powers_n<R>(t, tw[u >> log_s]);
        }
#pragma unroll
        for (int j = 0; j < R; ++j) v[b + B * j] = t[j];
    }
}

// outputs of a stage of radix RW at stride 2^LOGS -> inputs of the next stage of radix RR, through `smd`
// (2^LT + 2^(LT-4) doubles): real parts, then imaginary parts.  Every address is (a register per butterfly) + (a
// constant per point): swd(out_idx(u, j)) = swd(out_idx(u, 0)) + swd(j << LOGS) and swd(u + k Q) = swd(u) + swd(k Q) --
// the low bits of the first term never carry into the second (LOGS <= 4 or the term is a multiple of 16) -- so the
// points of a butterfly are addressed through the instructions' immediate offsets (the first version computed every
// position on its own: 30 % more vector instructions in the transforms).
template <int LT, int P, int RW, int RR, int LOGS>
__device__ __forceinline__ void reg_exchange(double2 (&v)[P], double * smd, int tid, bool mirror_w, bool mirror_r) {
    constexpr int T = (1 << LT) / P;
    constexpr int BW = P / RW, BR = P / RR;
    constexpr int QR = (1 << LT) / RR;
    constexpr int LRW = Log2<RW>::v;
    static_assert(LOGS + LRW >= 4, "reg_exchange: the butterfly index must start at bit 4 or above");
    double * pw[BW];
    const double * pr[BR];
#pragma unroll
    for (int b = 0; b < BW; ++b) pw[b] = smd + swd(out_idx(bfly_u<LT, P, RW>(tid, b, mirror_w), 0, LOGS, LRW));
#pragma unroll
    for (int b = 0; b < BR; ++b) pr[b] = smd + swd(bfly_u<LT, P, RR>(tid, b, mirror_r));
    tile_sync<T>();
#pragma unroll
    for (int b = 0; b < BW; ++b) {
#pragma unroll
        for (int j = 0; j < RW; ++j) pw[b][(j << LOGS) + ((j << LOGS) >> 4)] = v[b + BW * j].x;
    }
    tile_sync<T>();
#pragma unroll
    for (int b = 0; b < BR; ++b) {
#pragma unroll
        for (int k = 0; k < RR; ++k) v[b + BR * k].x = pr[b][k * (QR + QR / 16)];
    }
    tile_sync<T>();
#pragma unroll
    for (int b = 0; b < BW; ++b) {
#pragma unroll
        for (int j = 0; j < RW; ++j) pw[b][(j << LOGS) + ((j << LOGS) >> 4)] = v[b + BW * j].y;
    }
    tile_sync<T>();
#pragma unroll
    for (int b = 0; b < BR; ++b) {
#pragma unroll
        for (int k = 0; k < RR; ++k) v[b + BR * k].y = pr[b][k * (QR + QR / 16)];
    }
}

// One row of 2048 points in one wave, 32 points per lane: 2048 = 16 x 8 x 16.  In: v[i] = element l + 64 i, or
// (mirror_in) the layout a mirrored last stage left; out: the same, or (mirror_out) v[b + 2 j] = element
// 127 - (l + 64 b) + 128 j.
__device__ __forceinline__ void row_fft_2048(double2 (&v)[32], double * smd, int l, bool mirror_in, bool mirror_out,
                                             const double2 * __restrict__ s_w) {
    reg_butterflies<11, 32, 16>(v, l, mirror_in, 0, s_w);
    reg_exchange<11, 32, 16, 8, 0>(v, smd, l, mirror_in, false);
    reg_butterflies<11, 32, 8>(v, l, false, 4, s_w + 128);
    reg_exchange<11, 32, 8, 16, 4>(v, smd, l, false, mirror_out);
    reg_butterflies<11, 32, 16>(v, l, mirror_out, 7, nullptr);
}

// The kernel tables of one detector as the row pass keeps them in LDS (KTabSel<true>::make's layout), packed once per
// call so that a workgroup copies one contiguous block with 16-byte loads.
__global__ void k_pack_tables(const Params p, char * blob) {
    const int64_t kern = blockIdx.x;
    char * tab = blob + kern * (int64_t)p.tab_bytes;
    const int n_hint = (1 << p.log_n2) + 2;
    const int n_coef = 4 * (p.n_knot - 1);
    double * t_knots = reinterpret_cast<double *>(tab);
    double * t_mc = t_knots + p.n_knot;
    double * t_ac = t_mc + n_coef;
    uint16_t * t_hint = reinterpret_cast<uint16_t *>(t_ac + (p.ang_coef ? n_coef : 0));
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int i = tid; i < p.n_knot; i += nt) t_knots[i] = p.knots[i];
    for (int i = tid; i < n_coef; i += nt) t_mc[i] = p.mag_coef[kern * n_coef + i];
    if (p.ang_coef) {
        for (int i = tid; i < n_coef; i += nt) t_ac[i] = p.ang_coef[kern * n_coef + i];
    }
    const int n_hint_pad = (int)((tab + p.tab_bytes - reinterpret_cast<char *>(t_hint)) / 2);
    for (int i = tid; i < n_hint_pad; i += nt) t_hint[i] = (i < n_hint) ? p.knot_hint16[i] : (uint16_t)0;
}

// interval of bin k in the knot vector (kernel_interval of fft_device.hpp) with the two bins of the first block that a
// row pair can own -- g and N1 - g -- looked up when the kernel starts instead of where they are needed
template <typename H>
__device__ __forceinline__ int kernel_interval_pre(const KTab<H> & t, int k, int g, int lo_g, int lo_ng) {
    const double x = (double)k * t.fstep;
    const int q = k >> t.log_n1;
    const int h0 = (int)t.hint[q];
    const int h1 = (int)t.hint[q + 1];
    int lo = h0;
    if (h1 > h0) {
        if (t.knots[h0 + 1] <= x) {
            ++lo;
            while (lo < h1 && t.knots[lo + 1] <= x) ++lo;
        }
    }
    return (q == 0) ? ((k == g) ? lo_g : lo_ng) : lo;
}

// pass 2 for the row pairs (g, N1 - g), 0 < g < N1 / 2, N2 = 2048: 256 threads = two pairs, one row per wave
template <bool TLDS, bool CPLX>
__global__ __launch_bounds__(256, 2) void k_fft_rows_reg(const Params p) {
    extern __shared__ double2 sm[];           // 4 x 16 KB exchange buffers, then the kernel tables
    const int tid = threadIdx.x;
    const int l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int side = w & 1;                   // 0: row g, 1: row N1 - g
    const int b = blockIdx.y;
    const int n1 = 1 << p.log_n1;
    const int64_t m = (int64_t)n1 << 11;
    int g = 1 + 2 * (int)blockIdx.x + (w >> 1);
    const bool live = g < (n1 >> 1);
    if (!live) g = (n1 >> 1) - 1;             // an odd pair out repeats its neighbour's work and stores nothing
    const int64_t row = side ? (n1 - g) : g;
    RCLK_DECL;
    const int64_t kern = p.per_det ? (int64_t)(p.det0 + b) : 0;
    double2 * s_w = sm + 4 * (kPadRow / 2);                       // 144 stage twiddles
    char * s_tab = reinterpret_cast<char *>(s_w + 144);           // the kernel tables
    // Everything the kernel will ever ask global memory for is asked for here, together: the tables, then the row.
    uint4 tc[4];
    const int n_chunk = TLDS ? (p.tab_bytes >> 4) : 0;
    if (TLDS) {
        const uint4 * __restrict__ g_tab = reinterpret_cast<const uint4 *>(p.tab_blob + kern * (int64_t)p.tab_bytes);
#pragma unroll
        for (int j = 0; j < 4; ++j) tc[j] = (tid + 256 * j < n_chunk) ? g_tab[tid + 256 * j] : make_uint4(0, 0, 0, 0);
    }
    const double2 tws = (tid < 144) ? p.wrow[tid] : make_double2(0.0, 0.0);
    int lo_g = __builtin_amdgcn_readfirstlane(p.knot_hint0[g]);
    int lo_ng = __builtin_amdgcn_readfirstlane(p.knot_hint0[n1 - g]);
    // w_N^k = w_N^g w_4096^q = (w_N^g w_4096^l) w_64^r for bin k = g + N1 q, q = l + 64 r
    double2 w0 = cmul(tw_big(p.tb, g), p.tb.wtile[l]);
    const double2 * __restrict__ rowp = p.work + (int64_t)b * m + (row << 11) + l;
    double2 v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = rowp[64 * i];
    if (TLDS) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (tid + 256 * j < n_chunk) reinterpret_cast<uint4 *>(s_tab)[tid + 256 * j] = tc[j];
        }
    }
    if (tid < 144) s_w[tid] = tws;
    asm volatile("" : "+v"(w0.x), "+v"(w0.y), "+s"(lo_g), "+s"(lo_ng));   // here, not where they are first used
    __syncthreads();
    KTab<typename KTabSel<TLDS>::H> kt;
    if constexpr (TLDS) {
        const int n_coef = 4 * (p.n_knot - 1);
        const double * s_knots = reinterpret_cast<const double *>(s_tab);
        kt.knots = s_knots;
        kt.mc = s_knots + p.n_knot;
        kt.ac = CPLX ? kt.mc + n_coef : nullptr;
        kt.hint = reinterpret_cast<const typename KTabSel<TLDS>::H *>(kt.mc + n_coef * (CPLX ? 2 : 1));
        kt.hint0 = p.knot_hint0;
        kt.log_n1 = p.log_n1;
        kt.fstep = p.fstep;
    } else {
        kt = KTabSel<false>::make(p, kern, nullptr, tid, 256);
        if (!CPLX) kt.ac = nullptr;
    }
    double2 * own = sm + w * (kPadRow / 2);
    double2 * oth = sm + (w ^ 1) * (kPadRow / 2);
    double * smd = reinterpret_cast<double *>(own);
    RCLK_MARK(0);       // entry -> row and tables landed

    {
        row_fft_2048(v, smd, l, false, side != 0, s_w);
        RCLK_MARK(1);       // forward transform

        // Bin k = g + N1 q of row g (element q = l + 64 r of wave `side 0`: register r) pairs with bin M - k = element
        // 2047 - q of row N1 - g, which the mirrored last stage of wave `side 1` left in the same lane, in register
        // 30 + 2 (r & 1) - r.  Each wave works on ITS registers 0 .. 15 and hands registers 16 .. 31 to its partner
        // through its own buffer: own register i meets the partner's register 16 + jj, jj = 14 + 2 (i & 1) - i, on both
        // sides, so one section of code serves both waves (the roles za / zb are selected by `side`).
#pragma unroll
        for (int j = 0; j < 16; ++j) own[j * 64 + l] = v[16 + j];
        __syncthreads();
        RCLK_MARK(2);       // hand-over + barrier
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int jj = 14 + 2 * (i & 1) - i;
            const int r = side ? (16 + jj) : i;            // register of row g = the element's high bits
            const int k = g + n1 * (l + 64 * r);
            const double2 part = oth[jj * 64 + l];
            const double2 mine = v[i];
            // (component-wise selects: a conditional expression on the structs selects their ADDRESSES and pushes the
            // whole register array into scratch memory)
            double2 za = make_double2(side ? part.x : mine.x, side ? part.y : mine.y);
            double2 zb = make_double2(side ? mine.x : part.x, side ? mine.y : part.y);
            const double2 w64 = make_double2(side ? kC64[16 + jj] : kC64[i], side ? -kS64[16 + jj] : -kS64[i]);
            const double2 wk = cmul(w0, w64);
            pair_update_reg(za, zb, false, wk, kernel_eval(kt, kernel_interval_pre(kt, k, g, lo_g, lo_ng), k),
                            kernel_eval(kt, kernel_interval_pre(kt, (int)m - k, g, lo_g, lo_ng), (int)m - k), p.deconvolve);
            v[i] = make_double2(side ? zb.x : za.x, side ? zb.y : za.y);
            oth[jj * 64 + l] = make_double2(side ? za.x : zb.x, side ? za.y : zb.y);
        }
        RCLK_MARK(3);       // bin pairs
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) v[16 + j] = own[j * 64 + l];
        RCLK_MARK(4);       // barrier + take-back
    }
    row_fft_2048(v, smd, l, side != 0, false, s_w);
    RCLK_MARK(5);       // inverse transform

    if (live) {
        // the addresses are formed again from an opaque copy of the lane index: kept alive from the loads, 32 of them
        // would take 64 of the lane's registers through the whole kernel
        double2 * __restrict__ outp = p.work + (int64_t)b * m + (row << 11) + opaque_vgpr(l);
#pragma unroll
        for (int i = 0; i < 32; ++i) outp[64 * i] = v[i];
    }
    RCLK_MARK(6);       // stores
}

// The same pass with SIXTEEN points per lane: a row of 2048 in two waves, the row pair (g, N1 - g) in one workgroup of 256
// threads.  Half the registers (64 of data), 35 KB of exchange buffers per workgroup instead of 70: three workgroups per CU
// = three waves per SIMD where k_fft_rows_reg has two -- the pass is bound by how badly two waves per SIMD overlap their
// waits (profiles/r05_a section 3), not by instructions or bytes.  Element q = l + 128 i of a row sits in lane l (0 .. 127 of
// the row's half of the workgroup), register i; the mirrored last stage of the second row leaves element 127 - l + 128 j in
// register j, so bin q of row g meets bin 2047 - q of row N1 - g in the same lane, register 15 - i.  The exchanges
// synchronise with workgroup barriers (both rows run the same sequence).
__device__ __forceinline__ void row_fft_2048_p16(double2 (&v)[16], double * smd, int l, bool mirror_in, bool mirror_out,
                                                 const double2 * __restrict__ s_w) {
    reg_butterflies<11, 16, 16>(v, l, mirror_in, 0, s_w);
    reg_exchange<11, 16, 16, 8, 0>(v, smd, l, mirror_in, false);
    reg_butterflies<11, 16, 8>(v, l, false, 4, s_w + 128);
    reg_exchange<11, 16, 8, 16, 4>(v, smd, l, false, mirror_out);
    reg_butterflies<11, 16, 16>(v, l, mirror_out, 7, nullptr);
}

// HG (the default; TOAST_HIP_FFT_HINTS=lds turns it off): the 16-bit interval hints (4 KB, two look-ups per bin, the same for
// every workgroup: cache resident) stay in global memory -- 40 172 B of LDS instead of 44 272: FOUR workgroups per CU
// instead of three (cfg-3 call 18.59-18.65 -> 18.33-18.43 ms, the 2^23 call 42.5-42.7 -> 41.5 ms; profiles/r06_c)
template <bool TLDS, bool CPLX, bool HG = false>
__global__ __launch_bounds__(256, 3) void k_fft_rows_reg16(const Params p) {
    extern __shared__ double2 sm[];           // 2 x 17 KB exchange buffers (one per row), stage twiddles, kernel tables
    const int tid = threadIdx.x;
    const int l = tid & 127;
    const int side = __builtin_amdgcn_readfirstlane(tid >> 7);      // 0: row g, 1: row N1 - g
    const int b = blockIdx.y;
    const int n1 = 1 << p.log_n1;
    const int64_t m = (int64_t)n1 << 11;
    const int g = 1 + (int)blockIdx.x;
    const int64_t row = side ? (n1 - g) : g;
    const int64_t kern = p.per_det ? (int64_t)(p.det0 + b) : 0;
    double2 * s_w = sm + 2 * (kPadRow / 2);                        // 144 stage twiddles
    char * s_tab = reinterpret_cast<char *>(s_w + 144);           // the kernel tables
    uint4 tc[3];
    const int n_chunk = TLDS ? ((HG ? p.tab_copy_bytes : p.tab_bytes) >> 4) : 0;            // <= 768 (launch_rows_reg16)
    if (TLDS) {
        const uint4 * __restrict__ g_tab = reinterpret_cast<const uint4 *>(p.tab_blob + kern * (int64_t)p.tab_bytes);
#pragma unroll
        for (int j = 0; j < 3; ++j) tc[j] = (tid + 256 * j < n_chunk) ? g_tab[tid + 256 * j] : make_uint4(0, 0, 0, 0);
    }
    const double2 tws = (tid < 144) ? p.wrow[tid] : make_double2(0.0, 0.0);
    int lo_g = __builtin_amdgcn_readfirstlane(p.knot_hint0[g]);
    int lo_ng = __builtin_amdgcn_readfirstlane(p.knot_hint0[n1 - g]);
    // w_N^k = w_N^g w_4096^q = (w_N^g w_4096^l) w_32^r for bin k = g + N1 q, q = l + 128 r
    double2 w0 = cmul(tw_big(p.tb, g), p.tb.wtile[l]);
    const double2 * __restrict__ rowp = p.work + (int64_t)b * m + (row << 11) + l;
    double2 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = rowp[128 * i];
    if (TLDS) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (tid + 256 * j < n_chunk) reinterpret_cast<uint4 *>(s_tab)[tid + 256 * j] = tc[j];
        }
    }
    if (tid < 144) s_w[tid] = tws;
    asm volatile("" : "+v"(w0.x), "+v"(w0.y), "+s"(lo_g), "+s"(lo_ng));   // here, not where they are first used
    __syncthreads();
    KTab<typename KTabSel<TLDS>::H> kt;
    if constexpr (TLDS) {
        const int n_coef = 4 * (p.n_knot - 1);
        const double * s_knots = reinterpret_cast<const double *>(s_tab);
        kt.knots = s_knots;
        kt.mc = s_knots + p.n_knot;
        kt.ac = CPLX ? kt.mc + n_coef : nullptr;
        if constexpr (HG) kt.hint = p.knot_hint16;
        else kt.hint = reinterpret_cast<const typename KTabSel<TLDS>::H *>(kt.mc + n_coef * (CPLX ? 2 : 1));
        kt.hint0 = p.knot_hint0;
        kt.log_n1 = p.log_n1;
        kt.fstep = p.fstep;
    } else {
        kt = KTabSel<false>::make(p, kern, nullptr, tid, 256);
        if (!CPLX) kt.ac = nullptr;
    }
    double2 * own = sm + side * (kPadRow / 2);
    double2 * oth = sm + (side ^ 1) * (kPadRow / 2);
    double * smd = reinterpret_cast<double *>(own);

    row_fft_2048_p16(v, smd, l, false, side != 0, s_w);
    // each half works on ITS registers 0 .. 7 and hands registers 8 .. 15 to its partner through its own buffer: own
    // register i meets the partner's register 15 - i = 8 + jj, jj = 7 - i, on both sides
    __syncthreads();            // (the last exchange's reads of this buffer are done)
#pragma unroll
    for (int j = 0; j < 8; ++j) own[j * 128 + l] = v[8 + j];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int jj = 7 - i;
        const int r = side ? (8 + jj) : i;             // register of row g = the element's high bits
        const int k = g + n1 * (l + 128 * r);
        const double2 part = oth[jj * 128 + l];
        const double2 mine = v[i];
        double2 za = make_double2(side ? part.x : mine.x, side ? part.y : mine.y);
        double2 zb = make_double2(side ? mine.x : part.x, side ? mine.y : part.y);
        const double2 w32 = make_double2(side ? kC64[2 * (8 + jj)] : kC64[2 * i], side ? -kS64[2 * (8 + jj)] : -kS64[2 * i]);
        const double2 wk = cmul(w0, w32);
        pair_update_reg(za, zb, false, wk, kernel_eval(kt, kernel_interval_pre(kt, k, g, lo_g, lo_ng), k),
                        kernel_eval(kt, kernel_interval_pre(kt, (int)m - k, g, lo_g, lo_ng), (int)m - k), p.deconvolve);
        v[i] = make_double2(side ? zb.x : za.x, side ? zb.y : za.y);
        oth[jj * 128 + l] = make_double2(side ? za.x : zb.x, side ? za.y : zb.y);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) v[8 + j] = own[j * 128 + l];
    // (the next exchange starts with a barrier: taken back before the buffer is written again)
    row_fft_2048_p16(v, smd, l, side != 0, false, s_w);
    double2 * __restrict__ outp = p.work + (int64_t)b * m + (row << 11) + opaque_vgpr(l);
#pragma unroll
    for (int i = 0; i < 16; ++i) outp[128 * i] = v[i];
}

// ------------------------------------------------------------------------------------------
// column passes with the tile in registers
// ------------------------------------------------------------------------------------------
// Tile = N1 rows x C columns = 2^LT points, 32 per lane, T = 2^LT / 32 threads; element e = k1 C + c sits in thread
// e mod T, register e / T, so a lane owns ONE column (c = tid mod C) and a load / store instruction of a wave covers
// 64 / C consecutive rows of C adjacent columns: pieces of 16 C bytes.
//   n_fft 2^21:  N1 =  512, C = 8 (128-byte pieces), LT = 12, 128 threads, 512 = 16 x 4 x 8 (fft_fused.hip: C = 8);
//                pass 3: 256 threads of 16 points (k_fft_cols_inv16)
//   n_fft 2^22:  N1 = 1024, C = 8 (128-byte pieces), LT = 13, 256 threads, 1024 = 16 x 4 x 16 (fft_fused.hip: C = 4)
//   n_fft 2^23:  N1 = 2048, C = 4 ( 64-byte pieces), LT = 13, 256 threads, 2048 = 16 x 8 x 16 (fft_fused.hip: C = 2:
//                every 128-byte line is shared by four tiles and crosses HBM 2.3 times, profiles/r04_c section 6)
template <int LOGN>
struct ColPlan;
template <>
struct ColPlan<9> {
    static constexpr int LT = 12, R0 = 16, R1 = 4, R2 = 8;
};
template <>
struct ColPlan<10> {
    static constexpr int LT = 13, R0 = 16, R1 = 4, R2 = 16;     // (32 x 32: the inverse pass spilled 16 registers in its radix-32 butterflies)
};
template <>
struct ColPlan<11> {
    static constexpr int LT = 13, R0 = 16, R1 = 8, R2 = 16;
};

// stage twiddles of the column transform in the order the kernel keeps them in LDS: stage 1 (index u >> s0:
// w_n^h, h < Q0 >> s0), then stage 2 of a three-stage plan (index u >> s1: w_n^(R0 h), h < Q1 >> s1)
template <int LOGN>
constexpr int col_tw_count() {
    using PL = ColPlan<LOGN>;
    constexpr int s0 = PL::LT - LOGN;
    constexpr int q0 = (1 << PL::LT) / PL::R0;
    int n = q0 >> s0;
    if (PL::R2 > 1) {
        constexpr int s1 = s0 + Log2<PL::R0>::v;
        constexpr int q1 = (1 << PL::LT) / PL::R1;
        n += q1 >> s1;
    }
    return n;
}

template <int LOGN, int P>
__device__ __forceinline__ void col_fft(double2 (&v)[P], double * smd, int tid, const double2 * __restrict__ s_w) {
    using PL = ColPlan<LOGN>;
    constexpr int LT = PL::LT;
    constexpr int s0 = LT - LOGN;
    constexpr int s1 = s0 + Log2<PL::R0>::v;
    static_assert(PL::R0 <= P && PL::R1 <= P && PL::R2 <= P, "col_fft: a butterfly does not fit a lane");
    reg_butterflies<LT, P, PL::R0>(v, tid, false, s0, s_w);
    reg_exchange<LT, P, PL::R0, PL::R1, s0>(v, smd, tid, false, false);
    if constexpr (PL::R2 > 1) {
        constexpr int s2 = s1 + Log2<PL::R1>::v;
        constexpr int n0 = ((1 << LT) / PL::R0) >> s0;
        reg_butterflies<LT, P, PL::R1>(v, tid, false, s1, s_w + n0);
        reg_exchange<LT, P, PL::R1, PL::R2, s1>(v, smd, tid, false, false);
        reg_butterflies<LT, P, PL::R2>(v, tid, false, s2, nullptr);
    } else {
        reg_butterflies<LT, P, PL::R1>(v, tid, false, s1, nullptr);
    }
}

// padded_pair of fft_device.hpp in 32-bit index arithmetic (n_fft <= 2^24 here), in two steps: the pair
// (x[s + n_buffer], x[s + n_buffer + 1]) of the padded, apodised series (set_rfft_input, src/toast/fft.py:163-188),
// s even, is a pair of the timestream (pad_source: where; zero outside the padded range) times, in the mirrored parts,
// a pair of the window (pad_window: where) with the members swapped.  Branch-free.
// (masks instead of conditional expressions: the compiler turns nested conditionals on the indices into branches)
struct PadSel {
    int direct, left, right;      // all ones or zero; at most one is set
};
__device__ __forceinline__ PadSel pad_sel(int s, int n_samp, int n_reflect) {
    PadSel q;
    q.direct = -(int)((unsigned)s < (unsigned)n_samp);
    q.left = -(int)((unsigned)(s + n_reflect) < (unsigned)n_reflect);
    q.right = -(int)((unsigned)(s - n_samp) < (unsigned)n_reflect);
    return q;
}
__device__ __forceinline__ int pad_source(int s, int n_samp, int n_reflect) {
    const PadSel q = pad_sel(s, n_samp, n_reflect);
    return (s & q.direct) | ((-2 - s) & q.left) | ((2 * n_samp - 2 - s) & q.right);
}
__device__ __forceinline__ int pad_window(int s, int n_samp, int n_reflect) {
    const PadSel q = pad_sel(s, n_samp, n_reflect);
    return ((s + n_reflect) & q.left) | ((n_reflect - 2 - (s - n_samp)) & q.right);
}
__device__ __forceinline__ double2 pad_combine(double2 r, double2 a, int s, int n_samp, int n_reflect) {
    const PadSel q = pad_sel(s, n_samp, n_reflect);
    const double mx = r.y * (q.left ? a.x : a.y);
    const double my = r.x * (q.left ? a.y : a.x);
    const bool none = (q.direct | q.left | q.right) == 0;
    return make_double2(none ? 0.0 : (q.direct ? r.x : mx), none ? 0.0 : (q.direct ? r.y : my));
}

// pass 1 (INV = false) and pass 3 (INV = true), N2 = 2048, aligned timestreams (Params::aligned)
template <int LOGN, bool INV>
#if !defined(TOAST_FFT_WIN_ROUNDS)
#define TOAST_FFT_WIN_ROUNDS 1
#endif
#if !defined(TOAST_FFT_FWD_MINWG)
#define TOAST_FFT_FWD_MINWG 2
#endif
__global__ __launch_bounds__((1 << ColPlan<LOGN>::LT) / 32, (INV ? 2 : TOAST_FFT_FWD_MINWG)) void k_fft_cols_reg(const Params p) {
    constexpr int LT = ColPlan<LOGN>::LT;
    constexpr int T = (1 << LT) / 32;
    constexpr int LOGC = LT - LOGN;
    constexpr int DK = T >> LOGC;                      // rows between two registers of a lane
    constexpr int NTW = col_tw_count<LOGN>();
    static_assert(NTW <= T, "one stage twiddle per thread");
    extern __shared__ double smd[];                    // exchange buffer, then the stage twiddles
    double2 * s_w = reinterpret_cast<double2 *>(smd + (1 << LT) + (1 << (LT - 4)));
    const int tid = threadIdx.x;
    int b = blockIdx.y;
    unsigned bx = blockIdx.x;
    if (!INV && p.fwd_seq != nullptr) {
        // one-dimensional grid: workgroups go to the XCDs round robin; XCD x works through ITS tiles (fwd_seq) in
        // segments of fwd_k tiles, a segment for groups of fwd_g detectors, tile fastest -- the timestream lines that
        // neighbouring tiles of the chain share and the window entries that the same tile of every detector reads are
        // in that XCD's L2 when they are asked for again (Params::fwd_seq, chain_tile_order in fft_fused.hip)
        const unsigned xcd = bx & 7u, q = bx >> 3;
        const unsigned tpx = (unsigned)p.fwd_tiles >> 3;
        const unsigned per_seg = (unsigned)p.fwd_k * (unsigned)p.fwd_dets;
        const unsigned seg = q / per_seg, r = q - seg * per_seg;
        const unsigned per_grp = (unsigned)p.fwd_k * (unsigned)p.fwd_g;
        const unsigned dg = r / per_grp, r2 = r - dg * per_grp;
        const unsigned d_in = r2 / (unsigned)p.fwd_k, ti = r2 - d_in * (unsigned)p.fwd_k;
        bx = (unsigned)p.fwd_seq[xcd * tpx + seg * (unsigned)p.fwd_k + ti];
        b = (int)(dg * (unsigned)p.fwd_g + d_in);
    } else if (!INV && p.tile_order != nullptr) {
        bx = (unsigned)p.tile_order[bx];
    } else if (p.xcd_order && (gridDim.x & 7u) == 0u) {
        bx = (bx & 7u) * (gridDim.x >> 3) + (bx >> 3);
    }
    const int64_t j2 = ((int64_t)bx << LOGC) + (tid & ((1 << LOGC) - 1));      // this lane's column
    const int k10 = tid >> LOGC;                                                 // its first row
    const int64_t m = int64_t(1) << (LOGN + 11);
    double2 * __restrict__ work = p.work + (int64_t)b * m;
    double * __restrict__ row = p.tod + (int64_t)p.d_idx[p.det0 + b] * p.n_samp;
    // everything the kernel asks global memory for is asked for here: stage twiddles, the two factors of the four-step
    // twiddles w_M^(k1 j2) = w_N^(2 k1 j2), k1 = k10 + DK i: w0 wd^i, and the tile
    const double2 tws = (tid < NTW) ? p.wcol[tid] : make_double2(0.0, 0.0);
    double2 w0 = tw_big(p.tb, 2 * (int64_t)k10 * j2);
    double2 wd = tw_big(p.tb, 2 * (int64_t)DK * j2);
    double2 v[32];
    if (!INV) {
        const int s0 = 2 * ((k10 << 11) + (int)j2) - (int)p.n_buffer;
        const int n_samp = (int)p.n_samp, n_reflect = (int)p.n_reflect;
        // The 32 timestream pairs land in the tile's own registers.  The 32 window pairs would need as many again; they go
        // straight to LDS instead (global_load_lds_dwordx4 into the exchange buffer, idle until the first butterflies are
        // done; 1 KB per wave and instruction, lane-linear; 16 KB per wave = two halves of 8 pairs) in four rounds: rounds
        // 0 and 1 are in flight with the timestream loads, round r + 2 is issued when round r has been read back.
        // (Window pairs in registers, whole or in steps: the compiler hoists the steps together, runs out of registers
        // and spills in the middle of the loads -- load, wait, spill, load, wait, spill.)
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int si = opaque_vgpr(s0 + i * (DK << 12));
            v[i] = *reinterpret_cast<const double2 *>(row + (uint32_t)pad_source(si, n_samp, n_reflect));
            // one address at a time: hoisted together, the 32 (and the 32 of the window) take the registers the loaded
            // values need, and the allocator answers with load - wait - spill - load (nine round trips instead of one)
            __builtin_amdgcn_sched_barrier(0);
        }
        // ONE memory round trip for the padding: the first 16 window pairs land in registers (64 are free beside the tile),
        // the other 16 in this wave's 16 KB of the idle exchange buffer (global_load_lds_dwordx4, lane-linear 1 KB per
        // instruction).  TOAST_FFT_WIN_ROUNDS=4 restores the four double-buffered LDS rounds (three round trips).
        char * stage = reinterpret_cast<char *>(smd) + __builtin_amdgcn_readfirstlane(tid >> 6) * 16384;
        const double2 * staged = reinterpret_cast<const double2 *>(stage) + (tid & 63);
#if TOAST_FFT_WIN_ROUNDS == 1
        double2 wr[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int si = opaque_vgpr(s0 + i * (DK << 12));
            wr[i] = *reinterpret_cast<const double2 *>(p.apod + (uint32_t)pad_window(si, n_samp, n_reflect));
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int si = opaque_vgpr(s0 + (16 + i) * (DK << 12));
            const double * g = p.apod + (uint32_t)pad_window(si, n_samp, n_reflect);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                             (__attribute__((address_space(3))) void *)(stage + i * 1024), 16, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        {
            const int sb = opaque_vgpr(s0);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                v[k] = pad_combine(v[k], wr[k], sb + k * (DK << 12), n_samp, n_reflect);
                asm volatile("" : "+v"(v[k].x), "+v"(v[k].y));      // formed HERE (see below)
            }
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                double2 b4[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) b4[i] = staged[(4 * h + i) * 64];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k = 16 + 4 * h + i;
                    v[k] = pad_combine(v[k], b4[i], sb + k * (DK << 12), n_samp, n_reflect);
                    asm volatile("" : "+v"(v[k].x), "+v"(v[k].y));
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
        auto issue = [&](int r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int si = opaque_vgpr(s0 + (8 * r + i) * (DK << 12));
                const double * g = p.apod + (uint32_t)pad_window(si, n_samp, n_reflect);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                                 (__attribute__((address_space(3))) void *)(stage + ((r & 1) * 8 + i) * 1024), 16, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        auto take = [&](int r) {
            const int sb = opaque_vgpr(s0);     // (the lane masks are formed again where they are used)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                double2 b4[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) b4[i] = staged[((r & 1) * 8 + 4 * h + i) * 64];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k = 8 * r + 4 * h + i;
                    v[k] = pad_combine(v[k], b4[i], sb + k * (DK << 12), n_samp, n_reflect);
                    // the product is formed HERE: left alone, the compiler sinks it to its use -- the second butterfly of the
                    // first stage for every other point --, keeps 16 window pairs alive until then and spills as many
                    // timestream pairs straight from their loads (load, wait, spill, eight times over: 34 registers of
                    // scratch, 2.3 GB each way per launch at 2^23, and nine memory round trips where one was meant)
                    asm volatile("" : "+v"(v[k].x), "+v"(v[k].y));
                }
            }
        };
        issue(0);
        issue(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        take(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issue(2);
        take(1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issue(3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        take(2);
        take(3);
#endif
        __builtin_amdgcn_sched_barrier(0);      // the padding is finished before the first butterfly starts
    } else {
        const double2 * __restrict__ src = work + ((int64_t)k10 << 11) + j2;
#pragma unroll
        for (int i = 0; i < 32; ++i) v[i] = src[(int64_t)(DK * i) << 11];
    }
    if (tid < NTW) s_w[tid] = tws;
    asm volatile("" : "+v"(w0.x), "+v"(w0.y), "+v"(wd.x), "+v"(wd.y));
    // pass 1 needs the four-step factors after the transform: parked in LDS meanwhile (the transform leaves no register free)
    double2 * s_park = s_w + NTW;
    if (!INV) {
        s_park[tid] = w0;
        s_park[T + tid] = wd;
    }
    __syncthreads();
    if (INV) {
        double2 w = w0;
        v[0] = cmul(v[0], w);
#pragma unroll
        for (int i = 1; i < 32; ++i) {
            w = cmul(w, wd);
            v[i] = cmul(v[i], w);
            asm volatile("" : "+v"(v[i].x), "+v"(v[i].y));      // (formed here, not sunk into the butterflies: see pass 1)
        }
    }
    col_fft<LOGN, 32>(v, smd, tid, s_w);
    const int tid_tail = opaque_vgpr(tid);
    const int64_t j2t = ((int64_t)bx << LOGC) + (tid_tail & ((1 << LOGC) - 1));
    const int k10t = tid_tail >> LOGC;
    if (!INV) {
        w0 = s_park[tid_tail];
        wd = s_park[T + tid_tail];
        double2 w = w0;
        v[0] = cmul(v[0], w);
#pragma unroll
        for (int i = 1; i < 32; ++i) {
            w = cmul(w, wd);
            v[i] = cmul(v[i], w);
        }
        double2 * __restrict__ dst = work + ((int64_t)k10t << 11) + j2t;
        if (p.stream_hint & 1) {
#pragma unroll
            for (int i = 0; i < 32; ++i) store_nt(dst + ((int64_t)(DK * i) << 11), v[i]);
        } else {
#pragma unroll
            for (int i = 0; i < 32; ++i) dst[(int64_t)(DK * i) << 11] = v[i];
        }
    } else {
        // the transform ran on swapped data: Re z' = v.y, Im z' = v.x; crop + scale (fft.py:341-350); s and n_samp
        // are even: both samples of a pair are inside or outside together
        const int s0 = 2 * ((k10t << 11) + (int)j2t) - (int)p.n_buffer;
        const int n_samp = (int)p.n_samp;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int sidx = s0 + i * (DK << 12);
            if ((unsigned)sidx < (unsigned)n_samp) {
                *reinterpret_cast<double2 *>(row + sidx) = make_double2(v[i].y * p.scale, v[i].x * p.scale);
            }
        }
    }
}

// Pass 3 with SIXTEEN points per lane, for N1 = 512 (n_fft 2^21): the 4096-point tile in 256 threads, 120 registers, 35 KB
// of LDS: four workgroups = four waves per SIMD, where k_fft_cols_reg<9, true> has two (2024 against 2204 us per 512
// detectors; the LDS-tile kernel: 2252).  At N1 = 1024 / 2048 the tile's exchange buffer (70 KB) allows two workgroups
// either way and 16 points per lane lose (3175 against 2500 us at 2^23: profiles/r05_a section 10).
template <int LOGN>
__global__ __launch_bounds__((1 << ColPlan<LOGN>::LT) / 16, 4) void k_fft_cols_inv16(const Params p) {
    constexpr int P = 16;
    constexpr int LT = ColPlan<LOGN>::LT;
    constexpr int T = (1 << LT) / P;
    constexpr int LOGC = LT - LOGN;
    constexpr int DK = T >> LOGC;                      // rows between two registers of a lane
    constexpr int NTW = col_tw_count<LOGN>();
    static_assert(NTW <= T, "one stage twiddle per thread");
    extern __shared__ double smd[];                    // exchange buffer, then the stage twiddles
    double2 * s_w = reinterpret_cast<double2 *>(smd + (1 << LT) + (1 << (LT - 4)));
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    unsigned bx = blockIdx.x;
    if (p.xcd_order && (gridDim.x & 7u) == 0u) bx = (bx & 7u) * (gridDim.x >> 3) + (bx >> 3);
    const int64_t j2 = ((int64_t)bx << LOGC) + (tid & ((1 << LOGC) - 1));      // this lane's column
    const int k10 = tid >> LOGC;                                                 // its first row
    const int64_t m = int64_t(1) << (LOGN + 11);
    double2 * __restrict__ work = p.work + (int64_t)b * m;
    double * __restrict__ row = p.tod + (int64_t)p.d_idx[p.det0 + b] * p.n_samp;
    const double2 tws = (tid < NTW) ? p.wcol[tid] : make_double2(0.0, 0.0);
    double2 w0 = tw_big(p.tb, 2 * (int64_t)k10 * j2);
    double2 wd = tw_big(p.tb, 2 * (int64_t)DK * j2);
    double2 v[P];
    const double2 * __restrict__ src = work + ((int64_t)k10 << 11) + j2;
#pragma unroll
    for (int i = 0; i < P; ++i) v[i] = src[(int64_t)(DK * i) << 11];
    if (tid < NTW) s_w[tid] = tws;
    asm volatile("" : "+v"(w0.x), "+v"(w0.y), "+v"(wd.x), "+v"(wd.y));
    __syncthreads();
    {
        double2 w = w0;
        v[0] = cmul(v[0], w);
#pragma unroll
        for (int i = 1; i < P; ++i) {
            w = cmul(w, wd);
            v[i] = cmul(v[i], w);
            asm volatile("" : "+v"(v[i].x), "+v"(v[i].y));
        }
    }
    col_fft<LOGN, P>(v, smd, tid, s_w);
    const int tid_tail = opaque_vgpr(tid);
    const int64_t j2t = ((int64_t)bx << LOGC) + (tid_tail & ((1 << LOGC) - 1));
    const int k10t = tid_tail >> LOGC;
    const int s0 = 2 * ((k10t << 11) + (int)j2t) - (int)p.n_buffer;
    const int n_samp = (int)p.n_samp;
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int sidx = s0 + i * (DK << 12);
        if ((unsigned)sidx < (unsigned)n_samp) {
            *reinterpret_cast<double2 *>(row + sidx) = make_double2(v[i].y * p.scale, v[i].x * p.scale);
        }
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
bool rows_reg_supported(const Params & p) { return p.log_n2 == 11 && p.log_n1 >= 2; }

// LDS of one workgroup of k_fft_rows_reg: four exchange buffers + the kernel tables (tab_bytes, 0: tables in global memory)
size_t rows_reg_lds(size_t tab_bytes) { return 4 * kPadRow * sizeof(double) + 144 * sizeof(double2) + tab_bytes; }
// two workgroups per CU: 2 x (68 KB of exchange buffers + 2.25 KB of twiddles + tables) <= 160 KB
constexpr size_t kTabLdsReg = 9 * 1024 + 512;

void launch_rows_reg(const Params & p, unsigned n_det, bool tab_lds, size_t tab_bytes, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        const int most = (int)rows_reg_lds(kTabLdsReg);
        auto set = [&](const void * fn) { TH_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, most)); };
        set(reinterpret_cast<const void *>(&k_fft_rows_reg<true, false>));
        set(reinterpret_cast<const void *>(&k_fft_rows_reg<true, true>));
        set(reinterpret_cast<const void *>(&k_fft_rows_reg<false, false>));
        set(reinterpret_cast<const void *>(&k_fft_rows_reg<false, true>));
        attr_set = true;
    }
    const int n_pair = (1 << p.log_n1) / 2 - 1;                  // row pairs g = 1 .. N1 / 2 - 1
    if (n_pair <= 0) return;
    const dim3 grid((unsigned)((n_pair + 1) / 2), n_det);
    const bool cplx = p.ang_coef != nullptr;
    if (tab_lds && p.tab_blob != nullptr && tab_bytes <= kTabLdsReg) {
        if (cplx) hipLaunchKernelGGL((k_fft_rows_reg<true, true>), grid, dim3(256), rows_reg_lds(tab_bytes), st, p);
        else hipLaunchKernelGGL((k_fft_rows_reg<true, false>), grid, dim3(256), rows_reg_lds(tab_bytes), st, p);
    } else {
        if (cplx) hipLaunchKernelGGL((k_fft_rows_reg<false, true>), grid, dim3(256), rows_reg_lds(0), st, p);
        else hipLaunchKernelGGL((k_fft_rows_reg<false, false>), grid, dim3(256), rows_reg_lds(0), st, p);
    }
}

// k_fft_rows_reg16: one row pair per workgroup; LDS = two row buffers + twiddles + tables (<= 12 KB: three 16-byte chunks per thread)
size_t rows_reg16_lds(size_t tab_bytes) { return 2 * kPadRow * sizeof(double) + 144 * sizeof(double2) + tab_bytes; }

void launch_rows_reg16(const Params & p, unsigned n_det, bool tab_lds, size_t tab_bytes, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        const int most = (int)rows_reg16_lds(kTabLdsReg);
        auto set = [&](const void * fn) { TH_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, most)); };
        set(reinterpret_cast<const void *>(&k_fft_rows_reg16<true, false>));
        set(reinterpret_cast<const void *>(&k_fft_rows_reg16<true, true>));
        set(reinterpret_cast<const void *>(&k_fft_rows_reg16<false, false>));
        set(reinterpret_cast<const void *>(&k_fft_rows_reg16<false, true>));
        set(reinterpret_cast<const void *>(&k_fft_rows_reg16<true, false, true>));
        set(reinterpret_cast<const void *>(&k_fft_rows_reg16<true, true, true>));
        attr_set = true;
    }
    const int n_pair = (1 << p.log_n1) / 2 - 1;                  // row pairs g = 1 .. N1 / 2 - 1
    if (n_pair <= 0) return;
    const dim3 grid((unsigned)n_pair, n_det);
    const bool cplx = p.ang_coef != nullptr;
    // (TOAST_HIP_FFT_HINTS=lds: the hints in LDS with the other tables, three workgroups per CU -- rounds 5-6a)
    static const bool hints_global = [] {
        const char * e = std::getenv("TOAST_HIP_FFT_HINTS");
        return !(e != nullptr && std::string(e) == "lds");
    }();
    if (hints_global && tab_lds && p.tab_blob != nullptr && tab_bytes <= kTabLdsReg) {
        const size_t lds = rows_reg16_lds((size_t)p.tab_copy_bytes);
        if (cplx) hipLaunchKernelGGL((k_fft_rows_reg16<true, true, true>), grid, dim3(256), lds, st, p);
        else hipLaunchKernelGGL((k_fft_rows_reg16<true, false, true>), grid, dim3(256), lds, st, p);
    } else if (tab_lds && p.tab_blob != nullptr && tab_bytes <= kTabLdsReg) {
        if (cplx) hipLaunchKernelGGL((k_fft_rows_reg16<true, true>), grid, dim3(256), rows_reg16_lds(tab_bytes), st, p);
        else hipLaunchKernelGGL((k_fft_rows_reg16<true, false>), grid, dim3(256), rows_reg16_lds(tab_bytes), st, p);
    } else {
        if (cplx) hipLaunchKernelGGL((k_fft_rows_reg16<false, true>), grid, dim3(256), rows_reg16_lds(0), st, p);
        else hipLaunchKernelGGL((k_fft_rows_reg16<false, false>), grid, dim3(256), rows_reg16_lds(0), st, p);
    }
}

// column passes in registers: N2 = 2048, N1 = 512 / 1024 / 2048, aligned timestreams
// (n_fft 2^21, N1 = 512, has a kernel too -- TOAST_HIP_FFT_COLS=reg9 -- but there the LDS-tile kernels already move
// 128-byte pieces and pad their input in ONE memory round trip where pass 1 of this file needs three: 2.85 against 4.2 ms)
bool cols_reg_supported(const Params & p, int min_log_n1) {
    return p.log_n2 == 11 && p.log_n1 >= min_log_n1 && p.log_n1 <= 11 && p.aligned;
}
int cols_reg_log_c(int log_n1) { return (log_n1 == 9 ? 12 : 13) - log_n1; }

// stage twiddles of the column transform of length 2^log_n1 in the kernel's LDS order (host; `out` gets the exponents
// e of w_n^e, n = 2^log_n1); returns their number
int cols_reg_twiddles(int log_n1, int * out) {
    const int lt = (log_n1 == 9) ? 12 : 13;
    const int r0 = 16;
    const int s0 = lt - log_n1;
    int n = 0;
    const int q0 = (1 << lt) / r0;
    for (int h = 0; h < (q0 >> s0); ++h) out[n++] = h;
    {
        const int r1 = (log_n1 == 11) ? 8 : 4;
        const int s1 = s0 + 4;
        const int q1 = (1 << lt) / r1;
        for (int h = 0; h < (q1 >> s1); ++h) out[n++] = h << (s1 - s0);
    }
    return n;
}

template <int LOGN>
static void launch_cols_reg_n(const Params & p, bool inv, unsigned n_det, hipStream_t st) {
    constexpr int LT = ColPlan<LOGN>::LT;
    constexpr int T = (1 << LT) / 32;
    const size_t lds = ((size_t)(1 << LT) + (size_t)(1 << (LT - 4))) * sizeof(double) +
                       ((size_t)col_tw_count<LOGN>() + 2 * T) * sizeof(double2);
    static bool attr_set = false;
    if (!attr_set) {
        TH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fft_cols_reg<LOGN, false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        TH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fft_cols_reg<LOGN, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const unsigned n_tiles = (unsigned)(2048 >> (LT - LOGN));
    if constexpr (LOGN == 9) {
        if (inv) {
            const size_t lds16 = ((size_t)(1 << LT) + (size_t)(1 << (LT - 4))) * sizeof(double) +
                                 (size_t)col_tw_count<LOGN>() * sizeof(double2);
            static bool attr16 = false;
            if (!attr16) {
                TH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fft_cols_inv16<LOGN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16));
                attr16 = true;
            }
            hipLaunchKernelGGL((k_fft_cols_inv16<LOGN>), dim3(n_tiles, n_det), dim3((1 << LT) / 16), lds16, st, p);
            return;
        }
    }
    if (inv) {
        hipLaunchKernelGGL((k_fft_cols_reg<LOGN, true>), dim3(n_tiles, n_det), dim3(T), lds, st, p);
    } else if (p.fwd_seq != nullptr) {
        hipLaunchKernelGGL((k_fft_cols_reg<LOGN, false>), dim3(n_tiles * n_det), dim3(T), lds, st, p);
    } else {
        hipLaunchKernelGGL((k_fft_cols_reg<LOGN, false>), dim3(n_tiles, n_det), dim3(T), lds, st, p);
    }
}

void launch_cols_reg(const Params & p, bool inv, unsigned n_det, hipStream_t st) {
    if (p.log_n1 == 9) launch_cols_reg_n<9>(p, inv, n_det, st);
    else if (p.log_n1 == 10) launch_cols_reg_n<10>(p, inv, n_det, st);
    else launch_cols_reg_n<11>(p, inv, n_det, st);
}

void pack_tables(const Params & p, char * blob, int64_t n_kern, hipStream_t st) {
    hipLaunchKernelGGL(k_pack_tables, dim3((unsigned)n_kern), dim3(256), 0, st, p, blob);
}

#if defined(TOAST_FFT_REG_CLOCK)
void reg_ticks(unsigned long long * out, int reset) {
    TH_HIP(hipDeviceSynchronize());
    TH_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_reg_ticks), 16 * sizeof(unsigned long long)));
    if (reset) {
        unsigned long long z[16] = {0};
        TH_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_reg_ticks), z, sizeof(z)));
    }
}
#endif

}  // namespace fused_fft
}  // namespace toast_hip

#if defined(TOAST_FFT_REG_CLOCK)
extern "C" void toast_hip_fft_reg_ticks(unsigned long long * out, int reset) { toast_hip::fused_fft::reg_ticks(out, reset); }
#endif
