// fft_filter.hip -- FFT-based noise weighting of timestreams on gfx950 with rocFFT.
//
// Device counterpart of the reference's
//   * toast.fft.convolve(..., algorithm="numpy")  (src/toast/fft.py:163-212 padding/apodisation,
//     :190-212 kernel interpolation, :296-350 rfft -> multiply -> irfft -> crop), as driven by
//     ops.NoiseFilter (src/toast/ops/noise_filter.py:130-188), and
//   * FFTPlanReal1D (batched r2hc / hc2r in FFTW half-complex layout;
//     src/libtoast/include/toast/math_fft.hpp:24-82, src/libtoast/src/toast_math_fft_fftw.cpp:26-128).
//
// Pipeline per batch of B detectors (B chosen so the work buffers fit a budget):
//   k_fft_fill     tdata[b, :] = [0 .. | apodised mirror | tod | apodised mirror | .. 0]   (n_fft = 2^(ceil(log2 n)+1))
//   rocFFT D2Z     batched, out of place, interleaved hermitian output (n_fft/2 + 1 bins)
//   k_fft_kernel   F *= K (or /= K), K = |K|(f) exp(i arg K(f)) evaluated on the fly from the
//                  PCHIP piecewise cubics of |K| and arg K; Im F[nyquist] = 0; F[0] = 0
//   rocFFT Z2D     batched
//   k_fft_crop     tod[s] = tdata[b, n_buffer + s] / n_fft
// All stages are HBM streaming passes; the transforms themselves are rocFFT's.
#include <hip/hip_runtime.h>

#include <chrono>
#include <rocfft/rocfft.h>

#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <tuple>

#include "runtime.hpp"

using namespace toast_hip;

namespace toast_hip {
namespace fused_fft {
// fft_fused.hip: the three-pass hand-written pipeline (power-of-two lengths >= 8192)
bool supported(int64_t n_fft);
void set_points(int rows, int cols_fwd, int cols_inv);
void set_rows_split(int split);
void set_cols_reg(int on);
void set_rows_n2(int n2);
int mirror_tile_order_host(int64_t n_samp, int64_t n_buffer, int64_t n_reflect, int64_t n_tiles, int64_t cols_per_tile,
                           int32_t * order);
double pipeline_bytes_per_sample(int64_t n_samp, int64_t n_fft);
void convolve(double * d_tod, const int32_t * d_idx, int64_t n_det, int64_t n_samp, int64_t n_fft,
              int64_t n_buffer, int64_t n_reflect, double fstep, const double * d_knots, int64_t n_knot,
              const double * d_mag, const double * d_ang, int per_det, int deconvolve, const double * d_apod,
              int64_t max_batch, hipStream_t st);
}  // namespace fused_fft
}  // namespace toast_hip

namespace {

// TOAST_HIP_FFT=rocfft selects the rocFFT pipeline below for every length (default: the fused
// three-pass kernels wherever they apply, rocFFT for short or non power-of-two transforms).
int g_force_rocfft = -1;   // -1: not decided yet (environment), 0 / 1: set

bool use_fused(int64_t n_fft) {
    if (g_force_rocfft < 0) {
        const char * e = std::getenv("TOAST_HIP_FFT");
        g_force_rocfft = (e != nullptr && std::string(e) == "rocfft") ? 1 : 0;
    }
    return g_force_rocfft == 0 && fused_fft::supported(n_fft);
}

constexpr int kThreads = 256;

#define TH_ROCFFT(expr)                                                                   \
    do {                                                                                  \
        rocfft_status s_ = (expr);                                                        \
        if (s_ != rocfft_status_success) {                                                \
            std::ostringstream o_;                                                        \
            o_ << "rocFFT error " << (int)s_ << " at " << __FILE__ << ":" << __LINE__     \
               << " in " #expr;                                                           \
            throw ::toast_hip::Error(TOAST_HIP_ERR_DEVICE, o_.str());                     \
        }                                                                                 \
    } while (0)

struct Plan {
    rocfft_plan plan = nullptr;
    rocfft_execution_info info = nullptr;
    void * work = nullptr;
    size_t work_bytes = 0;
};

std::mutex g_mutex;
bool g_setup = false;
std::map<std::tuple<int, int64_t, int64_t, int>, Plan> g_plans;  // (device, length, batch, forward)

Plan & get_plan(int64_t length, int64_t batch, bool forward) {
    int dev = 0;
    TH_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_mutex);
    if (!g_setup) {
        TH_ROCFFT(rocfft_setup());
        g_setup = true;
    }
    auto key = std::make_tuple(dev, length, batch, forward ? 1 : 0);
    auto it = g_plans.find(key);
    if (it != g_plans.end()) return it->second;
    Plan p;
    rocfft_plan_description desc = nullptr;
    TH_ROCFFT(rocfft_plan_description_create(&desc));
    const size_t n_psd = (size_t)(length / 2 + 1);
    size_t one = 1;
    if (forward) {
        TH_ROCFFT(rocfft_plan_description_set_data_layout(desc, rocfft_array_type_real,
                                                          rocfft_array_type_hermitian_interleaved,
                                                          nullptr, nullptr, 1, &one, (size_t)length, 1,
                                                          &one, n_psd));
    } else {
        TH_ROCFFT(rocfft_plan_description_set_data_layout(desc, rocfft_array_type_hermitian_interleaved,
                                                          rocfft_array_type_real, nullptr, nullptr, 1,
                                                          &one, n_psd, 1, &one, (size_t)length));
    }
    const size_t len = (size_t)length;
    TH_ROCFFT(rocfft_plan_create(&p.plan, rocfft_placement_notinplace,
                                 forward ? rocfft_transform_type_real_forward
                                         : rocfft_transform_type_real_inverse,
                                 rocfft_precision_double, 1, &len, (size_t)batch, desc));
    TH_ROCFFT(rocfft_plan_description_destroy(desc));
    TH_ROCFFT(rocfft_execution_info_create(&p.info));
    TH_ROCFFT(rocfft_plan_get_work_buffer_size(p.plan, &p.work_bytes));
    if (p.work_bytes) {
        TH_HIP(hipMalloc(&p.work, p.work_bytes));
        TH_ROCFFT(rocfft_execution_info_set_work_buffer(p.info, p.work, p.work_bytes));
    }
    return g_plans.emplace(key, p).first->second;
}

void exec_plan(Plan & p, void * in, void * out, hipStream_t stream) {
    TH_ROCFFT(rocfft_execution_info_set_stream(p.info, stream));
    void * ib[1] = {in};
    void * ob[1] = {out};
    TH_ROCFFT(rocfft_execute(p.plan, ib, ob, p.info));
}

// Time-domain and Fourier-domain batch buffers: grow-only scratch slots of the memory manager
// (per process / device, failure-safe growth, the manager's block cache is flushed before giving up).
struct Scratch {
    int slot;
    void * get(size_t need) { return Manager::get().scratch(slot, need); }
};
Scratch g_tbuf{Manager::kScratchFftTime}, g_fbuf{Manager::kScratchFftFreq};

// ------------------------------------------------------------------------------------
// fill: src/toast/fft.py:163-188 (set_rfft_input)
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_fft_fill(const double * __restrict__ tod,
                                                       const int32_t * __restrict__ d_idx, int det0,
                                                       double * __restrict__ tdata,
                                                       const double * __restrict__ apod, int64_t n_samp,
                                                       int64_t n_fft, int64_t n_buffer, int64_t n_reflect) {
    const int b = blockIdx.y;
    const double * row = tod + (int64_t)d_idx[det0 + b] * n_samp;
    double * out = tdata + (int64_t)b * n_fft;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n_fft;
         i += (int64_t)gridDim.x * kThreads) {
        const int64_t s = i - n_buffer;  // sample index relative to the TOD start
        double v = 0.0;
        if (s >= 0 && s < n_samp) {
            v = row[s];
        } else if (s < 0 && s >= -n_reflect) {
            // tdata[n_buffer - n_reflect : n_buffer] = tod[n_reflect-1::-1] * apodize
            const int64_t j = s + n_reflect;  // 0 .. n_reflect-1 within the mirrored block
            v = row[n_reflect - 1 - j] * apod[j];
        } else if (s >= n_samp && s < n_samp + n_reflect) {
            // mirrored tail; the window is applied reversed (largest next to the data)
            const int64_t j = s - n_samp;  // 0 .. n_reflect-1
            v = row[n_samp - 1 - j] * apod[n_reflect - 1 - j];
        }
        out[i] = v;
    }
}

// Piecewise-cubic evaluation p(x) = c0 dx^3 + c1 dx^2 + c2 dx + c3 on the knot interval that
// contains x (first / last interval outside the knots: extrapolate=True), scipy PPoly order.
__device__ __forceinline__ double ppoly_eval(const double * __restrict__ knots, int n_knot,
                                             const double * __restrict__ coef, double x) {
    int lo = 0, hi = n_knot - 2;  // interval index range
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (knots[mid] <= x) {
            lo = mid;
        } else {
            hi = mid - 1;
        }
    }
    const double * c = coef + 4 * lo;
    const double dx = x - knots[lo];
    return ((c[0] * dx + c[1]) * dx + c[2]) * dx + c[3];
}

// multiply: src/toast/fft.py:190-212 (kernel = mag * exp(1j * ang)), :330-337
__global__ __launch_bounds__(kThreads) void k_fft_kernel(double2 * __restrict__ fdata, int det0,
                                                         int64_t n_psd, double fstep,
                                                         const double * __restrict__ knots, int n_knot,
                                                         const double * __restrict__ mag_coef,
                                                         const double * __restrict__ ang_coef,
                                                         int per_det, int deconvolve) {
    const int b = blockIdx.y;
    const int64_t kern = per_det ? (int64_t)(det0 + b) : 0;
    const double * mc = mag_coef + kern * 4 * (n_knot - 1);
    const double * ac = ang_coef ? ang_coef + kern * 4 * (n_knot - 1) : nullptr;
    double2 * f = fdata + (int64_t)b * n_psd;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n_psd;
         i += (int64_t)gridDim.x * kThreads) {
        double2 v = f[i];
        if (i == 0) {
            v = make_double2(0.0, 0.0);  // remove DC level
        } else {
            const double freq = (double)i * fstep;
            const double mag = ppoly_eval(knots, n_knot, mc, freq);
            double kr = mag, ki = 0.0;
            if (ac) {
                const double ang = ppoly_eval(knots, n_knot, ac, freq);
                kr = mag * cos(ang);
                ki = mag * sin(ang);
            }
            double2 r;
            if (deconvolve) {
                const double den = kr * kr + ki * ki;
                r.x = (v.x * kr + v.y * ki) / den;
                r.y = (v.y * kr - v.x * ki) / den;
            } else {
                r.x = v.x * kr - v.y * ki;
                r.y = v.x * ki + v.y * kr;
            }
            if (i == n_psd - 1) r.y = 0.0;  // Nyquist bin of a real transform is real
            v = r;
        }
        f[i] = v;
    }
}

__global__ __launch_bounds__(kThreads) void k_fft_crop(const double * __restrict__ tdata,
                                                       const int32_t * __restrict__ d_idx, int det0,
                                                       double * __restrict__ tod, int64_t n_samp,
                                                       int64_t n_fft, int64_t n_buffer, double norm) {
    const int b = blockIdx.y;
    double * row = tod + (int64_t)d_idx[det0 + b] * n_samp;
    const double * in = tdata + (int64_t)b * n_fft + n_buffer;
    for (int64_t s = (int64_t)blockIdx.x * kThreads + threadIdx.x; s < n_samp;
         s += (int64_t)gridDim.x * kThreads) {
        row[s] = in[s] * norm;
    }
}

// FFTW half-complex <-> interleaved hermitian (toast_math_fft_cufft.cpp:157-238 does this on
// the host): hc = r0, r1, .., r_{n/2}, i_{(n+1)/2-1}, .., i_1
__global__ __launch_bounds__(kThreads) void k_c2hc(const double2 * __restrict__ f, double * __restrict__ hc,
                                                   int64_t length, double scale) {
    const int b = blockIdx.y;
    const int64_t n_psd = length / 2 + 1;
    const double2 * in = f + (int64_t)b * n_psd;
    double * out = hc + (int64_t)b * length;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n_psd;
         i += (int64_t)gridDim.x * kThreads) {
        const double2 v = in[i];
        out[i] = v.x * scale;
        if (i > 0 && 2 * i < length) out[length - i] = v.y * scale;
    }
}

__global__ __launch_bounds__(kThreads) void k_hc2c(const double * __restrict__ hc, double2 * __restrict__ f,
                                                   int64_t length) {
    const int b = blockIdx.y;
    const int64_t n_psd = length / 2 + 1;
    const double * in = hc + (int64_t)b * length;
    double2 * out = f + (int64_t)b * n_psd;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n_psd;
         i += (int64_t)gridDim.x * kThreads) {
        const double im = (i > 0 && 2 * i < length) ? in[length - i] : 0.0;
        out[i] = make_double2(in[i], im);
    }
}

__global__ __launch_bounds__(kThreads) void k_scale(double * __restrict__ x, int64_t n, double s) {
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * kThreads) {
        x[i] *= s;
    }
}

// impulse of `value` at sample `pos` of every row
__global__ void k_impulse_set(double * __restrict__ rows, int64_t n_row, int64_t n_samp, int64_t pos, double value) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_row) rows[r * n_samp + pos] = value;
}

// One workgroup per row: ipeak = first index of max |x| (numpy.argmax), thr = 0.02 |x[ipeak]|,
// imin = max({j <= ipeak : |x_j| <= thr} u {0}), imax = min({j >= ipeak : |x_j| <= thr} u {n}), extent = imax - imin
// (the reference's two while loops, src/toast/fft.py:846-866).
__global__ __launch_bounds__(256) void k_impulse_extent(const double * __restrict__ rows, int64_t n_samp,
                                                        int32_t * __restrict__ extent) {
    const double * __restrict__ x = rows + (int64_t)blockIdx.x * n_samp;
    __shared__ double s_val[256];
    __shared__ int64_t s_idx[256];
    const int tid = threadIdx.x;
    double best = -1.0;
    int64_t bi = 0;
    for (int64_t j = tid; j < n_samp; j += 256) {
        const double a = fabs(x[j]);
        if (a > best) {        // strictly greater: the first occurrence wins inside a thread
            best = a;
            bi = j;
        }
    }
    s_val[tid] = best;
    s_idx[tid] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            const double ov = s_val[tid + s];
            const int64_t oi = s_idx[tid + s];
            if (ov > s_val[tid] || (ov == s_val[tid] && oi < s_idx[tid])) {
                s_val[tid] = ov;
                s_idx[tid] = oi;
            }
        }
        __syncthreads();
    }
    const int64_t ipeak = s_idx[0];
    const double thr = 0.02 * s_val[0];
    __syncthreads();
    int64_t lo = 0, hi = n_samp;
    for (int64_t j = tid; j < n_samp; j += 256) {
        const bool below = fabs(x[j]) <= thr;
        if (below && j <= ipeak && j > lo) lo = j;
        if (below && j >= ipeak && j < hi) hi = j;
    }
    s_idx[tid] = lo;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s && s_idx[tid + s] > s_idx[tid]) s_idx[tid] = s_idx[tid + s];
        __syncthreads();
    }
    const int64_t imin = s_idx[0];
    __syncthreads();
    s_idx[tid] = hi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s && s_idx[tid + s] < s_idx[tid]) s_idx[tid] = s_idx[tid + s];
        __syncthreads();
    }
    if (tid == 0) extent[blockIdx.x] = (int32_t)(s_idx[0] - imin);
}

// Flag extension of toast.fft.convolve / NoiseFilter (reference src/toast/utils.py:1055-1113 extend_flags, then
// src/toast/fft.py:935-945): every flagged run [s, e) ASSIGNS the mask to [max(s - b, 0), e + b) -- the end clipped to
// n - 1 when it reaches n, so the last sample is never assigned --, then the first and last b samples get the mask
// OR-ed in.  Sample j <= n - 2 is assigned iff a flagged sample lies in [j - b, j + b], i.e. iff the nearest flagged
// sample at or before j, or the nearest at or after j, is within b.  Two launches over (row, chunk of 2048 samples):
// the first leaves every chunk's first / last flagged sample and count in a small table, the second finds the nearest
// flagged samples outside its chunk from that table and inside it with a scan, and rewrites the chunk.  (The first
// version kept a prefix sum of the whole row in HBM -- 2.9 GB for cfg-3 -- and walked a row with one workgroup:
// 4.4 ms; this one reads the flags twice and writes them once.)
constexpr int kFlagChunk = 2048;      // 256 threads x 8 consecutive samples
constexpr int kFlagPer = kFlagChunk / 256;

// the thread's kFlagPer consecutive flags (OR-ed with the common row), bytes past the end of the row = 0: one 8-byte
// load per array when the piece is whole and aligned (rows of a multiple of 8 samples), byte loads otherwise
__device__ __forceinline__ void load_flags8(const uint8_t * __restrict__ f, const uint8_t * __restrict__ or_row,
                                            int64_t j0, int64_t n_samp, uint8_t (&val)[kFlagPer]) {
    static_assert(kFlagPer == 8, "one 64-bit word per thread");
    const bool whole = j0 + kFlagPer <= n_samp;
    uint64_t w = 0;
    if (whole && (reinterpret_cast<uintptr_t>(f + j0) & 7u) == 0) {
        w = *reinterpret_cast<const uint64_t *>(f + j0);
    } else {
#pragma unroll
        for (int k = 0; k < kFlagPer; ++k) {
            if (j0 + k < n_samp) w |= (uint64_t)f[j0 + k] << (8 * k);
        }
    }
    if (or_row != nullptr) {
        if (whole && (reinterpret_cast<uintptr_t>(or_row + j0) & 7u) == 0) {
            w |= *reinterpret_cast<const uint64_t *>(or_row + j0);
        } else {
#pragma unroll
            for (int k = 0; k < kFlagPer; ++k) {
                if (j0 + k < n_samp) w |= (uint64_t)or_row[j0 + k] << (8 * k);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kFlagPer; ++k) val[k] = (uint8_t)(w >> (8 * k));
}

__global__ __launch_bounds__(256) void k_flag_chunk_summary(const uint8_t * __restrict__ flags,
                                                            const int32_t * __restrict__ f_idx, int row0, int64_t n_samp,
                                                            uint8_t mask, const uint8_t * __restrict__ or_row,
                                                            int32_t * __restrict__ summary, int n_chunk) {
    const int r = row0 + blockIdx.y;
    const uint8_t * __restrict__ f = flags + (int64_t)f_idx[r] * n_samp;
    const int tid = threadIdx.x;
    const int64_t j0 = (int64_t)blockIdx.x * kFlagChunk + (int64_t)tid * kFlagPer;
    int32_t first = INT32_MAX, last = -1, cnt = 0;
    uint8_t val[kFlagPer];
    load_flags8(f, or_row, j0, n_samp, val);
#pragma unroll
    for (int k = 0; k < kFlagPer; ++k) {
        const int64_t j = j0 + k;
        if (j < n_samp && (val[k] & mask) != 0) {
            if (first == INT32_MAX) first = (int32_t)j;
            last = (int32_t)j;
            ++cnt;
        }
    }
    __shared__ int32_t s_first[256], s_last[256], s_cnt[256];
    s_first[tid] = first;
    s_last[tid] = last;
    s_cnt[tid] = cnt;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if (tid < s2) {
            s_first[tid] = min(s_first[tid], s_first[tid + s2]);
            s_last[tid] = max(s_last[tid], s_last[tid + s2]);
            s_cnt[tid] += s_cnt[tid + s2];
        }
        __syncthreads();
    }
    if (tid == 0) {
        int32_t * o = summary + ((int64_t)blockIdx.y * n_chunk + blockIdx.x) * 3;
        o[0] = s_first[0];
        o[1] = s_last[0];
        o[2] = s_cnt[0];
    }
}

__global__ __launch_bounds__(256) void k_extend_flags(uint8_t * __restrict__ flags, const int32_t * __restrict__ f_idx,
                                                      int row0, int64_t n_samp, uint8_t mask,
                                                      const int32_t * __restrict__ extent,
                                                      const int32_t * __restrict__ summary, int n_chunk, int edges,
                                                      const uint8_t * __restrict__ or_row) {
    const int r = row0 + blockIdx.y;
    uint8_t * __restrict__ f = flags + (int64_t)f_idx[r] * n_samp;
    const int64_t b = extent[r];
    const int tid = threadIdx.x;
    const int c = blockIdx.x;
    __shared__ int32_t s_a[256], s_b[256];
    __shared__ int64_t s_tot[256];
    // nearest flagged sample before / after this chunk, and the row's count, from the table of the first launch
    {
        const int32_t * __restrict__ row = summary + (int64_t)blockIdx.y * n_chunk * 3;
        int32_t before = -1, after = INT32_MAX;
        int64_t tot = 0;
        for (int i = tid; i < n_chunk; i += 256) {
            const int32_t fi = row[3 * i], la = row[3 * i + 1];
            tot += row[3 * i + 2];
            if (i < c && la > before) before = la;
            if (i > c && fi < after) after = fi;
        }
        s_a[tid] = before;
        s_b[tid] = after;
        s_tot[tid] = tot;
        __syncthreads();
        for (int s2 = 128; s2 > 0; s2 >>= 1) {
            if (tid < s2) {
                s_a[tid] = max(s_a[tid], s_a[tid + s2]);
                s_b[tid] = min(s_b[tid], s_b[tid + s2]);
                s_tot[tid] += s_tot[tid + s2];
            }
            __syncthreads();
        }
    }
    const int32_t prev_out = s_a[0], next_out = s_b[0];
    // a completely flagged row has no rising or falling edge: the reference returns without touching it, so the bits
    // outside the mask survive (utils.py:1078-1083)
    const bool extend = s_tot[0] != n_samp;
    __syncthreads();
    const int64_t j0 = (int64_t)c * kFlagChunk + (int64_t)tid * kFlagPer;
    uint8_t val[kFlagPer];
    bool set[kFlagPer];
    int32_t my_first = INT32_MAX, my_last = -1;
    load_flags8(f, or_row, j0, n_samp, val);
#pragma unroll
    for (int k = 0; k < kFlagPer; ++k) {
        const int64_t j = j0 + k;
        set[k] = j < n_samp && (val[k] & mask) != 0;
        if (set[k]) {
            if (my_first == INT32_MAX) my_first = (int32_t)j;
            my_last = (int32_t)j;
        }
    }
    // last flagged sample of the threads before this one, first flagged sample of the threads after it
    s_a[tid] = my_last;
    s_b[tid] = my_first;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const int32_t oa = (tid >= d) ? s_a[tid - d] : -1;
        const int32_t ob = (tid + d < 256) ? s_b[tid + d] : INT32_MAX;
        __syncthreads();
        s_a[tid] = max(s_a[tid], oa);
        s_b[tid] = min(s_b[tid], ob);
        __syncthreads();
    }
    int32_t prev = max(prev_out, (tid > 0) ? s_a[tid - 1] : -1);
    const int32_t next_right = min(next_out, (tid < 255) ? s_b[tid + 1] : INT32_MAX);
    // nearest flagged sample at or after j inside the thread: from the right
    int32_t next_in[kFlagPer];
    {
        int32_t nx = next_right;
#pragma unroll
        for (int k = kFlagPer - 1; k >= 0; --k) {
            if (set[k]) nx = (int32_t)(j0 + k);
            next_in[k] = nx;
        }
    }
    uint64_t w = 0;
#pragma unroll
    for (int k = 0; k < kFlagPer; ++k) {
        const int64_t j = j0 + k;
        if (set[k]) prev = (int32_t)j;
        const bool near = (prev >= 0 && j - prev <= b) || (next_in[k] != INT32_MAX && (int64_t)next_in[k] - j <= b);
        uint8_t v = val[k];
        if (extend && near && j <= n_samp - 2) v = mask;
        // f[:b] |= mask; f[-b:] |= mask  (Python semantics: b == 0 makes the second slice the whole array)
        if (edges && (b == 0 || j < b || j >= n_samp - b)) v |= mask;
        val[k] = v;
        w |= (uint64_t)v << (8 * k);
    }
    if (j0 + kFlagPer <= n_samp && (reinterpret_cast<uintptr_t>(f + j0) & 7u) == 0) {
        *reinterpret_cast<uint64_t *>(f + j0) = w;
    } else {
#pragma unroll
        for (int k = 0; k < kFlagPer; ++k) {
            if (j0 + k < n_samp) f[j0 + k] = val[k];
        }
    }
}

inline dim3 grid2(int64_t n, int64_t batch) {
    int64_t gx = (n + kThreads - 1) / kThreads;
    if (gx > 4096) gx = 4096;
    return dim3((unsigned)gx, (unsigned)batch, 1);
}

}  // namespace

extern "C" {

int64_t toast_hip_fft_length(int64_t n_samp) {
    // src/toast/fft.py:278-280: order = ceil(log2 n_samp); n_fft = 2^(order+1)
    int64_t order = 0;
    while ((int64_t(1) << order) < n_samp) ++order;
    return int64_t(1) << (order + 1);
}

void toast_hip_fft_select(int rocfft_only) { g_force_rocfft = rocfft_only ? 1 : 0; }

void toast_hip_fft_rows_split(int split) { toast_hip::fused_fft::set_rows_split(split); }
void toast_hip_fft_rows_n2(int n2) { toast_hip::fused_fft::set_rows_n2(n2); }
void toast_hip_fft_cols_reg(int on) { toast_hip::fused_fft::set_cols_reg(on); }

int toast_hip_fft_mirror_tile_order(int64_t n_samp, int64_t n_buffer, int64_t n_reflect, int64_t n_tiles,
                                    int64_t cols_per_tile, int32_t * order) {
    return toast_hip::fused_fft::mirror_tile_order_host(n_samp, n_buffer, n_reflect, n_tiles, cols_per_tile, order);
}

void toast_hip_fft_points(int rows, int cols_fwd, int cols_inv) {
    toast_hip::fused_fft::set_points(rows, cols_fwd, cols_inv);
}

int toast_hip_fft_fused(int64_t n_samp) { return use_fused(toast_hip_fft_length(n_samp)) ? 1 : 0; }

double toast_hip_fft_pipeline_bytes(int64_t n_samp) {
    const int64_t n_fft = toast_hip_fft_length(n_samp);
    if (use_fused(n_fft)) return fused_fft::pipeline_bytes_per_sample(n_samp, n_fft);
    // rocFFT pipeline: fill (8 + 8 r), 5 + 5 library passes of 16 r each way... counted from the
    // kernel trace (profiles/r02_a_fft_rocfft_baseline_rocprofv3.txt): 14 passes over 8 n_fft bytes,
    // read + write
    return 14.0 * 16.0 * (double)n_fft / (double)n_samp;
}

int toast_hip_fft_convolve_dev(double * d_tod, const int32_t * data_index, int64_t n_det,
                               int64_t n_samp, double rate, const double * knots, int64_t n_knot,
                               const double * mag_coef, const double * ang_coef, int64_t n_kernel,
                               int deconvolve, const double * apodize, int64_t n_apodize,
                               int64_t max_batch, void * stream) {
    return guarded([&] {
        if (n_det <= 0 || n_samp <= 0) return;
        if (n_knot < 2) fail_arg("fft_convolve: need at least two kernel frequencies");
        if (n_kernel != 1 && n_kernel != n_det) fail_arg("fft_convolve: n_kernel must be 1 or n_det");
        hipStream_t st = static_cast<hipStream_t>(stream);
        const int64_t n_fft = toast_hip_fft_length(n_samp);
        const int64_t n_psd = n_fft / 2 + 1;
        const int64_t n_buffer = (n_fft - n_samp) / 2;                          // fft.py:283
        const int64_t n_reflect = (n_buffer < n_samp) ? n_buffer : n_samp;      // fft.py:284
        if (n_apodize != n_reflect) fail_arg("fft_convolve: apodize window must have n_reflect entries");
        // numpy.fft.rfftfreq(n, d): k * (1 / (n d)) with d = 1 / rate
        const double fstep = 1.0 / ((double)n_fft * (1.0 / rate));

        static const bool host_timing = std::getenv("TOAST_HIP_FFT_HOST_TIMING") != nullptr;
        const auto ht0 = std::chrono::steady_clock::now();
        auto ht = [&](const char * what) {
            if (host_timing) {
                std::fprintf(stderr, "[toast_hip] fft_convolve_dev %-22s +%8.3f ms\n", what,
                             std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ht0).count());
            }
        };
        ParamBlock pb;
        const size_t o_idx = pb.push(data_index, sizeof(int32_t) * n_det);
        const size_t o_kn = pb.push(knots, sizeof(double) * n_knot);
        const size_t o_mc = pb.push(mag_coef, sizeof(double) * n_kernel * 4 * (n_knot - 1));
        const size_t o_ac = ang_coef ? pb.push(ang_coef, sizeof(double) * n_kernel * 4 * (n_knot - 1)) : 0;
        const size_t o_ap = pb.push(apodize, sizeof(double) * n_reflect);
        ht("pushed");
        const char * d = pb.commit(st);
        ht("committed");
        const int32_t * d_idx = (const int32_t *)(d + o_idx);
        const double * d_ac = ang_coef ? (const double *)(d + o_ac) : nullptr;

        if (use_fused(n_fft)) {
            fused_fft::convolve(d_tod, d_idx, n_det, n_samp, n_fft, n_buffer, n_reflect, fstep,
                                (const double *)(d + o_kn), n_knot, (const double *)(d + o_mc), d_ac,
                                (n_kernel == n_det && n_det > 1) ? 1 : 0, deconvolve,
                                (const double *)(d + o_ap), max_batch, st);
            ht("launched");
            return;
        }
        int64_t batch = (max_batch > 0) ? max_batch : 64;
        // keep the two work buffers under ~8 GB
        const int64_t cap = (int64_t)((size_t(8) << 30) / ((size_t)n_fft * 8 + (size_t)n_psd * 16));
        if (batch > cap) batch = cap > 0 ? cap : 1;
        if (batch > n_det) batch = n_det;
        double * tbuf = (double *)g_tbuf.get((size_t)batch * n_fft * sizeof(double));
        double2 * fbuf = (double2 *)g_fbuf.get((size_t)batch * n_psd * sizeof(double2));

        for (int64_t det0 = 0; det0 < n_det; det0 += batch) {
            const int64_t nb = (n_det - det0 < batch) ? (n_det - det0) : batch;
            Plan & fwd = get_plan(n_fft, nb, true);
            Plan & inv = get_plan(n_fft, nb, false);
            hipLaunchKernelGGL(k_fft_fill, grid2(n_fft, nb), dim3(kThreads), 0, st, d_tod, d_idx,
                               (int)det0, tbuf, (const double *)(d + o_ap), n_samp, n_fft, n_buffer,
                               n_reflect);
            exec_plan(fwd, tbuf, fbuf, st);
            hipLaunchKernelGGL(k_fft_kernel, grid2(n_psd, nb), dim3(kThreads), 0, st, fbuf, (int)det0,
                               n_psd, fstep, (const double *)(d + o_kn), (int)n_knot,
                               (const double *)(d + o_mc), d_ac, (n_kernel == n_det && n_det > 1) ? 1 : 0,
                               deconvolve);
            exec_plan(inv, fbuf, tbuf, st);
            hipLaunchKernelGGL(k_fft_crop, grid2(n_samp, nb), dim3(kThreads), 0, st, tbuf, d_idx,
                               (int)det0, d_tod, n_samp, n_fft, n_buffer, 1.0 / (double)n_fft);
            TH_HIP(hipGetLastError());
        }
    });
}

int toast_hip_fft_convolve(double * det_data, int64_t n_data_rows, const int32_t * data_index,
                           int64_t n_det, int64_t n_samp, double rate, const double * knots,
                           int64_t n_knot, const double * mag_coef, const double * ang_coef,
                           int64_t n_kernel, int deconvolve, const double * apodize,
                           int64_t n_apodize, int use_accel) {
    return guarded([&] {
        Manager::get().require_device();
        hipStream_t st = Manager::get().stream();
        Staging stg(use_accel != 0, st);
        double * d_tod = stg.inout(det_data, (size_t)(n_data_rows * n_samp));
        int rc = toast_hip_fft_convolve_dev(d_tod, data_index, n_det, n_samp, rate, knots, n_knot,
                                            mag_coef, ang_coef, n_kernel, deconvolve, apodize,
                                            n_apodize, 0, st);
        if (rc != TOAST_HIP_OK) throw Error(rc, toast_hip_last_error());
        stg.finish();
    });
}

// Width of every kernel's impulse response (reference src/toast/fft.py:836-872: an impulse of 100 in the middle of
// an empty timestream goes through the same convolution; walk left and right from the peak of |response| while it
// exceeds 2 % of the peak).  The reference does this on the host, one detector at a time; here the impulses are made,
// convolved and measured on the device in batches of rows of a scratch buffer and only the widths come back.
int toast_hip_fft_impulse_extents(int64_t n_det, int64_t n_samp, double rate, const double * knots, int64_t n_knot,
                                  const double * mag_coef, const double * ang_coef, int64_t n_kernel,
                                  int deconvolve, const double * apodize, int64_t n_apodize, int32_t * extents,
                                  void * stream) {
    return guarded([&] {
        if (n_det <= 0 || n_samp <= 0) return;
        if (n_kernel != 1 && n_kernel != n_det) fail_arg("fft_impulse_extents: n_kernel must be 1 or n_det");
        Manager::get().require_device();
        hipStream_t st = stream ? static_cast<hipStream_t>(stream) : Manager::get().stream();
        // a common kernel has one impulse response
        const int64_t n_resp = (n_kernel == 1) ? 1 : n_det;
        int64_t batch = (int64_t)((size_t(1) << 30) / ((size_t)n_samp * sizeof(double)));   // <= 1 GB of rows
        if (batch < 1) batch = 1;
        if (batch > 256) batch = 256;
        if (batch > n_resp) batch = n_resp;
        const size_t row_bytes = (((size_t)n_samp * sizeof(double)) + 255) & ~size_t(255);
        char * scratch = (char *)Manager::get().scratch(Manager::kScratchFftImpulse,
                                                        (size_t)batch * row_bytes + 256 + sizeof(int32_t) * batch, st);
        // rows must be contiguous [nb, n_samp] for the convolution: no padding between them
        double * d_rows = (double *)scratch;
        int32_t * d_ext = (int32_t *)(scratch + (((size_t)batch * n_samp * sizeof(double) + 255) & ~size_t(255)));
        std::vector<int32_t> idx((size_t)batch);
        for (int64_t i = 0; i < batch; ++i) idx[(size_t)i] = (int32_t)i;
        const int64_t n_coef = 4 * (n_knot - 1);
        for (int64_t r0 = 0; r0 < n_resp; r0 += batch) {
            const int64_t nb = (n_resp - r0 < batch) ? (n_resp - r0) : batch;
            TH_HIP(hipMemsetAsync(d_rows, 0, (size_t)nb * n_samp * sizeof(double), st));
            hipLaunchKernelGGL(k_impulse_set, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, d_rows, nb, n_samp,
                               n_samp / 2, 100.0);
            const int64_t nk = (n_kernel == 1) ? 1 : nb;
            int rc = toast_hip_fft_convolve_dev(d_rows, idx.data(), nb, n_samp, rate, knots, n_knot,
                                                mag_coef + ((n_kernel == 1) ? 0 : r0 * n_coef),
                                                ang_coef ? ang_coef + ((n_kernel == 1) ? 0 : r0 * n_coef) : nullptr, nk,
                                                deconvolve, apodize, n_apodize, 0, st);
            if (rc != TOAST_HIP_OK) throw Error(rc, toast_hip_last_error());
            hipLaunchKernelGGL(k_impulse_extent, dim3((unsigned)nb), dim3(256), 0, st, d_rows, n_samp, d_ext);
            TH_HIP(hipGetLastError());
            copy_to_host(extents + r0, d_ext, sizeof(int32_t) * nb, st);
        }
        if (n_kernel == 1) {
            for (int64_t i = 1; i < n_det; ++i) extents[i] = extents[0];
        }
    });
}

int toast_hip_fft_extend_flags(uint8_t * flags, int64_t n_flag_rows, const int32_t * flag_index, int64_t n_det,
                               int64_t n_samp, uint8_t mask, const int32_t * extents, int edges,
                               const uint8_t * or_row, int use_accel) {
    return guarded([&] {
        if (n_det <= 0 || n_samp <= 0) return;
        Manager::get().require_device();
        hipStream_t st = Manager::get().stream();
        Staging stg(use_accel != 0, st);
        uint8_t * d_flags = stg.inout(flags, (size_t)(n_flag_rows * n_samp));
        // the common row is a small per-call host array either way
        const uint8_t * d_or = (or_row != nullptr) ? stg.temp_in(or_row, (size_t)n_samp) : nullptr;
        ParamBlock pb;
        const size_t o_fi = pb.push(flag_index, sizeof(int32_t) * n_det);
        const size_t o_ex = pb.push(extents, sizeof(int32_t) * n_det);
        const char * d = pb.commit(st);
        const int64_t n_chunk = (n_samp + kFlagChunk - 1) / kFlagChunk;
        int64_t batch = 65535;     // rows per launch (grid y)
        if (batch > n_det) batch = n_det;
        int32_t * d_sum = (int32_t *)Manager::get().scratch(Manager::kScratchFftImpulse,
                                                           (size_t)batch * n_chunk * 3 * sizeof(int32_t), st);
        for (int64_t r0 = 0; r0 < n_det; r0 += batch) {
            const int64_t nb = (n_det - r0 < batch) ? (n_det - r0) : batch;
            const dim3 grid((unsigned)n_chunk, (unsigned)nb);
            hipLaunchKernelGGL(k_flag_chunk_summary, grid, dim3(256), 0, st, d_flags, (const int32_t *)(d + o_fi), (int)r0,
                               n_samp, mask, d_or, d_sum, (int)n_chunk);
            hipLaunchKernelGGL(k_extend_flags, grid, dim3(256), 0, st, d_flags, (const int32_t *)(d + o_fi), (int)r0, n_samp,
                               mask, (const int32_t *)(d + o_ex), d_sum, (int)n_chunk, edges ? 1 : 0, d_or);
        }
        TH_HIP(hipGetLastError());
        stg.finish();
    });
}

// FFTPlanReal1D counterpart: `count` real transforms of `length`, FFTW half-complex layout.
// forward: out = scale * r2hc(in);  backward: out = scale / length * hc2r(in)
int toast_hip_fft_r1d_dev(int forward, int64_t length, int64_t count, const double * d_in,
                          double * d_out, double scale, void * stream) {
    return guarded([&] {
        if (length <= 0 || count <= 0) return;
        hipStream_t st = static_cast<hipStream_t>(stream);
        const int64_t n_psd = length / 2 + 1;
        double2 * fbuf = (double2 *)g_fbuf.get((size_t)count * n_psd * sizeof(double2));
        if (forward) {
            double * tbuf = (double *)g_tbuf.get((size_t)count * length * sizeof(double));
            TH_HIP(hipMemcpyAsync(tbuf, d_in, (size_t)count * length * sizeof(double),
                                  hipMemcpyDeviceToDevice, st));  // rocFFT may overwrite its input
            exec_plan(get_plan(length, count, true), tbuf, fbuf, st);
            hipLaunchKernelGGL(k_c2hc, grid2(n_psd, count), dim3(kThreads), 0, st, fbuf, d_out, length,
                               scale);
        } else {
            hipLaunchKernelGGL(k_hc2c, grid2(n_psd, count), dim3(kThreads), 0, st, d_in, fbuf, length);
            exec_plan(get_plan(length, count, false), fbuf, d_out, st);
            hipLaunchKernelGGL(k_scale, grid2(count * length, 1), dim3(kThreads), 0, st, d_out,
                               count * length, scale / (double)length);
        }
        TH_HIP(hipGetLastError());
    });
}

int toast_hip_fft_r1d(int forward, int64_t length, int64_t count, const double * in, double * out,
                      double scale, int use_accel) {
    return guarded([&] {
        Manager::get().require_device();
        hipStream_t st = Manager::get().stream();
        Staging stg(use_accel != 0, st);
        const double * d_in = stg.in(in, (size_t)(length * count));
        double * d_out = stg.inout(out, (size_t)(length * count));
        int rc = toast_hip_fft_r1d_dev(forward, length, count, d_in, d_out, scale, st);
        if (rc != TOAST_HIP_OK) throw Error(rc, toast_hip_last_error());
        stg.finish();
    });
}

}  // extern "C"
