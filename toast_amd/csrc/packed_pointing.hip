// packed_pointing.hip -- the solver's pointing cache in 20 bytes per detector-sample instead of 33.
//
// Every PCG iteration of the offset-template map-maker sweeps the cached pointing twice (SolverLHS,
// src/toast/ops/mapmaker_solve.py:342-506: accumulate A^T N^-1 M a, then project M^T N^-1 (M a - A z)), and each sweep
// reads, per detector-sample, the pixel number (int64, 8 B), three Stokes weights (24 B) and a flag byte: 33 B, nothing
// else of that size.  None of it changes during a solve, and a good part of it is redundant:
//   * the intensity weight of a detector is ONE number (stokes_weights_IQU: weights[0] = cal,
//     src/toast/_libtoast/ops_stokes_weights.cpp:77-140) -- 8 B per sample for a per-detector constant;
//   * the kernels turn the global pixel number into an offset into the local map (division by n_pix_submap, a look-up
//     in global2local) every time; the offset itself fits 30 bits for every Nside <= 8192;
//   * the flags only say "skip this sample" -- two bits (the accumulation and the projection may look at different
//     flag arrays / masks).
// toast_hip_offset_pack_pointing_dev writes, once per solve, a 32-bit word per detector-sample (local pixel offset + 1,
// or 0, and the two flag bits) and the Q / U weights (16 B); it CHECKS that the intensity weight really is constant
// along each detector row and refuses otherwise (the caller then keeps using the arrays it has).  The two packed sweeps
// compute exactly what k_offset_accumulate_v2 / k_offset_scan_project_v2 compute from the original arrays -- same
// products, same order -- from 20 B per sample, without the division and the global2local gather.
#include "kernel_common.hpp"

#include "../../include/toast_hip.h"

using namespace toast_hip;

namespace {

constexpr uint32_t kPkIndex = 0x3fffffffu;   // local pixel offset + 1 (0: no pixel of the local map)
constexpr uint32_t kPkAccFlag = 0x40000000u;  // flagged for the accumulation (A^T N^-1)
constexpr uint32_t kPkProjFlag = 0x80000000u; // flagged for the projection (M^T N^-1)

__global__ __launch_bounds__(kThreads) void k_pack_pointing(
    const Chunk * __restrict__ chunks, int n_chunks, const int32_t * __restrict__ p_idx,
    const int32_t * __restrict__ w_idx, const int32_t * __restrict__ fa_idx, const int32_t * __restrict__ fp_idx,
    const int64_t * __restrict__ g2l, const int64_t * __restrict__ pixels, const double * __restrict__ weights,
    const uint8_t * __restrict__ dflags, uint8_t dmask, int use_d, const uint8_t * __restrict__ sflags, uint8_t smask,
    int use_s, const uint8_t * __restrict__ pflags, uint8_t pmask, int use_p, FastDiv nps_div, int64_t n_samp,
    uint32_t * __restrict__ key, double2 * __restrict__ qu, double * __restrict__ cal_out, int * __restrict__ status) {
    const int det = blockIdx.x;
    const int64_t * prow = pixels + (int64_t)p_idx[det] * n_samp;
    const double * wrow = weights + (int64_t)w_idx[det] * n_samp * 3;
    const uint8_t * darow = use_d ? dflags + (int64_t)fa_idx[det] * n_samp : nullptr;
    const uint8_t * dprow = use_p ? pflags + (int64_t)fp_idx[det] * n_samp : nullptr;
    uint32_t * krow = key + (int64_t)det * n_samp;
    double2 * qrow = qu + (int64_t)det * n_samp;
    const int64_t nps = nps_div.d;
    // the row's intensity weight: the one of its first sample in view; every sample is compared with it
    const double cal = wrow[3 * chunks[0].first];
    if (blockIdx.y == 0 && threadIdx.x == 0) cal_out[det] = cal;
    int bad = 0;
    for (int ci = blockIdx.y; ci < n_chunks; ci += gridDim.y) {
        const Chunk c = chunks[ci];
        for (int i = threadIdx.x; i < c.count; i += kThreads) {
            const int64_t s = c.first + i;
            const int64_t p = prow[s];
            const double w0 = wrow[3 * s], w1 = wrow[3 * s + 1], w2 = wrow[3 * s + 2];
            const uint8_t fd = use_d ? darow[s] : (uint8_t)0;
            const uint8_t fs = use_s ? sflags[s] : (uint8_t)0;
            const uint8_t fp = use_p ? dprow[s] : (uint8_t)0;
            const bool hit = p >= 0;
            const int64_t pp = hit ? p : 0;
            const int64_t gsm = fastdiv(pp, nps_div);
            const int64_t lsm = g2l[gsm];
            const bool local = hit && lsm >= 0;
            const int64_t off = local ? lsm * nps + (pp - gsm * nps) : -1;
            if (off + 1 > (int64_t)kPkIndex) bad |= 2;
            if (!(w0 == cal)) bad |= 1;
            uint32_t k = local ? (uint32_t)(off + 1) & kPkIndex : 0u;
            if (((fd & dmask) != 0) | ((fs & smask) != 0)) k |= kPkAccFlag;
            if ((fp & pmask) != 0) k |= kPkProjFlag;
            krow[s] = k;
            qrow[s] = make_double2(w1, w2);
        }
    }
    if (bad) atomicOr(status, bad);
}

// Pair words.  The two detectors of a co-pointing pair (rows 2b and 2b + 1 of a call) see the same pixel in every sample
// -- the pixel kernels evaluate it once for both -- so their words differ in the flag bits only.  When that holds for
// every sample in view of every pair (k_pair_check) and the offsets fit 28 bits (Nside <= 4096 full sky), row 2b is
// rewritten as ONE word per sample for both detectors (k_pair_merge: offset + 1 in bits 0-27, accumulation / projection
// flag of detector e in bits 28 + 2 e / 29 + 2 e) and the sweeps read 4 + 2 x 16 B per pair-sample = 18 B per
// detector-sample.  A last detector without a partner becomes a pair whose second member is flagged throughout.
constexpr uint32_t kPrIndex = 0x0fffffffu;
__device__ __forceinline__ uint32_t pr_acc_flag(int e) { return 1u << (28 + 2 * e); }
__device__ __forceinline__ uint32_t pr_proj_flag(int e) { return 1u << (29 + 2 * e); }

__global__ __launch_bounds__(kThreads) void k_pair_check(const Chunk * __restrict__ chunks, int n_chunks, int n_det,
                                                         const uint32_t * __restrict__ key, int64_t n_samp,
                                                         int * __restrict__ status) {
    const int d0 = 2 * blockIdx.x;
    const bool lone = d0 + 1 >= n_det;
    const uint32_t * ra = key + (int64_t)d0 * n_samp;
    const uint32_t * rb = key + (int64_t)(lone ? d0 : d0 + 1) * n_samp;
    int bad = 0;
    for (int ci = blockIdx.y; ci < n_chunks; ci += gridDim.y) {
        const Chunk c = chunks[ci];
        for (int i = threadIdx.x; i < c.count; i += kThreads) {
            const int64_t s = c.first + i;
            const uint32_t oa = ra[s] & kPkIndex, ob = rb[s] & kPkIndex;
            if (oa != ob || oa > kPrIndex) bad = 4;
        }
    }
    if (bad) atomicOr(status, bad);
}

__global__ __launch_bounds__(kThreads) void k_pair_merge(const Chunk * __restrict__ chunks, int n_chunks, int n_det,
                                                         uint32_t * __restrict__ key, int64_t n_samp) {
    const int d0 = 2 * blockIdx.x;
    const bool lone = d0 + 1 >= n_det;
    uint32_t * ra = key + (int64_t)d0 * n_samp;
    const uint32_t * rb = key + (int64_t)(lone ? d0 : d0 + 1) * n_samp;
    for (int ci = blockIdx.y; ci < n_chunks; ci += gridDim.y) {
        const Chunk c = chunks[ci];
        for (int i = threadIdx.x; i < c.count; i += kThreads) {
            const int64_t s = c.first + i;
            const uint32_t ka = ra[s], kb = lone ? (kPkAccFlag | kPkProjFlag) : rb[s];
            uint32_t w = ka & kPkIndex;
            if (ka & kPkAccFlag) w |= pr_acc_flag(0);
            if (ka & kPkProjFlag) w |= pr_proj_flag(0);
            if (kb & kPkAccFlag) w |= pr_acc_flag(1);
            if (kb & kPkProjFlag) w |= pr_proj_flag(1);
            ra[s] = w;
        }
    }
}

// Pair weights.  The two detectors of an orthogonal pair have Q / U weights that are negatives of each other up to a few
// units in the last place when they carry the same calibration and polarisation efficiency (stokes_weights_IQU:
// eta cal cos / sin of an angle that differs by pi).  q_a + q_b is then exact in double precision (Sterbenz) and a small
// multiple of one ulp: it fits a FLOAT exactly, and q_b = (q_a + q_b) - q_a exactly.  k_pair_weights checks that per
// sample in view and per component and writes the sums as float2: the sweeps then read 4 (key) + 16 (q_a, u_a) + 8
// (sums) B per pair-sample = 14 B per detector-sample instead of 18.  Where it does not hold -- a weight that is pure
// rounding noise around zero (1e-32: its sum with the partner's noise needs all 53 bits), the reference's NaN at a pole --
// the component gets a NaN MARKER and the sweep reads that sample's partner weight from the partner's own row, which
// stays where it was: lossless for every input.  The markers are counted; more than one component in a hundred (pairs of
// different calibration: every sample) and the call reports that the sums are not worth using.
__global__ __launch_bounds__(kThreads) void k_pair_weights(const Chunk * __restrict__ chunks, int n_chunks, int n_det,
                                                           const double2 * __restrict__ qu, int64_t n_samp,
                                                           float2 * __restrict__ corr, unsigned long long * __restrict__ n_marked) {
    const int d0 = 2 * blockIdx.x;
    const bool lone = d0 + 1 >= n_det;
    const double2 * ra = qu + (int64_t)d0 * n_samp;
    const double2 * rb = qu + (int64_t)(lone ? d0 : d0 + 1) * n_samp;
    float2 * cr = corr + (int64_t)blockIdx.x * n_samp;
    auto sum_or_marker = [](double a, double b, int & marked) {
        const double d = a + b;
        const float f = (float)d;
        if ((double)f == d && (double)f - a == b) return f;
        ++marked;
        return __builtin_nanf("");
    };
    int marked = 0;
    for (int ci = blockIdx.y; ci < n_chunks; ci += gridDim.y) {
        const Chunk c = chunks[ci];
        for (int i = threadIdx.x; i < c.count; i += kThreads) {
            const int64_t s = c.first + i;
            float2 f = make_float2(0.0f, 0.0f);
            if (!lone) {
                const double2 a = ra[s], b = rb[s];
                f.x = sum_or_marker(a.x, b.x, marked);
                f.y = sum_or_marker(a.y, b.y, marked);
            }
            cr[s] = f;
        }
    }
    if (marked) atomicAdd(n_marked, (unsigned long long)marked);
}

// The whole pack in ONE sweep over the cached pointing (round 6): both detectors of a pair per workgroup, the pair word
// (k_pair_merge's format), both rows of Q / U weights and the pair weight sums (k_pair_weights' float2 or NaN marker)
// written straight from the pixels, weights and flags -- 33 B read and 22 B written per detector-sample instead of the four
// passes pack -> check -> merge -> weights (53 + 8 + 12 + 40 / 2 B).  status: bit 0 / 1 as k_pack_pointing (intensity weight
// not constant along a row, offset beyond 30 bits), bit 2 as k_pair_check (the pair does not see the same pixel in some
// sample, or the offset needs more than 28 bits): the caller then packs the old way.  Row 2b + 1 of `key` is not written.
__global__ __launch_bounds__(kThreads) void k_pack_pairs(
    const Chunk * __restrict__ chunks, int n_chunks, int n_det, const int32_t * __restrict__ p_idx,
    const int32_t * __restrict__ w_idx, const int32_t * __restrict__ fa_idx, const int32_t * __restrict__ fp_idx,
    const int64_t * __restrict__ g2l, const int64_t * __restrict__ pixels, const double * __restrict__ weights,
    const uint8_t * __restrict__ dflags, uint8_t dmask, int use_d, const uint8_t * __restrict__ sflags, uint8_t smask,
    int use_s, const uint8_t * __restrict__ pflags, uint8_t pmask, int use_p, FastDiv nps_div, int64_t n_samp,
    uint32_t * __restrict__ key, double2 * __restrict__ qu, double * __restrict__ cal_out, float2 * __restrict__ corr,
    int * __restrict__ status, unsigned long long * __restrict__ n_marked) {
    constexpr int E = 2;
    const int d0 = E * blockIdx.x;
    const bool lone = d0 + 1 >= n_det;
    const int64_t * prow[E];
    const double * wrow[E];
    const uint8_t * darow[E];
    const uint8_t * dprow[E];
    double2 * qrow[E];
    double cal[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int det = (lone && e == 1) ? d0 : d0 + e;
        prow[e] = pixels + (int64_t)p_idx[det] * n_samp;
        wrow[e] = weights + (int64_t)w_idx[det] * n_samp * 3;
        darow[e] = use_d ? dflags + (int64_t)fa_idx[det] * n_samp : nullptr;
        dprow[e] = use_p ? pflags + (int64_t)fp_idx[det] * n_samp : nullptr;
        qrow[e] = qu + (int64_t)det * n_samp;
        cal[e] = wrow[e][3 * chunks[0].first];
        if (blockIdx.y == 0 && threadIdx.x == 0 && !(lone && e == 1)) cal_out[det] = cal[e];
    }
    uint32_t * krow = key + (int64_t)d0 * n_samp;
    float2 * crow = corr + (int64_t)blockIdx.x * n_samp;
    const int64_t nps = nps_div.d;
    auto sum_or_marker = [](double a, double b, int & marked) {
        const double d = a + b;
        const float f = (float)d;
        if ((double)f == d && (double)f - a == b) return f;
        ++marked;
        return __builtin_nanf("");
    };
    int bad = 0, marked = 0;
    for (int ci = blockIdx.y; ci < n_chunks; ci += gridDim.y) {
        const Chunk c = chunks[ci];
        for (int i = threadIdx.x; i < c.count; i += kThreads) {
            const int64_t s = c.first + i;
            const uint8_t fs = use_s ? sflags[s] : (uint8_t)0;
            uint32_t k[E];
            double w1[E], w2[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int64_t p = prow[e][s];
                const double w0 = wrow[e][3 * s];
                w1[e] = wrow[e][3 * s + 1];
                w2[e] = wrow[e][3 * s + 2];
                const uint8_t fd = use_d ? darow[e][s] : (uint8_t)0;
                const uint8_t fp = use_p ? dprow[e][s] : (uint8_t)0;
                const bool hit = p >= 0;
                const int64_t pp = hit ? p : 0;
                const int64_t gsm = fastdiv(pp, nps_div);
                const int64_t lsm = g2l[gsm];
                const bool local = hit && lsm >= 0;
                const int64_t off = local ? lsm * nps + (pp - gsm * nps) : -1;
                if (off + 1 > (int64_t)kPkIndex) bad |= 2;
                if (!(w0 == cal[e])) bad |= 1;
                k[e] = local ? (uint32_t)(off + 1) & kPkIndex : 0u;
                if (((fd & dmask) != 0) | ((fs & smask) != 0)) k[e] |= kPkAccFlag;
                if ((fp & pmask) != 0) k[e] |= kPkProjFlag;
            }
            if (lone) k[1] = (k[0] & kPkIndex) | kPkAccFlag | kPkProjFlag;
            const uint32_t oa = k[0] & kPkIndex, ob = k[1] & kPkIndex;
            if (oa != ob || oa > kPrIndex) bad |= 4;
            uint32_t w = oa & kPrIndex;
            if (k[0] & kPkAccFlag) w |= pr_acc_flag(0);
            if (k[0] & kPkProjFlag) w |= pr_proj_flag(0);
            if (k[1] & kPkAccFlag) w |= pr_acc_flag(1);
            if (k[1] & kPkProjFlag) w |= pr_proj_flag(1);
            krow[s] = w;
            qrow[0][s] = make_double2(w1[0], w2[0]);
            float2 f = make_float2(0.0f, 0.0f);
            if (!lone) {
                qrow[1][s] = make_double2(w1[1], w2[1]);
                f.x = sum_or_marker(w1[0], w1[1], marked);
                f.y = sum_or_marker(w2[0], w2[1], marked);
            }
            crow[s] = f;
        }
    }
    if (bad) atomicOr(status, bad);
    if (marked) atomicAdd(n_marked, (unsigned long long)marked);
}

// the partner's weight of one component: from the pair sum, or -- marker -- from the partner's own row
__device__ __forceinline__ double pair_partner(float sum, double mine, const double * __restrict__ partner_row_value) {
    return (sum == sum) ? (double)sum - mine : *partner_row_value;
}

// Wave-uniform amplitude look-up (UNI kernels, baselines of at least 128 samples): the 128 consecutive samples a wave
// handles in one trip (two per lane) lie in at most TWO baselines, so the amplitude values and flags are two scalars per
// detector -- fetched with scalar loads from wave-uniform addresses and chosen per lane by one compare -- instead of two
// 64-bit reciprocal divisions, four 64-bit index sums and eight gathers per lane (8 of the 12 vector memory instructions
// of a trip; profiles/r05_b: vector issue 87 % / 73 % busy).  Same values, same order of operations.
struct WaveSteps {
    int64_t step0;     // baseline (within its view) of the wave's first sample
    bool two;          // a second baseline starts inside the chunk after it ...
    int split;         // ... at this many samples from the wave's first one (4096: none within reach)
};

__device__ __forceinline__ WaveSteps wave_steps(int64_t s0, int base, int n_pair, int64_t vfirst, int64_t chunk_end,
                                                const FastDiv & step_div) {
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int jw = base + 64 * wv;
    if (jw > n_pair - 1) jw = n_pair - 1;       // (a wave past the chunk's end: all lanes idle, any valid address will do)
    const int64_t w0 = s0 + 2 * (int64_t)jw;
    WaveSteps w;
    w.step0 = fastdiv(w0 - vfirst, step_div);
    const int64_t next = vfirst + (w.step0 + 1) * step_div.d;
    w.two = next < chunk_end;
    const int64_t gap = next - w0;
    w.split = (w.two && gap < 4096) ? (int)gap : 4096;
    return w;
}

// byte flags[i] through the aligned dword that holds it: a SCALAR load when i is wave-uniform (a byte load is a vector
// instruction even then; a pointer rebuilt from an integer would be a flat one).  `flags` is 4-byte aligned (checked on the
// host: uniform_amplitudes), so the dword lies inside the allocation.
__device__ __forceinline__ uint32_t flag_byte_uniform(const uint8_t * __restrict__ flags, int64_t i) {
    const uint32_t w = reinterpret_cast<const uint32_t *>(flags)[i >> 2];
    return (w >> ((unsigned)(i & 3) * 8u)) & 0xffu;
}

// The accumulation from pair words: both detectors of a pair share the pixel, so their contributions are added before
// the run reduction (what k_offset_accumulate_pk<2> does when it finds the keys equal, without the second key stream).
// CORR: the partner's Q / U weights come from the pair sums (k_pair_weights) instead of its own row.
template <bool CORR, bool UNI>
__global__ __launch_bounds__(kThreads) void k_offset_accumulate_pr(
    const Chunk * __restrict__ chunks, int n_chunks, int n_det, const int64_t * __restrict__ view_first,
    const int64_t * __restrict__ view_aoff, FastDiv step_div, const int64_t * __restrict__ amp_offsets,
    const double * __restrict__ amps, const uint8_t * __restrict__ amp_flags, const double * __restrict__ det_scale,
    const double * __restrict__ cal, double * __restrict__ zmap, const uint32_t * __restrict__ key,
    const double2 * __restrict__ qu, int64_t n_samp, const float2 * __restrict__ corr) {
    constexpr int NNZ = 3, E = 2;
    const int det0 = E * blockIdx.x;
    const uint32_t * krow = key + (int64_t)det0 * n_samp;
    const float2 * crow = CORR ? corr + (int64_t)blockIdx.x * n_samp : nullptr;
    bool on[E];
    const double2 * qrow[E];
    double ds[E], cl[E];
    int64_t amp_offset[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        on[e] = det0 + e < n_det;
        const int det = on[e] ? det0 + e : det0;
        qrow[e] = qu + (int64_t)det * n_samp;
        ds[e] = det_scale[det];
        cl[e] = cal[det];
        amp_offset[e] = amp_offsets[det];
    }
    for (int ci = blockIdx.y; ci < n_chunks; ci += gridDim.y) {
        const Chunk c = chunks[ci];
        const int64_t vfirst = view_first[c.view];
        const int64_t vaoff = view_aoff[c.view];
        const int head = (int)(c.first & 1);
        const int64_t s0 = c.first + head;          // even
        const int n_pair = (c.count - head) >> 1;
        for (int base = 0; base < n_pair; base += kThreads) {
            const int j = base + threadIdx.x;
            const bool active = j < n_pair;
            const int64_t s = s0 + 2 * (int64_t)(active ? j : 0);
            const uint2 kk = *reinterpret_cast<const uint2 *>(krow + s);
            double2 qa[E], qb[E], av[E];
            uint8_t afa[E], afb[E];
            float4 cc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if constexpr (CORR) cc = *reinterpret_cast<const float4 *>(crow + s);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if (!(CORR && e == 1)) {
                    qa[e] = qrow[e][s];
                    qb[e] = qrow[e][s + 1];
                }
            }
            if constexpr (UNI) {
                const WaveSteps w = wave_steps(s0, base, n_pair, vfirst, c.first + c.count, step_div);
                const int rel = 2 * (int)(threadIdx.x & 63);
                const bool up_a = rel >= w.split, up_b = rel + 1 >= w.split;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int64_t i0 = amp_offset[e] + vaoff + w.step0, i1 = w.two ? i0 + 1 : i0;
                    const double a0 = amps[i0], a1 = amps[i1];
                    const uint32_t f0 = flag_byte_uniform(amp_flags, i0), f1 = flag_byte_uniform(amp_flags, i1);
                    av[e] = make_double2(up_a ? a1 : a0, up_b ? a1 : a0);
                    afa[e] = (uint8_t)(up_a ? f1 : f0);
                    afb[e] = (uint8_t)(up_b ? f1 : f0);
                }
            } else {
                const int64_t step_a = fastdiv(s - vfirst, step_div), step_b = fastdiv(s + 1 - vfirst, step_div);
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int64_t aa = amp_offset[e] + vaoff + step_a, ab = amp_offset[e] + vaoff + step_b;
                    afa[e] = amp_flags[aa];
                    afb[e] = amp_flags[ab];
                    av[e] = make_double2(amps[aa], amps[ab]);
                }
            }
            if constexpr (CORR) {
                const double * pa = reinterpret_cast<const double *>(qrow[1] + s);      // (read only behind a marker)
                qa[1] = make_double2(pair_partner(cc.x, qa[0].x, pa), pair_partner(cc.y, qa[0].y, pa + 1));
                qb[1] = make_double2(pair_partner(cc.z, qb[0].x, pa + 2), pair_partner(cc.w, qb[0].y, pa + 3));
            }
            const uint32_t ia = kk.x & kPrIndex, ib = kk.y & kPrIndex;
            double va[E][NNZ], vb[E][NNZ];
            bool any_a = false, any_b = false;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const bool good_a = active & on[e] & (ia != 0) & ((kk.x & pr_acc_flag(e)) == 0);
                const bool good_b = active & on[e] & (ib != 0) & ((kk.y & pr_acc_flag(e)) == 0);
                any_a |= good_a;
                any_b |= good_b;
                // tod = 0 + amplitude (unflagged amplitudes only), then * det_scale
                const double ta = (afa[e] == 0) ? (0.0 + av[e].x) : 0.0, tb = (afb[e] == 0) ? (0.0 + av[e].y) : 0.0;
                const double sa = ta * ds[e], sb = tb * ds[e];
                va[e][0] = good_a ? sa * cl[e] : 0.0;
                va[e][1] = good_a ? sa * qa[e].x : 0.0;
                va[e][2] = good_a ? sa * qa[e].y : 0.0;
                vb[e][0] = good_b ? sb * cl[e] : 0.0;
                vb[e][1] = good_b ? sb * qb[e].x : 0.0;
                vb[e][2] = good_b ? sb * qb[e].y : 0.0;
            }
            const int64_t kam = any_a ? (int64_t)ia - 1 : -1;
            const int64_t kbm = any_b ? (int64_t)ib - 1 : -1;
            double vam[NNZ], vbm[NNZ];
#pragma unroll
            for (int k = 0; k < NNZ; ++k) {
                vam[k] = va[0][k] + va[1][k];
                vbm[k] = vb[0][k] + vb[1][k];
            }
            scatter_runs2<NNZ>(kam, vam, kbm, vbm, zmap);
        }
        // the peeled first sample and the odd last one: lanes 0 and 1 of the workgroup
        const int tail = (c.count - head) & 1;
        if ((threadIdx.x == 0 && head) || (threadIdx.x == 1 && tail)) {
            const int64_t s = (threadIdx.x == 0) ? c.first : c.first + c.count - 1;
            const int64_t astep = fastdiv(s - vfirst, step_div);
            const uint32_t k = krow[s];
            const uint32_t idx = k & kPrIndex;
            for (int e = 0; e < E; ++e) {
                if (!on[e] || idx == 0 || (k & pr_acc_flag(e)) != 0) continue;
                const int64_t a = amp_offset[e] + vaoff + astep;
                const double t = (amp_flags[a] == 0) ? (0.0 + amps[a]) : 0.0;
                const double sd = t * ds[e];
                double2 q = qrow[CORR ? 0 : e][s];
                if (CORR && e == 1) {
                    const float2 f = crow[s];
                    const double * pa = reinterpret_cast<const double *>(qrow[1] + s);
                    q = make_double2(pair_partner(f.x, q.x, pa), pair_partner(f.y, q.y, pa + 1));
                }
                double * z = zmap + NNZ * ((int64_t)idx - 1);
                unsafeAtomicAdd(z, sd * cl[e]);
                unsafeAtomicAdd(z + 1, sd * q.x);
                unsafeAtomicAdd(z + 2, sd * q.y);
            }
        }
    }
}

// two doubles at an 8-byte aligned address (global loads need dword alignment only)
struct __attribute__((packed, aligned(8))) MapPair {
    double x, y;
};

// The projection from pair words: one key stream and ONE map gather per pair-sample.
template <bool CORR, bool UNI>
__global__ __launch_bounds__(kThreads) void k_offset_scan_project_pr(
    const Chunk * __restrict__ chunks, int n_chunks, int n_det, const int64_t * __restrict__ view_first,
    const int64_t * __restrict__ view_aoff, FastDiv step_div, const int64_t * __restrict__ amp_offsets,
    const double * __restrict__ amps_in, double * __restrict__ amps_out, const uint8_t * __restrict__ amp_flags,
    const double * __restrict__ det_w, const double * __restrict__ cal, const double * __restrict__ map,
    const uint32_t * __restrict__ key, const double2 * __restrict__ qu, int64_t n_samp, const float2 * __restrict__ corr) {
    constexpr int E = 2;
    const int det0 = E * blockIdx.x;
    const uint32_t * krow = key + (int64_t)det0 * n_samp;
    const float2 * crow = CORR ? corr + (int64_t)blockIdx.x * n_samp : nullptr;
    bool on[E];
    const double2 * qrow[E];
    double dw[E], cl[E];
    int64_t amp_offset[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        on[e] = det0 + e < n_det;
        const int det = on[e] ? det0 + e : det0;
        qrow[e] = qu + (int64_t)det * n_samp;
        dw[e] = det_w[det];
        cl[e] = cal[det];
        amp_offset[e] = amp_offsets[det];
    }
    auto finish = [&](bool hit, double av, double w0, double w1, double w2, double m0, double m1, double m2) {
        double sc = 0.0;
        sc += w0 * m0;
        sc += w1 * m1;
        sc += w2 * m2;
        sc *= 1.0;
        const double d = 0.0 + av;
        return hit ? d - sc : d;
    };
    for (int ci = blockIdx.y; ci < n_chunks; ci += gridDim.y) {
        const Chunk c = chunks[ci];
        const int64_t vfirst = view_first[c.view];
        const int64_t vaoff = view_aoff[c.view];
        const int head = (int)(c.first & 1);
        const int64_t s0 = c.first + head;          // even
        const int n_pair = (c.count - head) >> 1;
        for (int base = 0; base < n_pair; base += kThreads) {
            const int j = base + threadIdx.x;
            const bool active = j < n_pair;
            const int64_t s = s0 + 2 * (int64_t)(active ? j : 0);
            const uint2 kk = *reinterpret_cast<const uint2 *>(krow + s);
            const uint32_t ia = kk.x & kPrIndex, ib = kk.y & kPrIndex;
            const bool hit_a = ia != 0, hit_b = ib != 0;
            const double * ma = map + (hit_a ? 3 * ((int64_t)ia - 1) : 0);
            const double * mb = map + (hit_b ? 3 * ((int64_t)ib - 1) : 0);
            // (a map value is 24 bytes at an 8-byte aligned address: one 16-byte and one 8-byte request instead of three)
            const MapPair ha = *reinterpret_cast<const MapPair *>(ma), hb = *reinterpret_cast<const MapPair *>(mb);
            const double a0 = ha.x, a1 = ha.y, a2 = ma[2], b0 = hb.x, b1 = hb.y, b2 = mb[2];
            float4 cc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if constexpr (CORR) cc = *reinterpret_cast<const float4 *>(crow + s);
            const double2 qa0 = qrow[0][s], qb0 = qrow[0][s + 1];
            WaveSteps w = {0, false, 4096};
            int64_t step_a = 0, step_b = 0;
            bool up_a = false, up_b = false;
            if constexpr (UNI) {
                w = wave_steps(s0, base, n_pair, vfirst, c.first + c.count, step_div);
                const int rel = 2 * (int)(threadIdx.x & 63);
                up_a = rel >= w.split;
                up_b = rel + 1 >= w.split;
            } else {
                step_a = fastdiv(s - vfirst, step_div);
                step_b = fastdiv(s + 1 - vfirst, step_div);
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                int64_t aa, ab;
                uint8_t afa, afb;
                double2 av;
                if constexpr (UNI) {
                    const int64_t i0 = amp_offset[e] + vaoff + w.step0, i1 = w.two ? i0 + 1 : i0;
                    const double v0 = amps_in[i0], v1 = amps_in[i1];
                    const uint32_t f0 = flag_byte_uniform(amp_flags, i0), f1 = flag_byte_uniform(amp_flags, i1);
                    aa = up_a ? i1 : i0;
                    ab = up_b ? i1 : i0;
                    afa = (uint8_t)(up_a ? f1 : f0);
                    afb = (uint8_t)(up_b ? f1 : f0);
                    av = make_double2(up_a ? v1 : v0, up_b ? v1 : v0);
                } else {
                    aa = amp_offset[e] + vaoff + step_a;
                    ab = amp_offset[e] + vaoff + step_b;
                    afa = amp_flags[aa];
                    afb = amp_flags[ab];
                    av = make_double2(amps_in[aa], amps_in[ab]);
                }
                double2 qa = qa0, qb = qb0;
                if (e == 1) {
                    if constexpr (CORR) {
                        const double * pa = reinterpret_cast<const double *>(qrow[1] + s);
                        qa = make_double2(pair_partner(cc.x, qa0.x, pa), pair_partner(cc.y, qa0.y, pa + 1));
                        qb = make_double2(pair_partner(cc.z, qb0.x, pa + 2), pair_partner(cc.w, qb0.y, pa + 3));
                    } else {
                        qa = qrow[1][s];
                        qb = qrow[1][s + 1];
                    }
                }
                const double da = finish(hit_a, av.x, cl[e], qa.x, qa.y, a0, a1, a2);
                const double db = finish(hit_b, av.y, cl[e], qb.x, qb.y, b0, b1, b2);
                int64_t ka = (active && on[e] && afa == 0) ? aa : (int64_t)-1;
                int64_t kb = (active && on[e] && afb == 0) ? ab : (int64_t)-1;
                double va[1] = {(ka >= 0 && (kk.x & pr_proj_flag(e)) == 0) ? da * dw[e] : 0.0};
                double vb[1] = {(kb >= 0 && (kk.y & pr_proj_flag(e)) == 0) ? db * dw[e] : 0.0};
                scatter_runs2<1>(ka, va, kb, vb, amps_out);
            }
        }
        // the peeled first sample (lane 0) and the odd last one (lane 1)
        const int tail = (c.count - head) & 1;
        if ((threadIdx.x == 0 && head) || (threadIdx.x == 1 && tail)) {
            const int64_t s = (threadIdx.x == 0) ? c.first : c.first + c.count - 1;
            const int64_t astep = fastdiv(s - vfirst, step_div);
            const uint32_t k = krow[s];
            const uint32_t idx = k & kPrIndex;
            const bool hit = idx != 0;
            const double * m = map + (hit ? 3 * ((int64_t)idx - 1) : 0);
            for (int e = 0; e < E; ++e) {
                if (!on[e]) continue;
                const int64_t a = amp_offset[e] + vaoff + astep;
                if (amp_flags[a] != 0 || (k & pr_proj_flag(e)) != 0) continue;
                double2 q = qrow[CORR ? 0 : e][s];
                if (CORR && e == 1) {
                    const float2 f = crow[s];
                    const double * pa = reinterpret_cast<const double *>(qrow[1] + s);
                    q = make_double2(pair_partner(f.x, q.x, pa), pair_partner(f.y, q.y, pa + 1));
                }
                const double d = finish(hit, amps_in[a], cl[e], q.x, q.y, m[0], m[1], m[2]);
                unsafeAtomicAdd(amps_out + a, d * dw[e]);
            }
        }
    }
}

// k_offset_accumulate_v2 from the packed cache: zmap += A^T N^-1 (M a), two consecutive samples per lane, E = 2: the two
// detectors of a co-pointing pair in one workgroup (their contributions to a pixel are added before the atomics).
template <int E>
__global__ __launch_bounds__(kThreads) void k_offset_accumulate_pk(
    const Chunk * __restrict__ chunks, int n_chunks, int n_det, const int64_t * __restrict__ view_first,
    const int64_t * __restrict__ view_aoff, FastDiv step_div, const int64_t * __restrict__ amp_offsets,
    const double * __restrict__ amps, const uint8_t * __restrict__ amp_flags, const double * __restrict__ det_scale,
    const double * __restrict__ cal, double * __restrict__ zmap, const uint32_t * __restrict__ key,
    const double2 * __restrict__ qu, int64_t n_samp) {
    constexpr int NNZ = 3;
    const int det0 = E * blockIdx.x;
    bool on[E];
    const uint32_t * krow[E];
    const double2 * qrow[E];
    double ds[E], cl[E];
    int64_t amp_offset[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        on[e] = det0 + e < n_det;
        const int det = on[e] ? det0 + e : det0;
        krow[e] = key + (int64_t)det * n_samp;
        qrow[e] = qu + (int64_t)det * n_samp;
        ds[e] = det_scale[det];
        cl[e] = cal[det];
        amp_offset[e] = amp_offsets[det];
    }
    for (int ci = blockIdx.y; ci < n_chunks; ci += gridDim.y) {
        const Chunk c = chunks[ci];
        const int64_t vfirst = view_first[c.view];
        const int64_t vaoff = view_aoff[c.view];
        const int head = (int)(c.first & 1);
        const int64_t s0 = c.first + head;          // even
        const int n_pair = (c.count - head) >> 1;
        for (int base = 0; base < n_pair; base += kThreads) {
            const int j = base + threadIdx.x;
            const bool active = j < n_pair;
            const int64_t s = s0 + 2 * (int64_t)(active ? j : 0);
            const int64_t step_a = fastdiv(s - vfirst, step_div), step_b = fastdiv(s + 1 - vfirst, step_div);
            int64_t ka[E], kb[E];
            double va[E][NNZ], vb[E][NNZ];
            uint2 kk[E];
            double2 qa[E], qb[E], av[E];
            uint8_t afa[E], afb[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                kk[e] = *reinterpret_cast<const uint2 *>(krow[e] + s);
                qa[e] = qrow[e][s];
                qb[e] = qrow[e][s + 1];
                const int64_t aa = amp_offset[e] + vaoff + step_a, ab = amp_offset[e] + vaoff + step_b;
                afa[e] = amp_flags[aa];
                afb[e] = amp_flags[ab];
                av[e] = make_double2(amps[aa], amps[ab]);
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const uint32_t ia = kk[e].x & kPkIndex, ib = kk[e].y & kPkIndex;
                const bool good_a = active & on[e] & (ia != 0) & ((kk[e].x & kPkAccFlag) == 0);
                const bool good_b = active & on[e] & (ib != 0) & ((kk[e].y & kPkAccFlag) == 0);
                ka[e] = good_a ? (int64_t)ia - 1 : -1;
                kb[e] = good_b ? (int64_t)ib - 1 : -1;
                // tod = 0 + amplitude (unflagged amplitudes only), then * det_scale
                const double ta = (afa[e] == 0) ? (0.0 + av[e].x) : 0.0, tb = (afb[e] == 0) ? (0.0 + av[e].y) : 0.0;
                const double sa = ta * ds[e], sb = tb * ds[e];
                va[e][0] = good_a ? sa * cl[e] : 0.0;
                va[e][1] = good_a ? sa * qa[e].x : 0.0;
                va[e][2] = good_a ? sa * qa[e].y : 0.0;
                vb[e][0] = good_b ? sb * cl[e] : 0.0;
                vb[e][1] = good_b ? sb * qb[e].x : 0.0;
                vb[e][2] = good_b ? sb * qb[e].y : 0.0;
            }
            if constexpr (E == 2) {
                const bool mergeable = ((ka[0] == ka[1]) | (ka[0] < 0) | (ka[1] < 0)) &
                                       ((kb[0] == kb[1]) | (kb[0] < 0) | (kb[1] < 0));
                if (__all(mergeable)) {
                    const int64_t kam = (ka[0] >= 0) ? ka[0] : ka[1];
                    const int64_t kbm = (kb[0] >= 0) ? kb[0] : kb[1];
                    double vam[NNZ], vbm[NNZ];
#pragma unroll
                    for (int k = 0; k < NNZ; ++k) {
                        vam[k] = va[0][k] + va[1][k];
                        vbm[k] = vb[0][k] + vb[1][k];
                    }
                    scatter_runs2<NNZ>(kam, vam, kbm, vbm, zmap);
                    continue;
                }
            }
#pragma unroll
            for (int e = 0; e < E; ++e) scatter_runs2<NNZ>(ka[e], va[e], kb[e], vb[e], zmap);
        }
        // the peeled first sample and the odd last one: lanes 0 and 1 of the workgroup
        const int tail = (c.count - head) & 1;
        if ((threadIdx.x == 0 && head) || (threadIdx.x == 1 && tail)) {
            const int64_t s = (threadIdx.x == 0) ? c.first : c.first + c.count - 1;
            const int64_t astep = fastdiv(s - vfirst, step_div);
            for (int e = 0; e < E; ++e) {
                if (!on[e]) continue;
                const uint32_t k = krow[e][s];
                const uint32_t idx = k & kPkIndex;
                if (idx == 0 || (k & kPkAccFlag) != 0) continue;
                const int64_t a = amp_offset[e] + vaoff + astep;
                const double t = (amp_flags[a] == 0) ? (0.0 + amps[a]) : 0.0;
                const double sd = t * ds[e];
                const double2 q = qrow[e][s];
                double * z = zmap + NNZ * ((int64_t)idx - 1);
                unsafeAtomicAdd(z, sd * cl[e]);
                unsafeAtomicAdd(z + 1, sd * q.x);
                unsafeAtomicAdd(z + 2, sd * q.y);
            }
        }
    }
}

// k_offset_scan_project_v2 from the packed cache: a_out += M^T N^-1 (M a - A z)
__global__ __launch_bounds__(kThreads) void k_offset_scan_project_pk(
    const Chunk * __restrict__ chunks, int n_chunks, const int64_t * __restrict__ view_first,
    const int64_t * __restrict__ view_aoff, FastDiv step_div, const int64_t * __restrict__ amp_offsets,
    const double * __restrict__ amps_in, double * __restrict__ amps_out, const uint8_t * __restrict__ amp_flags,
    const double * __restrict__ det_w, const double * __restrict__ cal, const double * __restrict__ map,
    const uint32_t * __restrict__ key, const double2 * __restrict__ qu, int64_t n_samp) {
    const int det = blockIdx.x;
    const uint32_t * krow = key + (int64_t)det * n_samp;
    const double2 * qrow = qu + (int64_t)det * n_samp;
    const double dw = det_w[det];
    const double cl = cal[det];
    const int64_t amp_offset = amp_offsets[det];
    // one sample: same products in the same order as k_offset_scan_project(_v2)
    auto finish = [&](bool hit, double av, double w0, double w1, double w2, double m0, double m1, double m2) {
        double sc = 0.0;
        sc += w0 * m0;
        sc += w1 * m1;
        sc += w2 * m2;
        sc *= 1.0;
        const double d = 0.0 + av;
        return hit ? d - sc : d;
    };
    for (int ci = blockIdx.y; ci < n_chunks; ci += gridDim.y) {
        const Chunk c = chunks[ci];
        const int64_t vfirst = view_first[c.view];
        const int64_t abase = amp_offset + view_aoff[c.view];
        const int head = (int)(c.first & 1);
        const int64_t s0 = c.first + head;          // even
        const int n_pair = (c.count - head) >> 1;
        for (int base = 0; base < n_pair; base += kThreads) {
            const int j = base + threadIdx.x;
            const bool active = j < n_pair;
            const int64_t s = s0 + 2 * (int64_t)(active ? j : 0);
            const int64_t aa = abase + fastdiv(s - vfirst, step_div);
            const int64_t ab = abase + fastdiv(s + 1 - vfirst, step_div);
            const uint8_t afa = amp_flags[aa], afb = amp_flags[ab];
            const double2 av = make_double2(amps_in[aa], amps_in[ab]);
            const uint2 kk = *reinterpret_cast<const uint2 *>(krow + s);
            const double2 qa = qrow[s], qb = qrow[s + 1];
            const uint32_t ia = kk.x & kPkIndex, ib = kk.y & kPkIndex;
            const bool hit_a = ia != 0, hit_b = ib != 0;
            const double * ma = map + (hit_a ? 3 * ((int64_t)ia - 1) : 0);
            const double * mb = map + (hit_b ? 3 * ((int64_t)ib - 1) : 0);
            const double a0 = ma[0], a1 = ma[1], a2 = ma[2], b0 = mb[0], b1 = mb[1], b2 = mb[2];
            const double da = finish(hit_a, av.x, cl, qa.x, qa.y, a0, a1, a2);
            const double db = finish(hit_b, av.y, cl, qb.x, qb.y, b0, b1, b2);
            int64_t ka = (active && afa == 0) ? aa : (int64_t)-1;
            int64_t kb = (active && afb == 0) ? ab : (int64_t)-1;
            double va[1] = {(ka >= 0 && (kk.x & kPkProjFlag) == 0) ? da * dw : 0.0};
            double vb[1] = {(kb >= 0 && (kk.y & kPkProjFlag) == 0) ? db * dw : 0.0};
            scatter_runs2<1>(ka, va, kb, vb, amps_out);
        }
        // the peeled first sample (lane 0) and the odd last one (lane 1)
        const int tail = (c.count - head) & 1;
        if ((threadIdx.x == 0 && head) || (threadIdx.x == 1 && tail)) {
            const int64_t s = (threadIdx.x == 0) ? c.first : c.first + c.count - 1;
            const int64_t a = abase + fastdiv(s - vfirst, step_div);
            const uint32_t k = krow[s];
            if (amp_flags[a] == 0 && (k & kPkProjFlag) == 0) {
                const uint32_t idx = k & kPkIndex;
                const bool hit = idx != 0;
                const double * m = map + (hit ? 3 * ((int64_t)idx - 1) : 0);
                const double2 q = qrow[s];
                const double d = finish(hit, amps_in[a], cl, q.x, q.y, m[0], m[1], m[2]);
                unsafeAtomicAdd(amps_out + a, d * dw);
            }
        }
    }
}

}  // namespace

extern "C" {

int toast_hip_offset_pack_pointing_dev(
    const int64_t * d_g2l, int64_t n_pix_submap, const int32_t * pixel_index, const int64_t * d_pixels,
    const int32_t * weight_index, const double * d_weights, const int32_t * acc_flag_index, const uint8_t * d_det_flags,
    int64_t n_flag_samp, uint8_t det_flag_mask, const uint8_t * d_shared_flags, int64_t n_shared_flags,
    uint8_t shared_flag_mask, const int32_t * proj_flag_index, const uint8_t * d_proj_flags, int64_t n_proj_flag_samp,
    uint8_t proj_flag_mask, int64_t n_det, int64_t n_samp, const toast_hip_interval * intervals, int64_t n_view,
    uint32_t * d_key, double * d_qu, double * d_cal, int * packable, int * pair_words, void * stream) {
    return guarded([&] {
        if (packable == nullptr) fail_arg("offset_pack_pointing: packable must not be null");
        *packable = 0;
        if (pair_words != nullptr) *pair_words = 0;
        if (n_det <= 0) return;
        if (n_pix_submap <= 0) fail_arg("n_pix_submap must be positive");
        need_aligned(d_qu, "packed Q / U weights");
        need_aligned(d_key, "packed pixel words");
        const auto chunks = make_chunks(intervals, n_view, n_samp);
        if (chunks.empty()) return;
        hipStream_t st = as_stream(stream);
        const int use_d = (n_flag_samp == n_samp) ? 1 : 0;
        const int use_s = (n_shared_flags == n_samp) ? 1 : 0;
        const int use_p = (n_proj_flag_samp == n_samp) ? 1 : 0;
        std::vector<int32_t> fa(n_det, 0), fp(n_det, 0);
        if (use_d) std::memcpy(fa.data(), acc_flag_index, sizeof(int32_t) * n_det);
        if (use_p) std::memcpy(fp.data(), proj_flag_index, sizeof(int32_t) * n_det);
        ParamBlock pb;
        const size_t o_ch = pb.push_vec(chunks);
        const size_t o_pi = pb.push(pixel_index, sizeof(int32_t) * n_det);
        const size_t o_wi = pb.push(weight_index, sizeof(int32_t) * n_det);
        const size_t o_fa = pb.push_vec(fa);
        const size_t o_fp = pb.push_vec(fp);
        const char * d = pb.commit(st);
        int * d_status = static_cast<int *>(Manager::get().scratch(Manager::kScratchStatus, 64));
        TH_HIP(hipMemsetAsync(d_status, 0, sizeof(int), st));
        hipLaunchKernelGGL(k_pack_pointing, chunk_grid(n_det, chunks.size()), dim3(kThreads), 0, st,
                           (const Chunk *)(d + o_ch), (int)chunks.size(), (const int32_t *)(d + o_pi),
                           (const int32_t *)(d + o_wi), (const int32_t *)(d + o_fa), (const int32_t *)(d + o_fp), d_g2l,
                           d_pixels, d_weights, d_det_flags, det_flag_mask, use_d, d_shared_flags, shared_flag_mask, use_s,
                           d_proj_flags, proj_flag_mask, use_p, make_fastdiv(n_pix_submap), n_samp, d_key,
                           reinterpret_cast<double2 *>(d_qu), d_cal, d_status);
        check_launch();
        int status = 0;
        copy_to_host(&status, d_status, sizeof(int), st);     // (waits for the stream: once per solve)
        *packable = (status == 0) ? 1 : 0;
        if (status != 0 || pair_words == nullptr) return;
        const int rc = toast_hip_offset_pack_pairs_dev(d_key, n_det, n_samp, intervals, n_view, pair_words, stream);
        if (rc != 0) throw Error(rc, toast_hip_last_error());
    });
}

// Pack, pair words and pair weight sums in ONE sweep (k_pack_pairs) when the rows come in co-pointing pairs; the outputs
// are those of toast_hip_offset_pack_pointing_dev + toast_hip_offset_pack_pair_weights_dev, bit for bit (row 2b + 1 of
// d_key, which the pair-word sweeps never read, is left unwritten).  Pairs that do not share their pixels, rows of an odd
// length or d_corr == NULL: the separate passes as before (*pair_weights = 0, d_corr untouched).
int toast_hip_offset_pack_pointing_onepass_dev(
    const int64_t * d_g2l, int64_t n_pix_submap, const int32_t * pixel_index, const int64_t * d_pixels,
    const int32_t * weight_index, const double * d_weights, const int32_t * acc_flag_index, const uint8_t * d_det_flags,
    int64_t n_flag_samp, uint8_t det_flag_mask, const uint8_t * d_shared_flags, int64_t n_shared_flags,
    uint8_t shared_flag_mask, const int32_t * proj_flag_index, const uint8_t * d_proj_flags, int64_t n_proj_flag_samp,
    uint8_t proj_flag_mask, int64_t n_det, int64_t n_samp, const toast_hip_interval * intervals, int64_t n_view,
    uint32_t * d_key, double * d_qu, double * d_cal, float * d_corr, int * packable, int * pair_words, int * pair_weights,
    void * stream) {
    auto separate = [&] {
        return toast_hip_offset_pack_pointing_dev(d_g2l, n_pix_submap, pixel_index, d_pixels, weight_index, d_weights,
                                                  acc_flag_index, d_det_flags, n_flag_samp, det_flag_mask, d_shared_flags,
                                                  n_shared_flags, shared_flag_mask, proj_flag_index, d_proj_flags,
                                                  n_proj_flag_samp, proj_flag_mask, n_det, n_samp, intervals, n_view, d_key,
                                                  d_qu, d_cal, packable, pair_words, stream);
    };
    if (packable == nullptr || pair_words == nullptr || pair_weights == nullptr) {
        return guarded([&] { fail_arg("offset_pack_pointing_onepass: the three result words must not be null"); });
    }
    *pair_weights = 0;
    static const bool off = [] {
        const char * e = std::getenv("TOAST_HIP_PACK_ONEPASS");
        return e != nullptr && e[0] == '0';
    }();
    if (off || d_corr == nullptr || !pair_detectors() || n_det <= 0 || (n_samp & 1) != 0) return separate();
    int status = 0;
    const int rc = guarded([&] {
        *packable = 0;
        *pair_words = 0;
        if (n_pix_submap <= 0) fail_arg("n_pix_submap must be positive");
        need_aligned(d_qu, "packed Q / U weights");
        need_aligned(d_key, "packed pixel words");
        need_aligned(d_corr, "pair weight sums");
        const auto chunks = make_chunks(intervals, n_view, n_samp);
        if (chunks.empty()) return;
        hipStream_t st = as_stream(stream);
        const int use_d = (n_flag_samp == n_samp) ? 1 : 0;
        const int use_s = (n_shared_flags == n_samp) ? 1 : 0;
        const int use_p = (n_proj_flag_samp == n_samp) ? 1 : 0;
        std::vector<int32_t> fa(n_det, 0), fp(n_det, 0);
        if (use_d) std::memcpy(fa.data(), acc_flag_index, sizeof(int32_t) * n_det);
        if (use_p) std::memcpy(fp.data(), proj_flag_index, sizeof(int32_t) * n_det);
        ParamBlock pb;
        const size_t o_ch = pb.push_vec(chunks);
        const size_t o_pi = pb.push(pixel_index, sizeof(int32_t) * n_det);
        const size_t o_wi = pb.push(weight_index, sizeof(int32_t) * n_det);
        const size_t o_fa = pb.push_vec(fa);
        const size_t o_fp = pb.push_vec(fp);
        const char * d = pb.commit(st);
        char * scratch = static_cast<char *>(Manager::get().scratch(Manager::kScratchStatus, 64));
        TH_HIP(hipMemsetAsync(scratch, 0, 16, st));
        const dim3 gp((unsigned)((n_det + 1) / 2), chunk_grid(n_det, chunks.size()).y, 1);
        hipLaunchKernelGGL(k_pack_pairs, gp, dim3(kThreads), 0, st, (const Chunk *)(d + o_ch), (int)chunks.size(), (int)n_det,
                           (const int32_t *)(d + o_pi), (const int32_t *)(d + o_wi), (const int32_t *)(d + o_fa),
                           (const int32_t *)(d + o_fp), d_g2l, d_pixels, d_weights, d_det_flags, det_flag_mask, use_d,
                           d_shared_flags, shared_flag_mask, use_s, d_proj_flags, proj_flag_mask, use_p,
                           make_fastdiv(n_pix_submap), n_samp, d_key, reinterpret_cast<double2 *>(d_qu), d_cal,
                           reinterpret_cast<float2 *>(d_corr), reinterpret_cast<int *>(scratch),
                           reinterpret_cast<unsigned long long *>(scratch + 8));
        check_launch();
        unsigned long long words[2] = {0, 0};
        copy_to_host(words, scratch, sizeof(words), st);     // (waits for the stream: once per solve)
        status = (int)(words[0] & 0xffffffffull);
        if (status != 0) return;
        *packable = 1;
        *pair_words = 1;
        unsigned long long in_view = 0;
        for (const Chunk & c : chunks) in_view += (unsigned long long)c.count;
        const unsigned long long total = 2ull * in_view * (unsigned long long)(n_det / 2);
        *pair_weights = (words[1] * 100ull > total) ? 0 : 1;      // (as toast_hip_offset_pack_pair_weights_dev)
    });
    if (rc != 0) return rc;
    if ((status & 3) != 0) return 0;               // not packable at all (*packable = 0)
    if ((status & 4) != 0) return separate();      // the pairs do not co-point: plain words, every row on its own
    return 0;
}

// Co-pointing pairs: one word per pair-sample when every pair agrees on its pixels (see k_pair_check / k_pair_merge).
// Separate from the packing so that a caller can pack its rows in several calls (batches of detectors expanded from the
// boresight, never all in memory at once) and merge at the end.
int toast_hip_offset_pack_pairs_dev(uint32_t * d_key, int64_t n_det, int64_t n_samp, const toast_hip_interval * intervals,
                                    int64_t n_view, int * pair_words, void * stream) {
    return guarded([&] {
        if (pair_words == nullptr) fail_arg("offset_pack_pairs: pair_words must not be null");
        *pair_words = 0;
        if (n_det <= 0 || !pair_detectors()) return;
        const auto chunks = make_chunks(intervals, n_view, n_samp);
        if (chunks.empty()) return;
        hipStream_t st = as_stream(stream);
        ParamBlock pb;
        const size_t o_ch = pb.push_vec(chunks);
        const char * d = pb.commit(st);
        int * d_status = static_cast<int *>(Manager::get().scratch(Manager::kScratchStatus, 64));
        TH_HIP(hipMemsetAsync(d_status, 0, sizeof(int), st));
        const dim3 gp((unsigned)((n_det + 1) / 2), chunk_grid(n_det, chunks.size()).y, 1);
        hipLaunchKernelGGL(k_pair_check, gp, dim3(kThreads), 0, st, (const Chunk *)(d + o_ch), (int)chunks.size(), (int)n_det,
                           d_key, n_samp, d_status);
        check_launch();
        int status = 0;
        copy_to_host(&status, d_status, sizeof(int), st);
        if (status != 0) return;
        hipLaunchKernelGGL(k_pair_merge, gp, dim3(kThreads), 0, st, (const Chunk *)(d + o_ch), (int)chunks.size(), (int)n_det,
                           d_key, n_samp);
        check_launch();
        *pair_words = 1;
    });
}

// Pair weights: d_corr[(n_det + 1) / 2][n_samp] float2 = q_a + q_b, u_a + u_b of every pair-sample in view where that fits a
// float exactly and gives the partner's weight back exactly, a NaN marker where not (see k_pair_weights); *ok = 0 when more
// than one component in a hundred is marked (the sweeps then read both rows of d_qu as before).  d_qu stays as it is: the
// sweeps read the partner's row behind a marker.
int toast_hip_offset_pack_pair_weights_dev(const double * d_qu, float * d_corr, int64_t n_det, int64_t n_samp,
                                           const toast_hip_interval * intervals, int64_t n_view, int * ok, void * stream) {
    return guarded([&] {
        if (ok == nullptr) fail_arg("offset_pack_pair_weights: ok must not be null");
        *ok = 0;
        if (n_det <= 0 || !pair_detectors()) return;
        if ((n_samp & 1) != 0) fail_arg("offset_pack_pair_weights: rows of an even number of samples only");
        need_aligned(d_qu, "packed Q / U weights");
        need_aligned(d_corr, "pair weight sums");
        const auto chunks = make_chunks(intervals, n_view, n_samp);
        if (chunks.empty()) return;
        hipStream_t st = as_stream(stream);
        ParamBlock pb;
        const size_t o_ch = pb.push_vec(chunks);
        const char * d = pb.commit(st);
        unsigned long long * d_marked =
            reinterpret_cast<unsigned long long *>(static_cast<char *>(Manager::get().scratch(Manager::kScratchStatus, 64)) + 8);
        TH_HIP(hipMemsetAsync(d_marked, 0, sizeof(unsigned long long), st));
        const dim3 gp((unsigned)((n_det + 1) / 2), chunk_grid(n_det, chunks.size()).y, 1);
        hipLaunchKernelGGL(k_pair_weights, gp, dim3(kThreads), 0, st, (const Chunk *)(d + o_ch), (int)chunks.size(), (int)n_det,
                           reinterpret_cast<const double2 *>(d_qu), n_samp, reinterpret_cast<float2 *>(d_corr), d_marked);
        check_launch();
        unsigned long long marked = 0;
        copy_to_host(&marked, d_marked, sizeof(marked), st);
        // components in view: 2 per pair-sample; worth using while at most one in a hundred needs the partner's row (a
        // marked sample costs up to a 128-byte line of that row: 1 % of them ~1.3 B per sample against the 4 B saved)
        unsigned long long in_view = 0;
        for (const Chunk & c : chunks) in_view += (unsigned long long)c.count;
        const unsigned long long total = 2ull * in_view * (unsigned long long)(n_det / 2);
        const int status = (marked * 100ull > total) ? 1 : 0;
        *ok = (status == 0) ? 1 : 0;
    });
}

namespace {
// The wave-uniform amplitude look-up of the pair-word sweeps needs baselines of at least a wave's 128 samples
// (TOAST_HIP_PACKED_UNIFORM_AMPS=0 keeps the per-lane look-up: A/B runs, tests).
bool uniform_amplitudes(int64_t step_length, const uint8_t * d_amplitude_flags) {
    const char * e = std::getenv("TOAST_HIP_PACKED_UNIFORM_AMPS");
    if (e != nullptr && e[0] == '0') return false;
    // (the flags are read as aligned dwords: an interior pointer that is not 4-byte aligned keeps the per-lane look-up)
    return step_length >= 128 && (reinterpret_cast<uintptr_t>(d_amplitude_flags) & 3u) == 0;
}
}  // namespace

int toast_hip_offset_accumulate_packed_dev(
    int64_t step_length, const int64_t * amp_offsets, const int64_t * n_amp_views, const double * d_amplitudes,
    const uint8_t * d_amplitude_flags, double * d_zmap, const uint32_t * d_key, const double * d_qu, const double * d_cal,
    const double * det_scale, int pair_words, const float * d_pair_corr, int64_t n_det, int64_t n_samp,
    const toast_hip_interval * intervals, int64_t n_view, void * stream) {
    return guarded([&] {
        if (d_pair_corr != nullptr && !pair_words) fail_arg("offset_accumulate_packed: pair weight sums need pair words");
        if (n_det <= 0) return;
        if (step_length <= 0) fail_arg("step_length must be positive");
        if ((n_samp & 1) != 0) fail_arg("offset_accumulate_packed: rows of an even number of samples only");
        need_aligned(d_qu, "packed Q / U weights");
        need_aligned(d_key, "packed pixel words");
        const auto chunks = make_chunks(intervals, n_view, n_samp);
        if (chunks.empty()) return;
        const OffsetViews ov = offset_views(intervals, n_amp_views, n_view);
        ParamBlock pb;
        const size_t o_ch = pb.push_vec(chunks);
        const size_t o_vf = pb.push_vec(ov.first);
        const size_t o_va = pb.push_vec(ov.aoff);
        const size_t o_ao = pb.push(amp_offsets, sizeof(int64_t) * n_det);
        const size_t o_ds = pb.push(det_scale, sizeof(double) * n_det);
        hipStream_t st = as_stream(stream);
        const char * d = pb.commit(st);
        const dim3 grid = chunk_grid(n_det, chunks.size());
        const bool pr = pair_words != 0 || (pair_detectors() && n_det >= 2);
        const dim3 gp((unsigned)(pr ? (n_det + 1) / 2 : n_det), grid.y, 1);
        const bool uni = uniform_amplitudes(step_length, d_amplitude_flags);
#define TH_PK_ARGS                                                                                                  \
    (const Chunk *)(d + o_ch), (int)chunks.size(), (int)n_det, (const int64_t *)(d + o_vf), (const int64_t *)(d + o_va), \
        make_fastdiv(step_length), (const int64_t *)(d + o_ao), d_amplitudes, d_amplitude_flags,                    \
        (const double *)(d + o_ds), d_cal, d_zmap, d_key, reinterpret_cast<const double2 *>(d_qu), n_samp
        if (pair_words && d_pair_corr != nullptr) {
            need_aligned(d_pair_corr, "pair weight sums");
            const float2 * pc = reinterpret_cast<const float2 *>(d_pair_corr);
            if (uni) hipLaunchKernelGGL((k_offset_accumulate_pr<true, true>), gp, dim3(kThreads), 0, st, TH_PK_ARGS, pc);
            else hipLaunchKernelGGL((k_offset_accumulate_pr<true, false>), gp, dim3(kThreads), 0, st, TH_PK_ARGS, pc);
        } else if (pair_words) {
            const float2 * pc = nullptr;
            if (uni) hipLaunchKernelGGL((k_offset_accumulate_pr<false, true>), gp, dim3(kThreads), 0, st, TH_PK_ARGS, pc);
            else hipLaunchKernelGGL((k_offset_accumulate_pr<false, false>), gp, dim3(kThreads), 0, st, TH_PK_ARGS, pc);
        } else if (pr) {
            hipLaunchKernelGGL((k_offset_accumulate_pk<2>), gp, dim3(kThreads), 0, st, TH_PK_ARGS);
        } else {
            hipLaunchKernelGGL((k_offset_accumulate_pk<1>), gp, dim3(kThreads), 0, st, TH_PK_ARGS);
        }
#undef TH_PK_ARGS
        check_launch();
    });
}

int toast_hip_offset_scan_project_packed_dev(
    int64_t step_length, const int64_t * amp_offsets, const int64_t * n_amp_views, const double * d_amps_in,
    double * d_amps_out, const uint8_t * d_amplitude_flags, const double * d_map, const uint32_t * d_key,
    const double * d_qu, const double * d_cal, const double * det_weights, int pair_words, const float * d_pair_corr,
    int64_t n_det, int64_t n_samp, const toast_hip_interval * intervals, int64_t n_view, void * stream) {
    return guarded([&] {
        if (n_det <= 0) return;
        if (d_pair_corr != nullptr && !pair_words) fail_arg("offset_scan_project_packed: pair weight sums need pair words");
        if (step_length <= 0) fail_arg("step_length must be positive");
        if ((n_samp & 1) != 0) fail_arg("offset_scan_project_packed: rows of an even number of samples only");
        need_aligned(d_qu, "packed Q / U weights");
        need_aligned(d_key, "packed pixel words");
        const auto chunks = make_chunks(intervals, n_view, n_samp);
        if (chunks.empty()) return;
        const OffsetViews ov = offset_views(intervals, n_amp_views, n_view);
        ParamBlock pb;
        const size_t o_ch = pb.push_vec(chunks);
        const size_t o_vf = pb.push_vec(ov.first);
        const size_t o_va = pb.push_vec(ov.aoff);
        const size_t o_ao = pb.push(amp_offsets, sizeof(int64_t) * n_det);
        const size_t o_dw = pb.push(det_weights, sizeof(double) * n_det);
        hipStream_t st = as_stream(stream);
        const char * d = pb.commit(st);
        if (pair_words) {
            const dim3 gp((unsigned)((n_det + 1) / 2), chunk_grid(n_det, chunks.size()).y, 1);
#define TH_PR_ARGS                                                                                                   \
    (const Chunk *)(d + o_ch), (int)chunks.size(), (int)n_det, (const int64_t *)(d + o_vf), (const int64_t *)(d + o_va),  \
        make_fastdiv(step_length), (const int64_t *)(d + o_ao), d_amps_in, d_amps_out, d_amplitude_flags,            \
        (const double *)(d + o_dw), d_cal, d_map, d_key, reinterpret_cast<const double2 *>(d_qu), n_samp
            const bool uni = uniform_amplitudes(step_length, d_amplitude_flags);
            if (d_pair_corr != nullptr) {
                need_aligned(d_pair_corr, "pair weight sums");
                const float2 * pc = reinterpret_cast<const float2 *>(d_pair_corr);
                if (uni) hipLaunchKernelGGL((k_offset_scan_project_pr<true, true>), gp, dim3(kThreads), 0, st, TH_PR_ARGS, pc);
                else hipLaunchKernelGGL((k_offset_scan_project_pr<true, false>), gp, dim3(kThreads), 0, st, TH_PR_ARGS, pc);
            } else {
                const float2 * pc = nullptr;
                if (uni) hipLaunchKernelGGL((k_offset_scan_project_pr<false, true>), gp, dim3(kThreads), 0, st, TH_PR_ARGS, pc);
                else hipLaunchKernelGGL((k_offset_scan_project_pr<false, false>), gp, dim3(kThreads), 0, st, TH_PR_ARGS, pc);
            }
#undef TH_PR_ARGS
            check_launch();
            return;
        }
        hipLaunchKernelGGL(k_offset_scan_project_pk, chunk_grid(n_det, chunks.size()), dim3(kThreads), 0, st,
                           (const Chunk *)(d + o_ch), (int)chunks.size(), (const int64_t *)(d + o_vf),
                           (const int64_t *)(d + o_va), make_fastdiv(step_length), (const int64_t *)(d + o_ao), d_amps_in,
                           d_amps_out, d_amplitude_flags, (const double *)(d + o_dw), d_cal, d_map, d_key,
                           reinterpret_cast<const double2 *>(d_qu), n_samp);
        check_launch();
    });
}

}  // extern "C"
