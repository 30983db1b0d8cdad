// kernels.hip -- hand-written gfx950 kernels of the map-making hot path and their
// device-pointer level C entry points (toast_hip_*_dev in include/toast_hip.h).
//
// Launch shape shared by every per-sample kernel: one workgroup (256 threads = 4 wave64)
// per (chunk, detector); blockIdx.x = detector is the FAST grid index, so the workgroups
// resident at any moment cover all detectors over a short time span.  Detectors of one
// focalplane look at the same few square degrees at the same time, so the map working set
// of the resident workgroups is a few hundred KB and stays in every XCD's 4 MiB L2
// (time-major order; a detector-major grid would sweep the whole 300 MB map instead).
//
// Compiled with -ffp-contract=off (see build.py): the pixel path must reproduce the
// reference's separately rounded operations.
#include <hip/hip_runtime.h>

#include "kernel_common.hpp"

using namespace toast_hip;

namespace {

// ------------------------------------------------------------------------------------
// pointing_detector   [ref: ops_pointing_detector.cpp:33-68]
// R 32 B boresight (shared by all detectors, L2 resident) + 1 B flag, W 32 B.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_pointing_detector(
    const Chunk * __restrict__ chunks, int n_chunks, const double * __restrict__ fp,
    const int32_t * __restrict__ q_idx, const double * __restrict__ bore,
    double * __restrict__ quats, const uint8_t * __restrict__ flags, uint8_t mask, int use_flags,
    int64_t n_samp) {
    const int det = blockIdx.x;
    const double f[4] = {fp[4 * det], fp[4 * det + 1], fp[4 * det + 2], fp[4 * det + 3]};
    double * row = quats + (int64_t)q_idx[det] * n_samp * 4;
    for (int ci = blockIdx.y; ci < n_chunks; ci += gridDim.y) {
        const Chunk c = chunks[ci];
        for (int i = threadIdx.x; i < c.count; i += kThreads) {
            const int64_t s = c.first + i;
            double p[4] = {0.0, 0.0, 0.0, 1.0};
            if (!(use_flags && (flags[s] & mask))) {
                const Quat b = load_quat(bore + 4 * s);
                p[0] = b.x; p[1] = b.y; p[2] = b.z; p[3] = b.w;
            }
            double r[4];
            quat_mult(p, f, r);
            store_quat(row + 4 * s, r);
        }
    }
}

// ------------------------------------------------------------------------------------
// pixels_healpix   [ref: ops_pixels_healpix.cpp:586-666]
// R 32 B quaternion + 1 B flag, W 8 B pixel; hit_submaps written once per run of equal
// submaps inside a wave.
// ------------------------------------------------------------------------------------
// E = 2: detectors 2b and 2b + 1 of the call in one workgroup; the pixel arithmetic runs once for a co-pointing
// (orthogonally polarised) pair -- vec_to_pixel_pair, bit-identical to two separate evaluations.
template <bool NEST, int E>
__global__ __launch_bounds__(kThreads) void k_pixels_healpix(
    const Chunk * __restrict__ chunks, int n_chunks, int n_det, const int32_t * __restrict__ q_idx,
    const int32_t * __restrict__ p_idx, const double * __restrict__ quats,
    const uint8_t * __restrict__ flags, uint8_t mask, int use_flags, int64_t * __restrict__ pixels,
    uint8_t * __restrict__ hsub, FastDiv nps_div, int64_t nside, int factor, int64_t n_samp) {
    __shared__ double s_tab[2 * TOAST_ATAN_TABLE_N];
    if (threadIdx.x < 2 * TOAST_ATAN_TABLE_N) s_tab[threadIdx.x] = kAtanTab[threadIdx.x];
    __syncthreads();

    const double * qrow[E];
    int64_t * prow[E];
    bool valid[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        int det = E * blockIdx.x + e;
        valid[e] = det < n_det;
        if (!valid[e]) det = E * blockIdx.x;
        qrow[e] = quats + (int64_t)q_idx[det] * n_samp * 4;
        prow[e] = pixels + (int64_t)p_idx[det] * n_samp;
    }
    const int lane = threadIdx.x & 63;

    for (int ci = blockIdx.y; ci < n_chunks; ci += gridDim.y) {
        const Chunk c = chunks[ci];
        for (int base = 0; base < c.count; base += kThreads) {
            const int i = base + threadIdx.x;
            const bool active = i < c.count;
            const int64_t s = c.first + (active ? i : 0);
            double dir[E][3];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const Quat q = load_quat(qrow[e] + 4 * s);
                const double qa[4] = {q.x, q.y, q.z, q.w};
                quat_rotate_z(qa, dir[e]);
            }
            int64_t pix[E];
            if constexpr (E == 2) {
                vec_to_pixel_pair<NEST>(dir[0], dir[1], nside, factor, s_tab, pix[0], pix[1]);
            } else {
                pix[0] = vec_to_pixel<NEST>(dir[0], nside, factor, s_tab);
            }
            const bool flagged = use_flags && ((flags[s] & mask) != 0);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                int64_t sub = -1;
                if (flagged) {
                    pix[e] = -1;
                } else {
                    sub = fastdiv(pix[e], nps_div);
                }
                if (!active || !valid[e]) sub = -1;
                const int64_t prev = __shfl_up(sub, 1);
                if (active && valid[e]) {
                    prow[e][s] = pix[e];
                    if (sub >= 0 && (lane == 0 || prev != sub)) hsub[sub] = 1;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// healpix_vec2nest / healpix_vec2ring   [ref: ops_pixels_healpix.cpp:351-381, bindings :816-893]
// Pixel of n direction vectors: the device functions of the pointing kernels (vec_to_pixel).
// ------------------------------------------------------------------------------------
template <bool NEST>
__global__ __launch_bounds__(kThreads) void k_healpix_vec2pix(int64_t n, const double * __restrict__ vec,
                                                             int64_t * __restrict__ pix, int64_t nside, int factor) {
    __shared__ double s_tab[2 * TOAST_ATAN_TABLE_N];
    if (threadIdx.x < 2 * TOAST_ATAN_TABLE_N) s_tab[threadIdx.x] = kAtanTab[threadIdx.x];
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        const double v[3] = {vec[3 * i], vec[3 * i + 1], vec[3 * i + 2]};
        pix[i] = vec_to_pixel<NEST>(v, nside, factor, s_tab);
    }
}

// ------------------------------------------------------------------------------------
// healpix_ang2vec / vec2ang / ang2nest / ang2ring   [ref: ops_pixels_healpix.cpp:278-349, bindings :668-815]
// These go through sin / cos / acos / atan2 of the device math library; glibc's and the device's versions agree to
// an ulp, not bit for bit: vectors and angles are tolerance-class outputs, a pixel number can differ only for an angle
// within an ulp of a pixel or region boundary.  (The pointing kernels do not use them: their only transcendental is the
// double-double atan2 of hpix_math.hpp.)
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_healpix_ang2vec(int64_t n, const double * __restrict__ theta,
                                                             const double * __restrict__ phi,
                                                             double * __restrict__ vec) {
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        const double st = sin(theta[i]);
        vec[3 * i] = st * cos(phi[i]);
        vec[3 * i + 1] = st * sin(phi[i]);
        vec[3 * i + 2] = cos(theta[i]);
    }
}

__global__ __launch_bounds__(kThreads) void k_healpix_vec2ang(int64_t n, const double * __restrict__ vec,
                                                             double * __restrict__ theta, double * __restrict__ phi) {
    const double eps = 2.220446049250313e-16;
    const double pi = 3.14159265358979323846;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        const double x = vec[3 * i], y = vec[3 * i + 1], z = vec[3 * i + 2];
        const double norm = 1.0 / sqrt(x * x + y * y + z * z);
        const double th = acos(z * norm);
        const bool edge = (fabs(th) <= eps) || (fabs(pi - th) <= eps);
        const double pt = atan2(y, x);
        double ph = (pt < 0) ? pt + 2 * pi : pt;
        if (edge) ph = 0.0;
        theta[i] = th;
        phi[i] = ph;
    }
}

template <bool NEST>
__global__ __launch_bounds__(kThreads) void k_healpix_ang2pix(int64_t n, const double * __restrict__ theta,
                                                             const double * __restrict__ phi,
                                                             int64_t * __restrict__ pix, int64_t nside, int factor) {
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        // hpix_theta2z (:304-317): the square root is evaluated in both regions there, read only in the caps
        ZPhi a;
        a.z = cos(theta[i]);
        const double za = fabs(a.z);
        const int s = (a.z > 0.0) ? 1 : -1;
        a.region = (za <= TOAST_TWOTHIRDS) ? s : s + s;
        a.rtz = sqrt(3.0 * (1.0 - za));
        a.phi = phi[i];
        pix[i] = NEST ? zphi_to_nest(nside, factor, a) : zphi_to_ring(nside, factor, a);
    }
}

// ------------------------------------------------------------------------------------
// healpix_ring2nest / nest2ring / degrade_* / upgrade_*   [ref: ops_pixels_healpix.cpp:383-580, bindings :893-1150]
// op 0: ring -> nest, 1: nest -> ring, 2: nest >> 2 levels (degrade), 3: nest << 2 levels (upgrade),
// 4 / 5: the RING forms of 2 / 3 (through NEST at the input resolution and back at the output resolution)
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_healpix_convert(int op, int64_t n, const int64_t * __restrict__ in,
                                                             int64_t * __restrict__ out, int64_t nside, int factor,
                                                             int64_t levels) {
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        const int64_t p = in[i];
        int64_t r;
        if (op == 0) {
            r = ring_to_nest(nside, factor, p);
        } else if (op == 1) {
            r = nest_to_ring(nside, factor, p);
        } else if (op == 2) {
            r = p >> (2 * levels);
        } else if (op == 3) {
            r = p << (2 * levels);
        } else {
            const int64_t nest = ring_to_nest(nside, factor, p);
            if (op == 4) {
                r = nest_to_ring(nside >> levels, factor - (int)levels, nest >> (2 * levels));
            } else {
                r = nest_to_ring(nside << levels, factor + (int)levels, nest << (2 * levels));
            }
        }
        out[i] = r;
    }
}

// ------------------------------------------------------------------------------------
// stokes_weights   [ref: ops_stokes_weights.cpp:77-140, :459-505]
// ------------------------------------------------------------------------------------
template <bool HWP, int NOUT>   // NOUT = 3: (I, Q, U); 2: (Q, U) only (StokesWeights mode "QU")
__global__ __launch_bounds__(kThreads) void k_stokes_iqu(
    const Chunk * __restrict__ chunks, int n_chunks, const int32_t * __restrict__ q_idx,
    const int32_t * __restrict__ w_idx, const double * __restrict__ quats,
    double * __restrict__ weights, const double * __restrict__ hwp,
    const double * __restrict__ epsilon, const double * __restrict__ gamma,
    const double * __restrict__ cal, double usign, int64_t n_samp, int ref_nan) {
    const int det = blockIdx.x;
    const double eps = epsilon[det];
    const double eta = (1.0 - eps) / (1.0 + eps);
    const double cd = cal[det];
    const double gd = gamma[det];
    double c4g = 1.0, s4g = 0.0;
    if (HWP) sincos(4.0 * gd, &s4g, &c4g);
    const double * qrow = quats + (int64_t)q_idx[det] * n_samp * 4;
    double * wrow = weights + (int64_t)w_idx[det] * n_samp * NOUT;
    for (int ci = blockIdx.y; ci < n_chunks; ci += gridDim.y) {
        const Chunk c = chunks[ci];
        for (int i = threadIdx.x; i < c.count; i += kThreads) {
            const int64_t s = c.first + i;
            const Quat q = load_quat(qrow + 4 * s);
            const double qa[4] = {q.x, q.y, q.z, q.w};
            double c2a, s2a;
            stokes_cs2alpha(qa, c2a, s2a, ref_nan != 0);
            double * w = wrow + NOUT * s - (3 - NOUT);   // w[1], w[2] land in the last two slots
            if (HWP) {
                // ang = 2 (2 (gamma - hwp) - alpha) = beta - 2 alpha, beta = 4 gamma - 4 hwp
                double s4h, c4h, sb, cb;
                sincos(4.0 * hwp[s], &s4h, &c4h);
                hwp_rotation(c4g, s4g, c4h, s4h, cb, sb);
                const double cang = cb * c2a + sb * s2a;
                const double sang = sb * c2a - cb * s2a;
                if (NOUT == 3) w[0] = cd;
                w[1] = cang * eta * cd;
                w[2] = -sang * eta * cd * usign;
            } else {
                if (NOUT == 3) w[0] = cd;
                w[1] = c2a * eta * cd;
                w[2] = s2a * eta * cd * usign;
            }
        }
    }
}

__global__ __launch_bounds__(kThreads) void k_stokes_i(
    const Chunk * __restrict__ chunks, int n_chunks, const int32_t * __restrict__ w_idx,
    double * __restrict__ weights, const double * __restrict__ cal, int64_t n_samp) {
    const int det = blockIdx.x;
    const double cd = cal[det];
    double * wrow = weights + (int64_t)w_idx[det] * n_samp;
    for (int ci = blockIdx.y; ci < n_chunks; ci += gridDim.y) {
        const Chunk c = chunks[ci];
        for (int i = threadIdx.x; i < c.count; i += kThreads) wrow[c.first + i] = cd;
    }
}

// ------------------------------------------------------------------------------------
// scan_map   [ref: ops_scan_map.cpp:15-78]
// R 8 B pixel + 8*nnz B weights + 8 B tod (unless zeroing), W 8 B tod; map gather served by
// L2 (see file header).  `det_w` != nullptr fuses the PCG's diagonal noise weight.
// ------------------------------------------------------------------------------------
template <typename T, int NNZ>
__global__ __launch_bounds__(kThreads) void k_scan_map(
    const Chunk * __restrict__ chunks, int n_chunks, const int32_t * __restrict__ d_idx,
    const int32_t * __restrict__ p_idx, const int32_t * __restrict__ w_idx,
    const int64_t * __restrict__ g2l, const T * __restrict__ map, double * __restrict__ tod,
    const int64_t * __restrict__ pixels, const double * __restrict__ weights, int nnz_rt,
    FastDiv nps_div, double scale, int zero, int subtract, int mult,
    const double * __restrict__ det_w, int64_t n_samp, int det_major) {
    const int det = det_major ? blockIdx.y : blockIdx.x;
    const int ci0 = det_major ? blockIdx.x : blockIdx.y;
    const int cstride = det_major ? gridDim.x : gridDim.y;
    const int nnz = (NNZ > 0) ? NNZ : nnz_rt;
    double * drow = tod + (int64_t)d_idx[det] * n_samp;
    const int64_t * prow = pixels + (int64_t)p_idx[det] * n_samp;
    const double * wrow = weights + (int64_t)w_idx[det] * n_samp * nnz;
    const bool fuse = det_w != nullptr;
    const double dw = fuse ? det_w[det] : 1.0;
    const int64_t nps = nps_div.d;
    for (int ci = ci0; ci < n_chunks; ci += cstride) {
        const Chunk c = chunks[ci];
        for (int i = threadIdx.x; i < c.count; i += kThreads) {
            const int64_t s = c.first + i;
            // streaming loads first (one round trip), then the dependent global2local -> map gather
            const int64_t p = prow[s];
            double d = zero ? 0.0 : drow[s];
            const double * w = wrow + nnz * s;
            double wk[(NNZ > 0) ? NNZ : 1];
            if (NNZ > 0) {
#pragma unroll
                for (int k = 0; k < NNZ; ++k) wk[k] = w[k];
            }
            // (a pixel in a submap that is not local leaves the sample alone; the reference reads in front of the map)
            const int64_t gsm = fastdiv(p >= 0 ? p : 0, nps_div);
            const int64_t lsm = g2l[gsm];
            if ((p >= 0) & (lsm >= 0)) {
                const int64_t sub = p - gsm * nps;
                const T * m = map + nnz * (lsm * nps + sub);
                double v = 0.0;
                if (NNZ > 0) {
#pragma unroll
                    for (int k = 0; k < NNZ; ++k) v += wk[k] * (double)m[k];
                } else {
                    for (int k = 0; k < nnz; ++k) v += w[k] * (double)m[k];
                }
                v *= scale;
                if (subtract) {
                    d -= v;
                } else if (mult) {
                    d *= v;
                } else {
                    d += v;
                }
            }
            if (fuse) d *= dw;
            drow[s] = d;
        }
    }
}

// Two consecutive samples per lane: the pixel pair and the timestream pair are one 16-byte lane access each, the two
// weight triples three of them over 48 contiguous bytes, the result one 16-byte store (a wave covers 1 KiB / 3 KiB
// contiguous per load instruction instead of 512 B / three interleaved 24-byte strides).  Needs every row to start on a
// 16-byte boundary (even n_samp, checked by the launcher); a chunk that starts on an odd sample peels it, an odd tail is
// handled by one lane.  Per-sample arithmetic identical to k_scan_map (same expression, same order).
// One sample of scan_map in three straight-line stages, so that a lane holding several samples issues all their
// gathers of one stage together (a test on a loaded value must not guard the next load): a sample without
// a pixel (or whose submap is not local) gathers map element 0 and discards it.
struct ScanGather {
    int64_t off;   // element offset of the pixel's first map value
    bool hit;
};

__device__ __forceinline__ int64_t scan_submap(int64_t p, const FastDiv & nps_div) {
    return fastdiv(p >= 0 ? p : 0, nps_div);
}

__device__ __forceinline__ ScanGather scan_locate(int64_t p, int64_t gsm, int64_t lsm, const FastDiv & nps_div) {
    ScanGather g;
    g.hit = (p >= 0) & (lsm >= 0);
    g.off = g.hit ? 3 * (lsm * nps_div.d + (p - gsm * nps_div.d)) : 0;
    return g;
}

__device__ __forceinline__ double scan_combine(bool hit, double d, double w0, double w1, double w2, double m0, double m1,
                                               double m2, double scale, int subtract, int mult) {
    double v = 0.0;
    v += w0 * m0;
    v += w1 * m1;
    v += w2 * m2;
    v *= scale;
    const double r = subtract ? d - v : (mult ? d * v : d + v);
    return hit ? r : d;
}

// (round 6, tried and taken out: the global2local look-up of a wave's 128 samples as ONE scalar load when they all lie in the
//  same submap -- 6.19-6.23 against 6.08-6.14 ms in alternating processes, profiles/r06_b section 5: unlike the eight
//  amplitude gathers of the packed sweeps, these two sit at the head of the dependent chain pixel -> submap -> map value,
//  and the wave-wide agreement test adds a scalar round trip to it)
template <typename T>
__global__ __launch_bounds__(kThreads) void k_scan_map_v2(
    const Chunk * __restrict__ chunks, int n_chunks, const int32_t * __restrict__ d_idx,
    const int32_t * __restrict__ p_idx, const int32_t * __restrict__ w_idx,
    const int64_t * __restrict__ g2l, const T * __restrict__ map, double * __restrict__ tod,
    const int64_t * __restrict__ pixels, const double * __restrict__ weights,
    FastDiv nps_div, double scale, int zero, int subtract, int mult,
    const double * __restrict__ det_w, int64_t n_samp) {
    const int det = blockIdx.x;
    double * drow = tod + (int64_t)d_idx[det] * n_samp;
    const int64_t * prow = pixels + (int64_t)p_idx[det] * n_samp;
    const double * wrow = weights + (int64_t)w_idx[det] * n_samp * 3;
    const bool fuse = det_w != nullptr;
    const double dw = fuse ? det_w[det] : 1.0;
    for (int ci = blockIdx.y; ci < n_chunks; ci += gridDim.y) {
        const Chunk c = chunks[ci];
        const int head = (int)(c.first & 1);
        const int64_t s0 = c.first + head;          // even
        const int n_pair = (c.count - head) >> 1;
        for (int j = threadIdx.x; j < n_pair; j += kThreads) {
            const int64_t s = s0 + 2 * (int64_t)j;
            const longlong2 pp = *reinterpret_cast<const longlong2 *>(prow + s);
            double2 dd = make_double2(0.0, 0.0);
            if (!zero) dd = *reinterpret_cast<const double2 *>(drow + s);
            const double2 * wv = reinterpret_cast<const double2 *>(wrow + 3 * s);
            const double2 wa = wv[0], wb = wv[1], wc = wv[2];
            const int64_t ga = scan_submap(pp.x, nps_div), gb = scan_submap(pp.y, nps_div);
            const int64_t la = g2l[ga], lb = g2l[gb];
            const ScanGather qa = scan_locate(pp.x, ga, la, nps_div), qb = scan_locate(pp.y, gb, lb, nps_div);
            const T * ma = map + qa.off;
            const T * mb = map + qb.off;
            const T a0 = ma[0], a1 = ma[1], a2 = ma[2], b0 = mb[0], b1 = mb[1], b2 = mb[2];
            dd.x = scan_combine(qa.hit, dd.x, wa.x, wa.y, wb.x, (double)a0, (double)a1, (double)a2, scale, subtract, mult);
            dd.y = scan_combine(qb.hit, dd.y, wb.y, wc.x, wc.y, (double)b0, (double)b1, (double)b2, scale, subtract, mult);
            if (fuse) {
                dd.x *= dw;
                dd.y *= dw;
            }
            *reinterpret_cast<double2 *>(drow + s) = dd;
        }
        // the peeled first sample and the odd last one: lanes 0 and 1 of the workgroup
        const int tail = (c.count - head) & 1;
        if ((threadIdx.x == 0 && head) || (threadIdx.x == 1 && tail)) {
            const int64_t s = (threadIdx.x == 0) ? c.first : c.first + c.count - 1;
            const double * w = wrow + 3 * s;
            const int64_t p = prow[s];
            const int64_t gs = scan_submap(p, nps_div);
            const ScanGather q = scan_locate(p, gs, g2l[gs], nps_div);
            const T * m = map + q.off;
            double d = scan_combine(q.hit, zero ? 0.0 : drow[s], w[0], w[1], w[2], (double)m[0], (double)m[1], (double)m[2],
                                    scale, subtract, mult);
            if (fuse) d *= dw;
            drow[s] = d;
        }
    }
}

// ------------------------------------------------------------------------------------
// build_noise_weighted   [ref: ops_mapmaker_utils.cpp:15-86]
// R 8 B pixel + 8*nnz B weights + 8 B tod + 1 B det flag (+ shared flag); the scatter is
// reduced per run of equal pixels inside each wave (a satellite scan revisits the same pixel
// for ~10-20 consecutive samples), leaving one hardware fp64 atomic per run and component.
// ------------------------------------------------------------------------------------
template <int NNZ>
__global__ __launch_bounds__(kThreads) void k_build_noise_weighted(
    const Chunk * __restrict__ chunks, int n_chunks, const int32_t * __restrict__ p_idx,
    const int32_t * __restrict__ w_idx, const int32_t * __restrict__ d_idx,
    const int32_t * __restrict__ f_idx, const double * __restrict__ det_scale,
    const int64_t * __restrict__ g2l, double * __restrict__ zmap,
    const int64_t * __restrict__ pixels, const double * __restrict__ weights,
    const double * __restrict__ tod, const uint8_t * __restrict__ dflags, uint8_t dmask,
    int use_dflags, const uint8_t * __restrict__ sflags, uint8_t smask, int use_sflags,
    FastDiv nps_div, int64_t n_samp, int det_major) {
    const int det = det_major ? blockIdx.y : blockIdx.x;
    const int ci0 = det_major ? blockIdx.x : blockIdx.y;
    const int cstride = det_major ? gridDim.x : gridDim.y;
    const int64_t * prow = pixels + (int64_t)p_idx[det] * n_samp;
    const double * wrow = weights + (int64_t)w_idx[det] * n_samp * NNZ;
    const double * drow = tod + (int64_t)d_idx[det] * n_samp;
    const uint8_t * frow = use_dflags ? dflags + (int64_t)f_idx[det] * n_samp : nullptr;
    const double ds = det_scale[det];
    const int64_t nps = nps_div.d;
    for (int ci = ci0; ci < n_chunks; ci += cstride) {
        const Chunk c = chunks[ci];
        for (int base = 0; base < c.count; base += kThreads) {
            const int i = base + threadIdx.x;
            const bool active = i < c.count;
            const int64_t s = c.first + (active ? i : 0);
            int64_t key = -1;
            double v[NNZ];
#pragma unroll
            for (int k = 0; k < NNZ; ++k) v[k] = 0.0;
            if (active) {
                // every streaming load is issued before the first use: one memory round trip for
                // pixel / flags / tod / weights, a second one for global2local
                const int64_t p = prow[s];
                const uint8_t fd = use_dflags ? frow[s] : (uint8_t)0;
                const uint8_t fs = use_sflags ? sflags[s] : (uint8_t)0;
                const double t = drow[s];
                const double * w = wrow + NNZ * s;
                double wk[NNZ];
#pragma unroll
                for (int k = 0; k < NNZ; ++k) wk[k] = w[k];
                const bool good = (p >= 0) & ((fd & dmask) == 0) & ((fs & smask) == 0);
                if (good) {
                    const int64_t gsm = fastdiv(p, nps_div);
                    key = g2l[gsm] * nps + (p - gsm * nps);
                    const double sd = t * ds;
#pragma unroll
                    for (int k = 0; k < NNZ; ++k) v[k] = sd * wk[k];
                }
            }
            const bool tail = wave_run_reduce<NNZ>(key, v);
            if (tail && key >= 0) {
                double * z = zmap + NNZ * key;
#pragma unroll
                for (int k = 0; k < NNZ; ++k) unsafeAtomicAdd(z + k, v[k]);
            }
        }
    }
}

// Detector-pair variant.  The fp64 atomic unit, not HBM, bounds the scatter once runs get short
// (measured ~25 G atomics/s device-wide, independent of which XCD issues them:
// profiles/r01_b_tuning_experiments.txt §6), and focalplanes carry two orthogonally polarised
// detectors per pixel that see the same sky pixel at the same instant.  One workgroup therefore
// processes detectors (2b, 2b+1) of the call together: when every lane of the wave finds the
// two keys equal (or one of them invalid) the contributions are added *before* the run
// reduction -- half the scans and half the atomics; any other wave falls back to two passes.
template <int NNZ>
__global__ __launch_bounds__(kThreads) void k_build_noise_weighted_pair(
    const Chunk * __restrict__ chunks, int n_chunks, int n_det, const int32_t * __restrict__ p_idx,
    const int32_t * __restrict__ w_idx, const int32_t * __restrict__ d_idx,
    const int32_t * __restrict__ f_idx, const double * __restrict__ det_scale,
    const int64_t * __restrict__ g2l, double * __restrict__ zmap,
    const int64_t * __restrict__ pixels, const double * __restrict__ weights,
    const double * __restrict__ tod, const uint8_t * __restrict__ dflags, uint8_t dmask,
    int use_dflags, const uint8_t * __restrict__ sflags, uint8_t smask, int use_sflags,
    FastDiv nps_div, int64_t n_samp) {
    const int det0 = 2 * blockIdx.x;
    const bool two = det0 + 1 < n_det;
    const int det1 = two ? det0 + 1 : det0;
    const int64_t * prow[2] = {pixels + (int64_t)p_idx[det0] * n_samp, pixels + (int64_t)p_idx[det1] * n_samp};
    const double * wrow[2] = {weights + (int64_t)w_idx[det0] * n_samp * NNZ,
                              weights + (int64_t)w_idx[det1] * n_samp * NNZ};
    const double * drow[2] = {tod + (int64_t)d_idx[det0] * n_samp, tod + (int64_t)d_idx[det1] * n_samp};
    const uint8_t * frow[2] = {use_dflags ? dflags + (int64_t)f_idx[det0] * n_samp : nullptr,
                               use_dflags ? dflags + (int64_t)f_idx[det1] * n_samp : nullptr};
    const double ds[2] = {det_scale[det0], det_scale[det1]};
    const int64_t nps = nps_div.d;
    for (int ci = blockIdx.y; ci < n_chunks; ci += gridDim.y) {
        const Chunk c = chunks[ci];
        for (int base = 0; base < c.count; base += kThreads) {
            const int i = base + threadIdx.x;
            const bool active = i < c.count;
            const int64_t s = c.first + (active ? i : 0);
            int64_t key[2] = {-1, -1};
            double v[2][NNZ];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
#pragma unroll
                for (int k = 0; k < NNZ; ++k) v[e][k] = 0.0;
            }
            if (active) {
                // all streaming loads of both detectors first
                int64_t p[2];
                uint8_t fd[2];
                double t[2], wk[2][NNZ];
                const uint8_t fs = use_sflags ? sflags[s] : (uint8_t)0;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    p[e] = prow[e][s];
                    fd[e] = use_dflags ? frow[e][s] : (uint8_t)0;
                    t[e] = drow[e][s];
                    const double * w = wrow[e] + NNZ * s;
#pragma unroll
                    for (int k = 0; k < NNZ; ++k) wk[e][k] = w[k];
                }
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const bool good = (p[e] >= 0) & ((fd[e] & dmask) == 0) & ((fs & smask) == 0) & (e == 0 || two);
                    if (good) {
                        const int64_t gsm = fastdiv(p[e], nps_div);
                        key[e] = g2l[gsm] * nps + (p[e] - gsm * nps);
                        const double sd = t[e] * ds[e];
#pragma unroll
                        for (int k = 0; k < NNZ; ++k) v[e][k] = sd * wk[e][k];
                    }
                }
            }
            scatter_runs<NNZ, 2>(key, v, zmap);
        }
    }
}

// Two consecutive samples per lane and E = 1 or 2 detectors per workgroup (nnz = 3): 16-byte lane accesses for pixels
// and timestream, 3 x 16 bytes for the two weight triples, 2 bytes for the two flags; one segmented scan per 128
// samples (scatter_runs2).  Rows start on 16-byte boundaries (even n_samp, checked by the launcher); the odd first /
// last sample of a chunk is added by one lane with plain atomics.
template <int E>
__global__ __launch_bounds__(kThreads) void k_build_noise_weighted_v2(
    const Chunk * __restrict__ chunks, int n_chunks, int n_det, const int32_t * __restrict__ p_idx,
    const int32_t * __restrict__ w_idx, const int32_t * __restrict__ d_idx,
    const int32_t * __restrict__ f_idx, const double * __restrict__ det_scale,
    const int64_t * __restrict__ g2l, double * __restrict__ zmap,
    const int64_t * __restrict__ pixels, const double * __restrict__ weights,
    const double * __restrict__ tod, const uint8_t * __restrict__ dflags, uint8_t dmask,
    int use_dflags, const uint8_t * __restrict__ sflags, uint8_t smask, int use_sflags,
    FastDiv nps_div, int64_t n_samp) {
    constexpr int NNZ = 3;
    const int det0 = E * blockIdx.x;
    int dets[E];
    bool on[E];
    const int64_t * prow[E];
    const double * wrow[E];
    const double * drow[E];
    const uint8_t * frow[E];
    double ds[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        on[e] = det0 + e < n_det;
        dets[e] = on[e] ? det0 + e : det0;
        prow[e] = pixels + (int64_t)p_idx[dets[e]] * n_samp;
        wrow[e] = weights + (int64_t)w_idx[dets[e]] * n_samp * NNZ;
        drow[e] = tod + (int64_t)d_idx[dets[e]] * n_samp;
        frow[e] = use_dflags ? dflags + (int64_t)f_idx[dets[e]] * n_samp : nullptr;
        ds[e] = det_scale[dets[e]];
    }
    const int64_t nps = nps_div.d;
    const uint16_t dmask2 = (uint16_t)(dmask | (dmask << 8));
    const uint16_t smask2 = (uint16_t)(smask | (smask << 8));
    for (int ci = blockIdx.y; ci < n_chunks; ci += gridDim.y) {
        const Chunk c = chunks[ci];
        const int head = (int)(c.first & 1);
        const int64_t s0 = c.first + head;          // even
        const int n_pair = (c.count - head) >> 1;
        for (int base = 0; base < n_pair; base += kThreads) {
            const int j = base + threadIdx.x;
            const bool active = j < n_pair;
            const int64_t s = s0 + 2 * (int64_t)(active ? j : 0);
            int64_t ka[E], kb[E];
            double va[E][NNZ], vb[E][NNZ];
            // all streaming loads of both samples of all detectors first (inactive lanes re-read pair 0: same lines)
            longlong2 pp[E];
            double2 tt[E], w0[E], w1[E], w2[E];
            uint16_t fd[E];
            const uint16_t fs = use_sflags ? *reinterpret_cast<const uint16_t *>(sflags + s) : (uint16_t)0;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                pp[e] = *reinterpret_cast<const longlong2 *>(prow[e] + s);
                fd[e] = use_dflags ? *reinterpret_cast<const uint16_t *>(frow[e] + s) : (uint16_t)0;
                tt[e] = *reinterpret_cast<const double2 *>(drow[e] + s);
                const double2 * wv = reinterpret_cast<const double2 *>(wrow[e] + NNZ * s);
                w0[e] = wv[0];
                w1[e] = wv[1];
                w2[e] = wv[2];
            }
            // global2local of every sample in one round trip (sample without a pixel: submap 0, discarded)
            int64_t ga[E], gb[E], la[E], lb[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                ga[e] = fastdiv(pp[e].x >= 0 ? pp[e].x : 0, nps_div);
                gb[e] = fastdiv(pp[e].y >= 0 ? pp[e].y : 0, nps_div);
                la[e] = g2l[ga[e]];
                lb[e] = g2l[gb[e]];
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const uint16_t bad = (uint16_t)((fd[e] & dmask2) | (fs & smask2));
                // (la / lb < 0: the pixel lies in a submap that is not local -- no contribution, also not through the pair merge)
                const bool good_a = active & on[e] & (pp[e].x >= 0) & (la[e] >= 0) & ((bad & 0x00ff) == 0);
                const bool good_b = active & on[e] & (pp[e].y >= 0) & (lb[e] >= 0) & ((bad & 0xff00) == 0);
                ka[e] = good_a ? la[e] * nps + (pp[e].x - ga[e] * nps) : -1;
                kb[e] = good_b ? lb[e] * nps + (pp[e].y - gb[e] * nps) : -1;
                const double sa = tt[e].x * ds[e], sb = tt[e].y * ds[e];
                va[e][0] = good_a ? sa * w0[e].x : 0.0;
                va[e][1] = good_a ? sa * w0[e].y : 0.0;
                va[e][2] = good_a ? sa * w1[e].x : 0.0;
                vb[e][0] = good_b ? sb * w1[e].y : 0.0;
                vb[e][1] = good_b ? sb * w2[e].x : 0.0;
                vb[e][2] = good_b ? sb * w2[e].y : 0.0;
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
        // the peeled first sample (lane 0) and the odd last one (lane 1)
        const int tail = (c.count - head) & 1;
        if ((threadIdx.x == 0 && head) || (threadIdx.x == 1 && tail)) {
            const int64_t s = (threadIdx.x == 0) ? c.first : c.first + c.count - 1;
            const uint8_t fs = use_sflags ? sflags[s] : (uint8_t)0;
            for (int e = 0; e < E; ++e) {
                if (!on[e]) continue;
                const int64_t p = prow[e][s];
                const uint8_t fd = use_dflags ? frow[e][s] : (uint8_t)0;
                if ((p < 0) | ((fd & dmask) != 0) | ((fs & smask) != 0)) continue;
                const int64_t gsm = fastdiv(p, nps_div);
                const int64_t key = g2l[gsm] * nps + (p - gsm * nps);
                if (key < 0) continue;
                const double sd = drow[e][s] * ds[e];
                const double * w = wrow[e] + NNZ * s;
                double * z = zmap + NNZ * key;
                for (int k = 0; k < NNZ; ++k) unsafeAtomicAdd(z + k, sd * w[k]);
            }
        }
    }
}

// generic nnz (rare: nnz not in {1,2,3}): plain per-sample atomics
__global__ __launch_bounds__(kThreads) void k_build_noise_weighted_any(
    const Chunk * __restrict__ chunks, int n_chunks, const int32_t * __restrict__ p_idx,
    const int32_t * __restrict__ w_idx, const int32_t * __restrict__ d_idx,
    const int32_t * __restrict__ f_idx, const double * __restrict__ det_scale,
    const int64_t * __restrict__ g2l, double * __restrict__ zmap,
    const int64_t * __restrict__ pixels, const double * __restrict__ weights,
    const double * __restrict__ tod, const uint8_t * __restrict__ dflags, uint8_t dmask,
    int use_dflags, const uint8_t * __restrict__ sflags, uint8_t smask, int use_sflags,
    FastDiv nps_div, int64_t n_samp, int nnz) {
    const int det = blockIdx.x;
    const int64_t * prow = pixels + (int64_t)p_idx[det] * n_samp;
    const double * wrow = weights + (int64_t)w_idx[det] * n_samp * nnz;
    const double * drow = tod + (int64_t)d_idx[det] * n_samp;
    const uint8_t * frow = use_dflags ? dflags + (int64_t)f_idx[det] * n_samp : nullptr;
    const double ds = det_scale[det];
    const int64_t nps = nps_div.d;
    for (int ci = blockIdx.y; ci < n_chunks; ci += gridDim.y) {
        const Chunk c = chunks[ci];
        for (int i = threadIdx.x; i < c.count; i += kThreads) {
            const int64_t s = c.first + i;
            const int64_t p = prow[s];
            if (p < 0) continue;
            if (use_dflags && (frow[s] & dmask)) continue;
            if (use_sflags && (sflags[s] & smask)) continue;
            const int64_t gsm = fastdiv(p, nps_div);
            double * z = zmap + nnz * (g2l[gsm] * nps + (p - gsm * nps));
            const double sd = drow[s] * ds;
            const double * w = wrow + nnz * s;
            for (int k = 0; k < nnz; ++k) unsafeAtomicAdd(z + k, sd * w[k]);
        }
    }
}

// ------------------------------------------------------------------------------------
// Hit map and inverse pixel covariance accumulation (BuildHitMap / BuildInverseCovariance,
// src/toast/ops/mapmaker_utils/mapmaker_utils.py:100-210, :352-520; per-sample arithmetic of
// cov_accum_diag_hits / cov_accum_diag_invnpp, src/libtoast/src/toast_map_cov.cpp:66-153).
// Same run-reduced scatter as build_noise_weighted.  MODE 0: hits[pix] += 1 (int64);
// MODE 1: invnpp[pix, (j,k>=j)] += w_k * (w_j * det_scale).
// ------------------------------------------------------------------------------------
template <int NNZ, int MODE>
__global__ __launch_bounds__(kThreads) void k_build_cov(
    const Chunk * __restrict__ chunks, int n_chunks, const int32_t * __restrict__ p_idx,
    const int32_t * __restrict__ w_idx, const int32_t * __restrict__ f_idx,
    const double * __restrict__ det_scale, const int64_t * __restrict__ g2l,
    double * __restrict__ invcov, long long * __restrict__ hits, const int64_t * __restrict__ pixels,
    const double * __restrict__ weights, const uint8_t * __restrict__ dflags, uint8_t dmask,
    int use_dflags, const uint8_t * __restrict__ sflags, uint8_t smask, int use_sflags,
    FastDiv nps_div, int64_t n_samp) {
    constexpr int NV = (MODE == 0) ? 1 : NNZ * (NNZ + 1) / 2;
    const int det = blockIdx.x;
    const int64_t * prow = pixels + (int64_t)p_idx[det] * n_samp;
    const double * wrow = (MODE == 1) ? weights + (int64_t)w_idx[det] * n_samp * NNZ : nullptr;
    const uint8_t * frow = use_dflags ? dflags + (int64_t)f_idx[det] * n_samp : nullptr;
    const double ds = (MODE == 1) ? det_scale[det] : 1.0;
    const int64_t nps = nps_div.d;
    for (int ci = blockIdx.y; ci < n_chunks; ci += gridDim.y) {
        const Chunk c = chunks[ci];
        for (int base = 0; base < c.count; base += kThreads) {
            const int i = base + threadIdx.x;
            const bool active = i < c.count;
            const int64_t s = c.first + (active ? i : 0);
            int64_t key = -1;
            double v[NV];
#pragma unroll
            for (int k = 0; k < NV; ++k) v[k] = 0.0;
            if (active) {
                const int64_t p = prow[s];
                const uint8_t fd = use_dflags ? frow[s] : (uint8_t)0;
                const uint8_t fs = use_sflags ? sflags[s] : (uint8_t)0;
                double w[NNZ];
                if (MODE == 1) {
#pragma unroll
                    for (int k = 0; k < NNZ; ++k) w[k] = wrow[NNZ * s + k];
                }
                const bool good = (p >= 0) & ((fd & dmask) == 0) & ((fs & smask) == 0);
                if (good) {
                    const int64_t gsm = fastdiv(p, nps_div);
                    key = g2l[gsm] * nps + (p - gsm * nps);
                    if (MODE == 0) {
                        v[0] = 1.0;
                    } else {
                        int off = 0;
#pragma unroll
                        for (int j = 0; j < NNZ; ++j) {
                            const double sw = w[j] * ds;
#pragma unroll
                            for (int k = j; k < NNZ; ++k, ++off) v[off] = w[k] * sw;
                        }
                    }
                }
            }
            const bool tail = wave_run_reduce<NV>(key, v);
            if (tail && key >= 0) {
                if (MODE == 0) {
                    atomicAdd(reinterpret_cast<unsigned long long *>(hits + key),
                              (unsigned long long)(long long)v[0]);
                } else {
                    double * z = invcov + NV * key;
#pragma unroll
                    for (int k = 0; k < NV; ++k) unsafeAtomicAdd(z + k, v[k]);
                }
            }
        }
    }
}

// Inverse covariance with the detector-pair merged scatter (see k_build_noise_weighted_pair): the
// A / B detectors of a focalplane pixel look at the same sky pixel, so their packed products are
// summed before the run reduction and share one set of atomics.
// HITS: the hit map rides along as one more reduced value (a count, exact in fp64) and leaves as one integer atomic per
// run -- BuildHitMap's separate pass over pixels and flags (9 B per det-sample at 280 G samples/s) disappears.
template <int NNZ, bool HITS>
__global__ __launch_bounds__(kThreads) void k_build_cov_pair(
    const Chunk * __restrict__ chunks, int n_chunks, int n_det, const int32_t * __restrict__ p_idx,
    const int32_t * __restrict__ w_idx, const int32_t * __restrict__ f_idx,
    const double * __restrict__ det_scale, const int64_t * __restrict__ g2l,
    double * __restrict__ invcov, const int64_t * __restrict__ pixels,
    const double * __restrict__ weights, const uint8_t * __restrict__ dflags, uint8_t dmask,
    int use_dflags, const uint8_t * __restrict__ sflags, uint8_t smask, int use_sflags,
    FastDiv nps_div, int64_t n_samp, long long * __restrict__ hits) {
    constexpr int NC = NNZ * (NNZ + 1) / 2;
    constexpr int NV = NC + (HITS ? 1 : 0);
    const int det0 = 2 * blockIdx.x;
    const bool two = det0 + 1 < n_det;
    const int det1 = two ? det0 + 1 : det0;
    const int64_t * prow[2] = {pixels + (int64_t)p_idx[det0] * n_samp, pixels + (int64_t)p_idx[det1] * n_samp};
    const double * wrow[2] = {weights + (int64_t)w_idx[det0] * n_samp * NNZ,
                              weights + (int64_t)w_idx[det1] * n_samp * NNZ};
    const uint8_t * frow[2] = {use_dflags ? dflags + (int64_t)f_idx[det0] * n_samp : nullptr,
                               use_dflags ? dflags + (int64_t)f_idx[det1] * n_samp : nullptr};
    const double ds[2] = {det_scale[det0], det_scale[det1]};
    const int64_t nps = nps_div.d;
    for (int ci = blockIdx.y; ci < n_chunks; ci += gridDim.y) {
        const Chunk c = chunks[ci];
        for (int base = 0; base < c.count; base += kThreads) {
            const int i = base + threadIdx.x;
            const bool active = i < c.count;
            const int64_t s = c.first + (active ? i : 0);
            int64_t key[2] = {-1, -1};
            double v[2][NV];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
#pragma unroll
                for (int k = 0; k < NV; ++k) v[e][k] = 0.0;
            }
            if (active) {
                int64_t p[2];
                uint8_t fd[2];
                double wk[2][NNZ];
                const uint8_t fs = use_sflags ? sflags[s] : (uint8_t)0;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    p[e] = prow[e][s];
                    fd[e] = use_dflags ? frow[e][s] : (uint8_t)0;
                    const double * w = wrow[e] + NNZ * s;
#pragma unroll
                    for (int k = 0; k < NNZ; ++k) wk[e][k] = w[k];
                }
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const bool good = (p[e] >= 0) & ((fd[e] & dmask) == 0) & ((fs & smask) == 0) & (e == 0 || two);
                    if (good) {
                        const int64_t gsm = fastdiv(p[e], nps_div);
                        key[e] = g2l[gsm] * nps + (p[e] - gsm * nps);
                        int off = 0;
#pragma unroll
                        for (int j = 0; j < NNZ; ++j) {
                            const double sw = wk[e][j] * ds[e];
#pragma unroll
                            for (int k = j; k < NNZ; ++k, ++off) v[e][off] = wk[e][k] * sw;
                        }
                        if (HITS) v[e][NC] = 1.0;
                    }
                }
            }
            if (!HITS) {
                scatter_runs<NV, 2>(key, v, invcov);
            } else {
                // as scatter_runs, with the last value going to the integer hit map
                const bool mergeable = (key[0] == key[1]) | (key[0] < 0) | (key[1] < 0);
                if (__all(mergeable)) {
                    const int64_t km = (key[0] >= 0) ? key[0] : key[1];
                    double vm[NV];
#pragma unroll
                    for (int k = 0; k < NV; ++k) vm[k] = v[0][k] + v[1][k];
                    const bool tail = wave_run_reduce<NV>(km, vm);
                    if (tail && km >= 0) {
                        double * z = invcov + NC * km;
#pragma unroll
                        for (int k = 0; k < NC; ++k) unsafeAtomicAdd(z + k, vm[k]);
                        atomicAdd((unsigned long long *)(hits + km), (unsigned long long)__double2ll_rn(vm[NC]));
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const bool tail = wave_run_reduce<NV>(key[e], v[e]);
                        if (tail && key[e] >= 0) {
                            double * z = invcov + NC * key[e];
#pragma unroll
                            for (int k = 0; k < NC; ++k) unsafeAtomicAdd(z + k, v[e][k]);
                            atomicAdd((unsigned long long *)(hits + key[e]),
                                      (unsigned long long)__double2ll_rn(v[e][NC]));
                        }
                    }
                }
            }
        }
    }
}

// Inverse covariance (+ hits) with TWO consecutive samples per lane and the detector pair of a focalplane pixel per
// workgroup, like k_build_noise_weighted_v2<2>: one segmented scan per 128 samples instead of one per 64 -- the scan of
// the six packed products and the hit count is what the kernel spends its vector issue on (counters, round 5:
// k_build_cov_pair<3, true> 251 vector instructions per detector-sample, 74 % of the issue cycles, 3.8 TB/s).
// The hit count rides along as value NC and leaves as one integer atomic per run.
// SIG: three more values ride along -- the noise-weighted signal of the sample, (tod x scale) x w, as k_build_noise_weighted_v2
// forms it -- and leave as atomics into a second map (k_build_cov_pair_v2<HITS, true>: inverse covariance, hits and the
// right-hand side's A^T N^-1 d in ONE sweep over the pointing).
template <int NV, int NC, bool HITS, bool SIG>
__device__ __forceinline__ void cov_emit(int64_t key, const double (&v)[NV], double * __restrict__ invcov,
                                         long long * __restrict__ hits, double * __restrict__ zmap) {
    double * z = invcov + NC * key;
#pragma unroll
    for (int k = 0; k < NC; ++k) unsafeAtomicAdd(z + k, v[k]);
    if constexpr (HITS) atomicAdd((unsigned long long *)(hits + key), (unsigned long long)__double2ll_rn(v[NC]));
    if constexpr (SIG) {
        constexpr int ZO = NC + (HITS ? 1 : 0);
        double * m = zmap + 3 * key;
#pragma unroll
        for (int k = 0; k < 3; ++k) unsafeAtomicAdd(m + k, v[ZO + k]);
    }
}

// scatter_runs2 (kernel_common.hpp) for the covariance values: A before B in every lane
template <int NV, int NC, bool HITS, bool SIG>
__device__ __forceinline__ void cov_scatter_runs2(int64_t ka, double (&va)[NV], int64_t kb, double (&vb)[NV],
                                                  double * __restrict__ invcov, long long * __restrict__ hits,
                                                  double * __restrict__ zmap) {
    const int lane = threadIdx.x & 63;
    const bool same = ka == kb;
    const bool apart = !same & (ka >= 0);
    if (same) {
#pragma unroll
        for (int k = 0; k < NV; ++k) vb[k] += va[k];
    }
    if (__any(apart)) {
        const int64_t prev_b = dpp_i64<kDppWaveShr1>(kb);
        const bool give = apart & (lane > 0) & (prev_b == ka);
#pragma unroll
        for (int k = 0; k < NV; ++k) vb[k] += dpp_f64<kDppWaveShl1>(give ? va[k] : 0.0);
        if (apart & !give) cov_emit<NV, NC, HITS, SIG>(ka, va, invcov, hits, zmap);
    }
    const bool tail = wave_run_reduce<NV>(kb, vb);
    if (tail && kb >= 0) cov_emit<NV, NC, HITS, SIG>(kb, vb, invcov, hits, zmap);
}

template <bool HITS, bool SIG = false>
__global__ __launch_bounds__(kThreads) void k_build_cov_pair_v2(
    const Chunk * __restrict__ chunks, int n_chunks, int n_det, const int32_t * __restrict__ p_idx,
    const int32_t * __restrict__ w_idx, const int32_t * __restrict__ f_idx,
    const double * __restrict__ det_scale, const int64_t * __restrict__ g2l,
    double * __restrict__ invcov, const int64_t * __restrict__ pixels,
    const double * __restrict__ weights, const uint8_t * __restrict__ dflags, uint8_t dmask,
    int use_dflags, const uint8_t * __restrict__ sflags, uint8_t smask, int use_sflags,
    FastDiv nps_div, int64_t n_samp, long long * __restrict__ hits,
    const int32_t * __restrict__ d_idx = nullptr, const double * __restrict__ tod = nullptr,
    const double * __restrict__ sig_scale = nullptr, double * __restrict__ zmap = nullptr) {
    constexpr int NNZ = 3, NC = 6, ZO = NC + (HITS ? 1 : 0), NV = ZO + (SIG ? 3 : 0), E = 2;
    const int det0 = E * blockIdx.x;
    bool on[E];
    const int64_t * prow[E];
    const double * wrow[E];
    const double * drow[E];
    const uint8_t * frow[E];
    double ds[E], ss[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        on[e] = det0 + e < n_det;
        const int det = on[e] ? det0 + e : det0;
        prow[e] = pixels + (int64_t)p_idx[det] * n_samp;
        wrow[e] = weights + (int64_t)w_idx[det] * n_samp * NNZ;
        frow[e] = use_dflags ? dflags + (int64_t)f_idx[det] * n_samp : nullptr;
        ds[e] = det_scale[det];
        drow[e] = SIG ? tod + (int64_t)d_idx[det] * n_samp : nullptr;
        ss[e] = SIG ? sig_scale[det] : 0.0;
    }
    const int64_t nps = nps_div.d;
    const uint16_t dmask2 = (uint16_t)(dmask | (dmask << 8));
    const uint16_t smask2 = (uint16_t)(smask | (smask << 8));
    // packed upper triangle of (scale w) w^T, the operation order of k_build_cov_pair; the signal values in the operation
    // order of k_build_noise_weighted_v2 ((tod x scale) x w)
    auto products = [](double (&v)[NV], double wa, double wb, double wc, double scale, bool good, double t, double tscale) {
        const double sa = wa * scale, sb = wb * scale, sc = wc * scale;
        v[0] = good ? wa * sa : 0.0;
        v[1] = good ? wb * sa : 0.0;
        v[2] = good ? wc * sa : 0.0;
        v[3] = good ? wb * sb : 0.0;
        v[4] = good ? wc * sb : 0.0;
        v[5] = good ? wc * sc : 0.0;
        if constexpr (HITS) v[NC] = good ? 1.0 : 0.0;
        if constexpr (SIG) {
            const double st = t * tscale;
            v[ZO + 0] = good ? st * wa : 0.0;
            v[ZO + 1] = good ? st * wb : 0.0;
            v[ZO + 2] = good ? st * wc : 0.0;
        }
    };
    for (int ci = blockIdx.y; ci < n_chunks; ci += gridDim.y) {
        const Chunk c = chunks[ci];
        const int head = (int)(c.first & 1);
        const int64_t s0 = c.first + head;          // even
        const int n_pair = (c.count - head) >> 1;
        for (int base = 0; base < n_pair; base += kThreads) {
            const int j = base + threadIdx.x;
            const bool active = j < n_pair;
            const int64_t s = s0 + 2 * (int64_t)(active ? j : 0);
            // all streaming loads of both samples of both detectors first (inactive lanes re-read pair 0: same lines)
            longlong2 pp[E];
            double2 w0[E], w1[E], w2[E], tt[E];
            uint16_t fd[E];
            const uint16_t fs = use_sflags ? *reinterpret_cast<const uint16_t *>(sflags + s) : (uint16_t)0;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                pp[e] = *reinterpret_cast<const longlong2 *>(prow[e] + s);
                fd[e] = use_dflags ? *reinterpret_cast<const uint16_t *>(frow[e] + s) : (uint16_t)0;
                if constexpr (SIG) tt[e] = *reinterpret_cast<const double2 *>(drow[e] + s);
                else tt[e] = make_double2(0.0, 0.0);
                const double2 * wv = reinterpret_cast<const double2 *>(wrow[e] + NNZ * s);
                w0[e] = wv[0];
                w1[e] = wv[1];
                w2[e] = wv[2];
            }
            int64_t ga[E], gb[E], la[E], lb[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                ga[e] = fastdiv(pp[e].x >= 0 ? pp[e].x : 0, nps_div);
                gb[e] = fastdiv(pp[e].y >= 0 ? pp[e].y : 0, nps_div);
                la[e] = g2l[ga[e]];
                lb[e] = g2l[gb[e]];
            }
            int64_t ka[E], kb[E];
            double va[E][NV], vb[E][NV];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const uint16_t bad = (uint16_t)((fd[e] & dmask2) | (fs & smask2));
                const bool good_a = active & on[e] & (pp[e].x >= 0) & (la[e] >= 0) & ((bad & 0x00ff) == 0);
                const bool good_b = active & on[e] & (pp[e].y >= 0) & (lb[e] >= 0) & ((bad & 0xff00) == 0);
                ka[e] = good_a ? la[e] * nps + (pp[e].x - ga[e] * nps) : -1;
                kb[e] = good_b ? lb[e] * nps + (pp[e].y - gb[e] * nps) : -1;
                products(va[e], w0[e].x, w0[e].y, w1[e].x, ds[e], good_a, tt[e].x, ss[e]);
                products(vb[e], w1[e].y, w2[e].x, w2[e].y, ds[e], good_b, tt[e].y, ss[e]);
            }
            const bool mergeable = ((ka[0] == ka[1]) | (ka[0] < 0) | (ka[1] < 0)) &
                                   ((kb[0] == kb[1]) | (kb[0] < 0) | (kb[1] < 0));
            if (__all(mergeable)) {
                const int64_t kam = (ka[0] >= 0) ? ka[0] : ka[1];
                const int64_t kbm = (kb[0] >= 0) ? kb[0] : kb[1];
                double vam[NV], vbm[NV];
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    vam[k] = va[0][k] + va[1][k];
                    vbm[k] = vb[0][k] + vb[1][k];
                }
                cov_scatter_runs2<NV, NC, HITS, SIG>(kam, vam, kbm, vbm, invcov, hits, zmap);
                continue;
            }
#pragma unroll
            for (int e = 0; e < E; ++e) cov_scatter_runs2<NV, NC, HITS, SIG>(ka[e], va[e], kb[e], vb[e], invcov, hits, zmap);
        }
        // the peeled first sample (lane 0) and the odd last one (lane 1)
        const int tail = (c.count - head) & 1;
        if ((threadIdx.x == 0 && head) || (threadIdx.x == 1 && tail)) {
            const int64_t s = (threadIdx.x == 0) ? c.first : c.first + c.count - 1;
            const uint8_t fs = use_sflags ? sflags[s] : (uint8_t)0;
            for (int e = 0; e < E; ++e) {
                if (!on[e]) continue;
                const int64_t px = prow[e][s];
                const uint8_t fd = use_dflags ? frow[e][s] : (uint8_t)0;
                if ((px < 0) | ((fd & dmask) != 0) | ((fs & smask) != 0)) continue;
                const int64_t gsm = fastdiv(px, nps_div);
                const int64_t key = g2l[gsm] * nps + (px - gsm * nps);
                if (key < 0) continue;
                const double * w = wrow[e] + NNZ * s;
                double v[NV];
                products(v, w[0], w[1], w[2], ds[e], true, SIG ? drow[e][s] : 0.0, ss[e]);
                cov_emit<NV, NC, HITS, SIG>(key, v, invcov, hits, zmap);
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// Per-pixel inversion of the packed symmetric covariance through its eigen-decomposition
// (cov_eigendecompose_diag, src/libtoast/src/toast_map_cov.cpp:246-396): rcond = emin/emax;
// if rcond >= threshold the block becomes V diag(1/lambda) V^T, else zero (and rcond 0).
// Cyclic Jacobi rotations replace LAPACK dsyev (nnz <= 4, one thread per pixel).
// ------------------------------------------------------------------------------------
template <int NNZ>
__global__ __launch_bounds__(kThreads) void k_cov_invert(int64_t n_px, double * __restrict__ data,
                                                         double * __restrict__ cond, double threshold,
                                                         int invert) {
    constexpr int BLK = NNZ * (NNZ + 1) / 2;
    for (int64_t px = (int64_t)blockIdx.x * kThreads + threadIdx.x; px < n_px;
         px += (int64_t)gridDim.x * kThreads) {
        double * d = data + px * BLK;
        if (NNZ == 1) {
            if (cond) cond[px] = 1.0;
            if (invert && d[0] != 0) d[0] = 1.0 / d[0];
            continue;
        }
        double a[NNZ][NNZ], v[NNZ][NNZ];
        int off = 0;
#pragma unroll
        for (int k = 0; k < NNZ; ++k) {
#pragma unroll
            for (int m = k; m < NNZ; ++m, ++off) {
                a[k][m] = d[off];
                a[m][k] = d[off];
            }
        }
#pragma unroll
        for (int k = 0; k < NNZ; ++k) {
#pragma unroll
            for (int m = 0; m < NNZ; ++m) v[k][m] = (k == m) ? 1.0 : 0.0;
        }
        for (int sweep = 0; sweep < 30; ++sweep) {
            double offn = 0.0, diag = 0.0;
#pragma unroll
            for (int k = 0; k < NNZ; ++k) {
                diag += a[k][k] * a[k][k];
#pragma unroll
                for (int m = k + 1; m < NNZ; ++m) offn += a[k][m] * a[k][m];
            }
            if (offn <= 1e-34 * diag || offn == 0.0) break;
#pragma unroll
            for (int p = 0; p < NNZ - 1; ++p) {
#pragma unroll
                for (int q = p + 1; q < NNZ; ++q) {
                    const double apq = a[p][q];
                    if (apq == 0.0) continue;
                    const double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
                    const double t = ((theta >= 0) ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                    const double cs = 1.0 / sqrt(t * t + 1.0);
                    const double sn = t * cs;
#pragma unroll
                    for (int k = 0; k < NNZ; ++k) {
                        const double akp = a[k][p], akq = a[k][q];
                        a[k][p] = cs * akp - sn * akq;
                        a[k][q] = sn * akp + cs * akq;
                    }
#pragma unroll
                    for (int k = 0; k < NNZ; ++k) {
                        const double apk = a[p][k], aqk = a[q][k];
                        a[p][k] = cs * apk - sn * aqk;
                        a[q][k] = sn * apk + cs * aqk;
                    }
#pragma unroll
                    for (int k = 0; k < NNZ; ++k) {
                        const double vkp = v[k][p], vkq = v[k][q];
                        v[k][p] = cs * vkp - sn * vkq;
                        v[k][q] = sn * vkp + cs * vkq;
                    }
                }
            }
        }
        double emin = 1.0e100, emax = 0.0;
#pragma unroll
        for (int k = 0; k < NNZ; ++k) {
            if (a[k][k] < emin) emin = a[k][k];
            if (a[k][k] > emax) emax = a[k][k];
        }
        const double rc = (emax > 0.0) ? (emin / emax) : 0.0;
        const bool ok = rc >= threshold;
        if (invert) {
            off = 0;
#pragma unroll
            for (int k = 0; k < NNZ; ++k) {
#pragma unroll
                for (int m = k; m < NNZ; ++m, ++off) {
                    double acc = 0.0;
#pragma unroll
                    for (int e = 0; e < NNZ; ++e) acc += v[k][e] * v[m][e] / a[e][e];
                    d[off] = ok ? acc : 0.0;
                }
            }
        }
        if (cond) cond[px] = ok ? rc : 0.0;
    }
}

// ------------------------------------------------------------------------------------
// scan_mask: det_flags[d, s] |= value where mask[g2l[pix / nps], pix % nps] & bits
// (operator-level semantics of ScanMask, src/toast/ops/scan_map/scan_map.py:283-320 -- host
// NumPy in the reference; here the same pass on the device copies).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_scan_mask(
    const Chunk * __restrict__ chunks, int n_chunks, const int32_t * __restrict__ p_idx,
    const int32_t * __restrict__ f_idx, const int64_t * __restrict__ g2l,
    const uint8_t * __restrict__ mask, uint8_t bits, uint8_t value, const int64_t * __restrict__ pixels,
    uint8_t * __restrict__ flags, FastDiv nps_div, int64_t n_samp) {
    const int det = blockIdx.x;
    const int64_t * prow = pixels + (int64_t)p_idx[det] * n_samp;
    uint8_t * frow = flags + (int64_t)f_idx[det] * n_samp;
    const int64_t nps = nps_div.d;
    for (int ci = blockIdx.y; ci < n_chunks; ci += gridDim.y) {
        const Chunk c = chunks[ci];
        for (int i = threadIdx.x; i < c.count; i += kThreads) {
            const int64_t s = c.first + i;
            // straight-line loads (pixel and flag together, then global2local, then the mask byte): as nested tests
            // every sample waits for four memory round trips in a row
            const int64_t p = prow[s];
            const uint8_t f = frow[s];
            const bool hit = p >= 0;
            const int64_t pp = hit ? p : 0;
            const int64_t gsm = fastdiv(pp, nps_div);
            const int64_t lsm = g2l[gsm];
            const bool local = hit && lsm >= 0;
            const uint8_t mk = mask[(local ? lsm : 0) * nps + (pp - gsm * nps)];
            if (local && (mk & bits)) frow[s] = f | value;
        }
    }
}

// two consecutive samples per lane: one 16-byte pixel load, the two flag bytes as one 2-byte load / store, both
// global2local look-ups and both mask bytes in flight together (rows of an even length at 16-byte aligned addresses)
__global__ __launch_bounds__(kThreads) void k_scan_mask_v2(
    const Chunk * __restrict__ chunks, int n_chunks, const int32_t * __restrict__ p_idx,
    const int32_t * __restrict__ f_idx, const int64_t * __restrict__ g2l,
    const uint8_t * __restrict__ mask, uint8_t bits, uint8_t value, const int64_t * __restrict__ pixels,
    uint8_t * __restrict__ flags, FastDiv nps_div, int64_t n_samp) {
    const int det = blockIdx.x;
    const int64_t * prow = pixels + (int64_t)p_idx[det] * n_samp;
    uint8_t * frow = flags + (int64_t)f_idx[det] * n_samp;
    const int64_t nps = nps_div.d;
    auto masked = [&](int64_t p) {
        const bool hit = p >= 0;
        const int64_t pp = hit ? p : 0;
        const int64_t gsm = fastdiv(pp, nps_div);
        const int64_t lsm = g2l[gsm];
        const bool local = hit && lsm >= 0;
        const uint8_t mk = mask[(local ? lsm : 0) * nps + (pp - gsm * nps)];
        return local && (mk & bits);
    };
    for (int ci = blockIdx.y; ci < n_chunks; ci += gridDim.y) {
        const Chunk c = chunks[ci];
        const int head = (int)(c.first & 1);
        const int64_t s0 = c.first + head;          // even
        const int n_pair = (c.count - head) >> 1;
        for (int j = threadIdx.x; j < n_pair; j += kThreads) {
            const int64_t s = s0 + 2 * (int64_t)j;
            const longlong2 p = *reinterpret_cast<const longlong2 *>(prow + s);
            const uint16_t f = *reinterpret_cast<const uint16_t *>(frow + s);
            const bool ma = masked(p.x), mb = masked(p.y);
            const uint16_t g = (uint16_t)(f | (ma ? (uint16_t)value : (uint16_t)0) | (mb ? (uint16_t)((uint16_t)value << 8) : (uint16_t)0));
            if (g != f) *reinterpret_cast<uint16_t *>(frow + s) = g;
        }
        const int tail = (c.count - head) & 1;
        if ((threadIdx.x == 0 && head) || (threadIdx.x == 1 && tail)) {
            const int64_t s = (threadIdx.x == 0) ? c.first : c.first + c.count - 1;
            if (masked(prow[s])) frow[s] |= value;
        }
    }
}

// ------------------------------------------------------------------------------------
// Solver flags: one uint8 per det-sample = detector flag | shared flag (masked), the combined
// cut MapMaker hands to the binning and the templates while solving
// [ref: SolveAmplitudes._prepare_flagging, src/toast/ops/mapmaker_templates.py:764-810].
// Samples outside the intervals are set by the caller (memset) before this kernel.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_combine_flags(
    const Chunk * __restrict__ chunks, int n_chunks, const int32_t * __restrict__ o_idx,
    const int32_t * __restrict__ f_idx, uint8_t * __restrict__ out, const uint8_t * __restrict__ dflags,
    uint8_t dmask, int use_dflags, const uint8_t * __restrict__ sflags, uint8_t smask, int use_sflags,
    int64_t n_samp) {
    const int det = blockIdx.x;
    uint8_t * orow = out + (int64_t)o_idx[det] * n_samp;
    const uint8_t * frow = use_dflags ? dflags + (int64_t)f_idx[det] * n_samp : nullptr;
    for (int ci = blockIdx.y; ci < n_chunks; ci += gridDim.y) {
        const Chunk c = chunks[ci];
        for (int i = threadIdx.x; i < c.count; i += kThreads) {
            const int64_t s = c.first + i;
            const uint8_t fd = use_dflags ? frow[s] : (uint8_t)0;
            const uint8_t fs = use_sflags ? sflags[s] : (uint8_t)0;
            orow[s] = (((fd & dmask) != 0) | ((fs & smask) != 0)) ? 1 : 0;
        }
    }
}

// ------------------------------------------------------------------------------------
// Amplitude-vector algebra of the PCG on the device (axpby, flagged dot product): the
// Amplitudes arithmetic of src/toast/templates/amplitudes.py:400-565 for resident vectors.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_vec_axpby(int64_t n, double a, const double * __restrict__ x,
                                                        double b, double * __restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * kThreads) {
        // b == 0 overwrites (y may hold anything, including NaN)
        y[i] = (b == 0.0) ? a * x[i] : a * x[i] + b * y[i];
    }
}

// `partials` != nullptr (deterministic mode): every block stores its sum, k_vec_dot_final adds them in block
// order -- the result no longer depends on the order in which the blocks retire.
__global__ __launch_bounds__(kThreads) void k_vec_dot_final(int n_block, const double * __restrict__ partials,
                                                            double * __restrict__ result) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double t = 0.0;
        for (int b = 0; b < n_block; ++b) t += partials[b];
        *result = t;
    }
}

__global__ __launch_bounds__(kThreads) void k_vec_dot(int64_t n, const double * __restrict__ x,
                                                      const double * __restrict__ y,
                                                      const uint8_t * __restrict__ fx,
                                                      const uint8_t * __restrict__ fy,
                                                      double * __restrict__ result,
                                                      double * __restrict__ partials) {
    __shared__ double s_part[kThreads / 64];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * kThreads) {
        const bool good = (fx == nullptr || fx[i] == 0) && (fy == nullptr || fy[i] == 0);
        if (good) acc += x[i] * y[i];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_down(acc, d);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) t += s_part[w];
        if (partials != nullptr) {
            partials[blockIdx.x] = t;
        } else {
            unsafeAtomicAdd(result, t);
        }
    }
}

// ------------------------------------------------------------------------------------
// noise_weight   [ref: ops_noise_weight.cpp:71-96]
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_noise_weight(
    const Chunk * __restrict__ chunks, int n_chunks, const int32_t * __restrict__ d_idx,
    const double * __restrict__ det_w, double * __restrict__ tod, int64_t n_samp) {
    const int det = blockIdx.x;
    const double w = det_w[det];
    double * drow = tod + (int64_t)d_idx[det] * n_samp;
    for (int ci = blockIdx.y; ci < n_chunks; ci += gridDim.y) {
        const Chunk c = chunks[ci];
        for (int i = threadIdx.x; i < c.count; i += kThreads) drow[c.first + i] *= w;
    }
}

// two samples (16 bytes) per lane; rows start on 16-byte boundaries (even n_samp, checked by the launcher)
__global__ __launch_bounds__(kThreads) void k_noise_weight_v2(
    const Chunk * __restrict__ chunks, int n_chunks, const int32_t * __restrict__ d_idx,
    const double * __restrict__ det_w, double * __restrict__ tod, int64_t n_samp) {
    const int det = blockIdx.x;
    const double w = det_w[det];
    double * drow = tod + (int64_t)d_idx[det] * n_samp;
    for (int ci = blockIdx.y; ci < n_chunks; ci += gridDim.y) {
        const Chunk c = chunks[ci];
        const int head = (int)(c.first & 1);
        double2 * row2 = reinterpret_cast<double2 *>(drow + c.first + head);
        const int n_pair = (c.count - head) >> 1;
        for (int j = threadIdx.x; j < n_pair; j += kThreads) {
            double2 v = row2[j];
            v.x *= w;
            v.y *= w;
            row2[j] = v;
        }
        if (threadIdx.x == 0 && head) drow[c.first] *= w;
        if (threadIdx.x == 1 && ((c.count - head) & 1)) drow[c.first + c.count - 1] *= w;
    }
}

// ------------------------------------------------------------------------------------
// cov_accum_diag_hits / cov_accum_diag_invnpp   [ref: src/libtoast/src/toast_map_cov.cpp:66-153]
// The reference's kernels behind BuildHitMap / BuildInverseCovariance at the FFI level: one stream of samples with its
// (local submap, pixel in submap) index pair per sample (negative = skip); hits[hpx] += 1, invnpp[hpx] += upper
// triangle of (scale w) w^T.  Same run reduction + one atomic per run as k_build_cov.
// ------------------------------------------------------------------------------------
// MODE 0: hits, 1: inverse covariance (invnpp), 2: noise-weighted map (invnpp = zmap, `tod` = the signal)
template <int NNZ, int MODE>
__global__ __launch_bounds__(kThreads) void k_cov_accum(int64_t n_samp, const int64_t * __restrict__ submap,
                                                       const int64_t * __restrict__ subpix, int64_t subsize,
                                                       const double * __restrict__ weights, double scale,
                                                       double * __restrict__ invnpp, long long * __restrict__ hits,
                                                       const double * __restrict__ tod) {
    constexpr int NV = (MODE == 0) ? 1 : ((MODE == 2) ? NNZ : NNZ * (NNZ + 1) / 2);
    const int64_t n_round = (n_samp + kThreads - 1) / kThreads;
    for (int64_t r = blockIdx.x; r < n_round; r += gridDim.x) {
        const int64_t i = r * kThreads + threadIdx.x;
        int64_t key = -1;
        double v[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] = 0.0;
        if (i < n_samp) {
            const int64_t sm = submap[i], px = subpix[i];
            // (the reference tests isubmap = submap * subsize < 0, i.e. submap < 0)
            if (sm >= 0 && px >= 0) {
                key = sm * subsize + px;
                if (MODE == 0) {
                    v[0] = 1.0;
                } else if (MODE == 2) {
                    const double ss = scale * tod[i];
#pragma unroll
                    for (int k = 0; k < NNZ; ++k) v[k] = weights[i * NNZ + k] * ss;
                } else {
                    int off = 0;
#pragma unroll
                    for (int j = 0; j < NNZ; ++j) {
                        const double sw = weights[i * NNZ + j] * scale;
#pragma unroll
                        for (int k = j; k < NNZ; ++k, ++off) v[off] = weights[i * NNZ + k] * sw;
                    }
                }
            }
        }
        const bool tail = wave_run_reduce<NV>(key, v);
        if (tail && key >= 0) {
            if (MODE == 0) {
                atomicAdd(reinterpret_cast<unsigned long long *>(hits + key), (unsigned long long)(long long)v[0]);
            } else {
                double * z = invnpp + NV * key;
#pragma unroll
                for (int k = 0; k < NV; ++k) unsafeAtomicAdd(z + k, v[k]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// cov_mult_diag   [ref: src/libtoast/src/toast_map_cov.cpp:398-469]
// Per pixel, data1 <- packed upper triangle of the product of the two symmetric blocks: the reference expands both
// to full matrices, calls the batched dsymm (C = S1 S2 in column-major storage) and packs entry (k, m >= k) from
// C(m, k) = sum_j S1(m, j) S2(j, k).  LAPACK / BLAS are absent from the reference build here (it throws); the
// summation runs over j = 0 .. nnz-1 in order.
// ------------------------------------------------------------------------------------
template <int NNZ>
__global__ __launch_bounds__(kThreads) void k_cov_mult_diag(int64_t n_px, double * __restrict__ data1,
                                                            const double * __restrict__ data2) {
    constexpr int BLK = NNZ * (NNZ + 1) / 2;
    for (int64_t px = (int64_t)blockIdx.x * kThreads + threadIdx.x; px < n_px;
         px += (int64_t)gridDim.x * kThreads) {
        double * a = data1 + px * BLK;
        const double * b = data2 + px * BLK;
        if (NNZ == 1) {
            a[0] *= b[0];
            continue;
        }
        double s1[NNZ][NNZ], s2[NNZ][NNZ];
        int off = 0;
#pragma unroll
        for (int k = 0; k < NNZ; ++k) {
#pragma unroll
            for (int m = k; m < NNZ; ++m, ++off) {
                s1[k][m] = s1[m][k] = a[off];
                s2[k][m] = s2[m][k] = b[off];
            }
        }
        off = 0;
#pragma unroll
        for (int k = 0; k < NNZ; ++k) {
#pragma unroll
            for (int m = k; m < NNZ; ++m, ++off) {
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < NNZ; ++j) acc += s1[m][j] * s2[j][k];
                a[off] = acc;
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// cov_apply_diag   [ref: src/libtoast/src/toast_map_cov.cpp:471-528]
// One thread per pixel; reference accumulation order (row k, then the mirrored term).
// ------------------------------------------------------------------------------------
template <int NNZ>
__global__ __launch_bounds__(kThreads) void k_cov_apply_diag(int64_t n_px, const double * __restrict__ mat,
                                                             double * __restrict__ vec) {
    constexpr int BLK = NNZ * (NNZ + 1) / 2;
    for (int64_t px = (int64_t)blockIdx.x * kThreads + threadIdx.x; px < n_px;
         px += (int64_t)gridDim.x * kThreads) {
        const double * m = mat + px * BLK;
        double * v = vec + px * NNZ;
        double in[NNZ], t[NNZ];
#pragma unroll
        for (int k = 0; k < NNZ; ++k) {
            in[k] = v[k];
            t[k] = 0.0;
        }
        if (NNZ == 1) {
            v[0] = in[0] * m[0];
            continue;
        }
        int off = 0;
#pragma unroll
        for (int k = 0; k < NNZ; ++k) {
#pragma unroll
            for (int j = k; j < NNZ; ++j, ++off) {
                t[k] += m[off] * in[j];
                if (j != k) t[j] += m[off] * in[k];
            }
        }
#pragma unroll
        for (int k = 0; k < NNZ; ++k) v[k] = t[k];
    }
}

// ------------------------------------------------------------------------------------
// Offset template   [ref: template_offset.cpp:93-120, :243-290, :375-390]
// view_first / view_aoff: per-interval first sample and amplitude offset (host-built).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_offset_add_to_signal(
    const Chunk * __restrict__ chunks, int n_chunks, const int64_t * __restrict__ view_first,
    const int64_t * __restrict__ view_aoff, FastDiv step_div, const int64_t * __restrict__ amp_offsets,
    const int32_t * __restrict__ d_idx, const double * __restrict__ amps,
    const uint8_t * __restrict__ amp_flags, double * __restrict__ tod, int64_t n_samp) {
    const int det = blockIdx.x;
    double * drow = tod + (int64_t)d_idx[det] * n_samp;
    const int64_t amp_offset = amp_offsets[det];
    for (int ci = blockIdx.y; ci < n_chunks; ci += gridDim.y) {
        const Chunk c = chunks[ci];
        const int64_t vfirst = view_first[c.view];
        const int64_t abase = amp_offset + view_aoff[c.view];
        for (int i = threadIdx.x; i < c.count; i += kThreads) {
            const int64_t s = c.first + i;
            const int64_t a = abase + fastdiv(s - vfirst, step_div);
            if (amp_flags[a] == 0) drow[s] += amps[a];
        }
    }
}

__global__ __launch_bounds__(kThreads) void k_offset_project_signal(
    const Chunk * __restrict__ chunks, int n_chunks, const int64_t * __restrict__ view_first,
    const int64_t * __restrict__ view_aoff, FastDiv step_div, const int64_t * __restrict__ amp_offsets,
    const int32_t * __restrict__ d_idx, const int32_t * __restrict__ f_idx, double * __restrict__ amps,
    const uint8_t * __restrict__ amp_flags, const double * __restrict__ tod,
    const uint8_t * __restrict__ flags, uint8_t fmask, int use_flags, int64_t n_samp) {
    const int det = blockIdx.x;
    const double * drow = tod + (int64_t)d_idx[det] * n_samp;
    const uint8_t * frow = use_flags ? flags + (int64_t)f_idx[det] * n_samp : nullptr;
    const int64_t amp_offset = amp_offsets[det];
    for (int ci = blockIdx.y; ci < n_chunks; ci += gridDim.y) {
        const Chunk c = chunks[ci];
        const int64_t vfirst = view_first[c.view];
        const int64_t abase = amp_offset + view_aoff[c.view];
        for (int base = 0; base < c.count; base += kThreads) {
            const int i = base + threadIdx.x;
            const bool active = i < c.count;
            int64_t key = -1;
            double v[1] = {0.0};
            if (active) {
                const int64_t s = c.first + i;
                const int64_t a = abase + fastdiv(s - vfirst, step_div);
                // the three loads together (as nested tests: three memory round trips in a row)
                const uint8_t af = amp_flags[a];
                const uint8_t fl = use_flags ? frow[s] : (uint8_t)0;
                const double dv = drow[s];
                key = (af == 0) ? a : (int64_t)-1;
                v[0] = (af == 0 && (fl & fmask) == 0) ? dv : 0.0;
            }
            const bool tail = wave_run_reduce<1>(key, v);
            if (tail && key >= 0) unsafeAtomicAdd(amps + key, v[0]);
        }
    }
}

// Deterministic form of k_offset_project_signal (TOAST_HIP_DETERMINISTIC): one thread per amplitude adds its
// samples in increasing order onto the value already there -- the order of the reference's host loop
// (template_offset.cpp:243-290), so the projected amplitudes are bit-identical to it and from run to run.
__global__ __launch_bounds__(kThreads) void k_offset_project_signal_det(
    int n_view, const int64_t * __restrict__ view_first, const int64_t * __restrict__ view_last,
    const int64_t * __restrict__ view_aoff, const int64_t * __restrict__ view_namp, int64_t n_amp_det, int64_t step,
    const int64_t * __restrict__ amp_offsets, const int32_t * __restrict__ d_idx, const int32_t * __restrict__ f_idx,
    double * __restrict__ amps, const uint8_t * __restrict__ amp_flags, const double * __restrict__ tod,
    const uint8_t * __restrict__ flags, uint8_t fmask, int use_flags, int64_t n_samp) {
    const int det = blockIdx.y;
    int64_t b = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (b >= n_amp_det) return;
    int v = 0;
    while (v < n_view - 1 && b >= view_namp[v]) {
        b -= view_namp[v];
        ++v;
    }
    if (b >= view_namp[v]) return;
    const int64_t a = amp_offsets[det] + view_aoff[v] + b;
    if (amp_flags[a] != 0) return;
    const double * drow = tod + (int64_t)d_idx[det] * n_samp;
    const uint8_t * frow = use_flags ? flags + (int64_t)f_idx[det] * n_samp : nullptr;
    const int64_t s0 = view_first[v] + b * step;
    int64_t s1 = s0 + step;
    if (s1 > view_last[v]) s1 = view_last[v];
    double acc = amps[a];
    for (int64_t s = s0; s < s1; ++s) {
        const bool bad = use_flags && ((frow[s] & fmask) != 0);
        acc += bad ? 0.0 : drow[s];
    }
    amps[a] = acc;
}

// Number of flagged samples under every offset amplitude (the good-fraction cut and the
// preconditioner variances of Offset._initialize, offset.py:262-343), as doubles.
__global__ __launch_bounds__(kThreads) void k_offset_count_flagged(
    const Chunk * __restrict__ chunks, int n_chunks, const int64_t * __restrict__ view_first,
    const int64_t * __restrict__ view_aoff, FastDiv step_div, const int64_t * __restrict__ amp_offsets,
    const int32_t * __restrict__ f_idx, double * __restrict__ counts, const uint8_t * __restrict__ flags,
    uint8_t fmask, int64_t n_samp) {
    const int det = blockIdx.x;
    const uint8_t * frow = flags + (int64_t)f_idx[det] * n_samp;
    const int64_t amp_offset = amp_offsets[det];
    for (int ci = blockIdx.y; ci < n_chunks; ci += gridDim.y) {
        const Chunk c = chunks[ci];
        const int64_t vfirst = view_first[c.view];
        const int64_t abase = amp_offset + view_aoff[c.view];
        for (int base = 0; base < c.count; base += kThreads) {
            const int i = base + threadIdx.x;
            int64_t key = -1;
            double v[1] = {0.0};
            if (i < c.count) {
                const int64_t s = c.first + i;
                key = abase + fastdiv(s - vfirst, step_div);
                v[0] = ((frow[s] & fmask) != 0) ? 1.0 : 0.0;
            }
            const bool tail = wave_run_reduce<1>(key, v);
            if (tail && key >= 0 && v[0] != 0.0) unsafeAtomicAdd(counts + key, v[0]);
        }
    }
}

// The same count with SIXTEEN flag bytes per lane (one 16-byte load; baselines of at least 16 samples, so that a lane's
// samples lie in at most two of them): the one-byte-per-lane form spends a 64-bit reciprocal division and a segmented
// wave reduction on every byte -- 1.6 ms for the 0.74 GB of flags of cfg-3 (round 6: 0.3 ms).
struct __attribute__((packed, aligned(1))) FlagBytes16 {
    uint32_t w[4];
};
__global__ __launch_bounds__(kThreads) void k_offset_count_flagged16(
    const Chunk * __restrict__ chunks, int n_chunks, const int64_t * __restrict__ view_first,
    const int64_t * __restrict__ view_aoff, FastDiv step_div, const int64_t * __restrict__ amp_offsets,
    const int32_t * __restrict__ f_idx, double * __restrict__ counts, const uint8_t * __restrict__ flags,
    uint8_t fmask, int64_t n_samp) {
    const int det = blockIdx.x;
    const uint8_t * frow = flags + (int64_t)f_idx[det] * n_samp;
    const int64_t amp_offset = amp_offsets[det];
    const uint32_t m4 = 0x01010101u * (uint32_t)fmask;
    for (int ci = blockIdx.y; ci < n_chunks; ci += gridDim.y) {
        const Chunk c = chunks[ci];
        const int64_t vfirst = view_first[c.view];
        const int64_t abase = amp_offset + view_aoff[c.view];
        for (int base = 0; base < c.count; base += 16 * kThreads) {
            const int i = base + 16 * threadIdx.x;
            if (i >= c.count) continue;
            const int64_t s = c.first + i;
            const int n = (c.count - i < 16) ? (c.count - i) : 16;
            uint32_t w[4] = {0u, 0u, 0u, 0u};
            if (n == 16) {
                const FlagBytes16 v = *reinterpret_cast<const FlagBytes16 *>(frow + s);
#pragma unroll
                for (int k = 0; k < 4; ++k) w[k] = v.w[k];
            } else {
                for (int k = 0; k < n; ++k) w[k >> 2] |= (uint32_t)frow[s + k] << (8 * (k & 3));
            }
            const int64_t st = fastdiv(s - vfirst, step_div);
            const int64_t next = vfirst + (st + 1) * step_div.d - s;     // samples of this lane that belong to baseline st
            const int split = next < n ? (int)next : n;
            int lo = 0, hi = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int f = ((w[k >> 2] & m4) >> (8 * (k & 3))) & 0xff ? 1 : 0;      // (bytes past n are zero)
                lo += (k < split) ? f : 0;
                hi += (k < split) ? 0 : f;
            }
            if (lo) unsafeAtomicAdd(counts + abase + st, (double)lo);
            if (hi) unsafeAtomicAdd(counts + abase + st + 1, (double)hi);
        }
    }
}

// ------------------------------------------------------------------------------------
// Fused PCG left-hand side for offset templates (SolverLHS, src/toast/ops/mapmaker_solve.py:
// 342-506, with the Offset template kernels template_offset.cpp:93-120 / :243-290):
//
//   k_offset_accumulate :  zmap += A^T N^-1 (M a)          == add_to_signal + build_noise_weighted
//   k_offset_scan_project: a_out += M^T N^-1 (M a - A z)   == add_to_signal + scan_map(subtract)
//                                                              + noise_weight + project_signal
//
// The timestream M a is never materialised: 33 B per det-sample per kernel (pixel 8 + weights
// 24 + flag 1) instead of 8 + 41 and 8 + 48 + 9 in the reference's operator sequence.  Per
// sample the arithmetic is the same as in the unfused kernels.
// ------------------------------------------------------------------------------------
// E = detectors per workgroup: 2 merges the contributions of the detector pair (2b, 2b+1) before
// the run reduction whenever the whole wave sees equal keys (see k_build_noise_weighted_pair).
template <int NNZ, int E>
__global__ __launch_bounds__(kThreads) void k_offset_accumulate(
    const Chunk * __restrict__ chunks, int n_chunks, int n_det, const int64_t * __restrict__ view_first,
    const int64_t * __restrict__ view_aoff, FastDiv step_div, const int64_t * __restrict__ amp_offsets,
    const double * __restrict__ amps, const uint8_t * __restrict__ amp_flags,
    const int32_t * __restrict__ p_idx, const int32_t * __restrict__ w_idx,
    const int32_t * __restrict__ f_idx, const double * __restrict__ det_scale,
    const int64_t * __restrict__ g2l, double * __restrict__ zmap, const int64_t * __restrict__ pixels,
    const double * __restrict__ weights, const uint8_t * __restrict__ dflags, uint8_t dmask,
    int use_dflags, const uint8_t * __restrict__ sflags, uint8_t smask, int use_sflags,
    FastDiv nps_div, int64_t n_samp) {
    const int64_t * prow[E];
    const double * wrow[E];
    const uint8_t * frow[E];
    double ds[E];
    int64_t amp_offset[E];
    bool valid[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        int det = E * blockIdx.x + e;
        valid[e] = det < n_det;
        if (!valid[e]) det = E * blockIdx.x;
        prow[e] = pixels + (int64_t)p_idx[det] * n_samp;
        wrow[e] = weights + (int64_t)w_idx[det] * n_samp * NNZ;
        frow[e] = use_dflags ? dflags + (int64_t)f_idx[det] * n_samp : nullptr;
        ds[e] = det_scale[det];
        amp_offset[e] = amp_offsets[det];
    }
    const int64_t nps = nps_div.d;
    for (int ci = blockIdx.y; ci < n_chunks; ci += gridDim.y) {
        const Chunk c = chunks[ci];
        const int64_t vfirst = view_first[c.view];
        const int64_t vaoff = view_aoff[c.view];
        for (int base = 0; base < c.count; base += kThreads) {
            const int i = base + threadIdx.x;
            const bool active = i < c.count;
            const int64_t s = c.first + (active ? i : 0);
            int64_t key[E];
            double v[E][NNZ];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                key[e] = -1;
#pragma unroll
                for (int k = 0; k < NNZ; ++k) v[e][k] = 0.0;
            }
            if (active) {
                const uint8_t fs = use_sflags ? sflags[s] : (uint8_t)0;
                const int64_t astep = fastdiv(s - vfirst, step_div);
                int64_t p[E];
                uint8_t fd[E], af[E];
                double av[E], wk[E][NNZ];
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    p[e] = prow[e][s];
                    fd[e] = use_dflags ? frow[e][s] : (uint8_t)0;
                    const int64_t a = amp_offset[e] + vaoff + astep;
                    af[e] = amp_flags[a];
                    av[e] = amps[a];
                    const double * w = wrow[e] + NNZ * s;
#pragma unroll
                    for (int k = 0; k < NNZ; ++k) wk[e][k] = w[k];
                }
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const bool good = (p[e] >= 0) & ((fd[e] & dmask) == 0) & ((fs & smask) == 0) & valid[e];
                    if (good) {
                        const int64_t gsm = fastdiv(p[e], nps_div);
                        key[e] = g2l[gsm] * nps + (p[e] - gsm * nps);
                        // tod = 0 + amplitude (unflagged amplitudes only), then * det_scale
                        const double t = (af[e] == 0) ? (0.0 + av[e]) : 0.0;
                        const double sd = t * ds[e];
#pragma unroll
                        for (int k = 0; k < NNZ; ++k) v[e][k] = sd * wk[e][k];
                    }
                }
            }
            scatter_runs<NNZ, E>(key, v, zmap);
        }
    }
}

// SIGBUF: the signal is a timestream buffer instead of M a -- a_out += M^T N^-1 (d - A z), the right-hand side of the
// solver (SolverRHS: copy, scan_map(subtract), noise_weight and project_signal in one pass that leaves d untouched).
template <int NNZ, bool SIGBUF>
__global__ __launch_bounds__(kThreads) void k_offset_scan_project(
    const Chunk * __restrict__ chunks, int n_chunks, const int64_t * __restrict__ view_first,
    const int64_t * __restrict__ view_aoff, FastDiv step_div, const int64_t * __restrict__ amp_offsets,
    const double * __restrict__ amps_in, double * __restrict__ amps_out,
    const uint8_t * __restrict__ amp_flags, const int32_t * __restrict__ p_idx,
    const int32_t * __restrict__ w_idx, const int32_t * __restrict__ f_idx,
    const double * __restrict__ det_w, const int64_t * __restrict__ g2l,
    const double * __restrict__ map, const int64_t * __restrict__ pixels,
    const double * __restrict__ weights, const uint8_t * __restrict__ flags, uint8_t fmask,
    int use_flags, FastDiv nps_div, int64_t n_samp, const int32_t * __restrict__ s_idx,
    const double * __restrict__ signal) {
    const int det = blockIdx.x;
    const double * srow = SIGBUF ? signal + (int64_t)s_idx[det] * n_samp : nullptr;
    const int64_t * prow = pixels + (int64_t)p_idx[det] * n_samp;
    const double * wrow = weights + (int64_t)w_idx[det] * n_samp * NNZ;
    const uint8_t * frow = use_flags ? flags + (int64_t)f_idx[det] * n_samp : nullptr;
    const double dw = det_w[det];
    const int64_t nps = nps_div.d;
    const int64_t amp_offset = amp_offsets[det];
    for (int ci = blockIdx.y; ci < n_chunks; ci += gridDim.y) {
        const Chunk c = chunks[ci];
        const int64_t vfirst = view_first[c.view];
        const int64_t abase = amp_offset + view_aoff[c.view];
        for (int base = 0; base < c.count; base += kThreads) {
            const int i = base + threadIdx.x;
            const bool active = i < c.count;
            int64_t key = -1;
            double v[1] = {0.0};
            if (active) {
                // Straight-line form: with the flag and pixel tests as branches the compiler sinks the pixel, weight
                // and map loads into them and every sample waits for five memory round trips in a row (amplitude
                // flag -> sample flag -> pixel -> weights + global2local -> map); here everything that does not depend
                // on a loaded value is issued at once (three round trips: streams, global2local, map) and the tests
                // select values.  Flagged samples and samples without a pixel read map element 0 and discard it.
                const int64_t s = c.first + i;
                const int64_t a = abase + fastdiv(s - vfirst, step_div);
                const uint8_t af = amp_flags[a];
                const double av = SIGBUF ? srow[s] : amps_in[a];
                const uint8_t fl = use_flags ? frow[s] : (uint8_t)0;
                const int64_t p = prow[s];
                const double * w = wrow + NNZ * s;
                double wk[NNZ];
#pragma unroll
                for (int k = 0; k < NNZ; ++k) wk[k] = w[k];
                const int64_t pp = (p >= 0) ? p : 0;
                const int64_t gsm = fastdiv(pp, nps_div);
                int64_t lsm = g2l[gsm];
                const bool hit = (p >= 0) & (lsm >= 0);   // a pixel in a non-local submap scans nothing
                lsm = (lsm < 0) ? 0 : lsm;
                const double * m = map + NNZ * (lsm * nps + (pp - gsm * nps));
                double sc = 0.0;
#pragma unroll
                for (int k = 0; k < NNZ; ++k) sc += wk[k] * m[k];
                sc *= 1.0;
                double d = SIGBUF ? av : 0.0 + av;
                d = hit ? d - sc : d;
                const bool good = (fl & fmask) == 0;
                key = (af == 0) ? a : (int64_t)-1;
                v[0] = (af == 0 && good) ? d * dw : 0.0;
            }
            const bool tail = wave_run_reduce<1>(key, v);
            if (tail && key >= 0) unsafeAtomicAdd(amps_out + key, v[0]);
        }
    }
}

// k_offset_accumulate with two consecutive samples per lane (nnz = 3): k_build_noise_weighted_v2 with the timestream
// replaced by the amplitude of each sample's baseline (two look-ups per lane, equal except at a baseline boundary).
// SIG: the timestream is there too and the amplitude is SUBTRACTED from it -- zmap += A^T N^-1 (d - M a), the final binning
// of the template-cleaned signal (ApplyAmplitudes + BinMap, src/toast/ops/mapmaker.py:531-608) without writing the cleaned
// timestream: d - (0 + a) per sample, then the products of build_noise_weighted, in the reference's order.
template <int E, bool SIG>
__global__ __launch_bounds__(kThreads) void k_offset_accumulate_v2(
    const Chunk * __restrict__ chunks, int n_chunks, int n_det, const int64_t * __restrict__ view_first,
    const int64_t * __restrict__ view_aoff, FastDiv step_div, const int64_t * __restrict__ amp_offsets,
    const double * __restrict__ amps, const uint8_t * __restrict__ amp_flags,
    const int32_t * __restrict__ p_idx, const int32_t * __restrict__ w_idx,
    const int32_t * __restrict__ f_idx, const double * __restrict__ det_scale,
    const int64_t * __restrict__ g2l, double * __restrict__ zmap, const int64_t * __restrict__ pixels,
    const double * __restrict__ weights, const uint8_t * __restrict__ dflags, uint8_t dmask,
    int use_dflags, const uint8_t * __restrict__ sflags, uint8_t smask, int use_sflags,
    FastDiv nps_div, int64_t n_samp, const int32_t * __restrict__ d_idx, const double * __restrict__ signal) {
    constexpr int NNZ = 3;
    const int det0 = E * blockIdx.x;
    bool on[E];
    const int64_t * prow[E];
    const double * wrow[E];
    const uint8_t * frow[E];
    const double * srow[E];
    double ds[E];
    int64_t amp_offset[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        on[e] = det0 + e < n_det;
        const int det = on[e] ? det0 + e : det0;
        prow[e] = pixels + (int64_t)p_idx[det] * n_samp;
        wrow[e] = weights + (int64_t)w_idx[det] * n_samp * NNZ;
        frow[e] = use_dflags ? dflags + (int64_t)f_idx[det] * n_samp : nullptr;
        srow[e] = SIG ? signal + (int64_t)d_idx[det] * n_samp : nullptr;
        ds[e] = det_scale[det];
        amp_offset[e] = amp_offsets[det];
    }
    const int64_t nps = nps_div.d;
    const uint16_t dmask2 = (uint16_t)(dmask | (dmask << 8));
    const uint16_t smask2 = (uint16_t)(smask | (smask << 8));
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
            longlong2 pp[E];
            double2 w0[E], w1[E], w2[E], av[E], sg[E];
            uint16_t fd[E];
            uint8_t afa[E], afb[E];
            const uint16_t fs = use_sflags ? *reinterpret_cast<const uint16_t *>(sflags + s) : (uint16_t)0;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                pp[e] = *reinterpret_cast<const longlong2 *>(prow[e] + s);
                if constexpr (SIG) sg[e] = *reinterpret_cast<const double2 *>(srow[e] + s);
                fd[e] = use_dflags ? *reinterpret_cast<const uint16_t *>(frow[e] + s) : (uint16_t)0;
                const int64_t aa = amp_offset[e] + vaoff + step_a, ab = amp_offset[e] + vaoff + step_b;
                afa[e] = amp_flags[aa];
                afb[e] = amp_flags[ab];
                av[e] = make_double2(amps[aa], amps[ab]);
                const double2 * wv = reinterpret_cast<const double2 *>(wrow[e] + NNZ * s);
                w0[e] = wv[0];
                w1[e] = wv[1];
                w2[e] = wv[2];
            }
            int64_t ga[E], gb[E], la[E], lb[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                ga[e] = fastdiv(pp[e].x >= 0 ? pp[e].x : 0, nps_div);
                gb[e] = fastdiv(pp[e].y >= 0 ? pp[e].y : 0, nps_div);
                la[e] = g2l[ga[e]];
                lb[e] = g2l[gb[e]];
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const uint16_t bad = (uint16_t)((fd[e] & dmask2) | (fs & smask2));
                // (la / lb < 0: the pixel lies in a submap that is not local -- no contribution, also not through the pair merge)
                const bool good_a = active & on[e] & (pp[e].x >= 0) & (la[e] >= 0) & ((bad & 0x00ff) == 0);
                const bool good_b = active & on[e] & (pp[e].y >= 0) & (lb[e] >= 0) & ((bad & 0xff00) == 0);
                ka[e] = good_a ? la[e] * nps + (pp[e].x - ga[e] * nps) : -1;
                kb[e] = good_b ? lb[e] * nps + (pp[e].y - gb[e] * nps) : -1;
                // tod = 0 + amplitude (unflagged amplitudes only), then * det_scale
                double ta = (afa[e] == 0) ? (0.0 + av[e].x) : 0.0, tb = (afb[e] == 0) ? (0.0 + av[e].y) : 0.0;
                if constexpr (SIG) {
                    ta = sg[e].x - ta;
                    tb = sg[e].y - tb;
                }
                const double sa = ta * ds[e], sb = tb * ds[e];
                va[e][0] = good_a ? sa * w0[e].x : 0.0;
                va[e][1] = good_a ? sa * w0[e].y : 0.0;
                va[e][2] = good_a ? sa * w1[e].x : 0.0;
                vb[e][0] = good_b ? sb * w1[e].y : 0.0;
                vb[e][1] = good_b ? sb * w2[e].x : 0.0;
                vb[e][2] = good_b ? sb * w2[e].y : 0.0;
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
        const int tail = (c.count - head) & 1;
        if ((threadIdx.x == 0 && head) || (threadIdx.x == 1 && tail)) {
            const int64_t s = (threadIdx.x == 0) ? c.first : c.first + c.count - 1;
            const uint8_t fs = use_sflags ? sflags[s] : (uint8_t)0;
            const int64_t astep = fastdiv(s - vfirst, step_div);
            for (int e = 0; e < E; ++e) {
                if (!on[e]) continue;
                const int64_t p = prow[e][s];
                const uint8_t fd = use_dflags ? frow[e][s] : (uint8_t)0;
                if ((p < 0) | ((fd & dmask) != 0) | ((fs & smask) != 0)) continue;
                const int64_t gsm = fastdiv(p, nps_div);
                const int64_t key = g2l[gsm] * nps + (p - gsm * nps);
                if (key < 0) continue;
                const int64_t a = amp_offset[e] + vaoff + astep;
                double t = (amp_flags[a] == 0) ? (0.0 + amps[a]) : 0.0;
                if constexpr (SIG) t = srow[e][s] - t;
                const double sd = t * ds[e];
                const double * w = wrow[e] + NNZ * s;
                double * z = zmap + NNZ * key;
                for (int k = 0; k < NNZ; ++k) unsafeAtomicAdd(z + k, sd * w[k]);
            }
        }
    }
}

// Two consecutive samples per lane (nnz = 3; see k_scan_map_v2 / k_build_noise_weighted_v2): 16-byte lane accesses for
// pixels, weights (3 x) and -- SIGBUF -- the signal, the two flags as one 2-byte load; the global2local and map gathers
// of both samples go out together; the amplitude keys of the pair (equal except at a baseline boundary) are reduced over
// 128 samples per wave by one segmented scan (scatter_runs2).  Per-sample arithmetic as in k_offset_scan_project.
template <bool SIGBUF>
__global__ __launch_bounds__(kThreads) void k_offset_scan_project_v2(
    const Chunk * __restrict__ chunks, int n_chunks, const int64_t * __restrict__ view_first,
    const int64_t * __restrict__ view_aoff, FastDiv step_div, const int64_t * __restrict__ amp_offsets,
    const double * __restrict__ amps_in, double * __restrict__ amps_out,
    const uint8_t * __restrict__ amp_flags, const int32_t * __restrict__ p_idx,
    const int32_t * __restrict__ w_idx, const int32_t * __restrict__ f_idx,
    const double * __restrict__ det_w, const int64_t * __restrict__ g2l,
    const double * __restrict__ map, const int64_t * __restrict__ pixels,
    const double * __restrict__ weights, const uint8_t * __restrict__ flags, uint8_t fmask,
    int use_flags, FastDiv nps_div, int64_t n_samp, const int32_t * __restrict__ s_idx,
    const double * __restrict__ signal) {
    const int det = blockIdx.x;
    const double * srow = SIGBUF ? signal + (int64_t)s_idx[det] * n_samp : nullptr;
    const int64_t * prow = pixels + (int64_t)p_idx[det] * n_samp;
    const double * wrow = weights + (int64_t)w_idx[det] * n_samp * 3;
    const uint8_t * frow = use_flags ? flags + (int64_t)f_idx[det] * n_samp : nullptr;
    const double dw = det_w[det];
    const int64_t amp_offset = amp_offsets[det];
    const uint16_t fmask2 = (uint16_t)(fmask | (fmask << 8));
    // one sample: everything after the stream loads (global2local and the map value are gathered by the caller)
    auto finish = [&](bool hit, double av, double w0, double w1, double w2, double m0, double m1, double m2) {
        double sc = 0.0;
        sc += w0 * m0;
        sc += w1 * m1;
        sc += w2 * m2;
        sc *= 1.0;
        const double d = SIGBUF ? av : 0.0 + av;
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
            // streams (inactive lanes re-read pair 0: same lines) and the amplitude look-ups
            const int64_t aa = abase + fastdiv(s - vfirst, step_div);
            const int64_t ab = abase + fastdiv(s + 1 - vfirst, step_div);
            const uint8_t afa = amp_flags[aa], afb = amp_flags[ab];
            double2 av;
            if (SIGBUF) {
                av = *reinterpret_cast<const double2 *>(srow + s);
            } else {
                av = make_double2(amps_in[aa], amps_in[ab]);
            }
            const uint16_t fl = use_flags ? *reinterpret_cast<const uint16_t *>(frow + s) : (uint16_t)0;
            const longlong2 pp = *reinterpret_cast<const longlong2 *>(prow + s);
            const double2 * wv = reinterpret_cast<const double2 *>(wrow + 3 * s);
            const double2 wa = wv[0], wb = wv[1], wc = wv[2];
            const int64_t ga = scan_submap(pp.x, nps_div), gb = scan_submap(pp.y, nps_div);
            const int64_t la = g2l[ga], lb = g2l[gb];
            const ScanGather qa = scan_locate(pp.x, ga, la, nps_div), qb = scan_locate(pp.y, gb, lb, nps_div);
            const double * ma = map + qa.off;
            const double * mb = map + qb.off;
            const double a0 = ma[0], a1 = ma[1], a2 = ma[2], b0 = mb[0], b1 = mb[1], b2 = mb[2];
            const double da = finish(qa.hit, av.x, wa.x, wa.y, wb.x, a0, a1, a2);
            const double db = finish(qb.hit, av.y, wb.y, wc.x, wc.y, b0, b1, b2);
            const uint16_t bad = (uint16_t)(fl & fmask2);
            int64_t ka = (active && afa == 0) ? aa : (int64_t)-1;
            int64_t kb = (active && afb == 0) ? ab : (int64_t)-1;
            double va[1] = {(ka >= 0 && (bad & 0x00ff) == 0) ? da * dw : 0.0};
            double vb[1] = {(kb >= 0 && (bad & 0xff00) == 0) ? db * dw : 0.0};
            scatter_runs2<1>(ka, va, kb, vb, amps_out);
        }
        // the peeled first sample (lane 0) and the odd last one (lane 1)
        const int tail = (c.count - head) & 1;
        if ((threadIdx.x == 0 && head) || (threadIdx.x == 1 && tail)) {
            const int64_t s = (threadIdx.x == 0) ? c.first : c.first + c.count - 1;
            const int64_t a = abase + fastdiv(s - vfirst, step_div);
            if (amp_flags[a] == 0 && !(use_flags && (frow[s] & fmask))) {
                const int64_t p = prow[s];
                const int64_t gs = scan_submap(p, nps_div);
                const ScanGather q = scan_locate(p, gs, g2l[gs], nps_div);
                const double * m = map + q.off;
                const double * w = wrow + 3 * s;
                const double d = finish(q.hit, SIGBUF ? srow[s] : amps_in[a], w[0], w[1], w[2], m[0], m[1], m[2]);
                unsafeAtomicAdd(amps_out + a, d * dw);
            }
        }
    }
}

__global__ __launch_bounds__(kThreads) void k_offset_apply_diag_precond(
    int64_t n_amp, const double * __restrict__ var, const double * __restrict__ in,
    const uint8_t * __restrict__ flags, double * __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n_amp;
         i += (int64_t)gridDim.x * kThreads) {
        out[i] = (flags[i] == 0) ? in[i] * var[i] : 0.0;
    }
}

// ------------------------------------------------------------------------------------
// per-operation math probe (tests)
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_test_math(int op, int64_t n, const double * __restrict__ a,
                                                        const double * __restrict__ b,
                                                        double * __restrict__ out) {
    __shared__ double s_tab[2 * TOAST_ATAN_TABLE_N];
    if (threadIdx.x < 2 * TOAST_ATAN_TABLE_N) s_tab[threadIdx.x] = kAtanTab[threadIdx.x];
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * kThreads) {
        double r;
        if (op == 0) {
            r = atan2_dd(a[i], b[i], s_tab);
        } else if (op == 1) {
            r = f_sqrt(a[i]);
        } else {
            r = a[i] / b[i];
        }
        out[i] = r;
    }
}

}  // namespace

namespace {

inline bool rows_16b(const void * p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <typename T>
void launch_scan_map(dim3 grid, hipStream_t st, const Chunk * ch, int n_ch, const int32_t * di,
                     const int32_t * pi, const int32_t * wi, const int64_t * g2l, const void * map,
                     double * tod, const int64_t * pix, const double * w, int nnz, FastDiv dv,
                     double scale, int zero, int sub, int mult, const double * det_w,
                     int64_t n_samp) {
    const T * m = static_cast<const T *>(map);
    const int dm = det_major_grid() ? 1 : 0;
    if (nnz == 3 && !dm && vec2_lanes() && (n_samp & 1) == 0 && rows_16b(tod) && rows_16b(pix) && rows_16b(w)) {
        hipLaunchKernelGGL((k_scan_map_v2<T>), grid, dim3(kThreads), 0, st, ch, n_ch, di, pi, wi, g2l, m, tod, pix, w,
                           dv, scale, zero, sub, mult, det_w, n_samp);
        return;
    }
    if (dm) grid = dim3(grid.y, grid.x, 1);
    if (nnz == 3) {
        hipLaunchKernelGGL((k_scan_map<T, 3>), grid, dim3(kThreads), 0, st, ch, n_ch, di, pi, wi, g2l,
                           m, tod, pix, w, nnz, dv, scale, zero, sub, mult, det_w, n_samp, dm);
    } else if (nnz == 1) {
        hipLaunchKernelGGL((k_scan_map<T, 1>), grid, dim3(kThreads), 0, st, ch, n_ch, di, pi, wi, g2l,
                           m, tod, pix, w, nnz, dv, scale, zero, sub, mult, det_w, n_samp, dm);
    } else {
        hipLaunchKernelGGL((k_scan_map<T, 0>), grid, dim3(kThreads), 0, st, ch, n_ch, di, pi, wi, g2l,
                           m, tod, pix, w, nnz, dv, scale, zero, sub, mult, det_w, n_samp, dm);
    }
}

}  // namespace

// ------------------------------------------------------------------------------------
// C ABI, device-pointer level
// ------------------------------------------------------------------------------------
extern "C" {

int toast_hip_pointing_detector_dev(const double * focalplane, const double * d_boresight,
                                    const int32_t * quat_index, int64_t n_det, double * d_quats,
                                    int64_t n_samp, const toast_hip_interval * intervals,
                                    int64_t n_view, const uint8_t * d_shared_flags, int64_t n_flags,
                                    uint8_t mask, void * stream) {
    return guarded([&] {
        if (n_det <= 0) return;
        need_aligned(d_boresight, "boresight");
        need_aligned(d_quats, "quats");
        const auto chunks = make_chunks(intervals, n_view, n_samp);
        if (chunks.empty()) return;
        ParamBlock pb;
        const size_t o_ch = pb.push_vec(chunks);
        const size_t o_fp = pb.push(focalplane, sizeof(double) * 4 * n_det);
        const size_t o_qi = pb.push(quat_index, sizeof(int32_t) * n_det);
        const char * d = pb.commit(as_stream(stream));
        const int use_flags = (n_flags == n_samp) ? 1 : 0;
        hipLaunchKernelGGL(k_pointing_detector, chunk_grid(n_det, chunks.size()), dim3(kThreads), 0,
                           as_stream(stream), (const Chunk *)(d + o_ch), (int)chunks.size(),
                           (const double *)(d + o_fp), (const int32_t *)(d + o_qi), d_boresight,
                           d_quats, d_shared_flags, mask, use_flags, n_samp);
        check_launch();
    });
}

int toast_hip_pixels_healpix_dev(const int32_t * quat_index, int64_t n_det, const double * d_quats,
                                 const uint8_t * d_shared_flags, int64_t n_flags, uint8_t mask,
                                 const int32_t * pixel_index, int64_t * d_pixels, int64_t n_samp,
                                 const toast_hip_interval * intervals, int64_t n_view,
                                 uint8_t * d_hit_submaps, int64_t n_submap, int64_t n_pix_submap,
                                 int64_t nside, int nest, void * stream) {
    return guarded([&] {
        if (n_det <= 0) return;
        need_aligned(d_quats, "quats");
        const int factor = log2_exact(nside);
        if (n_pix_submap <= 0) fail_arg("n_pix_submap must be positive");
        if (n_submap * n_pix_submap < 12 * nside * nside) {
            fail_arg("hit_submaps is too short for this nside / n_pix_submap");
        }
        const auto chunks = make_chunks(intervals, n_view, n_samp);
        if (chunks.empty()) return;
        ParamBlock pb;
        const size_t o_ch = pb.push_vec(chunks);
        const size_t o_qi = pb.push(quat_index, sizeof(int32_t) * n_det);
        const size_t o_pi = pb.push(pixel_index, sizeof(int32_t) * n_det);
        const char * d = pb.commit(as_stream(stream));
        const int use_flags = (n_flags == n_samp) ? 1 : 0;
        const FastDiv dv = make_fastdiv(n_pix_submap);
        const bool pair = pair_detectors() && n_det >= 2;
        auto kern = pair ? (nest ? k_pixels_healpix<true, 2> : k_pixels_healpix<false, 2>)
                         : (nest ? k_pixels_healpix<true, 1> : k_pixels_healpix<false, 1>);
        hipLaunchKernelGGL(kern, chunk_grid(pair ? (n_det + 1) / 2 : n_det, chunks.size()), dim3(kThreads), 0,
                           as_stream(stream), (const Chunk *)(d + o_ch), (int)chunks.size(), (int)n_det,
                           (const int32_t *)(d + o_qi), (const int32_t *)(d + o_pi), d_quats,
                           d_shared_flags, mask, use_flags, d_pixels, d_hit_submaps, dv, nside,
                           factor, n_samp);
        check_launch();
    });
}

static int stokes_weights_pol_dev(int n_out, const int32_t * quat_index, int64_t n_det, const double * d_quats,
                                     const int32_t * weight_index, double * d_weights, int64_t n_samp,
                                     const double * d_hwp, int64_t n_hwp,
                                     const toast_hip_interval * intervals, int64_t n_view,
                                     const double * epsilon, const double * gamma, const double * cal,
                                     int iau, void * stream) {
    return guarded([&] {
        if (n_det <= 0) return;
        need_aligned(d_quats, "quats");
        const auto chunks = make_chunks(intervals, n_view, n_samp);
        if (chunks.empty()) return;
        ParamBlock pb;
        const size_t o_ch = pb.push_vec(chunks);
        const size_t o_qi = pb.push(quat_index, sizeof(int32_t) * n_det);
        const size_t o_wi = pb.push(weight_index, sizeof(int32_t) * n_det);
        const size_t o_e = pb.push(epsilon, sizeof(double) * n_det);
        const size_t o_g = pb.push(gamma, sizeof(double) * n_det);
        const size_t o_c = pb.push(cal, sizeof(double) * n_det);
        const char * d = pb.commit(as_stream(stream));
        const bool use_hwp = (n_hwp == n_samp);
        const double usign = iau ? -1.0 : 1.0;
        auto kern = (n_out == 3) ? (use_hwp ? k_stokes_iqu<true, 3> : k_stokes_iqu<false, 3>)
                                 : (use_hwp ? k_stokes_iqu<true, 2> : k_stokes_iqu<false, 2>);
        hipLaunchKernelGGL(kern, chunk_grid(n_det, chunks.size()), dim3(kThreads), 0,
                           as_stream(stream), (const Chunk *)(d + o_ch), (int)chunks.size(),
                           (const int32_t *)(d + o_qi), (const int32_t *)(d + o_wi), d_quats,
                           d_weights, d_hwp, (const double *)(d + o_e), (const double *)(d + o_g),
                           (const double *)(d + o_c), usign, n_samp, stokes_reference_nan() ? 1 : 0);
        check_launch();
    });
}

int toast_hip_stokes_weights_IQU_dev(const int32_t * quat_index, int64_t n_det, const double * d_quats,
                                     const int32_t * weight_index, double * d_weights, int64_t n_samp,
                                     const double * d_hwp, int64_t n_hwp,
                                     const toast_hip_interval * intervals, int64_t n_view,
                                     const double * epsilon, const double * gamma, const double * cal,
                                     int iau, void * stream) {
    return stokes_weights_pol_dev(3, quat_index, n_det, d_quats, weight_index, d_weights, n_samp, d_hwp, n_hwp, intervals,
                                  n_view, epsilon, gamma, cal, iau, stream);
}

int toast_hip_stokes_weights_QU_dev(const int32_t * quat_index, int64_t n_det, const double * d_quats,
                                    const int32_t * weight_index, double * d_weights, int64_t n_samp,
                                    const double * d_hwp, int64_t n_hwp,
                                    const toast_hip_interval * intervals, int64_t n_view,
                                    const double * epsilon, const double * gamma, const double * cal,
                                    int iau, void * stream) {
    return stokes_weights_pol_dev(2, quat_index, n_det, d_quats, weight_index, d_weights, n_samp, d_hwp, n_hwp, intervals,
                                  n_view, epsilon, gamma, cal, iau, stream);
}

int toast_hip_stokes_weights_I_dev(const int32_t * weight_index, int64_t n_det, double * d_weights,
                                   int64_t n_samp, const toast_hip_interval * intervals,
                                   int64_t n_view, const double * cal, void * stream) {
    return guarded([&] {
        if (n_det <= 0) return;
        const auto chunks = make_chunks(intervals, n_view, n_samp);
        if (chunks.empty()) return;
        ParamBlock pb;
        const size_t o_ch = pb.push_vec(chunks);
        const size_t o_wi = pb.push(weight_index, sizeof(int32_t) * n_det);
        const size_t o_c = pb.push(cal, sizeof(double) * n_det);
        const char * d = pb.commit(as_stream(stream));
        hipLaunchKernelGGL(k_stokes_i, chunk_grid(n_det, chunks.size()), dim3(kThreads), 0,
                           as_stream(stream), (const Chunk *)(d + o_ch), (int)chunks.size(),
                           (const int32_t *)(d + o_wi), d_weights, (const double *)(d + o_c), n_samp);
        check_launch();
    });
}


int toast_hip_scan_map_dev(int map_dtype, const int64_t * d_g2l, int64_t n_pix_submap,
                           const void * d_mapdata, int64_t nnz, double * d_det_data,
                           const int32_t * data_index, const int64_t * d_pixels,
                           const int32_t * pixel_index, const double * d_weights,
                           const int32_t * weight_index, int64_t n_det, int64_t n_samp,
                           const toast_hip_interval * intervals, int64_t n_view, double data_scale,
                           int should_zero, int should_subtract, int should_scale,
                           const double * det_weights, void * stream) {
    return guarded([&] {
        if (n_det <= 0) return;
        if (nnz <= 0) fail_arg("nnz must be positive");
        if (n_pix_submap <= 0) fail_arg("n_pix_submap must be positive");
        const auto chunks = make_chunks(intervals, n_view, n_samp);
        if (chunks.empty()) return;
        ParamBlock pb;
        const size_t o_ch = pb.push_vec(chunks);
        const size_t o_di = pb.push(data_index, sizeof(int32_t) * n_det);
        const size_t o_pi = pb.push(pixel_index, sizeof(int32_t) * n_det);
        const size_t o_wi = pb.push(weight_index, sizeof(int32_t) * n_det);
        const size_t o_dw = det_weights ? pb.push(det_weights, sizeof(double) * n_det) : 0;
        const char * d = pb.commit(as_stream(stream));
        const double * dw = det_weights ? (const double *)(d + o_dw) : nullptr;
        const FastDiv dv = make_fastdiv(n_pix_submap);
        const dim3 grid = chunk_grid(n_det, chunks.size());
        const Chunk * ch = (const Chunk *)(d + o_ch);
        const int32_t * di = (const int32_t *)(d + o_di);
        const int32_t * pi = (const int32_t *)(d + o_pi);
        const int32_t * wi = (const int32_t *)(d + o_wi);
        const int n_ch = (int)chunks.size();
        hipStream_t st = as_stream(stream);
        switch (map_dtype) {
            case TOAST_HIP_MAP_F64:
                launch_scan_map<double>(grid, st, ch, n_ch, di, pi, wi, d_g2l, d_mapdata, d_det_data,
                                        d_pixels, d_weights, (int)nnz, dv, data_scale, should_zero,
                                        should_subtract, should_scale, dw, n_samp);
                break;
            case TOAST_HIP_MAP_F32:
                launch_scan_map<float>(grid, st, ch, n_ch, di, pi, wi, d_g2l, d_mapdata, d_det_data,
                                       d_pixels, d_weights, (int)nnz, dv, data_scale, should_zero,
                                       should_subtract, should_scale, dw, n_samp);
                break;
            case TOAST_HIP_MAP_I64:
                launch_scan_map<int64_t>(grid, st, ch, n_ch, di, pi, wi, d_g2l, d_mapdata, d_det_data,
                                         d_pixels, d_weights, (int)nnz, dv, data_scale, should_zero,
                                         should_subtract, should_scale, dw, n_samp);
                break;
            case TOAST_HIP_MAP_I32:
                launch_scan_map<int32_t>(grid, st, ch, n_ch, di, pi, wi, d_g2l, d_mapdata, d_det_data,
                                         d_pixels, d_weights, (int)nnz, dv, data_scale, should_zero,
                                         should_subtract, should_scale, dw, n_samp);
                break;
            default:
                fail_arg("unknown map_dtype");
        }
        check_launch();
    });
}

int toast_hip_build_noise_weighted_dev(
    const int64_t * d_g2l, double * d_zmap, int64_t n_pix_submap, int64_t nnz,
    const int32_t * pixel_index, const int64_t * d_pixels, const int32_t * weight_index,
    const double * d_weights, const int32_t * data_index, const double * d_det_data,
    const int32_t * flag_index, const uint8_t * d_det_flags, int64_t n_flag_samp,
    const double * det_scale, uint8_t det_flag_mask, int64_t n_det, int64_t n_samp,
    const toast_hip_interval * intervals, int64_t n_view, const uint8_t * d_shared_flags,
    int64_t n_shared_flags, uint8_t shared_flag_mask, void * stream) {
    return guarded([&] {
        if (n_det <= 0) return;
        if (nnz <= 0) fail_arg("nnz must be positive");
        if (n_pix_submap <= 0) fail_arg("n_pix_submap must be positive");
        const auto chunks = make_chunks(intervals, n_view, n_samp);
        if (chunks.empty()) return;
        const int use_d = (n_flag_samp == n_samp) ? 1 : 0;
        const int use_s = (n_shared_flags == n_samp) ? 1 : 0;
        if (deterministic_mode()) {
            // debug mode: the reference host path's summation order, bit for bit (deterministic.hip)
            deterministic_scatter(0, d_g2l, d_zmap, n_pix_submap, nnz, pixel_index, d_pixels, weight_index,
                                  d_weights, data_index, d_det_data, flag_index, d_det_flags, use_d, det_scale,
                                  det_flag_mask, n_det, n_samp, intervals, n_view, d_shared_flags, use_s,
                                  shared_flag_mask, as_stream(stream));
            return;
        }
        std::vector<int32_t> fidx(n_det, 0);
        if (use_d) std::memcpy(fidx.data(), flag_index, sizeof(int32_t) * n_det);
        ParamBlock pb;
        const size_t o_ch = pb.push_vec(chunks);
        const size_t o_pi = pb.push(pixel_index, sizeof(int32_t) * n_det);
        const size_t o_wi = pb.push(weight_index, sizeof(int32_t) * n_det);
        const size_t o_di = pb.push(data_index, sizeof(int32_t) * n_det);
        const size_t o_fi = pb.push_vec(fidx);
        const size_t o_ds = pb.push(det_scale, sizeof(double) * n_det);
        const char * d = pb.commit(as_stream(stream));
        const FastDiv dv = make_fastdiv(n_pix_submap);
        const dim3 grid = chunk_grid(n_det, chunks.size());
        hipStream_t st = as_stream(stream);
#define TH_BNW_ARGS                                                                               \
    (const Chunk *)(d + o_ch), (int)chunks.size(), (const int32_t *)(d + o_pi),                   \
        (const int32_t *)(d + o_wi), (const int32_t *)(d + o_di), (const int32_t *)(d + o_fi),    \
        (const double *)(d + o_ds), d_g2l, d_zmap, d_pixels, d_weights, d_det_data, d_det_flags,  \
        det_flag_mask, use_d, d_shared_flags, shared_flag_mask, use_s, dv, n_samp
        const int dm = det_major_grid() ? 1 : 0;
        const dim3 g2 = dm ? dim3(grid.y, grid.x, 1) : grid;
        const bool v2 = vec2_lanes() && nnz == 3 && !dm && (n_samp & 1) == 0 && rows_16b(d_pixels) &&
                        rows_16b(d_weights) && rows_16b(d_det_data) && (!use_d || rows_16b(d_det_flags)) &&
                        (!use_s || rows_16b(d_shared_flags));
        if (v2) {
            const bool pr = pair_detectors() && n_det >= 2;
            const dim3 gp((unsigned)(pr ? (n_det + 1) / 2 : n_det), grid.y, 1);
            hipLaunchKernelGGL(pr ? k_build_noise_weighted_v2<2> : k_build_noise_weighted_v2<1>, gp, dim3(kThreads), 0, st,
                               (const Chunk *)(d + o_ch), (int)chunks.size(), (int)n_det, (const int32_t *)(d + o_pi),
                               (const int32_t *)(d + o_wi), (const int32_t *)(d + o_di), (const int32_t *)(d + o_fi),
                               (const double *)(d + o_ds), d_g2l, d_zmap, d_pixels, d_weights, d_det_data, d_det_flags,
                               det_flag_mask, use_d, d_shared_flags, shared_flag_mask, use_s, dv, n_samp);
        } else if (pair_detectors() && !dm && (nnz == 3 || nnz == 1) && n_det >= 2) {
            const dim3 gp((unsigned)((n_det + 1) / 2), grid.y, 1);
#define TH_BNW_PAIR_ARGS                                                                          \
    (const Chunk *)(d + o_ch), (int)chunks.size(), (int)n_det, (const int32_t *)(d + o_pi),       \
        (const int32_t *)(d + o_wi), (const int32_t *)(d + o_di), (const int32_t *)(d + o_fi),    \
        (const double *)(d + o_ds), d_g2l, d_zmap, d_pixels, d_weights, d_det_data, d_det_flags,  \
        det_flag_mask, use_d, d_shared_flags, shared_flag_mask, use_s, dv, n_samp
            if (nnz == 3) {
                hipLaunchKernelGGL(k_build_noise_weighted_pair<3>, gp, dim3(kThreads), 0, st, TH_BNW_PAIR_ARGS);
            } else {
                hipLaunchKernelGGL(k_build_noise_weighted_pair<1>, gp, dim3(kThreads), 0, st, TH_BNW_PAIR_ARGS);
            }
#undef TH_BNW_PAIR_ARGS
        } else if (nnz == 3) {
            hipLaunchKernelGGL(k_build_noise_weighted<3>, g2, dim3(kThreads), 0, st, TH_BNW_ARGS, dm);
        } else if (nnz == 1) {
            hipLaunchKernelGGL(k_build_noise_weighted<1>, g2, dim3(kThreads), 0, st, TH_BNW_ARGS, dm);
        } else if (nnz == 2) {
            hipLaunchKernelGGL(k_build_noise_weighted<2>, g2, dim3(kThreads), 0, st, TH_BNW_ARGS, dm);
        } else {
            hipLaunchKernelGGL(k_build_noise_weighted_any, grid, dim3(kThreads), 0, st, TH_BNW_ARGS,
                               (int)nnz);
        }
#undef TH_BNW_ARGS
        check_launch();
    });
}

int toast_hip_noise_weight_dev(double * d_det_data, int64_t n_samp, const int32_t * data_index,
                               int64_t n_det, const toast_hip_interval * intervals, int64_t n_view,
                               const double * detector_weights, void * stream) {
    return guarded([&] {
        if (n_det <= 0) return;
        const auto chunks = make_chunks(intervals, n_view, n_samp);
        if (chunks.empty()) return;
        ParamBlock pb;
        const size_t o_ch = pb.push_vec(chunks);
        const size_t o_di = pb.push(data_index, sizeof(int32_t) * n_det);
        const size_t o_w = pb.push(detector_weights, sizeof(double) * n_det);
        const char * d = pb.commit(as_stream(stream));
        const bool v2 = vec2_lanes() && (n_samp & 1) == 0 && rows_16b(d_det_data);
        hipLaunchKernelGGL(v2 ? k_noise_weight_v2 : k_noise_weight, chunk_grid(n_det, chunks.size()), dim3(kThreads), 0,
                           as_stream(stream), (const Chunk *)(d + o_ch), (int)chunks.size(),
                           (const int32_t *)(d + o_di), (const double *)(d + o_w), d_det_data, n_samp);
        check_launch();
    });
}

int toast_hip_healpix_ang2vec_dev(int64_t n, const double * d_theta, const double * d_phi, double * d_vec, void * stream) {
    return guarded([&] {
        if (n <= 0) return;
        hipLaunchKernelGGL(k_healpix_ang2vec, flat_grid(n), dim3(kThreads), 0, as_stream(stream), n, d_theta, d_phi, d_vec);
        check_launch();
    });
}

int toast_hip_healpix_vec2ang_dev(int64_t n, const double * d_vec, double * d_theta, double * d_phi, void * stream) {
    return guarded([&] {
        if (n <= 0) return;
        hipLaunchKernelGGL(k_healpix_vec2ang, flat_grid(n), dim3(kThreads), 0, as_stream(stream), n, d_vec, d_theta, d_phi);
        check_launch();
    });
}

int toast_hip_healpix_ang2pix_dev(int64_t nside, int nest, int64_t n, const double * d_theta, const double * d_phi,
                                 int64_t * d_pix, void * stream) {
    return guarded([&] {
        if (n <= 0) return;
        const int factor = log2_exact(nside);
        if (nest) {
            hipLaunchKernelGGL(k_healpix_ang2pix<true>, flat_grid(n), dim3(kThreads), 0, as_stream(stream), n, d_theta,
                               d_phi, d_pix, nside, factor);
        } else {
            hipLaunchKernelGGL(k_healpix_ang2pix<false>, flat_grid(n), dim3(kThreads), 0, as_stream(stream), n, d_theta,
                               d_phi, d_pix, nside, factor);
        }
        check_launch();
    });
}

int toast_hip_healpix_convert_dev(int op, int64_t nside, int64_t levels, int64_t n, const int64_t * d_in,
                                 int64_t * d_out, void * stream) {
    return guarded([&] {
        if (n <= 0) return;
        if (op < 0 || op > 5) fail_arg("healpix_convert: op must be 0..5");
        const int factor = log2_exact(nside);
        if (levels < 0 || ((op == 2 || op == 4) && levels > factor) || ((op == 3 || op == 5) && factor + levels > 29)) {
            fail_arg("healpix_convert: resolution change out of range");
        }
        hipLaunchKernelGGL(k_healpix_convert, flat_grid(n), dim3(kThreads), 0, as_stream(stream), op, n, d_in, d_out,
                           nside, factor, levels);
        check_launch();
    });
}

int toast_hip_healpix_vec2pix_dev(int64_t nside, int nest, int64_t n, const double * d_vec, int64_t * d_pix,
                                 void * stream) {
    return guarded([&] {
        if (n <= 0) return;
        const int factor = log2_exact(nside);
        if (nest) {
            hipLaunchKernelGGL(k_healpix_vec2pix<true>, flat_grid(n), dim3(kThreads), 0, as_stream(stream), n, d_vec,
                               d_pix, nside, factor);
        } else {
            hipLaunchKernelGGL(k_healpix_vec2pix<false>, flat_grid(n), dim3(kThreads), 0, as_stream(stream), n, d_vec,
                               d_pix, nside, factor);
        }
        check_launch();
    });
}

int toast_hip_cov_accum_diag_hits_dev(int64_t n_sub, int64_t subsize, int64_t n_samp, const int64_t * d_submap,
                                     const int64_t * d_subpix, int64_t * d_hits, void * stream) {
    return guarded([&] {
        if (n_sub < 1 || n_samp <= 0) return;
        hipLaunchKernelGGL((k_cov_accum<1, 0>), flat_grid(n_samp), dim3(kThreads), 0, as_stream(stream), n_samp, d_submap,
                           d_subpix, subsize, (const double *)nullptr, 1.0, (double *)nullptr,
                           reinterpret_cast<long long *>(d_hits), (const double *)nullptr);
        check_launch();
    });
}

int toast_hip_cov_accum_diag_invnpp_dev(int64_t n_sub, int64_t subsize, int64_t nnz, int64_t n_samp,
                                       const int64_t * d_submap, const int64_t * d_subpix, const double * d_weights,
                                       double scale, double * d_invnpp, void * stream) {
    return guarded([&] {
        if (n_sub < 1 || n_samp <= 0) return;
        const dim3 grid = flat_grid(n_samp);
        hipStream_t st = as_stream(stream);
#define TH_ACC(N)                                                                                              \
    hipLaunchKernelGGL((k_cov_accum<N, 1>), grid, dim3(kThreads), 0, st, n_samp, d_submap, d_subpix, subsize, \
                       d_weights, scale, d_invnpp, (long long *)nullptr, (const double *)nullptr)
        switch (nnz) {
            case 1: TH_ACC(1); break;
            case 2: TH_ACC(2); break;
            case 3: TH_ACC(3); break;
            case 4: TH_ACC(4); break;
            default: fail_arg("cov_accum_diag_invnpp: nnz must be 1..4");
        }
#undef TH_ACC
        check_launch();
    });
}

int toast_hip_cov_accum_zmap_dev(int64_t n_sub, int64_t subsize, int64_t nnz, int64_t n_samp, const int64_t * d_submap,
                                const int64_t * d_subpix, const double * d_weights, double scale, const double * d_tod,
                                double * d_zmap, void * stream) {
    return guarded([&] {
        if (n_sub < 1 || n_samp <= 0) return;
        const dim3 grid = flat_grid(n_samp);
        hipStream_t st = as_stream(stream);
#define TH_ACC(N)                                                                                              \
    hipLaunchKernelGGL((k_cov_accum<N, 2>), grid, dim3(kThreads), 0, st, n_samp, d_submap, d_subpix, subsize, \
                       d_weights, scale, d_zmap, (long long *)nullptr, d_tod)
        switch (nnz) {
            case 1: TH_ACC(1); break;
            case 2: TH_ACC(2); break;
            case 3: TH_ACC(3); break;
            case 4: TH_ACC(4); break;
            default: fail_arg("cov_accum_zmap: nnz must be 1..4");
        }
#undef TH_ACC
        check_launch();
    });
}

// global pixel -> (local submap, pixel in submap)   [ref: src/libtoast/include/toast/map_pixels.hpp:11-41]
__global__ __launch_bounds__(kThreads) void k_global_to_local(int64_t n, const int64_t * __restrict__ gl,
                                                             FastDiv nps_div, const int64_t * __restrict__ g2l,
                                                             int64_t * __restrict__ lsm, int64_t * __restrict__ lpx) {
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        const int64_t p = gl[i];
        int64_t sm = -1, px = -1;
        if (p >= 0) {
            const int64_t gsm = fastdiv(p, nps_div);
            px = p - gsm * nps_div.d;
            sm = g2l[gsm];
        }
        lsm[i] = sm;
        lpx[i] = px;
    }
}

int toast_hip_global_to_local_dev(int64_t n, const int64_t * d_global_pixels, int64_t n_pix_submap,
                                 const int64_t * d_global2local, int64_t * d_local_submaps, int64_t * d_local_pixels,
                                 void * stream) {
    return guarded([&] {
        if (n <= 0) return;
        if (n_pix_submap <= 0) fail_arg("n_pix_submap must be positive");
        hipLaunchKernelGGL(k_global_to_local, flat_grid(n), dim3(kThreads), 0, as_stream(stream), n, d_global_pixels,
                           make_fastdiv(n_pix_submap), d_global2local, d_local_submaps, d_local_pixels);
        check_launch();
    });
}

int toast_hip_cov_mult_diag_dev(int64_t n_sub, int64_t subsize, int64_t nnz, double * d_data1, const double * d_data2,
                               void * stream) {
    return guarded([&] {
        const int64_t n_px = n_sub * subsize;
        if (n_px <= 0) return;
        const dim3 grid = flat_grid(n_px);
        hipStream_t st = as_stream(stream);
        switch (nnz) {
            case 1: hipLaunchKernelGGL(k_cov_mult_diag<1>, grid, dim3(kThreads), 0, st, n_px, d_data1, d_data2); break;
            case 2: hipLaunchKernelGGL(k_cov_mult_diag<2>, grid, dim3(kThreads), 0, st, n_px, d_data1, d_data2); break;
            case 3: hipLaunchKernelGGL(k_cov_mult_diag<3>, grid, dim3(kThreads), 0, st, n_px, d_data1, d_data2); break;
            case 4: hipLaunchKernelGGL(k_cov_mult_diag<4>, grid, dim3(kThreads), 0, st, n_px, d_data1, d_data2); break;
            default: fail_arg("cov_mult_diag: nnz must be 1..4");
        }
        check_launch();
    });
}

int toast_hip_cov_apply_diag_dev(int64_t n_sub, int64_t subsize, int64_t nnz, const double * d_mat,
                                 double * d_vec, void * stream) {
    return guarded([&] {
        const int64_t n_px = n_sub * subsize;
        if (n_px <= 0) return;
        const dim3 grid = flat_grid(n_px);
        hipStream_t st = as_stream(stream);
        switch (nnz) {
            case 1: hipLaunchKernelGGL(k_cov_apply_diag<1>, grid, dim3(kThreads), 0, st, n_px, d_mat, d_vec); break;
            case 2: hipLaunchKernelGGL(k_cov_apply_diag<2>, grid, dim3(kThreads), 0, st, n_px, d_mat, d_vec); break;
            case 3: hipLaunchKernelGGL(k_cov_apply_diag<3>, grid, dim3(kThreads), 0, st, n_px, d_mat, d_vec); break;
            case 4: hipLaunchKernelGGL(k_cov_apply_diag<4>, grid, dim3(kThreads), 0, st, n_px, d_mat, d_vec); break;
            default: fail_arg("cov_apply_diag: nnz must be 1..4");
        }
        check_launch();
    });
}


int toast_hip_template_offset_add_to_signal_multi_dev(
    int64_t step_length, const int64_t * amp_offsets, const int64_t * n_amp_views,
    const double * d_amplitudes, const uint8_t * d_amplitude_flags, const int32_t * data_index,
    int64_t n_det, double * d_det_data, int64_t n_samp, const toast_hip_interval * intervals,
    int64_t n_view, void * stream) {
    return guarded([&] {
        if (n_det <= 0) return;
        if (step_length <= 0) fail_arg("step_length must be positive");
        const auto chunks = make_chunks(intervals, n_view, n_samp);
        if (chunks.empty()) return;
        const OffsetViews ov = offset_views(intervals, n_amp_views, n_view);
        ParamBlock pb;
        const size_t o_ch = pb.push_vec(chunks);
        const size_t o_vf = pb.push_vec(ov.first);
        const size_t o_va = pb.push_vec(ov.aoff);
        const size_t o_ao = pb.push(amp_offsets, sizeof(int64_t) * n_det);
        const size_t o_di = pb.push(data_index, sizeof(int32_t) * n_det);
        const char * d = pb.commit(as_stream(stream));
        hipLaunchKernelGGL(k_offset_add_to_signal, chunk_grid(n_det, chunks.size()), dim3(kThreads), 0,
                           as_stream(stream), (const Chunk *)(d + o_ch), (int)chunks.size(),
                           (const int64_t *)(d + o_vf), (const int64_t *)(d + o_va),
                           make_fastdiv(step_length), (const int64_t *)(d + o_ao),
                           (const int32_t *)(d + o_di), d_amplitudes, d_amplitude_flags, d_det_data, n_samp);
        check_launch();
    });
}

int toast_hip_template_offset_add_to_signal_dev(
    int64_t step_length, int64_t amp_offset, const int64_t * n_amp_views, const double * d_amplitudes,
    const uint8_t * d_amplitude_flags, int32_t data_index, double * d_det_data, int64_t n_samp,
    const toast_hip_interval * intervals, int64_t n_view, void * stream) {
    return toast_hip_template_offset_add_to_signal_multi_dev(step_length, &amp_offset, n_amp_views,
                                                             d_amplitudes, d_amplitude_flags, &data_index, 1,
                                                             d_det_data, n_samp, intervals, n_view, stream);
}

int toast_hip_template_offset_project_signal_multi_dev(
    const int32_t * data_index, const double * d_det_data, const int32_t * flag_index,
    const uint8_t * d_flag_data, uint8_t flag_mask, int64_t step_length, const int64_t * amp_offsets,
    const int64_t * n_amp_views, double * d_amplitudes, const uint8_t * d_amplitude_flags, int64_t n_det,
    int64_t n_samp, const toast_hip_interval * intervals, int64_t n_view, void * stream) {
    return guarded([&] {
        if (n_det <= 0) return;
        if (step_length <= 0) fail_arg("step_length must be positive");
        const auto chunks = make_chunks(intervals, n_view, n_samp);
        if (chunks.empty()) return;
        const OffsetViews ov = offset_views(intervals, n_amp_views, n_view);
        const int use_flags = (flag_index != nullptr && d_flag_data != nullptr && flag_index[0] >= 0) ? 1 : 0;
        std::vector<int32_t> fidx(n_det, 0);
        if (use_flags) std::memcpy(fidx.data(), flag_index, sizeof(int32_t) * n_det);
        ParamBlock pb;
        const size_t o_ch = pb.push_vec(chunks);
        const size_t o_vf = pb.push_vec(ov.first);
        const size_t o_va = pb.push_vec(ov.aoff);
        const size_t o_ao = pb.push(amp_offsets, sizeof(int64_t) * n_det);
        const size_t o_di = pb.push(data_index, sizeof(int32_t) * n_det);
        const size_t o_fi = pb.push_vec(fidx);
        if (deterministic_mode()) {
            std::vector<int64_t> vlast(n_view), vnamp(n_view);
            int64_t n_amp_det = 0;
            for (int64_t v = 0; v < n_view; ++v) {
                vlast[v] = intervals[v].last;
                vnamp[v] = n_amp_views[v];
                n_amp_det += n_amp_views[v];
            }
            const size_t o_vl = pb.push_vec(vlast);
            const size_t o_vn = pb.push_vec(vnamp);
            const char * dd = pb.commit(as_stream(stream));
            if (n_amp_det > 0) {
                hipLaunchKernelGGL(k_offset_project_signal_det,
                                   dim3((unsigned)((n_amp_det + kThreads - 1) / kThreads), (unsigned)n_det), dim3(kThreads),
                                   0, as_stream(stream), (int)n_view, (const int64_t *)(dd + o_vf),
                                   (const int64_t *)(dd + o_vl), (const int64_t *)(dd + o_va), (const int64_t *)(dd + o_vn),
                                   n_amp_det, step_length, (const int64_t *)(dd + o_ao), (const int32_t *)(dd + o_di),
                                   (const int32_t *)(dd + o_fi), d_amplitudes, d_amplitude_flags, d_det_data, d_flag_data,
                                   flag_mask, use_flags, n_samp);
                check_launch();
            }
            return;
        }
        const char * d = pb.commit(as_stream(stream));
        hipLaunchKernelGGL(k_offset_project_signal, chunk_grid(n_det, chunks.size()), dim3(kThreads), 0,
                           as_stream(stream), (const Chunk *)(d + o_ch), (int)chunks.size(),
                           (const int64_t *)(d + o_vf), (const int64_t *)(d + o_va),
                           make_fastdiv(step_length), (const int64_t *)(d + o_ao),
                           (const int32_t *)(d + o_di), (const int32_t *)(d + o_fi), d_amplitudes,
                           d_amplitude_flags, d_det_data, d_flag_data, flag_mask, use_flags, n_samp);
        check_launch();
    });
}

int toast_hip_template_offset_project_signal_dev(
    int32_t data_index, const double * d_det_data, int32_t flag_index, const uint8_t * d_flag_data,
    uint8_t flag_mask, int64_t step_length, int64_t amp_offset, const int64_t * n_amp_views,
    double * d_amplitudes, const uint8_t * d_amplitude_flags, int64_t n_samp,
    const toast_hip_interval * intervals, int64_t n_view, void * stream) {
    return toast_hip_template_offset_project_signal_multi_dev(
        &data_index, d_det_data, &flag_index, d_flag_data, flag_mask, step_length, &amp_offset, n_amp_views,
        d_amplitudes, d_amplitude_flags, 1, n_samp, intervals, n_view, stream);
}

namespace {
// zmap += A^T N^-1 (M a), or -- d_signal != nullptr -- zmap += A^T N^-1 (d - M a)
int offset_accumulate_impl(
    int64_t step_length, const int64_t * amp_offsets, const int64_t * n_amp_views,
    const double * d_amplitudes, const uint8_t * d_amplitude_flags, const int64_t * d_g2l, double * d_zmap,
    int64_t n_pix_submap, int64_t nnz, const int32_t * pixel_index, const int64_t * d_pixels,
    const int32_t * weight_index, const double * d_weights, const int32_t * flag_index,
    const uint8_t * d_det_flags, int64_t n_flag_samp, const double * det_scale, uint8_t det_flag_mask,
    int64_t n_det, int64_t n_samp, const toast_hip_interval * intervals, int64_t n_view,
    const uint8_t * d_shared_flags, int64_t n_shared_flags, uint8_t shared_flag_mask, const int32_t * data_index,
    const double * d_signal, void * stream) {
    return guarded([&] {
        if (n_det <= 0) return;
        if (step_length <= 0) fail_arg("step_length must be positive");
        if (nnz != 1 && nnz != 3) fail_arg("offset_accumulate: nnz must be 1 or 3");
        const auto chunks = make_chunks(intervals, n_view, n_samp);
        if (chunks.empty()) return;
        const OffsetViews ov = offset_views(intervals, n_amp_views, n_view);
        const int use_d = (n_flag_samp == n_samp) ? 1 : 0;
        const int use_s = (n_shared_flags == n_samp) ? 1 : 0;
        std::vector<int32_t> fidx(n_det, 0);
        if (use_d) std::memcpy(fidx.data(), flag_index, sizeof(int32_t) * n_det);
        ParamBlock pb;
        const size_t o_ch = pb.push_vec(chunks);
        const size_t o_vf = pb.push_vec(ov.first);
        const size_t o_va = pb.push_vec(ov.aoff);
        const size_t o_ao = pb.push(amp_offsets, sizeof(int64_t) * n_det);
        const size_t o_pi = pb.push(pixel_index, sizeof(int32_t) * n_det);
        const size_t o_wi = pb.push(weight_index, sizeof(int32_t) * n_det);
        const size_t o_fi = pb.push_vec(fidx);
        const size_t o_ds = pb.push(det_scale, sizeof(double) * n_det);
        const size_t o_di = d_signal != nullptr ? pb.push(data_index, sizeof(int32_t) * n_det) : o_pi;
        const char * d = pb.commit(as_stream(stream));
        const dim3 grid = chunk_grid(n_det, chunks.size());
        hipStream_t st = as_stream(stream);
#define TH_OA_ARGS                                                                                  \
    (const Chunk *)(d + o_ch), (int)chunks.size(), (int)n_det, (const int64_t *)(d + o_vf),         \
        (const int64_t *)(d + o_va), make_fastdiv(step_length), (const int64_t *)(d + o_ao),        \
        d_amplitudes, d_amplitude_flags, (const int32_t *)(d + o_pi), (const int32_t *)(d + o_wi),  \
        (const int32_t *)(d + o_fi), (const double *)(d + o_ds), d_g2l, d_zmap, d_pixels, d_weights, \
        d_det_flags, det_flag_mask, use_d, d_shared_flags, shared_flag_mask, use_s,                 \
        make_fastdiv(n_pix_submap), n_samp
#define TH_OA_ARGS2 TH_OA_ARGS, (const int32_t *)(d + o_di), d_signal
        const bool can2 = nnz == 3 && (n_samp & 1) == 0 && rows_16b(d_pixels) && rows_16b(d_weights) &&
                          (!use_d || rows_16b(d_det_flags)) && (!use_s || rows_16b(d_shared_flags));
        const bool v2 = vec2_lanes() && can2;
        if (d_signal != nullptr) {
            // the cleaned-signal form exists for the two-samples-per-lane kernel only (what cached IQU pointing gets); the
            // tuning switch that compares one with two samples per lane (TOAST_HIP_VEC2=0) does not take it away
            if (!(can2 && rows_16b(d_signal))) {
                fail_arg("offset_clean_accumulate: needs nnz = 3, an even number of samples per row and 16-byte aligned rows");
            }
            const bool pr = pair_detectors() && n_det >= 2;
            const dim3 gp((unsigned)(pr ? (n_det + 1) / 2 : n_det), grid.y, 1);
            if (pr) {
                hipLaunchKernelGGL((k_offset_accumulate_v2<2, true>), gp, dim3(kThreads), 0, st, TH_OA_ARGS2);
            } else {
                hipLaunchKernelGGL((k_offset_accumulate_v2<1, true>), gp, dim3(kThreads), 0, st, TH_OA_ARGS2);
            }
        } else if (v2) {
            const bool pr = pair_detectors() && n_det >= 2;
            const dim3 gp((unsigned)(pr ? (n_det + 1) / 2 : n_det), grid.y, 1);
            if (pr) {
                hipLaunchKernelGGL((k_offset_accumulate_v2<2, false>), gp, dim3(kThreads), 0, st, TH_OA_ARGS2);
            } else {
                hipLaunchKernelGGL((k_offset_accumulate_v2<1, false>), gp, dim3(kThreads), 0, st, TH_OA_ARGS2);
            }
        } else if (pair_detectors() && n_det >= 2) {
            const dim3 gp((unsigned)((n_det + 1) / 2), grid.y, 1);
            if (nnz == 3) {
                hipLaunchKernelGGL((k_offset_accumulate<3, 2>), gp, dim3(kThreads), 0, st, TH_OA_ARGS);
            } else {
                hipLaunchKernelGGL((k_offset_accumulate<1, 2>), gp, dim3(kThreads), 0, st, TH_OA_ARGS);
            }
        } else if (nnz == 3) {
            hipLaunchKernelGGL((k_offset_accumulate<3, 1>), grid, dim3(kThreads), 0, st, TH_OA_ARGS);
        } else {
            hipLaunchKernelGGL((k_offset_accumulate<1, 1>), grid, dim3(kThreads), 0, st, TH_OA_ARGS);
        }
#undef TH_OA_ARGS2
#undef TH_OA_ARGS
        check_launch();
    });
}
}  // namespace

int toast_hip_offset_accumulate_dev(
    int64_t step_length, const int64_t * amp_offsets, const int64_t * n_amp_views,
    const double * d_amplitudes, const uint8_t * d_amplitude_flags, const int64_t * d_g2l, double * d_zmap,
    int64_t n_pix_submap, int64_t nnz, const int32_t * pixel_index, const int64_t * d_pixels,
    const int32_t * weight_index, const double * d_weights, const int32_t * flag_index,
    const uint8_t * d_det_flags, int64_t n_flag_samp, const double * det_scale, uint8_t det_flag_mask,
    int64_t n_det, int64_t n_samp, const toast_hip_interval * intervals, int64_t n_view,
    const uint8_t * d_shared_flags, int64_t n_shared_flags, uint8_t shared_flag_mask, void * stream) {
    return offset_accumulate_impl(step_length, amp_offsets, n_amp_views, d_amplitudes, d_amplitude_flags, d_g2l, d_zmap,
                                  n_pix_submap, nnz, pixel_index, d_pixels, weight_index, d_weights, flag_index, d_det_flags,
                                  n_flag_samp, det_scale, det_flag_mask, n_det, n_samp, intervals, n_view, d_shared_flags,
                                  n_shared_flags, shared_flag_mask, nullptr, nullptr, stream);
}

int toast_hip_offset_clean_accumulate_dev(
    int64_t step_length, const int64_t * amp_offsets, const int64_t * n_amp_views,
    const double * d_amplitudes, const uint8_t * d_amplitude_flags, const int64_t * d_g2l, double * d_zmap,
    int64_t n_pix_submap, int64_t nnz, const int32_t * pixel_index, const int64_t * d_pixels,
    const int32_t * weight_index, const double * d_weights, const int32_t * data_index, const double * d_signal,
    const int32_t * flag_index, const uint8_t * d_det_flags, int64_t n_flag_samp, const double * det_scale,
    uint8_t det_flag_mask, int64_t n_det, int64_t n_samp, const toast_hip_interval * intervals, int64_t n_view,
    const uint8_t * d_shared_flags, int64_t n_shared_flags, uint8_t shared_flag_mask, void * stream) {
    if (d_signal == nullptr || data_index == nullptr) {
        return guarded([&] { fail_arg("offset_clean_accumulate: the signal and its row indices must not be null"); });
    }
    return offset_accumulate_impl(step_length, amp_offsets, n_amp_views, d_amplitudes, d_amplitude_flags, d_g2l, d_zmap,
                                  n_pix_submap, nnz, pixel_index, d_pixels, weight_index, d_weights, flag_index, d_det_flags,
                                  n_flag_samp, det_scale, det_flag_mask, n_det, n_samp, intervals, n_view, d_shared_flags,
                                  n_shared_flags, shared_flag_mask, data_index, d_signal, stream);
}

static int offset_scan_project_launch(
    int64_t step_length, const int64_t * amp_offsets, const int64_t * n_amp_views,
    const double * d_amplitudes_in, double * d_amplitudes_out, const uint8_t * d_amplitude_flags,
    const int64_t * d_g2l, const double * d_map, int64_t n_pix_submap, int64_t nnz,
    const int32_t * pixel_index, const int64_t * d_pixels, const int32_t * weight_index,
    const double * d_weights, const int32_t * flag_index, const uint8_t * d_flag_data, uint8_t flag_mask,
    const double * det_weights, int64_t n_det, int64_t n_samp, const toast_hip_interval * intervals,
    int64_t n_view, const int32_t * signal_index, const double * d_signal, void * stream,
    const char * fn = __builtin_FUNCTION()) {
    return guarded([&] {
        if (n_det <= 0) return;
        const bool sigbuf = d_signal != nullptr;
        if (sigbuf && signal_index == nullptr) fail_arg("offset_scan_project: signal_index is NULL");
        if (step_length <= 0) fail_arg("step_length must be positive");
        if (nnz != 1 && nnz != 3) fail_arg("offset_scan_project: nnz must be 1 or 3");
        const auto chunks = make_chunks(intervals, n_view, n_samp);
        if (chunks.empty()) return;
        const OffsetViews ov = offset_views(intervals, n_amp_views, n_view);
        const int use_flags = (flag_index != nullptr && d_flag_data != nullptr) ? 1 : 0;
        std::vector<int32_t> fidx(n_det, 0);
        if (use_flags) std::memcpy(fidx.data(), flag_index, sizeof(int32_t) * n_det);
        ParamBlock pb;
        const size_t o_ch = pb.push_vec(chunks);
        const size_t o_vf = pb.push_vec(ov.first);
        const size_t o_va = pb.push_vec(ov.aoff);
        const size_t o_ao = pb.push(amp_offsets, sizeof(int64_t) * n_det);
        const size_t o_pi = pb.push(pixel_index, sizeof(int32_t) * n_det);
        const size_t o_wi = pb.push(weight_index, sizeof(int32_t) * n_det);
        const size_t o_fi = pb.push_vec(fidx);
        const size_t o_dw = pb.push(det_weights, sizeof(double) * n_det);
        std::vector<int32_t> sidx(n_det, 0);
        if (sigbuf) std::memcpy(sidx.data(), signal_index, sizeof(int32_t) * n_det);
        const size_t o_si = pb.push_vec(sidx);
        const char * d = pb.commit(as_stream(stream));
        const dim3 grid = chunk_grid(n_det, chunks.size());
        hipStream_t st = as_stream(stream);
#define TH_OS_ARGS                                                                                  \
    (const Chunk *)(d + o_ch), (int)chunks.size(), (const int64_t *)(d + o_vf),                     \
        (const int64_t *)(d + o_va), make_fastdiv(step_length), (const int64_t *)(d + o_ao),        \
        d_amplitudes_in, d_amplitudes_out, d_amplitude_flags, (const int32_t *)(d + o_pi),          \
        (const int32_t *)(d + o_wi), (const int32_t *)(d + o_fi), (const double *)(d + o_dw), d_g2l, \
        d_map, d_pixels, d_weights, d_flag_data, flag_mask, use_flags, make_fastdiv(n_pix_submap),  \
        n_samp, (const int32_t *)(d + o_si), d_signal
        const bool v2 = vec2_lanes() && nnz == 3 && (n_samp & 1) == 0 && rows_16b(d_pixels) && rows_16b(d_weights) &&
                        (!use_flags || rows_16b(d_flag_data)) && (!sigbuf || rows_16b(d_signal));
        if (v2) {
            if (sigbuf) {
                hipLaunchKernelGGL((k_offset_scan_project_v2<true>), grid, dim3(kThreads), 0, st, TH_OS_ARGS);
            } else {
                hipLaunchKernelGGL((k_offset_scan_project_v2<false>), grid, dim3(kThreads), 0, st, TH_OS_ARGS);
            }
        } else if (sigbuf) {
            if (nnz == 3) {
                hipLaunchKernelGGL((k_offset_scan_project<3, true>), grid, dim3(kThreads), 0, st, TH_OS_ARGS);
            } else {
                hipLaunchKernelGGL((k_offset_scan_project<1, true>), grid, dim3(kThreads), 0, st, TH_OS_ARGS);
            }
        } else if (nnz == 3) {
            hipLaunchKernelGGL((k_offset_scan_project<3, false>), grid, dim3(kThreads), 0, st, TH_OS_ARGS);
        } else {
            hipLaunchKernelGGL((k_offset_scan_project<1, false>), grid, dim3(kThreads), 0, st, TH_OS_ARGS);
        }
#undef TH_OS_ARGS
        check_launch();
    }, fn);
}

int toast_hip_offset_scan_project_dev(
    int64_t step_length, const int64_t * amp_offsets, const int64_t * n_amp_views,
    const double * d_amplitudes_in, double * d_amplitudes_out, const uint8_t * d_amplitude_flags,
    const int64_t * d_g2l, const double * d_map, int64_t n_pix_submap, int64_t nnz,
    const int32_t * pixel_index, const int64_t * d_pixels, const int32_t * weight_index,
    const double * d_weights, const int32_t * flag_index, const uint8_t * d_flag_data, uint8_t flag_mask,
    const double * det_weights, int64_t n_det, int64_t n_samp, const toast_hip_interval * intervals,
    int64_t n_view, void * stream) {
    return offset_scan_project_launch(step_length, amp_offsets, n_amp_views, d_amplitudes_in, d_amplitudes_out,
                                      d_amplitude_flags, d_g2l, d_map, n_pix_submap, nnz, pixel_index, d_pixels,
                                      weight_index, d_weights, flag_index, d_flag_data, flag_mask, det_weights, n_det,
                                      n_samp, intervals, n_view, nullptr, nullptr, stream);
}

int toast_hip_offset_scan_project_signal_dev(
    int64_t step_length, const int64_t * amp_offsets, const int64_t * n_amp_views, const int32_t * signal_index,
    const double * d_signal, double * d_amplitudes_out, const uint8_t * d_amplitude_flags, const int64_t * d_g2l,
    const double * d_map, int64_t n_pix_submap, int64_t nnz, const int32_t * pixel_index, const int64_t * d_pixels,
    const int32_t * weight_index, const double * d_weights, const int32_t * flag_index, const uint8_t * d_flag_data,
    uint8_t flag_mask, const double * det_weights, int64_t n_det, int64_t n_samp,
    const toast_hip_interval * intervals, int64_t n_view, void * stream) {
    if (d_signal == nullptr) {
        set_last_error("offset_scan_project_signal: d_signal is NULL");
        return TOAST_HIP_ERR_ARG;
    }
    return offset_scan_project_launch(step_length, amp_offsets, n_amp_views, nullptr, d_amplitudes_out,
                                      d_amplitude_flags, d_g2l, d_map, n_pix_submap, nnz, pixel_index, d_pixels,
                                      weight_index, d_weights, flag_index, d_flag_data, flag_mask, det_weights, n_det,
                                      n_samp, intervals, n_view, signal_index, d_signal, stream);
}

int toast_hip_template_offset_apply_diag_precond_dev(const double * d_offset_var,
                                                     const double * d_amp_in,
                                                     const uint8_t * d_amplitude_flags,
                                                     double * d_amp_out, int64_t n_amp,
                                                     void * stream) {
    return guarded([&] {
        if (n_amp <= 0) return;
        hipLaunchKernelGGL(k_offset_apply_diag_precond, flat_grid(n_amp), dim3(kThreads), 0,
                           as_stream(stream), n_amp, d_offset_var, d_amp_in, d_amplitude_flags,
                           d_amp_out);
        check_launch();
    });
}

// the noise-weighted signal that may ride along with the inverse covariance (toast_hip_build_cov_hits_signal_dev)
struct CovSignal {
    const int32_t * index;      // host: row of every detector in d_signal
    const double * d_signal;    // [rows, n_samp]
    const double * scale;       // host: per detector (build_noise_weighted's det_scale)
    double * d_zmap;            // [n_local_pix, 3]
};

// d_hits != nullptr (mode 1 only): the hit map is accumulated by the same kernel when the pair-merged kernel applies;
// *hits_done tells the caller whether it was.  sig != nullptr: the same for A^T N^-1 d into sig->d_zmap (two samples per
// lane, hits in the same launch), *signal_done.
static int build_cov_launch(
    int mode /*0 hits, 1 inverse covariance*/, const int64_t * d_g2l, void * d_out, int64_t n_pix_submap,
    int64_t nnz, const int32_t * pixel_index, const int64_t * d_pixels, const int32_t * weight_index,
    const double * d_weights, const int32_t * flag_index, const uint8_t * d_det_flags,
    int64_t n_flag_samp, const double * det_scale, uint8_t det_flag_mask, int64_t n_det, int64_t n_samp,
    const toast_hip_interval * intervals, int64_t n_view, const uint8_t * d_shared_flags,
    int64_t n_shared_flags, uint8_t shared_flag_mask, void * stream, int64_t * d_hits_in, bool * hits_done,
    const CovSignal * sig = nullptr, bool * signal_done = nullptr, const char * fn = __builtin_FUNCTION()) {
    if (hits_done != nullptr) *hits_done = false;
    if (signal_done != nullptr) *signal_done = false;
    return guarded([&] {
        int64_t * d_hits = d_hits_in;
        if (!(mode == 1 && nnz == 3 && pair_detectors() && n_det >= 2 && !deterministic_mode())) d_hits = nullptr;
        if (n_det <= 0) return;
        if (n_pix_submap <= 0) fail_arg("n_pix_submap must be positive");
        if (mode == 1 && (nnz < 1 || nnz > 3)) fail_arg("build_inverse_covariance: nnz must be 1..3");
        const auto chunks = make_chunks(intervals, n_view, n_samp);
        if (chunks.empty()) return;
        const int use_d = (n_flag_samp == n_samp) ? 1 : 0;
        const int use_s = (n_shared_flags == n_samp) ? 1 : 0;
        std::vector<int32_t> fidx(n_det, 0), widx(n_det, 0);
        std::vector<double> dscale(n_det, 1.0);
        if (use_d) std::memcpy(fidx.data(), flag_index, sizeof(int32_t) * n_det);
        if (mode == 1) {
            std::memcpy(widx.data(), weight_index, sizeof(int32_t) * n_det);
            std::memcpy(dscale.data(), det_scale, sizeof(double) * n_det);
        }
        if (mode == 1 && deterministic_mode()) {
            // (the hit map is integer: its atomic accumulation is order-independent already)
            deterministic_scatter(1, d_g2l, (double *)d_out, n_pix_submap, nnz, pixel_index, d_pixels, widx.data(),
                                  d_weights, nullptr, nullptr, fidx.data(), d_det_flags, use_d, dscale.data(),
                                  det_flag_mask, n_det, n_samp, intervals, n_view, d_shared_flags, use_s,
                                  shared_flag_mask, as_stream(stream));
            return;
        }
        ParamBlock pb;
        const size_t o_ch = pb.push_vec(chunks);
        const size_t o_pi = pb.push(pixel_index, sizeof(int32_t) * n_det);
        const size_t o_wi = pb.push_vec(widx);
        const size_t o_fi = pb.push_vec(fidx);
        const size_t o_ds = pb.push_vec(dscale);
        const size_t o_di = (sig != nullptr && mode == 1) ? pb.push(sig->index, sizeof(int32_t) * n_det) : 0;
        const size_t o_ss = (sig != nullptr && mode == 1) ? pb.push(sig->scale, sizeof(double) * n_det) : 0;
        const char * d = pb.commit(as_stream(stream));
        const FastDiv dv = make_fastdiv(n_pix_submap);
        const dim3 grid = chunk_grid(n_det, chunks.size());
        hipStream_t st = as_stream(stream);
#define TH_COV_ARGS                                                                              \
    (const Chunk *)(d + o_ch), (int)chunks.size(), (const int32_t *)(d + o_pi),                  \
        (const int32_t *)(d + o_wi), (const int32_t *)(d + o_fi), (const double *)(d + o_ds),    \
        d_g2l, (double *)d_out, (long long *)d_out, d_pixels, d_weights, d_det_flags,            \
        det_flag_mask, use_d, d_shared_flags, shared_flag_mask, use_s, dv, n_samp
        if (mode == 0) {
            hipLaunchKernelGGL((k_build_cov<1, 0>), grid, dim3(kThreads), 0, st, TH_COV_ARGS);
        } else if (nnz == 3 && pair_detectors() && n_det >= 2) {
            const dim3 gp((unsigned)((n_det + 1) / 2), grid.y, 1);
            // two samples per lane when every row starts on a 16-byte boundary (TOAST_HIP_VEC2=0: one sample per lane)
            const bool v2 = vec2_lanes() && (n_samp & 1) == 0 && rows_16b(d_pixels) && rows_16b(d_weights) &&
                            (!use_d || rows_16b(d_det_flags)) && (!use_s || rows_16b(d_shared_flags));
            if (v2 && d_hits != nullptr && sig != nullptr && rows_16b(sig->d_signal)) {
                const char * d2 = d;
                hipLaunchKernelGGL((k_build_cov_pair_v2<true, true>), gp, dim3(kThreads), 0, st,
                                   (const Chunk *)(d2 + o_ch), (int)chunks.size(), (int)n_det, (const int32_t *)(d2 + o_pi),
                                   (const int32_t *)(d2 + o_wi), (const int32_t *)(d2 + o_fi), (const double *)(d2 + o_ds),
                                   d_g2l, (double *)d_out, d_pixels, d_weights, d_det_flags, det_flag_mask, use_d,
                                   d_shared_flags, shared_flag_mask, use_s, dv, n_samp, (long long *)d_hits,
                                   (const int32_t *)(d2 + o_di), sig->d_signal, (const double *)(d2 + o_ss), sig->d_zmap);
                if (signal_done != nullptr) *signal_done = true;
            } else if (v2) {
                hipLaunchKernelGGL(d_hits != nullptr ? k_build_cov_pair_v2<true> : k_build_cov_pair_v2<false>, gp,
                                   dim3(kThreads), 0, st, (const Chunk *)(d + o_ch), (int)chunks.size(), (int)n_det,
                                   (const int32_t *)(d + o_pi), (const int32_t *)(d + o_wi), (const int32_t *)(d + o_fi),
                                   (const double *)(d + o_ds), d_g2l, (double *)d_out, d_pixels, d_weights, d_det_flags,
                                   det_flag_mask, use_d, d_shared_flags, shared_flag_mask, use_s, dv, n_samp,
                                   (long long *)d_hits, (const int32_t *)nullptr, (const double *)nullptr,
                                   (const double *)nullptr, (double *)nullptr);
            } else if (d_hits != nullptr) {
                hipLaunchKernelGGL((k_build_cov_pair<3, true>), gp, dim3(kThreads), 0, st, (const Chunk *)(d + o_ch),
                                   (int)chunks.size(), (int)n_det, (const int32_t *)(d + o_pi),
                                   (const int32_t *)(d + o_wi), (const int32_t *)(d + o_fi), (const double *)(d + o_ds),
                                   d_g2l, (double *)d_out, d_pixels, d_weights, d_det_flags, det_flag_mask, use_d,
                                   d_shared_flags, shared_flag_mask, use_s, dv, n_samp, (long long *)d_hits);
            } else {
                hipLaunchKernelGGL((k_build_cov_pair<3, false>), gp, dim3(kThreads), 0, st, (const Chunk *)(d + o_ch),
                                   (int)chunks.size(), (int)n_det, (const int32_t *)(d + o_pi),
                                   (const int32_t *)(d + o_wi), (const int32_t *)(d + o_fi), (const double *)(d + o_ds),
                                   d_g2l, (double *)d_out, d_pixels, d_weights, d_det_flags, det_flag_mask, use_d,
                                   d_shared_flags, shared_flag_mask, use_s, dv, n_samp, (long long *)nullptr);
            }
        } else if (nnz == 3) {
            hipLaunchKernelGGL((k_build_cov<3, 1>), grid, dim3(kThreads), 0, st, TH_COV_ARGS);
        } else if (nnz == 2) {
            hipLaunchKernelGGL((k_build_cov<2, 1>), grid, dim3(kThreads), 0, st, TH_COV_ARGS);
        } else {
            hipLaunchKernelGGL((k_build_cov<1, 1>), grid, dim3(kThreads), 0, st, TH_COV_ARGS);
        }
#undef TH_COV_ARGS
        check_launch();
        if (hits_done != nullptr) *hits_done = d_hits != nullptr;
    }, fn);
}

int toast_hip_build_cov_dev(
    int mode /*0 hits, 1 inverse covariance*/, const int64_t * d_g2l, void * d_out, int64_t n_pix_submap,
    int64_t nnz, const int32_t * pixel_index, const int64_t * d_pixels, const int32_t * weight_index,
    const double * d_weights, const int32_t * flag_index, const uint8_t * d_det_flags,
    int64_t n_flag_samp, const double * det_scale, uint8_t det_flag_mask, int64_t n_det, int64_t n_samp,
    const toast_hip_interval * intervals, int64_t n_view, const uint8_t * d_shared_flags,
    int64_t n_shared_flags, uint8_t shared_flag_mask, void * stream) {
    return build_cov_launch(mode, d_g2l, d_out, n_pix_submap, nnz, pixel_index, d_pixels, weight_index, d_weights,
                            flag_index, d_det_flags, n_flag_samp, det_scale, det_flag_mask, n_det, n_samp, intervals,
                            n_view, d_shared_flags, n_shared_flags, shared_flag_mask, stream, nullptr, nullptr);
}

int toast_hip_build_cov_hits_dev(
    const int64_t * d_g2l, double * d_invcov, int64_t * d_hits, int64_t n_pix_submap, int64_t nnz,
    const int32_t * pixel_index, const int64_t * d_pixels, const int32_t * weight_index, const double * d_weights,
    const int32_t * flag_index, const uint8_t * d_det_flags, int64_t n_flag_samp, const double * det_scale,
    uint8_t det_flag_mask, int64_t n_det, int64_t n_samp, const toast_hip_interval * intervals, int64_t n_view,
    const uint8_t * d_shared_flags, int64_t n_shared_flags, uint8_t shared_flag_mask, void * stream) {
    bool hits_done = false;
    int rc = build_cov_launch(1, d_g2l, d_invcov, n_pix_submap, nnz, pixel_index, d_pixels, weight_index, d_weights,
                              flag_index, d_det_flags, n_flag_samp, det_scale, det_flag_mask, n_det, n_samp, intervals,
                              n_view, d_shared_flags, n_shared_flags, shared_flag_mask, stream, d_hits, &hits_done);
    if (rc != TOAST_HIP_OK || hits_done || n_det <= 0) return rc;
    // no pair-merged kernel for this call (nnz != 3, a single detector, pairing off, deterministic mode): the hit
    // map in its own pass
    return build_cov_launch(0, d_g2l, d_hits, n_pix_submap, nnz, pixel_index, d_pixels, weight_index, d_weights,
                            flag_index, d_det_flags, n_flag_samp, det_scale, det_flag_mask, n_det, n_samp, intervals,
                            n_view, d_shared_flags, n_shared_flags, shared_flag_mask, stream, nullptr, nullptr);
}

int toast_hip_build_cov_hits_signal_dev(
    const int64_t * d_g2l, double * d_invcov, int64_t * d_hits, double * d_zmap, int64_t n_pix_submap, int64_t nnz,
    const int32_t * pixel_index, const int64_t * d_pixels, const int32_t * weight_index, const double * d_weights,
    const int32_t * data_index, const double * d_det_data, const int32_t * flag_index, const uint8_t * d_det_flags,
    int64_t n_flag_samp, const double * det_scale, const double * data_scale, uint8_t det_flag_mask, int64_t n_det,
    int64_t n_samp, const toast_hip_interval * intervals, int64_t n_view, const uint8_t * d_shared_flags,
    int64_t n_shared_flags, uint8_t shared_flag_mask, int * fused, void * stream) {
    if (fused != nullptr) *fused = 0;
    const CovSignal sig{data_index, d_det_data, data_scale, d_zmap};
    bool hits_done = false, signal_done = false;
    int rc = build_cov_launch(1, d_g2l, d_invcov, n_pix_submap, nnz, pixel_index, d_pixels, weight_index, d_weights,
                              flag_index, d_det_flags, n_flag_samp, det_scale, det_flag_mask, n_det, n_samp, intervals,
                              n_view, d_shared_flags, n_shared_flags, shared_flag_mask, stream, d_hits, &hits_done,
                              (d_zmap != nullptr && d_det_data != nullptr) ? &sig : nullptr, &signal_done);
    if (rc != TOAST_HIP_OK || n_det <= 0) return rc;
    if (!hits_done) {
        rc = build_cov_launch(0, d_g2l, d_hits, n_pix_submap, nnz, pixel_index, d_pixels, weight_index, d_weights,
                              flag_index, d_det_flags, n_flag_samp, det_scale, det_flag_mask, n_det, n_samp, intervals,
                              n_view, d_shared_flags, n_shared_flags, shared_flag_mask, stream, nullptr, nullptr);
        if (rc != TOAST_HIP_OK) return rc;
    }
    if (fused != nullptr) *fused = signal_done ? 1 : 0;
    if (signal_done || d_zmap == nullptr || d_det_data == nullptr) return rc;
    // no fused kernel for this call: the map of the signal in its own sweep
    return toast_hip_build_noise_weighted_dev(d_g2l, d_zmap, n_pix_submap, nnz, pixel_index, d_pixels, weight_index, d_weights,
                                              data_index, d_det_data, flag_index, d_det_flags, n_flag_samp, data_scale,
                                              det_flag_mask, n_det, n_samp, intervals, n_view, d_shared_flags,
                                              n_shared_flags, shared_flag_mask, stream);
}

int toast_hip_cov_eigendecompose_diag_dev(int64_t n_sub, int64_t subsize, int64_t nnz, double * d_data,
                                          double * d_cond, double threshold, int invert, void * stream) {
    return guarded([&] {
        const int64_t n_px = n_sub * subsize;
        if (n_px <= 0) return;
        const dim3 grid = flat_grid(n_px);
        hipStream_t st = as_stream(stream);
        switch (nnz) {
            case 1: hipLaunchKernelGGL(k_cov_invert<1>, grid, dim3(kThreads), 0, st, n_px, d_data, d_cond, threshold, invert); break;
            case 2: hipLaunchKernelGGL(k_cov_invert<2>, grid, dim3(kThreads), 0, st, n_px, d_data, d_cond, threshold, invert); break;
            case 3: hipLaunchKernelGGL(k_cov_invert<3>, grid, dim3(kThreads), 0, st, n_px, d_data, d_cond, threshold, invert); break;
            case 4: hipLaunchKernelGGL(k_cov_invert<4>, grid, dim3(kThreads), 0, st, n_px, d_data, d_cond, threshold, invert); break;
            default: fail_arg("cov_eigendecompose_diag: nnz must be 1..4");
        }
        check_launch();
    });
}

int toast_hip_scan_mask_dev(const int64_t * d_g2l, const uint8_t * d_mask, int64_t n_pix_submap,
                            uint8_t mask_bits, uint8_t flag_value, const int32_t * pixel_index,
                            const int64_t * d_pixels, const int32_t * flag_index, uint8_t * d_det_flags,
                            int64_t n_det, int64_t n_samp, const toast_hip_interval * intervals,
                            int64_t n_view, void * stream) {
    return guarded([&] {
        if (n_det <= 0) return;
        if (n_pix_submap <= 0) fail_arg("n_pix_submap must be positive");
        const auto chunks = make_chunks(intervals, n_view, n_samp);
        if (chunks.empty()) return;
        ParamBlock pb;
        const size_t o_ch = pb.push_vec(chunks);
        const size_t o_pi = pb.push(pixel_index, sizeof(int32_t) * n_det);
        const size_t o_fi = pb.push(flag_index, sizeof(int32_t) * n_det);
        const char * d = pb.commit(as_stream(stream));
        const bool v2 = vec2_lanes() && (n_samp & 1) == 0 && rows_16b(d_pixels) &&
                        (reinterpret_cast<uintptr_t>(d_det_flags) & 1) == 0;
        hipLaunchKernelGGL(v2 ? k_scan_mask_v2 : k_scan_mask, chunk_grid(n_det, chunks.size()), dim3(kThreads), 0,
                           as_stream(stream), (const Chunk *)(d + o_ch), (int)chunks.size(),
                           (const int32_t *)(d + o_pi), (const int32_t *)(d + o_fi), d_g2l, d_mask,
                           mask_bits, flag_value, d_pixels, d_det_flags, make_fastdiv(n_pix_submap), n_samp);
        check_launch();
    });
}

int toast_hip_offset_count_flagged_dev(int64_t step_length, const int64_t * amp_offsets,
                                       const int64_t * n_amp_views, double * d_counts, const int32_t * flag_index,
                                       const uint8_t * d_det_flags, uint8_t flag_mask, int64_t n_det, int64_t n_samp,
                                       const toast_hip_interval * intervals, int64_t n_view, void * stream) {
    return guarded([&] {
        if (n_det <= 0) return;
        if (step_length <= 0) fail_arg("step_length must be positive");
        const auto chunks = make_chunks(intervals, n_view, n_samp);
        if (chunks.empty()) return;
        const OffsetViews ov = offset_views(intervals, n_amp_views, n_view);
        ParamBlock pb;
        const size_t o_ch = pb.push_vec(chunks);
        const size_t o_vf = pb.push_vec(ov.first);
        const size_t o_va = pb.push_vec(ov.aoff);
        const size_t o_ao = pb.push(amp_offsets, sizeof(int64_t) * n_det);
        const size_t o_fi = pb.push(flag_index, sizeof(int32_t) * n_det);
        hipStream_t st = as_stream(stream);
        const char * d = pb.commit(st);
        // (counts are whole numbers added as doubles: the order of the additions does not matter)
        static const bool wide = [] {
            const char * e = std::getenv("TOAST_HIP_COUNT_FLAGGED16");
            return !(e != nullptr && e[0] == '0');
        }();
        auto kern = (wide && step_length >= 16) ? k_offset_count_flagged16 : k_offset_count_flagged;
        hipLaunchKernelGGL(kern, chunk_grid(n_det, chunks.size()), dim3(kThreads), 0, st,
                           (const Chunk *)(d + o_ch), (int)chunks.size(), (const int64_t *)(d + o_vf),
                           (const int64_t *)(d + o_va), make_fastdiv(step_length), (const int64_t *)(d + o_ao),
                           (const int32_t *)(d + o_fi), d_counts, d_det_flags, flag_mask, n_samp);
        check_launch();
    });
}

// mask[i] |= bit where value[i] < threshold: the pixels with a poor condition number that SolveAmplitudes turns into
// sample flags (src/toast/ops/mapmaker_templates.py:902-939: rcond_mask[rcond < threshold] = 1), without a host pass.
__global__ __launch_bounds__(kThreads) void k_threshold_mask(int64_t n, const double * __restrict__ value,
                                                              double threshold, uint8_t bit,
                                                              uint8_t * __restrict__ mask) {
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        if (value[i] < threshold) mask[i] |= bit;
    }
}

int toast_hip_threshold_mask_dev(int64_t n, const double * d_value, double threshold, uint8_t bit, uint8_t * d_mask,
                                 void * stream) {
    return guarded([&] {
        if (n <= 0) return;
        hipLaunchKernelGGL(k_threshold_mask, flat_grid(n), dim3(kThreads), 0, as_stream(stream), n, d_value, threshold,
                           bit, d_mask);
        check_launch();
    });
}

// Offset template set-up (reference src/toast/templates/offset.py:262-343): from the number of flagged samples under every
// baseline to its flag and its variance.  Detector d of the call owns amplitudes amp_offsets[d] + j, j < n_len, of
// lengths amp_len[j] samples; its noise weight is det_weight[d].  An amplitude is cut (flag 1, variance 0) when the
// fraction of good samples is <= good_fraction, the detector has no weight or the baseline is empty; else
// variance = 1 / (det_weight * n_good).
__global__ __launch_bounds__(kThreads) void k_offset_variance(int64_t n_det, int64_t n_len,
                                                               const int64_t * __restrict__ amp_offsets,
                                                               const double * __restrict__ det_weight,
                                                               const int64_t * __restrict__ amp_len,
                                                               const double * __restrict__ n_bad, double good_fraction,
                                                               uint8_t * __restrict__ flags, double * __restrict__ var) {
    const int64_t d = blockIdx.y;
    const int64_t base = amp_offsets[d];
    const double w = det_weight[d];
    for (int64_t j = (int64_t)blockIdx.x * kThreads + threadIdx.x; j < n_len; j += (int64_t)gridDim.x * kThreads) {
        const int64_t a = base + j;
        const int64_t len = amp_len[j];
        const double n_good = (double)len - rint(n_bad[a]);
        const bool cut = (n_good / (double)(len > 1 ? len : 1) <= good_fraction) || (w <= 0.0) || (len == 0);
        flags[a] = cut ? (uint8_t)1 : (uint8_t)0;
        var[a] = cut ? 0.0 : 1.0 / (w * n_good);
    }
}

int toast_hip_offset_variance_dev(int64_t n_det, int64_t n_len, const int64_t * amp_offsets, const double * det_weight,
                                  const int64_t * amp_len, const double * d_n_bad, double good_fraction,
                                  uint8_t * d_amp_flags, double * d_variance, void * stream) {
    return guarded([&] {
        if (n_det <= 0 || n_len <= 0) return;
        ParamBlock pb;
        const size_t o_ao = pb.push(amp_offsets, sizeof(int64_t) * n_det);
        const size_t o_dw = pb.push(det_weight, sizeof(double) * n_det);
        const size_t o_al = pb.push(amp_len, sizeof(int64_t) * n_len);
        hipStream_t st = as_stream(stream);
        const char * d = pb.commit(st);
        int64_t gx = (n_len + kThreads - 1) / kThreads;
        if (gx > 1024) gx = 1024;
        hipLaunchKernelGGL(k_offset_variance, dim3((unsigned)gx, (unsigned)n_det), dim3(kThreads), 0, st, n_det, n_len,
                           (const int64_t *)(d + o_ao), (const double *)(d + o_dw), (const int64_t *)(d + o_al), d_n_bad,
                           good_fraction, d_amp_flags, d_variance);
        check_launch();
    });
}

int toast_hip_combine_flags_dev(uint8_t * d_out, const int32_t * out_index, const uint8_t * d_det_flags,
                                int64_t n_flag_samp, const int32_t * flag_index, uint8_t det_flag_mask,
                                const uint8_t * d_shared_flags, int64_t n_shared_flags, uint8_t shared_flag_mask,
                                int64_t n_det, int64_t n_samp, const toast_hip_interval * intervals,
                                int64_t n_view, int64_t n_out_rows, int outside_value, void * stream) {
    return guarded([&] {
        if (n_det <= 0) return;
        hipStream_t st0 = as_stream(stream);
        if (outside_value >= 0 && n_out_rows > 0) {
            TH_HIP(hipMemsetAsync(d_out, outside_value & 0xff, (size_t)(n_out_rows * n_samp), st0));
        }
        const auto chunks = make_chunks(intervals, n_view, n_samp);
        if (chunks.empty()) return;
        const int use_d = (n_flag_samp == n_samp) ? 1 : 0;
        const int use_s = (n_shared_flags == n_samp) ? 1 : 0;
        std::vector<int32_t> fidx(n_det, 0);
        if (use_d) std::memcpy(fidx.data(), flag_index, sizeof(int32_t) * n_det);
        ParamBlock pb;
        const size_t o_ch = pb.push_vec(chunks);
        const size_t o_oi = pb.push(out_index, sizeof(int32_t) * n_det);
        const size_t o_fi = pb.push_vec(fidx);
        hipStream_t st = as_stream(stream);
        const char * d = pb.commit(st);
        hipLaunchKernelGGL(k_combine_flags, chunk_grid(n_det, chunks.size()), dim3(kThreads), 0, st,
                           (const Chunk *)(d + o_ch), (int)chunks.size(), (const int32_t *)(d + o_oi),
                           (const int32_t *)(d + o_fi), d_out, d_det_flags, det_flag_mask, use_d, d_shared_flags,
                           shared_flag_mask, use_s, n_samp);
        check_launch();
    });
}

}  // extern "C"

namespace {
// n_block blocks of block_bytes: dst block dst_index[i] <- src block src_index[i] (one workgroup column per block)
__global__ __launch_bounds__(kThreads) void k_block_move(char * __restrict__ dst, const char * __restrict__ src,
                                                         const int64_t * __restrict__ dst_index,
                                                         const int64_t * __restrict__ src_index, int64_t block_bytes) {
    const int64_t i = blockIdx.x;
    char * d = dst + dst_index[i] * block_bytes;
    const char * s = src + src_index[i] * block_bytes;
    if ((block_bytes & 15) == 0) {
        const int64_t n16 = block_bytes >> 4;
        for (int64_t k = (int64_t)blockIdx.y * kThreads + threadIdx.x; k < n16; k += (int64_t)gridDim.y * kThreads) {
            reinterpret_cast<uint4 *>(d)[k] = reinterpret_cast<const uint4 *>(s)[k];
        }
    } else {
        for (int64_t k = (int64_t)blockIdx.y * kThreads + threadIdx.x; k < block_bytes; k += (int64_t)gridDim.y * kThreads) d[k] = s[k];
    }
}
}  // namespace

extern "C" {

// Submaps from one layout into another: local submaps of a map into their places in the union of all ranks' submaps and
// back (PixelData.sync_alltoallv on the device for ranks that hold DIFFERENT local submaps, toast_amd/pixels.py).
int toast_hip_block_move_dev(void * d_dst, const void * d_src, int64_t n_block, int64_t block_bytes,
                             const int64_t * dst_index, const int64_t * src_index, void * stream) {
    return guarded([&] {
        if (n_block <= 0 || block_bytes <= 0) return;
        if (((block_bytes & 15) == 0) && ((reinterpret_cast<uintptr_t>(d_dst) | reinterpret_cast<uintptr_t>(d_src)) & 15) != 0) {
            fail_arg("block_move: blocks of a multiple of 16 bytes need 16-byte aligned buffers");
        }
        hipStream_t st = as_stream(stream);
        ParamBlock pb;
        const size_t o_d = pb.push(dst_index, sizeof(int64_t) * (size_t)n_block);
        const size_t o_s = pb.push(src_index, sizeof(int64_t) * (size_t)n_block);
        const char * d = pb.commit(st);
        int64_t per = (block_bytes / 16 + kThreads - 1) / kThreads;
        if (per < 1) per = 1;
        if (per > 64) per = 64;
        hipLaunchKernelGGL(k_block_move, dim3((unsigned)n_block, (unsigned)per), dim3(kThreads), 0, st,
                           static_cast<char *>(d_dst), static_cast<const char *>(d_src), (const int64_t *)(d + o_d),
                           (const int64_t *)(d + o_s), block_bytes);
        check_launch();
    });
}

int toast_hip_copy_dev(void * d_dst, const void * d_src, size_t nbytes, void * stream) {
    return guarded([&] {
        if (nbytes == 0) return;
        TH_HIP(hipMemcpyAsync(d_dst, d_src, nbytes, hipMemcpyDeviceToDevice, as_stream(stream)));
    });
}

int toast_hip_memset_dev(void * d_dst, int value, size_t nbytes, void * stream) {
    return guarded([&] {
        if (nbytes == 0) return;
        TH_HIP(hipMemsetAsync(d_dst, value & 0xff, nbytes, as_stream(stream)));
    });
}

int toast_hip_vec_axpby_dev(int64_t n, double a, const double * d_x, double b, double * d_y, void * stream) {
    return guarded([&] {
        if (n <= 0) return;
        hipLaunchKernelGGL(k_vec_axpby, flat_grid(n), dim3(kThreads), 0, as_stream(stream), n, a, d_x, b, d_y);
        check_launch();
    });
}

int toast_hip_vec_dot_dev(int64_t n, const double * d_x, const double * d_y, const uint8_t * d_flags_x,
                          const uint8_t * d_flags_y, double * result, void * stream) {
    return guarded([&] {
        double * d_res = (double *)Manager::get().scratch(Manager::kScratchDot, sizeof(double) * 1032);
        hipStream_t st = as_stream(stream);
        TH_HIP(hipMemsetAsync(d_res, 0, sizeof(double), st));
        if (n > 0) {
            dim3 grid = flat_grid(n);
            if (grid.x > 1024) grid.x = 1024;
            double * d_part = deterministic_mode() ? d_res + 8 : nullptr;
            hipLaunchKernelGGL(k_vec_dot, grid, dim3(kThreads), 0, st, n, d_x, d_y, d_flags_x, d_flags_y, d_res, d_part);
            if (d_part != nullptr) {
                hipLaunchKernelGGL(k_vec_dot_final, dim3(1), dim3(kThreads), 0, st, (int)grid.x, d_part, d_res);
            }
            check_launch();
        }
        copy_to_host(result, d_res, sizeof(double), st);
    });
}

int toast_hip_test_math_dev(int op, int64_t n, const double * d_a, const double * d_b, double * d_out,
                            void * stream) {
    return guarded([&] {
        if (n <= 0) return;
        hipLaunchKernelGGL(k_test_math, flat_grid(n), dim3(kThreads), 0, as_stream(stream), op, n, d_a,
                           d_b, d_out);
        check_launch();
    });
}

}  // extern "C"

// ------------------------------------------------------------------------------------
namespace {

// Placement probe of the memory manager (TOAST_HIP_ALLOC=probe, runtime.cpp): one read + write pass over a new block
// with the access pattern of the timestream kernels (1024 rows in flight, 1024-sample chunks per workgroup); returns
// the time in ms.  The block holds nothing yet.
__global__ __launch_bounds__(kThreads) void k_probe_stream(double * __restrict__ p, int64_t row_len, double one) {
    double * row = p + (int64_t)blockIdx.x * row_len;
    for (int64_t c0 = (int64_t)blockIdx.y * 1024; c0 < row_len; c0 += (int64_t)gridDim.y * 1024) {
        for (int i = threadIdx.x; i < 1024 && c0 + i < row_len; i += kThreads) row[c0 + i] = row[c0 + i] * one;
    }
}

// The same pass over rows that are dealt round-robin to up to four separate ranges (row r lives in range r % nb):
// placement experiments -- does it matter where in HBM the rows that are in flight together sit?
struct ProbeBases {
    double * p[4];
};
__global__ __launch_bounds__(kThreads) void k_probe_stream_split(ProbeBases b, int nb, int64_t row_len, double one,
                                                                 unsigned long long * __restrict__ clk) {
    // clk[0] / clk[1]: earliest start and latest end of any workgroup in ticks of the device's constant-rate clock
    // (hipDeviceAttributeWallClockRate, 100 MHz): the pass is timed where it runs -- host-side event handling, a
    // profiler's interception of the dispatch or a busy host thread do not enter the rate
    // Only the workgroups that are dispatched FIRST (blockIdx.y == 0) and LAST (blockIdx.y == gridDim.y - 1; x is the fast
    // index of the dispatch order) touch the clock words: 2048 atomics instead of one pair per workgroup -- 2 x 262 144
    // atomics on ONE L2 line serialise into 5.2 ms of a 0.75 ms pass, every chunk then "runs" at 0.72 TB/s and no two
    // can be told apart (the first form of this kernel, profiles/r06_a section 1).
    const bool first = clk != nullptr && blockIdx.y == 0 && threadIdx.x == 0;
    const bool last = clk != nullptr && blockIdx.y == gridDim.y - 1;
    if (first) atomicMin(clk, (unsigned long long)wall_clock64());
    const int r = blockIdx.x;
    double * row = b.p[r % nb] + (int64_t)(r / nb) * row_len;
    for (int64_t c0 = (int64_t)blockIdx.y * 1024; c0 < row_len; c0 += (int64_t)gridDim.y * 1024) {
        for (int i = threadIdx.x; i < 1024 && c0 + i < row_len; i += kThreads) row[c0 + i] = row[c0 + i] * one;
    }
    if (last) {
        __builtin_amdgcn_s_waitcnt(0);       // the stores of this lane have been acknowledged
        __syncthreads();
        if (threadIdx.x == 0) atomicMax(clk + 1, (unsigned long long)wall_clock64());
    }
}

// The byte mix of scan_map / build_noise_weighted WITHOUT their gathers, scans and atomics, in the launch shape of the
// *_v2 kernels (detector = blockIdx.x, 1024-sample chunks along y, two samples per lane): 16 B of pixels, 48 B of weights and
// 16 B of a timestream read per lane and trip; 16 B written when `out` is given.  What the arrays' places in HBM let a
// kernel of this byte mix reach (bench.py: roofline.stream_ceiling).
__global__ __launch_bounds__(kThreads) void k_probe_byte_mix(const int64_t * __restrict__ pixels,
                                                             const double * __restrict__ weights,
                                                             const double * __restrict__ tod, double * __restrict__ out,
                                                             double * __restrict__ sink, int64_t n_samp) {
    const int64_t row = (int64_t)blockIdx.x * n_samp;
    double acc = 0.0;
    for (int64_t c0 = (int64_t)blockIdx.y * 1024; c0 < n_samp; c0 += (int64_t)gridDim.y * 1024) {
        for (int64_t s = c0 + 2 * (int64_t)threadIdx.x; s < c0 + 1024 && s + 1 < n_samp; s += 2 * kThreads) {
            const longlong2 pp = *reinterpret_cast<const longlong2 *>(pixels + row + s);
            const double2 tt = *reinterpret_cast<const double2 *>(tod + row + s);
            const double2 * wv = reinterpret_cast<const double2 *>(weights + 3 * (row + s));
            const double2 w0 = wv[0], w1 = wv[1], w2 = wv[2];
            const double a = tt.x * (w0.x + w0.y + w1.x) + (double)(pp.x & 1);
            const double b = tt.y * (w1.y + w2.x + w2.y) + (double)(pp.y & 1);
            if (out != nullptr) *reinterpret_cast<double2 *>(out + row + s) = make_double2(a, b);
            else acc += a + b;
        }
    }
    if (out == nullptr && acc == 1.2345e301) *sink = acc;      // (never: keeps the loads alive)
}

__global__ void k_probe_clock_reset(unsigned long long * clk) {
    clk[0] = ~0ull;
    clk[1] = 0ull;
}

}  // namespace

namespace toast_hip {
// One read + write pass over up to four ranges, rows dealt round-robin; the best of three passes in ms.  Timed by the
// device's constant-rate clock inside the kernel (first workgroup in, last workgroup out); HIP events only when the
// clock's rate cannot be asked for.  *used_clock (optional): which of the two it was.
double probe_stream_split_ms(void * const * bases, int nb, size_t bytes_each, hipStream_t st, bool * used_clock) {
    if (nb < 1 || nb > 4) fail_arg("probe_stream_split: 1 .. 4 ranges");
    const int64_t rows = 1024;
    const int64_t rows_each = rows / nb;
    const int64_t row_len = (int64_t)(bytes_each / sizeof(double)) / rows_each;
    if (row_len < 1024) return 0.0;
    ProbeBases b;
    for (int k = 0; k < 4; ++k) b.p[k] = static_cast<double *>(bases[k < nb ? k : 0]);
    int64_t gy = (row_len + 1023) / 1024;
    if (gy > 65535) gy = 65535;
    // the clock words: 16 bytes of device memory per process and device, never freed (probes run at set-up, from the
    // slab builder's thread and from the caller's: one at a time)
    static std::mutex mu;
    static std::map<int, unsigned long long *> words;
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0, khz = 0;
    TH_HIP(hipGetDevice(&dev));
    unsigned long long * clk = nullptr;
    static const bool events_only = [] {
        const char * e = std::getenv("TOAST_HIP_PROBE_CLOCK");
        return e != nullptr && e[0] == '0';
    }();
    if (!events_only && hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) == hipSuccess && khz > 0) {
        auto it = words.find(dev);
        if (it == words.end()) {
            void * w = nullptr;
            if (hipMalloc(&w, 2 * sizeof(unsigned long long)) == hipSuccess) {
                it = words.emplace(dev, static_cast<unsigned long long *>(w)).first;
            } else {
                (void)hipGetLastError();
            }
        }
        if (it != words.end()) clk = it->second;
    } else {
        (void)hipGetLastError();
    }
    if (used_clock != nullptr) *used_clock = clk != nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (clk == nullptr) {
        TH_HIP(hipEventCreate(&e0));
        TH_HIP(hipEventCreate(&e1));
    }
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        if (clk != nullptr) {
            hipLaunchKernelGGL(k_probe_clock_reset, dim3(1), dim3(1), 0, st, clk);
        } else {
            TH_HIP(hipEventRecord(e0, st));
        }
        hipLaunchKernelGGL(k_probe_stream_split, dim3((unsigned)(rows_each * nb), (unsigned)gy), dim3(kThreads), 0, st, b, nb,
                           row_len, 1.0, clk);
        double ms = 0.0;
        if (clk != nullptr) {
            unsigned long long t[2] = {0, 0};
            TH_HIP(hipMemcpyAsync(t, clk, sizeof t, hipMemcpyDeviceToHost, st));
            TH_HIP(hipStreamSynchronize(st));
            ms = t[1] > t[0] ? (double)(t[1] - t[0]) / (double)khz : 0.0;
        } else {
            TH_HIP(hipEventRecord(e1, st));
            TH_HIP(hipEventSynchronize(e1));
            float f = 0.0f;
            TH_HIP(hipEventElapsedTime(&f, e0, e1));
            ms = (double)f;
        }
        if (ms > 0.0 && ms < best) best = ms;
    }
    if (e0 != nullptr) (void)hipEventDestroy(e0);
    if (e1 != nullptr) (void)hipEventDestroy(e1);
    return best < 1e29 ? best : 0.0;
}

void probe_byte_mix(const int64_t * pixels, const double * weights, const double * tod, double * out, int64_t n_det,
                    int64_t n_samp, hipStream_t st) {
    if (n_det <= 0 || n_samp < 2) return;
    if ((n_samp & 1) != 0) fail_arg("probe_byte_mix: even row length");
    int64_t gy = (n_samp + 1023) / 1024;
    if (gy > 65535) gy = 65535;
    static double * sink = nullptr;
    if (sink == nullptr) TH_HIP(hipMalloc(reinterpret_cast<void **>(&sink), sizeof(double)));
    hipLaunchKernelGGL(k_probe_byte_mix, dim3((unsigned)n_det, (unsigned)gy), dim3(kThreads), 0, st, pixels, weights, tod, out,
                       sink, n_samp);
    check_launch();
}

double probe_stream_ms(void * block, size_t bytes, hipStream_t st) {
    const int64_t rows = 1024;
    const int64_t row_len = (int64_t)(bytes / sizeof(double)) / rows;
    if (row_len < 1024) return 0.0;
    hipEvent_t e0, e1;
    TH_HIP(hipEventCreate(&e0));
    TH_HIP(hipEventCreate(&e1));
    int64_t gy = (row_len + 1023) / 1024;
    if (gy > 65535) gy = 65535;
    float best = 1e30f;
    for (int rep = 0; rep < 2; ++rep) {   // the first pass also pays the first touch of the block
        TH_HIP(hipEventRecord(e0, st));
        hipLaunchKernelGGL(k_probe_stream, dim3((unsigned)rows, (unsigned)gy), dim3(kThreads), 0, st,
                           static_cast<double *>(block), row_len, 1.0);
        TH_HIP(hipEventRecord(e1, st));
        TH_HIP(hipEventSynchronize(e1));
        float ms = 0.0f;
        TH_HIP(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return (double)best;
}
}  // namespace toast_hip
