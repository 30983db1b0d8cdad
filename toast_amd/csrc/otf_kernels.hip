// otf_kernels.hip -- "pointing on the fly" variants of the map-domain kernels.
//
// The reference's memory-saving mode (BinMap.full_pointing = False, src/toast/ops/
// mapmaker_binning.py:265-271, and SolverLHS, src/toast/ops/mapmaker_solve.py:476-489) runs
// [PointingDetectorSimple, PixelsHealpix, StokesWeights] again inside every pass over the
// data, one detector at a time, through 64 B/det-sample of scratch.  Here the same three steps
// run in registers inside the accumulate / scan kernels:
//
//     boresight (32 B per *time* sample, shared by all detectors -> L2)  ->  detector quaternion
//     [ops_pointing_detector.cpp:33-68]  ->  pixel [ops_pixels_healpix.cpp:586-666]  +  Stokes
//     weights [ops_stokes_weights.cpp:77-140]  ->  zmap scatter / map gather
//
// so a det-sample costs 9 B (A^T: signal + flag) or 16 B (A: signal read + write) of HBM traffic
// instead of 41 / 48 B, and 0-2 B in the offset-template forms, at the price of ~300 fp64
// operations.  The per-sample arithmetic is the device functions of hpix_math.hpp that the
// stand-alone kernels use, so pixels are bit-identical and weights identical to the cached path.
#include "kernel_common.hpp"

namespace {

// Device view of toast_hip_otf_pointing.
struct OtfDev {
    const double * bore;      // [n_samp, 4]
    const uint8_t * pflags;   // shared flags seen by the pointing operators
    const double * hwp;       // [n_samp] (MODE 2 only)
    const double * hwp_tab;   // optional [n_samp, 2] = (cos 4 hwp, sin 4 hwp), see toast_hip_hwp_table_dev
    const double * fp;        // [n_det, 4]
    const double * eps;       // [n_det]
    const double * gamma;     // [n_det]
    const double * cal;       // [n_det]
    const int64_t * g2l;      // global submap -> local submap
    const int32_t * cpix;     // PIX 1: cached local pixel index [rows, n_samp] (int32, -1 = none)
    const int32_t * cpix_idx; // PIX 1: row of each detector in cpix
    double usign;
    int64_t nside;
    FastDiv nps_div;
    int factor;
    int use_pflags;
    uint8_t pmask;
    int ref_nan;              // reproduce the reference's NaN Q / U weights at the poles
};

struct DetConst {
    double f[4];
    double eta, cd, gd;
    double c4g, s4g;        // cos / sin of 4 gamma (HWP modulation, see hwp_rotation)
    const int32_t * crow;   // PIX 1: this detector's row of cached local pixel indices
};

__device__ __forceinline__ DetConst det_const(const OtfDev & P, int det) {
    DetConst D;
#pragma unroll
    for (int k = 0; k < 4; ++k) D.f[k] = P.fp[4 * det + k];
    const double eps = P.eps[det];
    D.eta = (1.0 - eps) / (1.0 + eps);
    D.cd = P.cal[det];
    D.gd = P.gamma[det];
    sincos(4.0 * D.gd, &D.s4g, &D.c4g);
    D.crow = nullptr;
    return D;
}

// MODE 0: intensity only (nnz 1); 1: IQU without HWP; 2: IQU with HWP.
template <int MODE>
struct ModeNnz {
    static constexpr int value = (MODE == 0) ? 1 : 3;
};

// cos / sin of four times the HWP angle of time sample s: from the per-observation table when
// the caller provides one (identical values: the table is filled with the same sincos), else
// evaluated here.
template <int MODE>
__device__ __forceinline__ void hwp_cs4(const OtfDev & P, int64_t s, double & c4h, double & s4h) {
    c4h = 1.0;
    s4h = 0.0;
    if constexpr (MODE == 2) {
        if (P.hwp_tab != nullptr) {
            const double2 t = *reinterpret_cast<const double2 *>(P.hwp_tab + 2 * s);
            c4h = t.x;
            s4h = t.y;
        } else {
            sincos(4.0 * P.hwp[s], &s4h, &c4h);
        }
    }
}

// stokes_weights of one det-sample from its detector quaternion (computed for flagged samples too, like the
// stand-alone kernel)
template <int MODE>
__device__ __forceinline__ void otf_weights(const OtfDev & P, const DetConst & D, const double * r, double c4h,
                                            double s4h, double (&w)[ModeNnz<MODE>::value]) {
    if constexpr (MODE == 0) {
        w[0] = D.cd;
    } else {
        double c2a, s2a;
        stokes_cs2alpha(r, c2a, s2a, P.ref_nan != 0);
        if constexpr (MODE == 2) {
            double sb, cb;
            hwp_rotation(D.c4g, D.s4g, c4h, s4h, cb, sb);
            const double cang = cb * c2a + sb * s2a;
            const double sang = sb * c2a - cb * s2a;
            w[0] = D.cd;
            w[1] = cang * D.eta * D.cd;
            w[2] = -sang * D.eta * D.cd * P.usign;
        } else {
            w[0] = D.cd;
            w[1] = c2a * D.eta * D.cd;
            w[2] = s2a * D.eta * D.cd * P.usign;
        }
    }
}

// LOCAL map index (n_pix_submap * local_submap + pixel-in-submap; -1 when the boresight sample is
// flagged) and Stokes weights of one det-sample.
//   PIX 0: the pixel is computed here (pointing_detector -> pixels_healpix -> global2local);
//   PIX 1: the local index is read from a compact int32 cache (4 B instead of the 8 B global
//          pixel + global2local lookup) and only the weights are evaluated on the fly;
//   PIX 2: like 0 but the GLOBAL pixel number is returned (quaternion-free pixels_healpix);
//   PIX 3: no pixel at all, weights only (quaternion-free stokes_weights).
// c4h / s4h: cos / sin of four times the HWP angle of this time sample (MODE 2), evaluated once
// per sample by the caller and shared by the detectors of a workgroup.
template <bool NEST, int MODE, int PIX>
__device__ __forceinline__ int64_t otf_point(const OtfDev & P, const DetConst & D, int64_t s,
                                             const double * s_tab, double c4h, double s4h,
                                             double (&w)[ModeNnz<MODE>::value]) {
    int64_t lidx = -1;
    if constexpr (PIX == 1) lidx = D.crow[s];
    if constexpr (PIX == 1 && MODE == 0) {
        w[0] = D.cd;
        return lidx;
    }
    const Quat b = load_quat(P.bore + 4 * s);
    const uint8_t fl = P.use_pflags ? P.pflags[s] : (uint8_t)0;
    const bool flagged = (fl & P.pmask) != 0;
    // pointing_detector: a flagged boresight sample is replaced by the identity rotation
    double p[4] = {0.0, 0.0, 0.0, 1.0};
    if (!flagged) {
        p[0] = b.x; p[1] = b.y; p[2] = b.z; p[3] = b.w;
    }
    double r[4];
    quat_mult(p, D.f, r);
    otf_weights<MODE>(P, D, r, c4h, s4h, w);
    if constexpr (PIX == 1) {
        return lidx;
    } else if constexpr (PIX == 3) {
        return 0;
    } else {
        // pixels_healpix
        if (flagged) return -1;
        double dir[3];
        quat_rotate_z(r, dir);
        const int64_t pix = vec_to_pixel<NEST>(dir, P.nside, P.factor, s_tab);
        if constexpr (PIX == 2) return pix;
        const int64_t gsm = fastdiv(pix, P.nps_div);
        return P.g2l[gsm] * P.nps_div.d + (pix - gsm * P.nps_div.d);
    }
}

// The same for the two detectors of one workgroup (E = 2: detectors 2b and 2b + 1 of the call, the two orthogonally
// polarised detectors of a focalplane pixel in the usual ordering).  The boresight sample is loaded once; when the
// pixel is computed here (PIX 0 / 2) both directions go through vec_to_pixel_pair, which evaluates the pixel
// arithmetic ONCE when the two lines of sight agree to rounding (bit-identical to two separate evaluations by
// construction and on 3e9 host-checked pairs, tests/devmath_host.cpp) -- the pixel is ~60 % of the per-sample
// instructions of the on-the-fly kernels.  Unrelated detectors just take the separate path.
template <bool NEST, int MODE, int PIX>
__device__ __forceinline__ void otf_point_pair(const OtfDev & P, const DetConst & D0, const DetConst & D1, int64_t s,
                                               const double * s_tab, double c4h, double s4h,
                                               double (&w0)[ModeNnz<MODE>::value], double (&w1)[ModeNnz<MODE>::value],
                                               int64_t & idx0, int64_t & idx1) {
    if constexpr (PIX == 1 || PIX == 3) {
        idx0 = otf_point<NEST, MODE, PIX>(P, D0, s, s_tab, c4h, s4h, w0);
        idx1 = otf_point<NEST, MODE, PIX>(P, D1, s, s_tab, c4h, s4h, w1);
    } else {
        const Quat b = load_quat(P.bore + 4 * s);
        const uint8_t fl = P.use_pflags ? P.pflags[s] : (uint8_t)0;
        const bool flagged = (fl & P.pmask) != 0;
        double p[4] = {0.0, 0.0, 0.0, 1.0};
        if (!flagged) {
            p[0] = b.x; p[1] = b.y; p[2] = b.z; p[3] = b.w;
        }
        double r0[4], r1[4];
        quat_mult(p, D0.f, r0);
        quat_mult(p, D1.f, r1);
        otf_weights<MODE>(P, D0, r0, c4h, s4h, w0);
        otf_weights<MODE>(P, D1, r1, c4h, s4h, w1);
        idx0 = idx1 = -1;
        if (flagged) return;
        double dir0[3], dir1[3];
        quat_rotate_z(r0, dir0);
        quat_rotate_z(r1, dir1);
        int64_t pix0, pix1;
        vec_to_pixel_pair<NEST>(dir0, dir1, P.nside, P.factor, s_tab, pix0, pix1);
        if constexpr (PIX == 2) {
            idx0 = pix0;
            idx1 = pix1;
        } else {
            const int64_t gsm0 = fastdiv(pix0, P.nps_div);
            idx0 = P.g2l[gsm0] * P.nps_div.d + (pix0 - gsm0 * P.nps_div.d);
            idx1 = idx0;
            if (pix1 != pix0) {
                const int64_t gsm1 = fastdiv(pix1, P.nps_div);
                idx1 = P.g2l[gsm1] * P.nps_div.d + (pix1 - gsm1 * P.nps_div.d);
            }
        }
    }
}

// Offset-template addressing (template_offset.cpp:57-63, :93-120)
struct OffsetDev {
    const int64_t * view_first;
    const int64_t * view_aoff;
    const int64_t * amp_offsets;
    const double * amps_in;
    double * amps_out;
    const uint8_t * amp_flags;
    FastDiv step_div;
};

// ------------------------------------------------------------------------------------
// A^T:  zmap += P^T N^-1 d      SIG 0: d = timestream buffer (build_noise_weighted)
//                               SIG 1: d = M a, offset amplitudes (k_offset_accumulate)
//                               SIG 2: d = timestream - M a (ApplyAmplitudes + the final BinMap in one pass, round 6)
// ------------------------------------------------------------------------------------
template <bool NEST, int MODE, int SIG, int PIX, int E>
__global__ __launch_bounds__(kThreads) void k_otf_accumulate(
    const Chunk * __restrict__ chunks, int n_chunks, int n_det, OtfDev P, OffsetDev O,
    const int32_t * __restrict__ d_idx, const double * __restrict__ tod,
    const int32_t * __restrict__ f_idx, const uint8_t * __restrict__ dflags, uint8_t dmask,
    int use_dflags, const uint8_t * __restrict__ sflags, uint8_t smask, int use_sflags,
    const double * __restrict__ det_scale, double * __restrict__ zmap, int64_t n_samp) {
    constexpr int NNZ = ModeNnz<MODE>::value;
    __shared__ double s_tab[2 * TOAST_ATAN_TABLE_N];
    if (threadIdx.x < 2 * TOAST_ATAN_TABLE_N) s_tab[threadIdx.x] = kAtanTab[threadIdx.x];
    __syncthreads();

    // E detectors per workgroup (pair merging, see k_build_noise_weighted_pair)
    DetConst D[E];
    const double * drow[E];
    const uint8_t * frow[E];
    double ds[E];
    int64_t amp_offset[E];
    bool valid[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        int det = E * blockIdx.x + e;
        valid[e] = det < n_det;
        if (!valid[e]) det = E * blockIdx.x;
        D[e] = det_const(P, det);
        if (PIX == 1) D[e].crow = P.cpix + (int64_t)P.cpix_idx[det] * n_samp;
        drow[e] = (SIG != 1) ? tod + (int64_t)d_idx[det] * n_samp : nullptr;
        frow[e] = use_dflags ? dflags + (int64_t)f_idx[det] * n_samp : nullptr;
        ds[e] = det_scale[det];
        amp_offset[e] = (SIG != 0) ? O.amp_offsets[det] : 0;
    }
    for (int ci = blockIdx.y; ci < n_chunks; ci += gridDim.y) {
        const Chunk c = chunks[ci];
        int64_t vfirst = 0, vaoff = 0;
        if (SIG != 0) {
            vfirst = O.view_first[c.view];
            vaoff = O.view_aoff[c.view];
        }
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
                int64_t astep = 0;
                if (SIG != 0) astep = fastdiv(s - vfirst, O.step_div);
                double c4h, s4h;
                hwp_cs4<MODE>(P, s, c4h, s4h);
                uint8_t fd[E];
                double t[E];
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    fd[e] = use_dflags ? frow[e][s] : (uint8_t)0;
                    if (SIG == 0) {
                        t[e] = drow[e][s];
                    } else {
                        const int64_t a = amp_offset[e] + vaoff + astep;
                        const uint8_t af = O.amp_flags[a];
                        const double av = O.amps_in[a];
                        t[e] = (af == 0) ? (0.0 + av) : 0.0;
                        if (SIG == 2) t[e] = drow[e][s] - t[e];      // d - (0 + a): the template-cleaned sample
                    }
                }
                double wk[E][NNZ];
                int64_t pidx[E];
                if constexpr (E == 2) {
                    otf_point_pair<NEST, MODE, PIX>(P, D[0], D[1], s, s_tab, c4h, s4h, wk[0], wk[1], pidx[0], pidx[1]);
                } else {
                    pidx[0] = otf_point<NEST, MODE, PIX>(P, D[0], s, s_tab, c4h, s4h, wk[0]);
                }
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const bool good = (pidx[e] >= 0) & ((fd[e] & dmask) == 0) & ((fs & smask) == 0) & valid[e];
                    if (good) {
                        key[e] = pidx[e];
                        const double sd = t[e] * ds[e];
#pragma unroll
                        for (int k = 0; k < NNZ; ++k) v[e][k] = sd * wk[e][k];
                    }
                }
            }
            scatter_runs<NNZ, E>(key, v, zmap);
        }
    }
}

// ------------------------------------------------------------------------------------
// A:  SIG 0:  d = (zero ? 0 : d) -/+ scale * P m ; d *= det_w     (scan_map [+ noise_weight])
//     SIG 1:  a_out += M^T N^-1 (M a - P m)                        (k_offset_scan_project)
//     SIG 2:  a_out += M^T N^-1 (d - P m), d only read             (the tail of SolverRHS in one pass)
// ------------------------------------------------------------------------------------
template <bool NEST, int MODE, int SIG, int PIX, int E>
__global__ __launch_bounds__(kThreads) void k_otf_scan(
    const Chunk * __restrict__ chunks, int n_chunks, int n_det, OtfDev P, OffsetDev O,
    const int32_t * __restrict__ d_idx, double * __restrict__ tod, double scale, int zero,
    int subtract, const int32_t * __restrict__ f_idx, const uint8_t * __restrict__ flags,
    uint8_t fmask, int use_flags, const double * __restrict__ det_w,
    const double * __restrict__ map, int64_t n_samp) {
    constexpr int NNZ = ModeNnz<MODE>::value;
    __shared__ double s_tab[2 * TOAST_ATAN_TABLE_N];
    if (threadIdx.x < 2 * TOAST_ATAN_TABLE_N) s_tab[threadIdx.x] = kAtanTab[threadIdx.x];
    __syncthreads();

    // E detectors per workgroup: with E = 2 the pixel of a co-pointing detector pair is evaluated once (otf_point_pair)
    const bool fuse = det_w != nullptr;
    DetConst D[E];
    double * drow[E];
    const uint8_t * frow[E];
    double dw[E];
    int64_t amp_offset[E];
    bool valid[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        int det = E * blockIdx.x + e;
        valid[e] = det < n_det;
        if (!valid[e]) det = E * blockIdx.x;
        D[e] = det_const(P, det);
        if (PIX == 1) D[e].crow = P.cpix + (int64_t)P.cpix_idx[det] * n_samp;
        drow[e] = (SIG != 1) ? tod + (int64_t)d_idx[det] * n_samp : nullptr;   // SIG 2: read only
        frow[e] = (SIG != 0 && use_flags) ? flags + (int64_t)f_idx[det] * n_samp : nullptr;
        dw[e] = fuse ? det_w[det] : 1.0;
        amp_offset[e] = (SIG != 0) ? O.amp_offsets[det] : 0;
    }
    for (int ci = blockIdx.y; ci < n_chunks; ci += gridDim.y) {
        const Chunk c = chunks[ci];
        int64_t vfirst = 0, vaoff = 0;
        if (SIG != 0) {
            vfirst = O.view_first[c.view];
            vaoff = O.view_aoff[c.view];
        }
        for (int base = 0; base < c.count; base += kThreads) {
            const int i = base + threadIdx.x;
            const bool active = i < c.count;
            const int64_t s = c.first + (active ? i : 0);
            if (SIG == 0) {
                if (!active) continue;
                double d[E];
#pragma unroll
                for (int e = 0; e < E; ++e) d[e] = zero ? 0.0 : drow[e][s];
                double c4h, s4h;
                hwp_cs4<MODE>(P, s, c4h, s4h);
                double wk[E][NNZ];
                int64_t pidx[E];
                if constexpr (E == 2) {
                    otf_point_pair<NEST, MODE, PIX>(P, D[0], D[1], s, s_tab, c4h, s4h, wk[0], wk[1], pidx[0], pidx[1]);
                } else {
                    pidx[0] = otf_point<NEST, MODE, PIX>(P, D[0], s, s_tab, c4h, s4h, wk[0]);
                }
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    if (pidx[e] >= 0) {
                        const double * m = map + NNZ * pidx[e];
                        double v = 0.0;
#pragma unroll
                        for (int k = 0; k < NNZ; ++k) v += wk[e][k] * m[k];
                        v *= scale;
                        if (subtract) {
                            d[e] -= v;
                        } else {
                            d[e] += v;
                        }
                    }
                    if (fuse) d[e] *= dw[e];
                    if (valid[e]) drow[e][s] = d[e];
                }
            } else {
                int64_t key[E];
                double v[E][1];
                double av[E];
                bool need[E];
                bool any = false;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    key[e] = -1;
                    v[e][0] = 0.0;
                    av[e] = 0.0;
                    need[e] = false;
                    if (active && valid[e]) {
                        const int64_t a = amp_offset[e] + vaoff + fastdiv(s - vfirst, O.step_div);
                        const uint8_t af = O.amp_flags[a];
                        av[e] = (SIG == 2) ? drow[e][s] : O.amps_in[a];
                        const uint8_t fl = use_flags ? frow[e][s] : (uint8_t)0;
                        if (af == 0) {
                            key[e] = a;
                            need[e] = (fl & fmask) == 0;
                        }
                    }
                    any = any || need[e];
                }
                if (any) {
                    double c4h, s4h;
                    hwp_cs4<MODE>(P, s, c4h, s4h);
                    double wk[E][NNZ];
                    int64_t pidx[E];
                    if constexpr (E == 2) {
                        otf_point_pair<NEST, MODE, PIX>(P, D[0], D[1], s, s_tab, c4h, s4h, wk[0], wk[1], pidx[0],
                                                        pidx[1]);
                    } else {
                        pidx[0] = otf_point<NEST, MODE, PIX>(P, D[0], s, s_tab, c4h, s4h, wk[0]);
                    }
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        if (need[e]) {
                            double d = (SIG == 2) ? av[e] : 0.0 + av[e];
                            if (pidx[e] >= 0) {
                                const double * m = map + NNZ * pidx[e];
                                double sc = 0.0;
#pragma unroll
                                for (int k = 0; k < NNZ; ++k) sc += wk[e][k] * m[k];
                                sc *= 1.0;
                                d -= sc;
                            }
                            v[e][0] = d * dw[e];
                        }
                    }
                }
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const bool tail = wave_run_reduce<1>(key[e], v[e]);
                    if (tail && key[e] >= 0) unsafeAtomicAdd(O.amps_out + key[e], v[e][0]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------
struct OtfHost {
    OtfDev dev;
    int mode;
    bool nest;
    bool compact;
    size_t o_fp, o_eps, o_gamma, o_cal, o_ci;
};

// Validate the descriptor and stage its per-detector arrays in the parameter block.
OtfHost otf_prepare(const toast_hip_otf_pointing * pt, int64_t n_det, int64_t n_samp, int64_t n_pix_submap,
                    const int64_t * d_g2l, ParamBlock & pb) {
    if (pt == nullptr) fail_arg("otf pointing descriptor is NULL");
    if (pt->nnz != 1 && pt->nnz != 3) fail_arg("otf pointing: nnz must be 1 (I) or 3 (IQU)");
    if (n_pix_submap <= 0) fail_arg("n_pix_submap must be positive");
    if (pt->d_boresight == nullptr || pt->focalplane == nullptr) fail_arg("otf pointing: boresight / focalplane missing");
    need_aligned(pt->d_boresight, "boresight");
    OtfHost h;
    h.nest = pt->nest != 0;
    const bool hwp = (pt->nnz == 3) && (pt->n_hwp == n_samp) && (pt->d_hwp != nullptr);
    h.mode = (pt->nnz == 1) ? 0 : (hwp ? 2 : 1);
    std::vector<double> zeros(n_det, 0.0), ones(n_det, 1.0);
    h.o_fp = pb.push(pt->focalplane, sizeof(double) * 4 * n_det);
    h.o_eps = pb.push(pt->epsilon ? pt->epsilon : zeros.data(), sizeof(double) * n_det);
    h.o_gamma = pb.push(pt->gamma ? pt->gamma : zeros.data(), sizeof(double) * n_det);
    h.o_cal = pb.push(pt->cal ? pt->cal : ones.data(), sizeof(double) * n_det);
    h.compact = pt->d_compact_pixels != nullptr;
    h.o_ci = 0;
    if (h.compact) {
        if (pt->compact_index == nullptr) fail_arg("otf pointing: compact_index is required with d_compact_pixels");
        h.o_ci = pb.push(pt->compact_index, sizeof(int32_t) * n_det);
    }
    OtfDev & d = h.dev;
    d.g2l = d_g2l;
    d.cpix = pt->d_compact_pixels;
    d.cpix_idx = nullptr;
    d.bore = pt->d_boresight;
    d.pflags = pt->d_shared_flags;
    d.hwp = pt->d_hwp;
    d.hwp_tab = hwp ? pt->d_hwp_table : nullptr;
    if (d.hwp_tab != nullptr) need_aligned(d.hwp_tab, "hwp table");
    d.usign = pt->IAU ? -1.0 : 1.0;
    d.ref_nan = stokes_reference_nan() ? 1 : 0;
    d.nside = pt->nside;
    d.nps_div = make_fastdiv(n_pix_submap);
    d.factor = log2_exact(pt->nside);
    d.use_pflags = (pt->n_shared_flags == n_samp && pt->d_shared_flags != nullptr) ? 1 : 0;
    d.pmask = pt->shared_flag_mask;
    return h;
}

void otf_bind(OtfHost & h, const char * base) {
    h.dev.fp = (const double *)(base + h.o_fp);
    h.dev.eps = (const double *)(base + h.o_eps);
    h.dev.gamma = (const double *)(base + h.o_gamma);
    h.dev.cal = (const double *)(base + h.o_cal);
    if (h.compact) h.dev.cpix_idx = (const int32_t *)(base + h.o_ci);
}

template <int SIG, int PIX, int E, typename... Args>
void launch_accumulate_pix(const OtfHost & h, dim3 grid, hipStream_t st, Args... args) {
#define TH_OTF_CASE(N, M)                                                                        \
    hipLaunchKernelGGL((k_otf_accumulate<N, M, SIG, PIX, E>), grid, dim3(kThreads), 0, st, args...)
    // with cached pixels the ordering scheme plays no role: one instantiation serves both
    const bool nest = (PIX == 1) ? true : h.nest;
    if (nest) {
        if (h.mode == 0) TH_OTF_CASE(true, 0);
        else if (h.mode == 1) TH_OTF_CASE(true, 1);
        else TH_OTF_CASE(true, 2);
    } else if constexpr (PIX == 0) {
        if (h.mode == 0) TH_OTF_CASE(false, 0);
        else if (h.mode == 1) TH_OTF_CASE(false, 1);
        else TH_OTF_CASE(false, 2);
    }
#undef TH_OTF_CASE
}

// grid.x = detectors of the call; halved here when detector pairs share a workgroup
template <int SIG, typename... Args>
void launch_accumulate(const OtfHost & h, dim3 grid, hipStream_t st, Args... args) {
    const bool pair = pair_detectors() && grid.x >= 2;
    if (pair) grid.x = (grid.x + 1) / 2;
    if (h.compact) {
        if (pair) launch_accumulate_pix<SIG, 1, 2>(h, grid, st, args...);
        else launch_accumulate_pix<SIG, 1, 1>(h, grid, st, args...);
    } else {
        if (pair) launch_accumulate_pix<SIG, 0, 2>(h, grid, st, args...);
        else launch_accumulate_pix<SIG, 0, 1>(h, grid, st, args...);
    }
}

template <int SIG, int PIX, int E, typename... Args>
void launch_scan_pix(const OtfHost & h, dim3 grid, hipStream_t st, Args... args) {
#define TH_OTF_CASE(N, M)                                                                        \
    hipLaunchKernelGGL((k_otf_scan<N, M, SIG, PIX, E>), grid, dim3(kThreads), 0, st, args...)
    const bool nest = (PIX == 1) ? true : h.nest;
    if (nest) {
        if (h.mode == 0) TH_OTF_CASE(true, 0);
        else if (h.mode == 1) TH_OTF_CASE(true, 1);
        else TH_OTF_CASE(true, 2);
    } else if constexpr (PIX == 0) {
        if (h.mode == 0) TH_OTF_CASE(false, 0);
        else if (h.mode == 1) TH_OTF_CASE(false, 1);
        else TH_OTF_CASE(false, 2);
    }
#undef TH_OTF_CASE
}

// grid.x = detectors of the call; halved when detector pairs share a workgroup (only where the pixel is computed
// in the kernel: with the compact pixel cache there is nothing to share)
template <int SIG, typename... Args>
void launch_scan(const OtfHost & h, dim3 grid, hipStream_t st, Args... args) {
    if (h.compact) {
        launch_scan_pix<SIG, 1, 1>(h, grid, st, args...);
    } else if (pair_detectors() && grid.x >= 2) {
        grid.x = (grid.x + 1) / 2;
        launch_scan_pix<SIG, 0, 2>(h, grid, st, args...);
    } else {
        launch_scan_pix<SIG, 0, 1>(h, grid, st, args...);
    }
}

__global__ __launch_bounds__(kThreads) void k_hwp_table(int64_t n, const double * __restrict__ hwp,
                                                        double * __restrict__ tab) {
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        double s4h, c4h;
        sincos(4.0 * hwp[i], &s4h, &c4h);
        *reinterpret_cast<double2 *>(tab + 2 * i) = make_double2(c4h, s4h);
    }
}

// ------------------------------------------------------------------------------------
// Quaternion-free pointing expansion: the outputs of pixels_healpix / stokes_weights written
// straight from the boresight, without the [n_det, n_samp, 4] detector-quaternion buffer in
// between (cfg-3: 23.6 GB that is never allocated, written or read twice).
// ------------------------------------------------------------------------------------
template <bool NEST, int E>
__global__ __launch_bounds__(kThreads) void k_otf_pixels(
    const Chunk * __restrict__ chunks, int n_chunks, int n_det, OtfDev P, const int32_t * __restrict__ p_idx,
    int64_t * __restrict__ pixels, uint8_t * __restrict__ hsub, int64_t n_samp) {
    __shared__ double s_tab[2 * TOAST_ATAN_TABLE_N];
    if (threadIdx.x < 2 * TOAST_ATAN_TABLE_N) s_tab[threadIdx.x] = kAtanTab[threadIdx.x];
    __syncthreads();
    DetConst D[E];
    int64_t * prow[E];
    bool valid[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        int det = E * blockIdx.x + e;
        valid[e] = det < n_det;
        if (!valid[e]) det = E * blockIdx.x;
        D[e] = det_const(P, det);
        prow[e] = pixels + (int64_t)p_idx[det] * n_samp;
    }
    const int lane = threadIdx.x & 63;
    for (int ci = blockIdx.y; ci < n_chunks; ci += gridDim.y) {
        const Chunk c = chunks[ci];
        for (int base = 0; base < c.count; base += kThreads) {
            const int i = base + threadIdx.x;
            const bool active = i < c.count;
            const int64_t s = c.first + (active ? i : 0);
            double w[E][1];
            int64_t pix[E];
            if constexpr (E == 2) {
                otf_point_pair<NEST, 0, 2>(P, D[0], D[1], s, s_tab, 1.0, 0.0, w[0], w[1], pix[0], pix[1]);
            } else {
                pix[0] = otf_point<NEST, 0, 2>(P, D[0], s, s_tab, 1.0, 0.0, w[0]);
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                int64_t sub = (pix[e] >= 0 && active && valid[e]) ? fastdiv(pix[e], P.nps_div) : -1;
                const int64_t prev = __shfl_up(sub, 1);
                if (active && valid[e]) {
                    prow[e][s] = pix[e];
                    if (sub >= 0 && (lane == 0 || prev != sub)) hsub[sub] = 1;
                }
            }
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(kThreads) void k_otf_weights(
    const Chunk * __restrict__ chunks, int n_chunks, OtfDev P, const int32_t * __restrict__ w_idx,
    double * __restrict__ weights, int64_t n_samp) {
    constexpr int NNZ = ModeNnz<MODE>::value;
    const int det = blockIdx.x;
    const DetConst D = det_const(P, det);
    double * wrow = weights + (int64_t)w_idx[det] * n_samp * NNZ;
    for (int ci = blockIdx.y; ci < n_chunks; ci += gridDim.y) {
        const Chunk c = chunks[ci];
        for (int i = threadIdx.x; i < c.count; i += kThreads) {
            const int64_t s = c.first + i;
            double c4h, s4h, w[NNZ];
            hwp_cs4<MODE>(P, s, c4h, s4h);
            (void)otf_point<true, MODE, 3>(P, D, s, nullptr, c4h, s4h, w);
#pragma unroll
            for (int k = 0; k < NNZ; ++k) wrow[NNZ * s + k] = w[k];
        }
    }
}

// boresight -> int32 local map indices directly (no int64 pixel buffer at all)
template <bool NEST, int E>
__global__ __launch_bounds__(kThreads) void k_otf_compact_pixels(
    const Chunk * __restrict__ chunks, int n_chunks, int n_det, OtfDev P, const int32_t * __restrict__ c_idx,
    int32_t * __restrict__ cpix, int64_t n_samp) {
    __shared__ double s_tab[2 * TOAST_ATAN_TABLE_N];
    if (threadIdx.x < 2 * TOAST_ATAN_TABLE_N) s_tab[threadIdx.x] = kAtanTab[threadIdx.x];
    __syncthreads();
    DetConst D[E];
    int32_t * crow[E];
    bool valid[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        int det = E * blockIdx.x + e;
        valid[e] = det < n_det;
        if (!valid[e]) det = E * blockIdx.x;
        D[e] = det_const(P, det);
        crow[e] = cpix + (int64_t)c_idx[det] * n_samp;
    }
    for (int ci = blockIdx.y; ci < n_chunks; ci += gridDim.y) {
        const Chunk c = chunks[ci];
        for (int i = threadIdx.x; i < c.count; i += kThreads) {
            const int64_t s = c.first + i;
            double w[E][1];
            int64_t l[E];
            if constexpr (E == 2) {
                otf_point_pair<NEST, 0, 0>(P, D[0], D[1], s, s_tab, 1.0, 0.0, w[0], w[1], l[0], l[1]);
            } else {
                l[0] = otf_point<NEST, 0, 0>(P, D[0], s, s_tab, 1.0, 0.0, w[0]);
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if (l[e] < 0) l[e] = -1;   // flagged sample or pixel in a submap that is not local
                if (valid[e]) crow[e][s] = (int32_t)l[e];
            }
        }
    }
}

// int64 global pixels -> int32 local map indices (the compact cache read by PIX 1)
__global__ __launch_bounds__(kThreads) void k_compact_pixels(
    const Chunk * __restrict__ chunks, int n_chunks, const int32_t * __restrict__ p_idx,
    const int32_t * __restrict__ c_idx, const int64_t * __restrict__ pixels,
    int32_t * __restrict__ cpix, const int64_t * __restrict__ g2l, FastDiv nps_div, int64_t n_samp) {
    const int det = blockIdx.x;
    const int64_t * prow = pixels + (int64_t)p_idx[det] * n_samp;
    int32_t * crow = cpix + (int64_t)c_idx[det] * n_samp;
    for (int ci = blockIdx.y; ci < n_chunks; ci += gridDim.y) {
        const Chunk c = chunks[ci];
        for (int i = threadIdx.x; i < c.count; i += kThreads) {
            const int64_t s = c.first + i;
            const int64_t p = prow[s];
            int64_t l = -1;
            if (p >= 0) {
                const int64_t gsm = fastdiv(p, nps_div);
                const int64_t lsm = g2l[gsm];
                if (lsm >= 0) l = lsm * nps_div.d + (p - gsm * nps_div.d);
            }
            crow[s] = (int32_t)l;
        }
    }
}

}  // namespace

extern "C" {

int toast_hip_otf_build_noise_weighted_dev(
    const toast_hip_otf_pointing * pointing, const int64_t * d_g2l, double * d_zmap, int64_t n_pix_submap,
    const int32_t * data_index, const double * d_det_data, const int32_t * flag_index,
    const uint8_t * d_det_flags, int64_t n_flag_samp, const double * det_scale, uint8_t det_flag_mask,
    int64_t n_det, int64_t n_samp, const toast_hip_interval * intervals, int64_t n_view,
    const uint8_t * d_shared_flags, int64_t n_shared_flags, uint8_t shared_flag_mask, void * stream) {
    return guarded([&] {
        if (n_det <= 0) return;
        const auto chunks = make_chunks(intervals, n_view, n_samp);
        if (chunks.empty()) return;
        ParamBlock pb;
        OtfHost h = otf_prepare(pointing, n_det, n_samp, n_pix_submap, d_g2l, pb);
        const int use_d = (n_flag_samp == n_samp) ? 1 : 0;
        const int use_s = (n_shared_flags == n_samp) ? 1 : 0;
        std::vector<int32_t> fidx(n_det, 0);
        if (use_d) std::memcpy(fidx.data(), flag_index, sizeof(int32_t) * n_det);
        const size_t o_ch = pb.push_vec(chunks);
        const size_t o_di = pb.push(data_index, sizeof(int32_t) * n_det);
        const size_t o_fi = pb.push_vec(fidx);
        const size_t o_ds = pb.push(det_scale, sizeof(double) * n_det);
        hipStream_t st = as_stream(stream);
        const char * d = pb.commit(st);
        otf_bind(h, d);
        OffsetDev off{};
        launch_accumulate<0>(h, chunk_grid(n_det, chunks.size()), st, (const Chunk *)(d + o_ch),
                             (int)chunks.size(), (int)n_det, h.dev, off, (const int32_t *)(d + o_di), d_det_data,
                             (const int32_t *)(d + o_fi), d_det_flags, det_flag_mask, use_d, d_shared_flags,
                             shared_flag_mask, use_s, (const double *)(d + o_ds), d_zmap, n_samp);
        check_launch();
    });
}

int toast_hip_otf_scan_map_dev(const toast_hip_otf_pointing * pointing, const int64_t * d_g2l,
                               const double * d_map, int64_t n_pix_submap, double * d_det_data,
                               const int32_t * data_index, int64_t n_det, int64_t n_samp,
                               const toast_hip_interval * intervals, int64_t n_view, double data_scale,
                               int should_zero, int should_subtract, const double * det_weights,
                               void * stream) {
    return guarded([&] {
        if (n_det <= 0) return;
        const auto chunks = make_chunks(intervals, n_view, n_samp);
        if (chunks.empty()) return;
        ParamBlock pb;
        OtfHost h = otf_prepare(pointing, n_det, n_samp, n_pix_submap, d_g2l, pb);
        const size_t o_ch = pb.push_vec(chunks);
        const size_t o_di = pb.push(data_index, sizeof(int32_t) * n_det);
        const size_t o_dw = det_weights ? pb.push(det_weights, sizeof(double) * n_det) : 0;
        hipStream_t st = as_stream(stream);
        const char * d = pb.commit(st);
        otf_bind(h, d);
        OffsetDev off{};
        launch_scan<0>(h, chunk_grid(n_det, chunks.size()), st, (const Chunk *)(d + o_ch), (int)chunks.size(),
                       (int)n_det, h.dev, off, (const int32_t *)(d + o_di), d_det_data, data_scale, should_zero ? 1 : 0,
                       should_subtract ? 1 : 0, (const int32_t *)nullptr, (const uint8_t *)nullptr, (uint8_t)0, 0,
                       det_weights ? (const double *)(d + o_dw) : (const double *)nullptr, d_map, n_samp);
        check_launch();
    });
}

int toast_hip_otf_offset_accumulate_dev(
    const toast_hip_otf_pointing * pointing, int64_t step_length, const int64_t * amp_offsets,
    const int64_t * n_amp_views, const double * d_amplitudes, const uint8_t * d_amplitude_flags,
    const int64_t * d_g2l, double * d_zmap, int64_t n_pix_submap, const int32_t * flag_index,
    const uint8_t * d_det_flags, int64_t n_flag_samp, const double * det_scale, uint8_t det_flag_mask,
    int64_t n_det, int64_t n_samp, const toast_hip_interval * intervals, int64_t n_view,
    const uint8_t * d_shared_flags, int64_t n_shared_flags, uint8_t shared_flag_mask, void * stream) {
    return guarded([&] {
        if (n_det <= 0) return;
        if (step_length <= 0) fail_arg("step_length must be positive");
        const auto chunks = make_chunks(intervals, n_view, n_samp);
        if (chunks.empty()) return;
        const OffsetViews ov = offset_views(intervals, n_amp_views, n_view);
        ParamBlock pb;
        OtfHost h = otf_prepare(pointing, n_det, n_samp, n_pix_submap, d_g2l, pb);
        const int use_d = (n_flag_samp == n_samp) ? 1 : 0;
        const int use_s = (n_shared_flags == n_samp) ? 1 : 0;
        std::vector<int32_t> fidx(n_det, 0);
        if (use_d) std::memcpy(fidx.data(), flag_index, sizeof(int32_t) * n_det);
        const size_t o_ch = pb.push_vec(chunks);
        const size_t o_vf = pb.push_vec(ov.first);
        const size_t o_va = pb.push_vec(ov.aoff);
        const size_t o_ao = pb.push(amp_offsets, sizeof(int64_t) * n_det);
        const size_t o_fi = pb.push_vec(fidx);
        const size_t o_ds = pb.push(det_scale, sizeof(double) * n_det);
        hipStream_t st = as_stream(stream);
        const char * d = pb.commit(st);
        otf_bind(h, d);
        OffsetDev off{(const int64_t *)(d + o_vf), (const int64_t *)(d + o_va), (const int64_t *)(d + o_ao),
                      d_amplitudes, nullptr, d_amplitude_flags, make_fastdiv(step_length)};
        launch_accumulate<1>(h, chunk_grid(n_det, chunks.size()), st, (const Chunk *)(d + o_ch),
                             (int)chunks.size(), (int)n_det, h.dev, off, (const int32_t *)nullptr, (const double *)nullptr,
                             (const int32_t *)(d + o_fi), d_det_flags, det_flag_mask, use_d, d_shared_flags,
                             shared_flag_mask, use_s, (const double *)(d + o_ds), d_zmap, n_samp);
        check_launch();
    });
}

int toast_hip_otf_offset_clean_accumulate_dev(
    const toast_hip_otf_pointing * pointing, int64_t step_length, const int64_t * amp_offsets,
    const int64_t * n_amp_views, const double * d_amplitudes, const uint8_t * d_amplitude_flags,
    const int64_t * d_g2l, double * d_zmap, int64_t n_pix_submap, const int32_t * data_index, const double * d_signal,
    const int32_t * flag_index, const uint8_t * d_det_flags, int64_t n_flag_samp, const double * det_scale,
    uint8_t det_flag_mask, int64_t n_det, int64_t n_samp, const toast_hip_interval * intervals, int64_t n_view,
    const uint8_t * d_shared_flags, int64_t n_shared_flags, uint8_t shared_flag_mask, void * stream) {
    return guarded([&] {
        if (n_det <= 0) return;
        if (step_length <= 0) fail_arg("step_length must be positive");
        if (d_signal == nullptr || data_index == nullptr) fail_arg("otf_offset_clean_accumulate: the signal and its row indices must not be null");
        const auto chunks = make_chunks(intervals, n_view, n_samp);
        if (chunks.empty()) return;
        const OffsetViews ov = offset_views(intervals, n_amp_views, n_view);
        ParamBlock pb;
        OtfHost h = otf_prepare(pointing, n_det, n_samp, n_pix_submap, d_g2l, pb);
        const int use_d = (n_flag_samp == n_samp) ? 1 : 0;
        const int use_s = (n_shared_flags == n_samp) ? 1 : 0;
        std::vector<int32_t> fidx(n_det, 0);
        if (use_d) std::memcpy(fidx.data(), flag_index, sizeof(int32_t) * n_det);
        const size_t o_ch = pb.push_vec(chunks);
        const size_t o_vf = pb.push_vec(ov.first);
        const size_t o_va = pb.push_vec(ov.aoff);
        const size_t o_ao = pb.push(amp_offsets, sizeof(int64_t) * n_det);
        const size_t o_di = pb.push(data_index, sizeof(int32_t) * n_det);
        const size_t o_fi = pb.push_vec(fidx);
        const size_t o_ds = pb.push(det_scale, sizeof(double) * n_det);
        hipStream_t st = as_stream(stream);
        const char * d = pb.commit(st);
        otf_bind(h, d);
        OffsetDev off{(const int64_t *)(d + o_vf), (const int64_t *)(d + o_va), (const int64_t *)(d + o_ao),
                      d_amplitudes, nullptr, d_amplitude_flags, make_fastdiv(step_length)};
        launch_accumulate<2>(h, chunk_grid(n_det, chunks.size()), st, (const Chunk *)(d + o_ch),
                             (int)chunks.size(), (int)n_det, h.dev, off, (const int32_t *)(d + o_di), d_signal,
                             (const int32_t *)(d + o_fi), d_det_flags, det_flag_mask, use_d, d_shared_flags,
                             shared_flag_mask, use_s, (const double *)(d + o_ds), d_zmap, n_samp);
        check_launch();
    });
}

int toast_hip_otf_offset_scan_project_dev(
    const toast_hip_otf_pointing * pointing, int64_t step_length, const int64_t * amp_offsets,
    const int64_t * n_amp_views, const double * d_amplitudes_in, double * d_amplitudes_out,
    const uint8_t * d_amplitude_flags, const int64_t * d_g2l, const double * d_map, int64_t n_pix_submap,
    const int32_t * flag_index, const uint8_t * d_det_flags, int64_t n_flag_samp, uint8_t det_flag_mask,
    const double * det_weights, int64_t n_det, int64_t n_samp, const toast_hip_interval * intervals,
    int64_t n_view, void * stream) {
    return guarded([&] {
        if (n_det <= 0) return;
        if (step_length <= 0) fail_arg("step_length must be positive");
        if (det_weights == nullptr) fail_arg("det_weights is required");
        const auto chunks = make_chunks(intervals, n_view, n_samp);
        if (chunks.empty()) return;
        const OffsetViews ov = offset_views(intervals, n_amp_views, n_view);
        ParamBlock pb;
        OtfHost h = otf_prepare(pointing, n_det, n_samp, n_pix_submap, d_g2l, pb);
        const int use_f = (n_flag_samp == n_samp) ? 1 : 0;
        std::vector<int32_t> fidx(n_det, 0);
        if (use_f) std::memcpy(fidx.data(), flag_index, sizeof(int32_t) * n_det);
        const size_t o_ch = pb.push_vec(chunks);
        const size_t o_vf = pb.push_vec(ov.first);
        const size_t o_va = pb.push_vec(ov.aoff);
        const size_t o_ao = pb.push(amp_offsets, sizeof(int64_t) * n_det);
        const size_t o_fi = pb.push_vec(fidx);
        const size_t o_dw = pb.push(det_weights, sizeof(double) * n_det);
        hipStream_t st = as_stream(stream);
        const char * d = pb.commit(st);
        otf_bind(h, d);
        OffsetDev off{(const int64_t *)(d + o_vf), (const int64_t *)(d + o_va), (const int64_t *)(d + o_ao),
                      d_amplitudes_in, d_amplitudes_out, d_amplitude_flags, make_fastdiv(step_length)};
        launch_scan<1>(h, chunk_grid(n_det, chunks.size()), st, (const Chunk *)(d + o_ch), (int)chunks.size(),
                       (int)n_det, h.dev, off, (const int32_t *)nullptr, (double *)nullptr, 1.0, 0, 1,
                       (const int32_t *)(d + o_fi), d_det_flags, det_flag_mask, use_f,
                       (const double *)(d + o_dw), d_map, n_samp);
        check_launch();
    });
}

int toast_hip_otf_offset_scan_project_signal_dev(
    const toast_hip_otf_pointing * pointing, int64_t step_length, const int64_t * amp_offsets,
    const int64_t * n_amp_views, const int32_t * signal_index, const double * d_signal, double * d_amplitudes_out,
    const uint8_t * d_amplitude_flags, const int64_t * d_g2l, const double * d_map, int64_t n_pix_submap,
    const int32_t * flag_index, const uint8_t * d_det_flags, int64_t n_flag_samp, uint8_t det_flag_mask,
    const double * det_weights, int64_t n_det, int64_t n_samp, const toast_hip_interval * intervals,
    int64_t n_view, void * stream) {
    return guarded([&] {
        if (n_det <= 0) return;
        if (step_length <= 0) fail_arg("step_length must be positive");
        if (det_weights == nullptr) fail_arg("det_weights is required");
        if (d_signal == nullptr || signal_index == nullptr) fail_arg("signal and signal_index are required");
        const auto chunks = make_chunks(intervals, n_view, n_samp);
        if (chunks.empty()) return;
        const OffsetViews ov = offset_views(intervals, n_amp_views, n_view);
        ParamBlock pb;
        OtfHost h = otf_prepare(pointing, n_det, n_samp, n_pix_submap, d_g2l, pb);
        const int use_f = (n_flag_samp == n_samp) ? 1 : 0;
        std::vector<int32_t> fidx(n_det, 0);
        if (use_f) std::memcpy(fidx.data(), flag_index, sizeof(int32_t) * n_det);
        const size_t o_ch = pb.push_vec(chunks);
        const size_t o_vf = pb.push_vec(ov.first);
        const size_t o_va = pb.push_vec(ov.aoff);
        const size_t o_ao = pb.push(amp_offsets, sizeof(int64_t) * n_det);
        const size_t o_fi = pb.push_vec(fidx);
        const size_t o_dw = pb.push(det_weights, sizeof(double) * n_det);
        const size_t o_si = pb.push(signal_index, sizeof(int32_t) * n_det);
        hipStream_t st = as_stream(stream);
        const char * d = pb.commit(st);
        otf_bind(h, d);
        OffsetDev off{(const int64_t *)(d + o_vf), (const int64_t *)(d + o_va), (const int64_t *)(d + o_ao),
                      nullptr, d_amplitudes_out, d_amplitude_flags, make_fastdiv(step_length)};
        launch_scan<2>(h, chunk_grid(n_det, chunks.size()), st, (const Chunk *)(d + o_ch), (int)chunks.size(),
                       (int)n_det, h.dev, off, (const int32_t *)(d + o_si), const_cast<double *>(d_signal), 1.0, 0, 1,
                       (const int32_t *)(d + o_fi), d_det_flags, det_flag_mask, use_f,
                       (const double *)(d + o_dw), d_map, n_samp);
        check_launch();
    });
}

int toast_hip_otf_pixels_healpix_dev(const toast_hip_otf_pointing * pointing, const int32_t * pixel_index,
                                     int64_t * d_pixels, int64_t n_det, int64_t n_samp,
                                     const toast_hip_interval * intervals, int64_t n_view,
                                     uint8_t * d_hit_submaps, int64_t n_submap, int64_t n_pix_submap,
                                     void * stream) {
    return guarded([&] {
        if (n_det <= 0) return;
        if (n_pix_submap <= 0) fail_arg("n_pix_submap must be positive");
        if (n_submap * n_pix_submap < 12 * pointing->nside * pointing->nside) {
            fail_arg("hit_submaps is too short for this nside / n_pix_submap");
        }
        const auto chunks = make_chunks(intervals, n_view, n_samp);
        if (chunks.empty()) return;
        ParamBlock pb;
        toast_hip_otf_pointing pt = *pointing;
        pt.d_compact_pixels = nullptr;
        pt.nnz = 1;
        OtfHost h = otf_prepare(&pt, n_det, n_samp, n_pix_submap, nullptr, pb);
        const size_t o_ch = pb.push_vec(chunks);
        const size_t o_pi = pb.push(pixel_index, sizeof(int32_t) * n_det);
        hipStream_t st = as_stream(stream);
        const char * d = pb.commit(st);
        otf_bind(h, d);
        const bool pair = pair_detectors() && n_det >= 2;
        auto kern = pair ? (h.nest ? k_otf_pixels<true, 2> : k_otf_pixels<false, 2>)
                         : (h.nest ? k_otf_pixels<true, 1> : k_otf_pixels<false, 1>);
        hipLaunchKernelGGL(kern, chunk_grid(pair ? (n_det + 1) / 2 : n_det, chunks.size()), dim3(kThreads), 0, st,
                           (const Chunk *)(d + o_ch), (int)chunks.size(), (int)n_det, h.dev,
                           (const int32_t *)(d + o_pi), d_pixels, d_hit_submaps, n_samp);
        check_launch();
    });
}

int toast_hip_otf_stokes_weights_dev(const toast_hip_otf_pointing * pointing, const int32_t * weight_index,
                                     double * d_weights, int64_t n_det, int64_t n_samp,
                                     const toast_hip_interval * intervals, int64_t n_view, void * stream) {
    return guarded([&] {
        if (n_det <= 0) return;
        const auto chunks = make_chunks(intervals, n_view, n_samp);
        if (chunks.empty()) return;
        ParamBlock pb;
        toast_hip_otf_pointing pt = *pointing;
        pt.d_compact_pixels = nullptr;
        OtfHost h = otf_prepare(&pt, n_det, n_samp, 1, nullptr, pb);
        const size_t o_ch = pb.push_vec(chunks);
        const size_t o_wi = pb.push(weight_index, sizeof(int32_t) * n_det);
        hipStream_t st = as_stream(stream);
        const char * d = pb.commit(st);
        otf_bind(h, d);
        const dim3 grid = chunk_grid(n_det, chunks.size());
#define TH_OTF_W(M)                                                                               \
    hipLaunchKernelGGL(k_otf_weights<M>, grid, dim3(kThreads), 0, st, (const Chunk *)(d + o_ch),  \
                       (int)chunks.size(), h.dev, (const int32_t *)(d + o_wi), d_weights, n_samp)
        if (h.mode == 0) TH_OTF_W(0);
        else if (h.mode == 1) TH_OTF_W(1);
        else TH_OTF_W(2);
#undef TH_OTF_W
        check_launch();
    });
}

int toast_hip_hwp_table_dev(const double * d_hwp, int64_t n_samp, double * d_table, void * stream) {
    return guarded([&] {
        if (n_samp <= 0) return;
        need_aligned(d_table, "hwp table");
        hipLaunchKernelGGL(k_hwp_table, flat_grid(n_samp), dim3(kThreads), 0, as_stream(stream), n_samp, d_hwp,
                           d_table);
        check_launch();
    });
}

int toast_hip_otf_compact_pixels_dev(const toast_hip_otf_pointing * pointing, const int64_t * d_g2l,
                                     int64_t n_pix_submap, int64_t n_local_submap,
                                     const int32_t * compact_index, int32_t * d_compact_pixels, int64_t n_det,
                                     int64_t n_samp, const toast_hip_interval * intervals, int64_t n_view,
                                     void * stream) {
    return guarded([&] {
        if (n_det <= 0) return;
        if (n_local_submap * n_pix_submap > (int64_t)INT32_MAX) {
            fail_arg("compact pixels: the local map has more than 2^31-1 pixels");
        }
        const auto chunks = make_chunks(intervals, n_view, n_samp);
        if (chunks.empty()) return;
        ParamBlock pb;
        toast_hip_otf_pointing pt = *pointing;
        pt.d_compact_pixels = nullptr;   // the pixels are computed here
        pt.nnz = 1;
        OtfHost h = otf_prepare(&pt, n_det, n_samp, n_pix_submap, d_g2l, pb);
        const size_t o_ch = pb.push_vec(chunks);
        const size_t o_ci = pb.push(compact_index, sizeof(int32_t) * n_det);
        hipStream_t st = as_stream(stream);
        const char * d = pb.commit(st);
        otf_bind(h, d);
        const bool pair = pair_detectors() && n_det >= 2;
        auto kern = pair ? (h.nest ? k_otf_compact_pixels<true, 2> : k_otf_compact_pixels<false, 2>)
                         : (h.nest ? k_otf_compact_pixels<true, 1> : k_otf_compact_pixels<false, 1>);
        hipLaunchKernelGGL(kern, chunk_grid(pair ? (n_det + 1) / 2 : n_det, chunks.size()), dim3(kThreads), 0, st,
                           (const Chunk *)(d + o_ch), (int)chunks.size(), (int)n_det, h.dev,
                           (const int32_t *)(d + o_ci), d_compact_pixels, n_samp);
        check_launch();
    });
}

int toast_hip_compact_pixels_dev(const int64_t * d_g2l, int64_t n_pix_submap, int64_t n_local_submap,
                                 const int32_t * pixel_index, const int64_t * d_pixels,
                                 const int32_t * compact_index, int32_t * d_compact_pixels, int64_t n_det,
                                 int64_t n_samp, const toast_hip_interval * intervals, int64_t n_view,
                                 void * stream) {
    return guarded([&] {
        if (n_det <= 0) return;
        if (n_pix_submap <= 0) fail_arg("n_pix_submap must be positive");
        if (n_local_submap * n_pix_submap > (int64_t)INT32_MAX) {
            fail_arg("compact pixels: the local map has more than 2^31-1 pixels");
        }
        const auto chunks = make_chunks(intervals, n_view, n_samp);
        if (chunks.empty()) return;
        ParamBlock pb;
        const size_t o_ch = pb.push_vec(chunks);
        const size_t o_pi = pb.push(pixel_index, sizeof(int32_t) * n_det);
        const size_t o_ci = pb.push(compact_index, sizeof(int32_t) * n_det);
        hipStream_t st = as_stream(stream);
        const char * d = pb.commit(st);
        hipLaunchKernelGGL(k_compact_pixels, chunk_grid(n_det, chunks.size()), dim3(kThreads), 0, st,
                           (const Chunk *)(d + o_ch), (int)chunks.size(), (const int32_t *)(d + o_pi),
                           (const int32_t *)(d + o_ci), d_pixels, d_compact_pixels, d_g2l,
                           make_fastdiv(n_pix_submap), n_samp);
        check_launch();
    });
}

}  // extern "C"
