// hpix_math.hpp -- per-sample pointing math of the MI355X map-making path.
//
// Device functions for gfx950 (also compilable by g++ for the CPU self-check in
// tests/devmath_host.cpp, which is how bit-parity of this arithmetic with the reference
// is verified without a GPU).  What each function mirrors in the reference
// (paths relative to /root/reference/src/toast/_libtoast/):
//
//   quat_rotate                   ops_pixels_healpix.cpp:50-76, ops_stokes_weights.cpp:21-48
//   quat_mult                     ops_pointing_detector.cpp:21-31
//   atan2_dd                      libm atan2 as called at ops_pixels_healpix.cpp:118
//   zphi_from_vec                 ops_pixels_healpix.cpp:104-120
//   phi_to_tt                     ops_pixels_healpix.cpp:44-48, 132-139
//   zphi_to_nest / zphi_to_ring   ops_pixels_healpix.cpp:122-208 / 210-276
//   stokes_alpha                  ops_stokes_weights.cpp:50-75
//
// Bit-exactness strategy (HEALPix indices must equal libtoast's CPU result exactly):
//   * this file is always compiled with -ffp-contract=off, so every *, + below is one
//     IEEE-754 operation, in the reference's order; fma() appears only where written;
//   * f64 division and sqrt are correctly rounded on gfx950 (checked on the GPU by
//     tests/test_gpu_math.py), as they are on x86-64;
//   * the only libm transcendental on the pixel path is atan2.  glibc's is accurate to
//     well under 1 ulp; atan2_dd evaluates atan2 in double-double (~2^-66 relative) and
//     rounds once, so the two agree except when the exact value lies within ~1e-4 ulp of
//     a rounding boundary, and a differing phi changes the pixel only if the sample also
//     sits within 1 ulp of a pixel edge.  Tests count mismatches (expected, observed: 0).
//   * the Morton interleave is done with shift/mask steps (bit-identical to the
//     reference's 256-entry table, without the memory traffic).
#pragma once

#include <cstdint>

#if defined(__HIPCC__)
# include <hip/hip_runtime.h>
# define TOAST_HD __host__ __device__ __forceinline__
# define TOAST_HD_CONST static __device__ const
#else
# include <cmath>
# define TOAST_HD inline
# define TOAST_HD_CONST static const
#endif

namespace toast_hip {

#include "atan_table.inc"

#define TOAST_TWOTHIRDS 0.66666666666666666667  // ops_pixels_healpix.cpp:16
#define TOAST_TWOPI (2 * 3.14159265358979323846)  // 2 * M_PI
#define TOAST_2_OVER_PI 0.63661977236758134308   // M_2_PI

TOAST_HD double f_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
TOAST_HD double f_abs(double a) { return __builtin_fabs(a); }
TOAST_HD double f_sqrt(double a) { return __builtin_sqrt(a); }
TOAST_HD double f_floor(double a) { return __builtin_floor(a); }

// ------------------------------------------------------------------ double-double
struct dd {
    double hi;
    double lo;
};

TOAST_HD dd two_sum(double a, double b) {
    const double s = a + b;
    const double bb = s - a;
    return dd{s, (a - (s - bb)) + (b - bb)};
}

TOAST_HD dd quick_two_sum(double a, double b) {  // requires |a| >= |b| or a == 0
    const double s = a + b;
    return dd{s, b - (s - a)};
}

TOAST_HD dd two_prod(double a, double b) {
    const double p = a * b;
    return dd{p, f_fma(a, b, -p)};
}

TOAST_HD dd dd_add(dd a, dd b) {
    dd s = two_sum(a.hi, b.hi);
    s.lo += a.lo + b.lo;
    return quick_two_sum(s.hi, s.lo);
}

TOAST_HD dd dd_sub(dd a, dd b) { return dd_add(a, dd{-b.hi, -b.lo}); }

// a + b when |a.hi| >= |b.hi| is known (3 additions fewer).
TOAST_HD dd dd_add_ordered(dd a, dd b) {
    dd s = quick_two_sum(a.hi, b.hi);
    s.lo += a.lo + b.lo;
    return quick_two_sum(s.hi, s.lo);
}

// 1/b to ~1 ulp for normal positive b without the IEEE division sequence: integer-seeded
// Newton iteration.  Only + * fma and an integer subtract, so host and device agree bit for
// bit.  (An IEEE f64 divide costs ~35 instructions on gfx950, this is 12.)
TOAST_HD double recip_newton(double b) {
    union {
        double d;
        int64_t i;
    } u;
    u.d = b;
    u.i = 0x7FDE623822FC16E6ll - u.i;  // ~ 1/b within 12 %
    double r = u.d;
    r = r * f_fma(-b, r, 2.0);  // error 1.4e-2
    r = r * f_fma(-b, r, 2.0);  // 2e-4
    r = r * f_fma(-b, r, 2.0);  // 4e-8
    r = r * f_fma(-b, r, 2.0);  // 2e-15
    r = f_fma(r, f_fma(-b, r, 1.0), r);  // ~1 ulp
    return r;
}

// a / b with a ~1 ulp reciprocal and fma residuals (~2^-100 relative); b.hi > 0, normal.
TOAST_HD dd dd_div(dd a, dd b) {
    const double r = recip_newton(b.hi);
    const double q = a.hi * r;
    const double e = f_fma(-q, b.hi, a.hi);
    const double ql = ((e + a.lo) - q * b.lo) * r;
    return quick_two_sum(q, ql);
}

// atan2 evaluated in double-double and rounded once.  `tab` points at the interleaved
// {hi, lo} atan(i/32) table (an LDS copy on the device, kAtanTab itself on the host).
TOAST_HD double atan2_dd(double y, double x, const double * tab) {
    const double ax = f_abs(x);
    const double ay = f_abs(y);
    const bool xneg = __builtin_signbit(x);
    const bool yneg = __builtin_signbit(y);
    if (!(ax == ax) || !(ay == ay)) return x + y;  // NaN in, NaN out
    const bool swap = ay > ax;
    const double big = swap ? ay : ax;
    const double small = swap ? ax : ay;
    dd a;  // atan(small / big), in [0, pi/4]
    if (small == 0.0) {
        a = dd{0.0, 0.0};  // covers (0,0) too
    } else if (big == small) {
        a = dd{tab[64], tab[65]};
    } else {
        const dd t = dd_div(dd{small, 0.0}, dd{big, 0.0});
        const int i = (int)(t.hi * 32.0 + 0.5);
        dd u = t;
        if (i != 0) {
            // u = (t - c) / (1 + t c); t.hi - c is exact because |t - c| <= 1/64 <= c/2
            const double c = (double)i * 0.03125;
            // |t.hi - c| is 0 or >= ulp(t.hi)/2 >= |t.lo|;  1 >= t c
            const dd num = quick_two_sum(t.hi - c, t.lo);
            const dd tc = two_prod(t.hi, c);
            dd den = quick_two_sum(1.0, tc.hi);
            den.lo += tc.lo + t.lo * c;
            u = dd_div(num, den);
        }
        // atan(u) = u + u * p(u^2) for |u| <= 1/64 (next term u^13/13 < 2^-75 relative)
        const double s = u.hi * u.hi;
        double p = -1.0 / 11.0;
        p = f_fma(p, s, 1.0 / 9.0);
        p = f_fma(p, s, -1.0 / 7.0);
        p = f_fma(p, s, 1.0 / 5.0);
        p = f_fma(p, s, -1.0 / 3.0);
        p = p * s;
        const dd au = quick_two_sum(u.hi, u.lo + u.hi * p);
        // atan(i/32) >= 1/32 > |au| for i >= 1; the table entry is exactly 0 for i == 0
        a = (i == 0) ? au : dd_add_ordered(dd{tab[2 * i], tab[2 * i + 1]}, au);
    }
    // pi/2 and pi dominate a (a <= pi/4 resp. <= pi/2)
    if (swap) a = dd_add_ordered(dd{kHalfPi_HI, kHalfPi_LO}, dd{-a.hi, -a.lo});
    if (xneg) a = dd_add_ordered(dd{kPi_HI, kPi_LO}, dd{-a.hi, -a.lo});
    const double r = a.hi + a.lo;
    return yneg ? -r : r;
}

// atan2 in plain double precision, for the Ziv-style fast path of the pixel computation
// (vec_to_pixel below): absolute error < 2^-46 = 1.4e-14 for finite arguments with a normal
// |larger| component, anything else comes out as NaN or as a value the caller's safety check
// rejects.  t = small / big with a 4-step Newton reciprocal (2e-15 relative), atan(t) = t P(t^2)
// with the 17-term Chebyshev fit of atan(sqrt(s)) / sqrt(s) on [0, 1] (7.6e-15 absolute,
// mpmath.chebyfit at 60 digits), octant fix-ups like atan2_dd.  ~55 FP64 instructions against
// ~220 for the double-double evaluation.
#if defined(__HIPCC__)
# define TOAST_CONST_TABLE static __constant__ const
#else
# define TOAST_CONST_TABLE static const
#endif
TOAST_CONST_TABLE double kAtanFastPoly[17] = {
    7.06314263588804606e-05,  -6.77410539964507603e-04, 3.06599570281080062e-03,  -8.79243554221763486e-03,
    1.81985704853957868e-02,  -2.95890121943443718e-02, 4.05177897712506591e-02,  -4.97799667595504730e-02,
    5.79419231688184516e-02,  -6.64614579022125423e-02, 7.68881166240129449e-02,  -9.09048964779903268e-02,
    1.11110775849717722e-01,  -1.42857126460503997e-01, 1.99999999574818260e-01,  -3.33333333328939774e-01,
    9.99999999999992450e-01};
#define TOAST_ATAN2_FAST_ERR 1.1368683772161603e-13   /* 2^-43: bound used by the caller, 8x the budget */
TOAST_HD double atan2_fast(double y, double x) {
    const double ax = f_abs(x);
    const double ay = f_abs(y);
    const bool swap = ay > ax;
    const double big = swap ? ay : ax;
    const double small = swap ? ax : ay;
    union {
        double d;
        int64_t i;
    } u;
    u.d = big;
    u.i = 0x7FDE623822FC16E6ll - u.i;
    double r = u.d;
    r = r * f_fma(-big, r, 2.0);
    r = r * f_fma(-big, r, 2.0);
    r = r * f_fma(-big, r, 2.0);
    r = r * f_fma(-big, r, 2.0);
    const double t = small * r;
    const double s = t * t;
    // coefficients through the scalar unit (constant address space -> s_load -> SGPR operands): as
    // literals the compiler parks all seventeen in VGPRs for the lifetime of the sample loop
    double p = kAtanFastPoly[0];
#pragma unroll
    for (int k = 1; k < 17; ++k) p = f_fma(p, s, kAtanFastPoly[k]);
    double a = t * p;
    if (swap) a = kHalfPi_HI - a;
    if (__builtin_signbit(x)) a = kPi_HI - a;
    return __builtin_signbit(y) ? -a : a;
}

// ------------------------------------------------------------------ quaternions
// Rotate v by unit quaternion q = [x, y, z, w]; literal operation order of the reference.
TOAST_HD void quat_rotate(const double * q, const double * v, double * out) {
    const double xw = q[3] * q[0], yw = q[3] * q[1], zw = q[3] * q[2];
    const double x2 = -q[0] * q[0], xy = q[0] * q[1], xz = q[0] * q[2];
    const double y2 = -q[1] * q[1], yz = q[1] * q[2], z2 = -q[2] * q[2];
    out[0] = 2 * ((y2 + z2) * v[0] + (xy - zw) * v[1] + (yw + xz) * v[2]) + v[0];
    out[1] = 2 * ((zw + xy) * v[0] + (x2 + z2) * v[1] + (yz - xw) * v[2]) + v[1];
    out[2] = 2 * ((xz - yw) * v[0] + (xw + yz) * v[1] + (x2 + y2) * v[2]) + v[2];
}

// quat_rotate(q, (0,0,1)) for finite q, bit-identical to the general form: the reference's
// `(..) * 0.0` terms only contribute signed zeros, which vanish against a non-zero third term,
// and `+ 0.0` reproduces the sign of a zero result (x + (+0) == +0 for x == -0).
TOAST_HD void quat_rotate_z(const double * q, double * out) {
    const double xw = q[3] * q[0], yw = q[3] * q[1];
    const double x2 = -q[0] * q[0], xz = q[0] * q[2];
    const double y2 = -q[1] * q[1], yz = q[1] * q[2];
    out[0] = 2 * (yw + xz) + 0.0;
    out[1] = 2 * (yz - xw) + 0.0;
    out[2] = 2 * (x2 + y2) + 1.0;
}

// r = p * q (scalar last), reference term order.
TOAST_HD void quat_mult(const double * p, const double * q, double * r) {
    r[0] = p[0] * q[3] + p[1] * q[2] - p[2] * q[1] + p[3] * q[0];
    r[1] = -p[0] * q[2] + p[1] * q[3] + p[2] * q[0] + p[3] * q[1];
    r[2] = p[0] * q[1] - p[1] * q[0] + p[2] * q[3] + p[3] * q[2];
    r[3] = -p[0] * q[0] - p[1] * q[1] - p[2] * q[2] + p[3] * q[3];
}

// ------------------------------------------------------------------ HEALPix
struct ZPhi {
    double phi;
    double z;
    double rtz;
    int region;  // sign(z) * (1 if |z| <= 2/3 else 2)
};

// everything of hpix_vec2zphi that does not need phi
TOAST_HD ZPhi zphi_head(const double * v) {
    ZPhi o;
    o.z = v[2];
    const double za = f_abs(o.z);
    const int s = (o.z > 0.0) ? 1 : -1;
    o.region = (za <= TOAST_TWOTHIRDS) ? s : s + s;
    // the reference always evaluates sqrt(3 (1 - |z|)) but reads it only in the polar caps
    o.rtz = (o.region == 1 || o.region == -1) ? 0.0 : f_sqrt(3.0 * (1.0 - za));
    o.phi = 0.0;
    return o;
}

TOAST_HD ZPhi zphi_from_vec(const double * v, const double * atan_tab) {
    ZPhi o = zphi_head(v);
    o.phi = atan2_dd(v[1], v[0], atan_tab);
    return o;
}

TOAST_HD double phi_to_tt(double phi) {
    const double tol = 10.0 * 2.220446049250313e-16;
    const double period = TOAST_TWOPI;
    // phi / period, correctly rounded without a divide (Markstein): y = RN(1/period),
    // q0 = RN(phi y), r = phi - period q0 (exact by fma), q = RN(q0 + r y).  Exact IEEE quotient
    // for every phi because the significand of period is not all ones; checked against the
    // hardware / x86 division in tests (test_devmath_host.py, test_gpu_math.py).
    const double inv = 1.0 / TOAST_TWOPI;
    const double q0 = phi * inv;
    // (q0 itself when it is a zero: keeps the sign of -0 / period)
    const double div = (q0 == 0.0) ? q0 : f_fma(f_fma(-q0, period, phi), inv, q0);
    double pm = period * (div - (double)((int64_t)div));
    if ((pm < tol) && (pm > -tol)) pm = 0.0;
    return (pm >= 0.0) ? pm * TOAST_2_OVER_PI : pm * TOAST_2_OVER_PI + 4.0;
}

// bit k of the low 32 bits of v -> bit 2k  (== reference utab composition, :20-27, :78-85)
TOAST_HD uint64_t spread_bits(uint64_t v) {
    v &= 0xffffffffull;
    v = (v | (v << 16)) & 0x0000ffff0000ffffull;
    v = (v | (v << 8)) & 0x00ff00ff00ff00ffull;
    v = (v | (v << 4)) & 0x0f0f0f0f0f0f0f0full;
    v = (v | (v << 2)) & 0x3333333333333333ull;
    v = (v | (v << 1)) & 0x5555555555555555ull;
    return v;
}

// 16-bit variant for nside <= 8192 (x, y < 2^13): all-32-bit integer path
TOAST_HD uint32_t spread_bits16(uint32_t v) {
    v &= 0xffffu;
    v = (v | (v << 8)) & 0x00ff00ffu;
    v = (v | (v << 4)) & 0x0f0f0f0fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}

template <typename I>
TOAST_HD I morton_interleave(I x, I y);
template <>
TOAST_HD int64_t morton_interleave<int64_t>(int64_t x, int64_t y) {
    return (int64_t)(spread_bits((uint64_t)x) | (spread_bits((uint64_t)y) << 1));
}
template <>
TOAST_HD int32_t morton_interleave<int32_t>(int32_t x, int32_t y) {
    return (int32_t)(spread_bits16((uint32_t)x) | (spread_bits16((uint32_t)y) << 1));
}

// Integer type I = int32_t is exact for nside <= 8192 (12 nside^2 < 2^31, every intermediate
// below fits); int64_t is the general path.  f64 -> integer conversions truncate like the
// reference's (int64_t) casts.
template <typename I>
TOAST_HD I zphi_to_nest_t(I nside, int factor, const ZPhi & a) {
    const double tt = phi_to_tt(a.phi);
    const double dn = (double)nside;
    const I nm1 = nside - 1;
    I x, y, face;
    if (a.region == 1 || a.region == -1) {
        const double t1 = 0.5 * dn + dn * tt;
        const double t2 = (0.75 * dn) * a.z;
        const I jp = (I)(t1 - t2);
        const I jm = (I)(t1 + t2);
        const I ifp = jp >> factor;
        const I ifm = jm >> factor;
        if (ifp == ifm) {
            face = (ifp == 4) ? (I)4 : ifp + 4;
        } else if (ifp < ifm) {
            face = ifp;
        } else {
            face = ifm + 8;
        }
        x = jm & nm1;
        y = nm1 - (jp & nm1);
    } else {
        const I ntt = (I)tt;
        const double tp = tt - (double)ntt;
        const double t1 = dn * a.rtz;
        I jp = (I)(tp * t1);
        I jm = (I)((1.0 - tp) * t1);
        if (jp >= nside) jp = nm1;
        if (jm >= nside) jm = nm1;
        if (a.z >= 0) {
            face = ntt;
            x = nm1 - jm;
            y = nm1 - jp;
        } else {
            face = ntt + 8;
            x = jp;
            y = jm;
        }
    }
    return morton_interleave<I>(x, y) + (face << (2 * factor));
}

template <typename I>
TOAST_HD I zphi_to_ring_t(I nside, int /*factor*/, const ZPhi & a) {
    const double tt = phi_to_tt(a.phi);
    const double dn = (double)nside;
    const I n4 = 4 * nside;
    if (a.region == 1 || a.region == -1) {
        const I ncap = 2 * (nside * nside - nside);
        const double t1 = 0.5 * dn + dn * tt;
        const double t2 = (0.75 * dn) * a.z;
        const I jp = (I)(t1 - t2);
        const I jm = (I)(t1 + t2);
        const I ir = (nside + 1) + jp - jm;
        const I kshift = 1 - (ir & 1);
        I ip = (jp + jm - nside + kshift + 1) >> 1;
        // ip % (4 nside) with C truncation semantics; 4 nside is a power of two
        ip = (ip >= 0) ? (ip & (n4 - 1)) : -((-ip) & (n4 - 1));
        return ncap + ((ir - 1) * n4 + ip);
    }
    const double tp = tt - f_floor(tt);
    const double t1 = dn * a.rtz;
    const I jp = (I)(tp * t1);
    const I jm = (I)((1.0 - tp) * t1);
    const I ir = jp + jm + 1;
    I ip = (I)(tt * (double)ir);
    // longpart = ip / (4 ir); tt < 4 (+rounding) so the quotient is 0 or 1 (ip >= 0)
    const I four_ir = 4 * ir;
    if (ip >= four_ir) ip -= (ip >= 2 * four_ir) ? ip / four_ir : 1;
    const I npix = 12 * nside * nside;
    return (a.region > 0) ? (2 * ir * (ir - 1) + ip) : (npix - 2 * ir * (ir + 1) + ip);
}

TOAST_HD int64_t zphi_to_nest(int64_t nside, int factor, const ZPhi & a) {
    if (nside <= 8192) return (int64_t)zphi_to_nest_t<int32_t>((int32_t)nside, factor, a);
    return zphi_to_nest_t<int64_t>(nside, factor, a);
}

TOAST_HD int64_t zphi_to_ring(int64_t nside, int factor, const ZPhi & a) {
    if (nside <= 8192) return (int64_t)zphi_to_ring_t<int32_t>((int32_t)nside, factor, a);
    return zphi_to_ring_t<int64_t>(nside, factor, a);
}

// ------------------------------------------------------------------ Ziv-style fast path
// The pixel index is a piecewise-constant function of phi.  pixel_checked evaluates the SAME
// arithmetic as phi_to_tt + zphi_to_nest_t / zphi_to_ring_t on an approximate phi (atan2_fast,
// |phi~ - phi| <= delta) and reports `safe` only if no decision of that arithmetic can come out
// differently for ANY phi' with |phi' - phi~| <= delta: every operand of a float -> integer
// truncation must be farther than `bound` from an integer and the zero-snap / sign test of
// phi_to_tt farther than delta from its thresholds.  Error propagation (every operation is
// monotone; dn = nside, a power of two, so dn * tt is exact):
//   |d tt| <= (2 / pi) delta + 3 ulp(4)            <= delta          (delta = 2^-43 >> 1e-15)
//   equatorial  t1 -+ t2:   dn |d tt| + 2 ulp(5 dn)  <= 1.03 dn delta
//   polar  tp t1, (1 - tp) t1 (t1 = dn rtz <= dn):    <= 1.01 dn delta;   tt itself: delta
//   ring polar  tt ir (ir < 2 dn):                    <= 2.01 dn delta
// bound = 4 dn delta (and 4 delta for tt) covers all of them with a factor >= 2 to spare.  NaNs
// fail every comparison, so non-finite input is never "safe".  When safe, the returned index is
// by construction the one the double-double path produces; otherwise the caller recomputes with
// atan2_dd.  At nside 1024 a sample is unsafe with probability ~3e-9.
template <typename I, bool NEST>
TOAST_HD I pixel_checked(I nside, int factor, const ZPhi & a, double delta, bool & safe) {
    const double dn = (double)nside;
    const double bound = 4.0 * dn * delta;
    const double hi = 1.0 - bound;
    // phi_to_tt, with the distance of pm to the snap interval checked
    const double tol = 10.0 * 2.220446049250313e-16;
    const double period = TOAST_TWOPI;
    const double inv = 1.0 / TOAST_TWOPI;
    const double q0 = a.phi * inv;
    const double div = (q0 == 0.0) ? q0 : f_fma(f_fma(-q0, period, a.phi), inv, q0);
    double pm = period * (div - (double)((int64_t)div));
    bool ok = f_abs(pm) > tol + 2.0 * delta;
    if ((pm < tol) && (pm > -tol)) pm = 0.0;
    const double tt = (pm >= 0.0) ? pm * TOAST_2_OVER_PI : pm * TOAST_2_OVER_PI + 4.0;
    const I nm1 = nside - 1;
    I result;
    if (a.region == 1 || a.region == -1) {
        const double t1 = 0.5 * dn + dn * tt;
        const double t2 = (0.75 * dn) * a.z;
        const double wp = t1 - t2, wm = t1 + t2;
        const I jp = (I)wp;
        const I jm = (I)wm;
        const double fp = wp - (double)jp, fm = wm - (double)jm;
        ok = ok && (fp > bound) && (fp < hi) && (fm > bound) && (fm < hi);
        if (NEST) {
            const I ifp = jp >> factor;
            const I ifm = jm >> factor;
            I face;
            if (ifp == ifm) {
                face = (ifp == 4) ? (I)4 : ifp + 4;
            } else if (ifp < ifm) {
                face = ifp;
            } else {
                face = ifm + 8;
            }
            const I x = jm & nm1;
            const I y = nm1 - (jp & nm1);
            result = morton_interleave<I>(x, y) + (face << (2 * factor));
        } else {
            const I n4 = 4 * nside;
            const I ncap = 2 * (nside * nside - nside);
            const I ir = (nside + 1) + jp - jm;
            const I kshift = 1 - (ir & 1);
            I ip = (jp + jm - nside + kshift + 1) >> 1;
            ip = (ip >= 0) ? (ip & (n4 - 1)) : -((-ip) & (n4 - 1));
            result = ncap + ((ir - 1) * n4 + ip);
        }
    } else {
        const I ntt = (I)tt;
        const double tp = tt - (double)ntt;      // == tt - floor(tt) for tt >= 0
        const double tb = 4.0 * delta;
        ok = ok && (tp > tb) && (tp < 1.0 - tb);
        const double t1 = dn * a.rtz;
        const double wp = tp * t1, wm = (1.0 - tp) * t1;
        I jp = (I)wp;
        I jm = (I)wm;
        const double fp = wp - (double)jp, fm = wm - (double)jm;
        ok = ok && (fp > bound) && (fp < hi) && (fm > bound) && (fm < hi);
        if (NEST) {
            if (jp >= nside) jp = nm1;
            if (jm >= nside) jm = nm1;
            I x, y, face;
            if (a.z >= 0) {
                face = ntt;
                x = nm1 - jm;
                y = nm1 - jp;
            } else {
                face = ntt + 8;
                x = jp;
                y = jm;
            }
            result = morton_interleave<I>(x, y) + (face << (2 * factor));
        } else {
            const I ir = jp + jm + 1;
            const double wi = tt * (double)ir;
            I ip = (I)wi;
            const double fi = wi - (double)ip;
            ok = ok && (fi > bound) && (fi < hi);
            const I four_ir = 4 * ir;
            if (ip >= four_ir) ip -= (ip >= 2 * four_ir) ? ip / four_ir : 1;
            const I npix = 12 * nside * nside;
            result = (a.region > 0) ? (2 * ir * (ir - 1) + ip) : (npix - 2 * ir * (ir + 1) + ip);
        }
    }
    safe = ok;
    return result;
}

// The double-double path, kept out of line: it runs for ~1e-9 of the samples and must not set the
// register allocation of the kernels that inline vec_to_pixel.
#if defined(TOAST_PIXEL_SLOW_INLINE)   /* experiment switch, profiles/r02_c */
# define TOAST_NOINLINE TOAST_HD
#elif defined(__HIPCC__)
# define TOAST_NOINLINE __host__ __device__ __attribute__((noinline))
#else
# define TOAST_NOINLINE __attribute__((noinline))
#endif
// (the direction goes in by value: a pointer argument would force the caller's vector into scratch
// memory for EVERY sample -- 16 B of extra HBM writes per sample, measured with WRITE_SIZE)
template <bool NEST>
TOAST_NOINLINE int64_t vec_to_pixel_slow(double vx, double vy, double vz, int64_t nside, int factor,
                                         const double * atan_tab) {
    const double v[3] = {vx, vy, vz};
    const ZPhi a = zphi_from_vec(v, atan_tab);
    return NEST ? zphi_to_nest(nside, factor, a) : zphi_to_ring(nside, factor, a);
}

// Pixel of a direction vector: fast path first, double-double atan2 only where the fast result
// is not provably the same.  Bit-identical to zphi_from_vec + zphi_to_nest / zphi_to_ring.
template <bool NEST>
TOAST_HD int64_t vec_to_pixel(const double * v, int64_t nside, int factor, const double * atan_tab) {
    ZPhi a = zphi_head(v);
    a.phi = atan2_fast(v[1], v[0]);
    bool safe;
    int64_t pix;
    if (nside <= 8192) {
        pix = (int64_t)pixel_checked<int32_t, NEST>((int32_t)nside, factor, a, TOAST_ATAN2_FAST_ERR, safe);
    } else {
        pix = pixel_checked<int64_t, NEST>(nside, factor, a, TOAST_ATAN2_FAST_ERR, safe);
    }
    if (__builtin_expect(!safe, 0)) pix = vec_to_pixel_slow<NEST>(v[0], v[1], v[2], nside, factor, atan_tab);
    return pix;
}

// Pixels of TWO directions that are expected to coincide: the two orthogonally polarised detectors of one
// focalplane pixel look along the same line of sight, and their direction vectors differ only by the rounding of two
// different quaternion products (measured: <= 1.6e-15 per component).  When v1 lies within kPairDirTol = 2^-48 of v0 the
// checked fast path of v0 already decides v1's pixel: pixel_checked() reports `safe` only if every decision of the
// pixel arithmetic is the same for ALL phi within delta = 2^-43 of its approximate phi, with a margin of
// bound = 4 nside delta on every truncation operand.  For v1 (reference phi_1 = libm atan2(v1) within 1 ulp of exact):
//   |phi_1 - phi~| <= 2^-46 (atan2_fast) + sqrt(2) 2^-48 / r + 1 ulp  <  2^-43      for r = |(x, y)| >= 2^-3
//   equatorial  t2 = 0.75 nside z:          |d t2| <= 0.75 nside 2^-48          = 0.02 nside delta
//   polar  t1 = nside sqrt(3 (1 - |z|)):    |d t1| <= 1.5 nside 2^-48 / rtz     <= 0.31 nside delta  for 1 - |z| >= 2^-7
// all inside the factor the bound holds in reserve (hpix_math.hpp, "Ziv-style fast path": 1.03 of 4 used), every
// operation being monotone in z as it is in phi; the region test |z| <= 2/3 is guarded separately.  Outside these
// conditions (within 7 degrees of a pole, NaNs, unrelated detectors, an unsafe v0) v1 goes through vec_to_pixel on
// its own.  Result: bit-identical to two vec_to_pixel calls (tests/devmath_host.cpp devmath_sweep_pair: 3e9 random
// and pixel-edge adversarial pairs, 0 mismatches).
#define TOAST_PAIR_DIR_TOL 3.552713678800501e-15      /* 2^-48 */
#define TOAST_PAIR_R2_MIN 0.015625                    /* 2^-6: r >= 2^-3 */
#define TOAST_PAIR_ZA_MAX 0.9921875                   /* 1 - 2^-7 */
#define TOAST_PAIR_REGION_GUARD 7.105427357601002e-15 /* 2^-47 */
template <bool NEST>
TOAST_HD void vec_to_pixel_pair(const double * v0, const double * v1, int64_t nside, int factor,
                                const double * atan_tab, int64_t & pix0, int64_t & pix1) {
    ZPhi a = zphi_head(v0);
    a.phi = atan2_fast(v0[1], v0[0]);
    bool safe;
    int64_t pix;
    if (nside <= 8192) {
        pix = (int64_t)pixel_checked<int32_t, NEST>((int32_t)nside, factor, a, TOAST_ATAN2_FAST_ERR, safe);
    } else {
        pix = pixel_checked<int64_t, NEST>(nside, factor, a, TOAST_ATAN2_FAST_ERR, safe);
    }
    const double za = f_abs(v0[2]);
    const double r2 = v0[0] * v0[0] + v0[1] * v0[1];
    const bool close = (f_abs(v1[0] - v0[0]) <= TOAST_PAIR_DIR_TOL) && (f_abs(v1[1] - v0[1]) <= TOAST_PAIR_DIR_TOL) &&
                       (f_abs(v1[2] - v0[2]) <= TOAST_PAIR_DIR_TOL) && (r2 >= TOAST_PAIR_R2_MIN) &&
                       (za <= TOAST_PAIR_ZA_MAX) && (f_abs(za - TOAST_TWOTHIRDS) > TOAST_PAIR_REGION_GUARD);
    if (__builtin_expect(!safe, 0)) pix = vec_to_pixel_slow<NEST>(v0[0], v0[1], v0[2], nside, factor, atan_tab);
    pix0 = pix;
    if (safe && close) {
        pix1 = pix;
    } else {
        pix1 = vec_to_pixel<NEST>(v1, nside, factor, atan_tab);
    }
}

// quat_rotate(q, (1,0,0)) for finite q (same reasoning as quat_rotate_z).
TOAST_HD void quat_rotate_x(const double * q, double * out) {
    const double yw = q[3] * q[1], zw = q[3] * q[2];
    const double xy = q[0] * q[1], xz = q[0] * q[2];
    const double y2 = -q[1] * q[1], z2 = -q[2] * q[2];
    out[0] = 2 * (y2 + z2) + 1.0;
    out[1] = 2 * (zw + xy) + 0.0;
    out[2] = 2 * (xz - yw) + 0.0;
}

// cos(2 alpha), sin(2 alpha) of the detector polarisation angle alpha of the reference
// (ops_stokes_weights.cpp:50-75: alpha = atan2(alpha_y, alpha_x) with the meridian vector
// built from cos / sin of atan2(vd1, vd0)), evaluated algebraically:
//   cos(atan2(y, x)) = x / r,  sin(atan2(y, x)) = y / r,
//   cos 2a = (ax^2 - ay^2) / (ax^2 + ay^2),  sin 2a = 2 ax ay / (ax^2 + ay^2)
// i.e. one reciprocal instead of two atan2 and two sincos (see the scaling argument below).  Weights are a
// tolerance-class output (reference tests: assert_allclose), agreement ~1e-15 absolute.
// `reference_nan`: reproduce the one place where the reference formulation is NOT finite -- within
// rounding of a pole vd2 * vd2 can exceed 1, its -sqrt(1 - vd2 * vd2) is NaN and so are alpha and
// the Q / U weights (ops_stokes_weights.cpp:66-75).  The library's default since round 4 (results identical to the
// reference's on the same inputs); TOAST_HIP_STOKES_REFERENCE_NAN=0 / toast_hip_set_stokes_reference_nan(0) selects
// finite weights of modulus eta * cal there instead.
TOAST_HD void stokes_cs2alpha(const double * q, double & c2a, double & s2a, bool reference_nan = false) {
    double vd[3], vo[3];
    quat_rotate_z(q, vd);
    quat_rotate_x(q, vo);
    if (reference_nan && (1.0 - vd[2] * vd[2] < 0.0)) {
        c2a = s2a = __builtin_nan("");
        return;
    }
    const double r2 = vd[0] * vd[0] + vd[1] * vd[1];   // sin^2(theta)
    double ax, ay;
    if (r2 > 0.0) {
        // alpha = atan2(ay, ax) only enters through cos 2a / sin 2a, which do not change when
        // (ax, ay) is scaled by s = sin(theta) > 0.  With s * meridian = (vd2 vd0, vd2 vd1, -s^2)
        // and s * (vd x meridian) = (-vd1, vd0, 0) for the unit vector vd, the scaled components
        // need neither the square roots nor the 1/s of the reference formulation:
        ax = vd[2] * (vd[0] * vo[0] + vd[1] * vo[1]) - r2 * vo[2];
        ay = vd[0] * vo[1] - vd[1] * vo[0];
    } else {
        // on the pole atan2(+-0, +-0) is 0 or pi: the reference's expressions, literally
        const double cxy = __builtin_signbit(vd[0]) ? -1.0 : 1.0;
        const double sxy = 0.0;
        const double vm_x = vd[2] * cxy;
        const double vm_y = vd[2] * sxy;
        const double vm_z = -f_sqrt(1.0 - vd[2] * vd[2]);
        ay = (vd[0] * (vm_y * vo[2] - vm_z * vo[1]) - vd[1] * (vm_x * vo[2] - vm_z * vo[0]) +
              vd[2] * (vm_x * vo[1] - vm_y * vo[0]));
        ax = (vm_x * vo[0] + vm_y * vo[1] + vm_z * vo[2]);
    }
    const double n2 = ax * ax + ay * ay;
    c2a = 1.0;  // alpha = atan2(0, 0) = 0
    s2a = 0.0;
    if (n2 > 0.0) {
        const double ninv = recip_newton(n2);
        c2a = (ax * ax - ay * ay) * ninv;
        s2a = (2.0 * ax * ay) * ninv;
    }
}

// cos / sin of the HWP modulation angle  beta = 2 (2 (gamma - hwp)) = 4 gamma - 4 hwp
// (ops_stokes_weights.cpp:96-99) by the angle-addition formulas, from cos / sin of 4 gamma
// (per detector) and of 4 hwp (per time sample, shared by all detectors): no transcendental
// per det-sample.  4 x is exact in binary floating point; agreement with sincos(beta) ~3e-16.
TOAST_HD void hwp_rotation(double c4g, double s4g, double c4h, double s4h, double & cb, double & sb) {
    cb = c4g * c4h + s4g * s4h;
    sb = s4g * c4h - c4g * s4h;
}

// ------------------------------------------------------------------ RING <-> NEST
// bit 2k of v -> bit k (the inverse of spread_bits; == the reference's ctab composition hpix_pix2xy, :87-102)
TOAST_HD uint64_t compress_bits(uint64_t v) {
    v &= 0x5555555555555555ull;
    v = (v | (v >> 1)) & 0x3333333333333333ull;
    v = (v | (v >> 2)) & 0x0f0f0f0f0f0f0f0full;
    v = (v | (v >> 4)) & 0x00ff00ff00ff00ffull;
    v = (v | (v >> 8)) & 0x0000ffff0000ffffull;
    v = (v | (v >> 16)) & 0x00000000ffffffffull;
    return v;
}

// ops_pixels_healpix.cpp:383-473 (pixel numbers only: integer arithmetic and two exactly rounded square roots)
TOAST_HD int64_t ring_to_nest(int64_t nside, int factor, int64_t ringpix) {
    const int64_t npix = 12 * nside * nside;
    const int64_t ncap = 2 * (nside * nside - nside);
    const int64_t jr_tab[12] = {2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4};
    const int64_t jp_tab[12] = {1, 3, 5, 7, 0, 2, 4, 6, 1, 3, 5, 7};
    int64_t fc, nr, kshift, iring, iphi;
    if (ringpix < ncap) {
        iring = (int64_t)(0.5 * (1.0 + f_sqrt((double)(1 + 2 * ringpix))));
        iphi = (ringpix + 1) - 2 * iring * (iring - 1);
        kshift = 0;
        nr = iring;
        fc = 0;
        int64_t tmp = iphi - 1;
        if (tmp >= 2 * iring) {
            fc = 2;
            tmp -= 2 * iring;
        }
        if (tmp >= iring) ++fc;
    } else if (ringpix < npix - ncap) {
        const int64_t ip = ringpix - ncap;
        iring = (ip >> (factor + 2)) + nside;
        iphi = (ip & (4 * nside - 1)) + 1;
        kshift = (iring + nside) & 1;
        nr = nside;
        const int64_t ire = iring - nside + 1;
        const int64_t irm = 2 * nside + 2 - ire;
        const int64_t ifm = (iphi - (ire / 2) + nside - 1) >> factor;
        const int64_t ifp = (iphi - (irm / 2) + nside - 1) >> factor;
        if (ifp == ifm) {
            fc = (ifp == 4) ? 4 : ifp + 4;
        } else if (ifp < ifm) {
            fc = ifp;
        } else {
            fc = ifm + 8;
        }
    } else {
        const int64_t ip = npix - ringpix;
        iring = (int64_t)(0.5 * (1.0 + f_sqrt((double)(2 * ip - 1))));
        iphi = 4 * iring + 1 - (ip - 2 * iring * (iring - 1));
        kshift = 0;
        nr = iring;
        iring = 4 * nside - iring;
        fc = 8;
        int64_t tmp = iphi - 1;
        if (tmp >= 2 * nr) {
            fc = 10;
            tmp -= 2 * nr;
        }
        if (tmp >= nr) ++fc;
    }
    const int64_t irt = iring - jr_tab[fc] * nside + 1;
    int64_t ipt = 2 * iphi - jp_tab[fc] * nr - kshift - 1;
    if (ipt >= 2 * nside) ipt -= 8 * nside;
    const int64_t x = (ipt - irt) >> 1;
    const int64_t y = (-(ipt + irt)) >> 1;
    return morton_interleave<int64_t>(x, y) + (fc << (2 * factor));
}

// ops_pixels_healpix.cpp:475-520
TOAST_HD int64_t nest_to_ring(int64_t nside, int factor, int64_t nestpix) {
    const int64_t npix = 12 * nside * nside;
    const int64_t ncap = 2 * (nside * nside - nside);
    const int64_t jr_tab[12] = {2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4};
    const int64_t jp_tab[12] = {1, 3, 5, 7, 0, 2, 4, 6, 1, 3, 5, 7};
    const int64_t fc = nestpix >> (2 * factor);
    const uint64_t in_face = (uint64_t)(nestpix & (nside * nside - 1));
    const int64_t x = (int64_t)compress_bits(in_face);
    const int64_t y = (int64_t)compress_bits(in_face >> 1);
    const int64_t jr = jr_tab[fc] * nside - x - y - 1;
    int64_t nr, n_before, kshift;
    if (jr < nside) {
        nr = jr;
        n_before = 2 * nr * (nr - 1);
        kshift = 0;
    } else if (jr > 3 * nside) {
        nr = 4 * nside - jr;
        n_before = npix - 2 * (nr + 1) * nr;
        kshift = 0;
    } else {
        nr = nside;
        n_before = ncap + (jr - nside) * 4 * nside;
        kshift = (jr - nside) & 1;
    }
    int64_t jp = (jp_tab[fc] * nr + x - y + 1 + kshift) / 2;
    if (jp > 4 * nside) {
        jp -= 4 * nside;
    } else if (jp < 1) {
        jp += 4 * nside;
    }
    return n_before + jp - 1;
}

// ------------------------------------------------------------------ division by a run-time constant
// q = n / d for 0 <= n < 2^63 with one 64x64->high multiply.  mul = floor(2^(63+s)/d) + 1,
// s = ceil(log2 d): the error term n*e/(d 2^(63+s)) < 1/d, so the floor is exact.
struct FastDiv {
    uint64_t mul;
    int32_t shift;  // s - 1, or -1 for d == 1
    int64_t d;
};

inline FastDiv make_fastdiv(int64_t d) {
    FastDiv f;
    f.d = d;
    if (d <= 1) {
        f.mul = 0;
        f.shift = -1;
        return f;
    }
    int s = 0;
    while ((int64_t(1) << s) < d) ++s;
    const unsigned __int128 num = (unsigned __int128)1 << (63 + s);
    f.mul = (uint64_t)(num / (unsigned __int128)d) + 1;
    f.shift = s - 1;
    return f;
}

TOAST_HD uint64_t mulhi_u64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

TOAST_HD int64_t fastdiv(int64_t n, const FastDiv & f) {
    return (f.shift < 0) ? n : (int64_t)(mulhi_u64((uint64_t)n, f.mul) >> f.shift);
}

}  // namespace toast_hip
