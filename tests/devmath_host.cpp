// Host build of toast_amd/csrc/hpix_math.hpp (the device pointing math) so that its
// bit-parity with the oracle can be measured on a CPU: tests/test_devmath_host.py.
// Test harness only -- the product never runs this code on the host.
#include <cmath>
#include <cstdint>
#include "../toast_amd/csrc/hpix_math.hpp"

using namespace toast_hip;

extern "C" {

void devmath_atan2(int64_t n, const double * y, const double * x, double * out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) out[i] = atan2_dd(y[i], x[i], kAtanTab);
}

void devmath_pixels(int64_t n, const double * quats, int64_t nside, int nest, int64_t * pix) {
    int factor = 0;
    while (nside != (int64_t(1) << factor)) ++factor;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double dir[3];
        quat_rotate_z(quats + 4 * i, dir);
        const ZPhi a = zphi_from_vec(dir, kAtanTab);
        pix[i] = nest ? zphi_to_nest(nside, factor, a) : zphi_to_ring(nside, factor, a);
    }
}

void devmath_vec2pix(int64_t n, const double * vec, int64_t nside, int nest, int64_t * pix) {
    int factor = 0;
    while (nside != (int64_t(1) << factor)) ++factor;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const ZPhi a = zphi_from_vec(vec + 3 * i, kAtanTab);
        pix[i] = nest ? zphi_to_nest(nside, factor, a) : zphi_to_ring(nside, factor, a);
    }
}

// the shipped pixel path (Ziv-style: plain-double atan2, double-double only when not provably equal)
void devmath_pixels_fast(int64_t n, const double * quats, int64_t nside, int nest, int64_t * pix) {
    int factor = 0;
    while (nside != (int64_t(1) << factor)) ++factor;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double dir[3];
        quat_rotate_z(quats + 4 * i, dir);
        pix[i] = nest ? vec_to_pixel<true>(dir, nside, factor, kAtanTab) : vec_to_pixel<false>(dir, nside, factor, kAtanTab);
    }
}

void devmath_vec2pix_fast(int64_t n, const double * vec, int64_t nside, int nest, int64_t * pix) {
    int factor = 0;
    while (nside != (int64_t(1) << factor)) ++factor;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        pix[i] = nest ? vec_to_pixel<true>(vec + 3 * i, nside, factor, kAtanTab)
                      : vec_to_pixel<false>(vec + 3 * i, nside, factor, kAtanTab);
    }
}

void devmath_atan2_fast(int64_t n, const double * y, const double * x, double * out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) out[i] = atan2_fast(y[i], x[i]);
}

// Self-contained sweep: n pseudo-random unit vectors (splitmix64 from `seed`; every 8th one pushed
// onto a decision boundary |z| = 2/3, a face meridian or a pole within a few ulp), pixel by the
// fast path vs the double-double path.  Returns the number of mismatches; *n_fallback counts the
// samples the fast path handed to the slow one (*n_fallback_random: among the 7/8 of purely
// random directions), *max_err the largest |atan2_fast - atan2_dd|.
int64_t devmath_sweep(int64_t n, uint64_t seed, int64_t nside, int nest, int64_t * n_fallback,
                      int64_t * n_fallback_random, double * max_err) {
    int factor = 0;
    while (nside != (int64_t(1) << factor)) ++factor;
    int64_t bad = 0, fb = 0, fbr = 0;
    double worst = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : bad, fb, fbr) reduction(max : worst)
    for (int64_t i = 0; i < n; ++i) {
        uint64_t st = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1);
        auto next = [&st]() {
            uint64_t z = (st += 0x9E3779B97F4A7C15ull);
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            return z ^ (z >> 31);
        };
        auto unif = [&next]() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); };
        double z = 2.0 * unif() - 1.0;
        double phi = TOAST_TWOPI * unif() - 3.14159265358979323846;
        const uint64_t kind = next() & 63;
        if (kind == 0) z = (next() & 1) ? TOAST_TWOTHIRDS : -TOAST_TWOTHIRDS;
        if (kind == 1) phi = 1.5707963267948966 * (double)((int64_t)(next() % 5) - 2);
        if (kind == 2) z = (next() & 1) ? 1.0 : -1.0;
        if (kind == 3) phi = 0.0;
        if (kind < 8) {   // nudge by up to +-4 ulp
            const int64_t k = (int64_t)(next() % 9) - 4;
            union { double d; int64_t i; } w;
            w.d = (kind & 1) ? phi : z;
            w.i += k;
            if (kind & 1) phi = w.d; else z = w.d;
        }
        const double rxy = f_sqrt((1.0 - z) * (1.0 + z) > 0 ? (1.0 - z) * (1.0 + z) : 0.0);
        const double v[3] = {rxy * std::cos(phi), rxy * std::sin(phi), z};
        ZPhi a = zphi_from_vec(v, kAtanTab);
        const int64_t want = nest ? zphi_to_nest(nside, factor, a) : zphi_to_ring(nside, factor, a);
        const int64_t got = nest ? vec_to_pixel<true>(v, nside, factor, kAtanTab) : vec_to_pixel<false>(v, nside, factor, kAtanTab);
        bad += (got != want);
        ZPhi b = zphi_head(v);
        b.phi = atan2_fast(v[1], v[0]);
        bool safe;
        if (nside <= 8192) {
            (void)(nest ? pixel_checked<int32_t, true>((int32_t)nside, factor, b, TOAST_ATAN2_FAST_ERR, safe)
                        : pixel_checked<int32_t, false>((int32_t)nside, factor, b, TOAST_ATAN2_FAST_ERR, safe));
        } else {
            (void)(nest ? pixel_checked<int64_t, true>(nside, factor, b, TOAST_ATAN2_FAST_ERR, safe)
                        : pixel_checked<int64_t, false>(nside, factor, b, TOAST_ATAN2_FAST_ERR, safe));
        }
        fb += !safe;
        fbr += (!safe && kind >= 8);
        const double e = f_abs(b.phi - a.phi);
        if (e == e && e > worst) worst = e;
    }
    *n_fallback = fb;
    *n_fallback_random = fbr;
    *max_err = worst;
    return bad;
}

// Detector pairs (vec_to_pixel_pair): direction v0 as in devmath_sweep plus a family sitting on PIXEL EDGES (an
// equatorial jp / jm boundary, a polar-cap jp or jm boundary or an in-ring index boundary of the RING scheme,
// reconstructed from integer targets, then nudged by a few ulp), partner v1 = v0 + per-component noise of amplitude 2^-53 .. 2^-47 (the last one beyond the sharing
// tolerance).  Both pixels must equal the double-double path evaluated on each direction separately.  Returns the
// number of mismatches; *n_shared = pairs whose second pixel was taken from the first.
int64_t devmath_sweep_pair(int64_t n, uint64_t seed, int64_t nside, int nest, int64_t * n_shared, int64_t * n_differ) {
    int factor = 0;
    while (nside != (int64_t(1) << factor)) ++factor;
    int64_t bad = 0, shared = 0, differ = 0;
    const double dn = (double)nside;
#pragma omp parallel for schedule(static) reduction(+ : bad, shared, differ)
    for (int64_t i = 0; i < n; ++i) {
        uint64_t st = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1);
        auto next = [&st]() {
            uint64_t z = (st += 0x9E3779B97F4A7C15ull);
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            return z ^ (z >> 31);
        };
        auto unif = [&next]() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); };
        double z = 2.0 * unif() - 1.0;
        double phi = TOAST_TWOPI * unif() - 3.14159265358979323846;
        const uint64_t kind = next() & 15;
        if (kind == 0) z = (next() & 1) ? TOAST_TWOTHIRDS : -TOAST_TWOTHIRDS;
        if (kind == 1) phi = 1.5707963267948966 * (double)((int64_t)(next() % 5) - 2);
        if (kind == 2) z = (next() & 1) ? 1.0 : -1.0;
        if (kind == 3 || kind == 4) {
            // equatorial pixel edge: t1 -+ t2 = integer J with t1 = dn / 2 + dn tt, t2 = 0.75 dn z
            z = (2.0 * unif() - 1.0) * TOAST_TWOTHIRDS;
            const double t2 = 0.75 * dn * z;
            const double J = std::floor(unif() * 4.0 * dn);
            double tt = (kind == 3) ? (J + t2 - 0.5 * dn) / dn : (J - t2 - 0.5 * dn) / dn;
            tt -= 4.0 * std::floor(tt / 4.0);
            phi = tt * 1.5707963267948966;
            if (phi > 3.14159265358979323846) phi -= TOAST_TWOPI;
        }
        if (kind == 5) {
            // polar cap: tp * dn * rtz = integer J  (rtz = sqrt(3 (1 - |z|)))
            const double za = TOAST_TWOTHIRDS + unif() * (1.0 - TOAST_TWOTHIRDS) * 0.999;
            z = (next() & 1) ? za : -za;
            const double t1 = dn * f_sqrt(3.0 * (1.0 - za));
            const double J = std::floor(unif() * t1);
            const double tp = (t1 > 0.0) ? J / t1 : 0.0;
            const double tt = (double)(next() & 3) + tp;
            phi = tt * 1.5707963267948966;
            if (phi > 3.14159265358979323846) phi -= TOAST_TWOPI;
        }
        if (kind == 6 || kind == 7) {
            // polar cap, the OTHER truncations: (1 - tp) t1 = integer J (jm edge), or tt (jp + jm + 1) = integer (the
            // ring scheme's in-ring index)
            const double za = TOAST_TWOTHIRDS + unif() * (1.0 - TOAST_TWOTHIRDS) * 0.999;
            z = (next() & 1) ? za : -za;
            const double t1 = dn * f_sqrt(3.0 * (1.0 - za));
            double tp = unif();
            const double ntt = (double)(next() & 3);
            if (kind == 6) {
                const double J = std::floor(unif() * t1);
                tp = (t1 > 0.0) ? 1.0 - J / t1 : 0.0;
                if (tp >= 1.0) tp = 0.0;
            } else {
                const double ir = std::floor(tp * t1) + std::floor((1.0 - tp) * t1) + 1.0;
                const double K = std::floor((ntt + tp) * ir);
                const double tp2 = K / ir - ntt;
                if (tp2 >= 0.0 && tp2 < 1.0) tp = tp2;
            }
            phi = (ntt + tp) * 1.5707963267948966;
            if (phi > 3.14159265358979323846) phi -= TOAST_TWOPI;
        }
        if (kind < 8) {   // nudge by up to +-4 ulp
            const int64_t k = (int64_t)(next() % 9) - 4;
            union { double d; int64_t i; } w;
            const bool on_phi = (kind == 1) || (kind >= 3);
            w.d = on_phi ? phi : z;
            w.i += k;
            if (on_phi) phi = w.d; else z = w.d;
        }
        const double rxy = f_sqrt((1.0 - z) * (1.0 + z) > 0 ? (1.0 - z) * (1.0 + z) : 0.0);
        const double v0[3] = {rxy * std::cos(phi), rxy * std::sin(phi), z};
        static const double amps[4] = {1.1102230246251565e-16, 8.881784197001252e-16, 1.7763568394002505e-15,
                                       7.105427357601002e-15};
        const double amp = amps[next() & 3];
        double v1[3];
        for (int k = 0; k < 3; ++k) v1[k] = v0[k] + amp * (2.0 * unif() - 1.0);
        if ((next() & 63) == 0) v1[next() % 3] = v0[0];      // an unrelated partner now and then
        const ZPhi a0 = zphi_from_vec(v0, kAtanTab);
        const ZPhi a1 = zphi_from_vec(v1, kAtanTab);
        const int64_t want0 = nest ? zphi_to_nest(nside, factor, a0) : zphi_to_ring(nside, factor, a0);
        const int64_t want1 = nest ? zphi_to_nest(nside, factor, a1) : zphi_to_ring(nside, factor, a1);
        int64_t got0, got1;
        if (nest) {
            vec_to_pixel_pair<true>(v0, v1, nside, factor, kAtanTab, got0, got1);
        } else {
            vec_to_pixel_pair<false>(v0, v1, nside, factor, kAtanTab, got0, got1);
        }
        bad += (got0 != want0) + (got1 != want1);
        differ += (want0 != want1);
        // sharing rate: re-derive the decision the function took
        ZPhi b = zphi_head(v0);
        b.phi = atan2_fast(v0[1], v0[0]);
        bool safe;
        if (nside <= 8192) {
            (void)(nest ? pixel_checked<int32_t, true>((int32_t)nside, factor, b, TOAST_ATAN2_FAST_ERR, safe)
                        : pixel_checked<int32_t, false>((int32_t)nside, factor, b, TOAST_ATAN2_FAST_ERR, safe));
        } else {
            (void)(nest ? pixel_checked<int64_t, true>(nside, factor, b, TOAST_ATAN2_FAST_ERR, safe)
                        : pixel_checked<int64_t, false>(nside, factor, b, TOAST_ATAN2_FAST_ERR, safe));
        }
        const double za = f_abs(v0[2]);
        const bool close = (f_abs(v1[0] - v0[0]) <= TOAST_PAIR_DIR_TOL) && (f_abs(v1[1] - v0[1]) <= TOAST_PAIR_DIR_TOL) &&
                           (f_abs(v1[2] - v0[2]) <= TOAST_PAIR_DIR_TOL) &&
                           (v0[0] * v0[0] + v0[1] * v0[1] >= TOAST_PAIR_R2_MIN) && (za <= TOAST_PAIR_ZA_MAX) &&
                           (f_abs(za - TOAST_TWOTHIRDS) > TOAST_PAIR_REGION_GUARD);
        shared += (safe && close);
    }
    *n_shared = shared;
    *n_differ = differ;
    return bad;
}

void devmath_ring2nest(int64_t n, int64_t nside, const int64_t * in, int64_t * out) {
    int factor = 0;
    while (nside != (int64_t(1) << factor)) ++factor;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) out[i] = ring_to_nest(nside, factor, in[i]);
}

void devmath_nest2ring(int64_t n, int64_t nside, const int64_t * in, int64_t * out) {
    int factor = 0;
    while (nside != (int64_t(1) << factor)) ++factor;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) out[i] = nest_to_ring(nside, factor, in[i]);
}

void devmath_stokes(int64_t n, const double * quats, double * c2a, double * s2a) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) stokes_cs2alpha(quats + 4 * i, c2a[i], s2a[i]);
}

void devmath_recip(int64_t n, const double * b, double * out) {
    for (int64_t i = 0; i < n; ++i) out[i] = recip_newton(b[i]);
}

// the correctly rounded phi / (2 pi) used by phi_to_tt, exposed through its effect: returns
// the intermediate quotient
void devmath_div_twopi(int64_t n, const double * phi, double * out) {
    const double period = TOAST_TWOPI;
    const double inv = 1.0 / TOAST_TWOPI;
    for (int64_t i = 0; i < n; ++i) {
        const double q0 = phi[i] * inv;
        out[i] = (q0 == 0.0) ? q0 : f_fma(f_fma(-q0, period, phi[i]), inv, q0);
    }
}

void devmath_fastdiv(int64_t n, const int64_t * num, int64_t d, int64_t * q) {
    const FastDiv f = make_fastdiv(d);
    for (int64_t i = 0; i < n; ++i) q[i] = fastdiv(num[i], f);
}

}
