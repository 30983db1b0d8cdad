// capi_host.cpp -- host-pointer level of the C ABI (include/toast_hip.h): the entry points
// that take exactly what the reference's pybind11 bindings take.  Each one resolves its
// large arrays to device pointers -- through the memory manager when use_accel != 0 (the
// reference's `omgr.device_ptr(...)` lookups, e.g. ops_scan_map.cpp:166-171), or by staging
// temporary device copies otherwise -- and forwards to the *_dev kernel launcher.
// No computation happens on the host.
#include "runtime.hpp"

using namespace toast_hip;

namespace {

struct Call {
    Staging st;
    hipStream_t stream;
    explicit Call(int use_accel)
        : st((Manager::get().require_device(), use_accel != 0), Manager::get().stream()),
          stream(Manager::get().stream()) {}
    void check(int rc) {
        if (rc != TOAST_HIP_OK) throw Error(rc, toast_hip_last_error());
    }
};

// An optional array is "absent" when its length differs from n_samp
// (ops_pixels_healpix.cpp:1204-1211); absent arrays are passed to the kernels as NULL.
template <typename T>
const T * optional_in(Staging & st, const T * host, int64_t n, int64_t n_samp, int64_t rows = 1) {
    if (n != n_samp || host == nullptr) return nullptr;
    return st.in(host, (size_t)(rows * n_samp));
}

// global2local is mapped per call by the reference (`map(to: raw_global2local...)`,
// ops_scan_map.cpp:187): use a registered copy when there is one, else stage it.
const int64_t * resolve_g2l(Staging & st, const int64_t * host, int64_t n_submap, bool accel) {
    if (accel) {
        void * p = Manager::get().find(host);
        if (p != nullptr) return static_cast<const int64_t *>(p);
        // content-addressed upload: repeated calls with the same table cost nothing
        ParamBlock pb;
        pb.push(host, sizeof(int64_t) * (size_t)n_submap);
        return reinterpret_cast<const int64_t *>(pb.commit(Manager::get().stream()));
    }
    return st.in(host, (size_t)n_submap);
}

}  // namespace

extern "C" {

int toast_hip_pointing_detector(const double * focalplane, const double * boresight,
                                const int32_t * quat_index, int64_t n_det, double * quats,
                                int64_t n_quat_rows, int64_t n_samp,
                                const toast_hip_interval * intervals, int64_t n_view,
                                const uint8_t * shared_flags, int64_t n_flags, uint8_t mask,
                                int use_accel) {
    return guarded([&] {
        Call c(use_accel);
        const double * d_bore = c.st.in(boresight, (size_t)(4 * n_samp));
        double * d_quats = c.st.inout(quats, (size_t)(n_quat_rows * n_samp * 4));
        const uint8_t * d_flags = optional_in(c.st, shared_flags, n_flags, n_samp);
        c.check(toast_hip_pointing_detector_dev(focalplane, d_bore, quat_index, n_det, d_quats, n_samp,
                                                intervals, n_view, d_flags, d_flags ? n_samp : 0,
                                                mask, c.stream));
        c.st.finish();
    });
}

int toast_hip_pixels_healpix(const int32_t * quat_index, int64_t n_det, const double * quats,
                             int64_t n_quat_rows, const uint8_t * shared_flags, int64_t n_flags,
                             uint8_t mask, const int32_t * pixel_index, int64_t * pixels,
                             int64_t n_pixel_rows, int64_t n_samp,
                             const toast_hip_interval * intervals, int64_t n_view,
                             uint8_t * hit_submaps, int64_t n_submap, int64_t n_pix_submap,
                             int64_t nside, int nest, int use_accel) {
    return guarded([&] {
        Call c(use_accel);
        const double * d_quats = c.st.in(quats, (size_t)(n_quat_rows * n_samp * 4));
        int64_t * d_pix = c.st.inout(pixels, (size_t)(n_pixel_rows * n_samp));
        const uint8_t * d_flags = optional_in(c.st, shared_flags, n_flags, n_samp);
        // hit_submaps is `map(tofrom:)` in the reference (ops_pixels_healpix.cpp:1263): always a
        // per-call device copy, OR-ed with the host contents.
        Staging hs(false, c.stream);
        uint8_t * d_hsub = hs.temp_inout(hit_submaps, (size_t)n_submap);
        c.check(toast_hip_pixels_healpix_dev(quat_index, n_det, d_quats, d_flags, d_flags ? n_samp : 0,
                                             mask, pixel_index, d_pix, n_samp, intervals, n_view,
                                             d_hsub, n_submap, n_pix_submap, nside, nest, c.stream));
        hs.finish();
        c.st.finish();
    });
}

int toast_hip_stokes_weights_IQU(const int32_t * quat_index, int64_t n_det, const double * quats,
                                 int64_t n_quat_rows, const int32_t * weight_index, double * weights,
                                 int64_t n_weight_rows, int64_t n_samp, const double * hwp,
                                 int64_t n_hwp, const toast_hip_interval * intervals, int64_t n_view,
                                 const double * epsilon, const double * gamma, const double * cal,
                                 int iau, int use_accel) {
    return guarded([&] {
        Call c(use_accel);
        const double * d_quats = c.st.in(quats, (size_t)(n_quat_rows * n_samp * 4));
        double * d_w = c.st.inout(weights, (size_t)(n_weight_rows * n_samp * 3));
        const double * d_hwp = optional_in(c.st, hwp, n_hwp, n_samp);
        c.check(toast_hip_stokes_weights_IQU_dev(quat_index, n_det, d_quats, weight_index, d_w, n_samp,
                                                 d_hwp, d_hwp ? n_samp : 0, intervals, n_view, epsilon,
                                                 gamma, cal, iau, c.stream));
        c.st.finish();
    });
}

int toast_hip_stokes_weights_QU(const int32_t * quat_index, int64_t n_det, const double * quats,
                                int64_t n_quat_rows, const int32_t * weight_index, double * weights,
                                int64_t n_weight_rows, int64_t n_samp, const double * hwp,
                                int64_t n_hwp, const toast_hip_interval * intervals, int64_t n_view,
                                const double * epsilon, const double * gamma, const double * cal,
                                int iau, int use_accel) {
    return guarded([&] {
        Call c(use_accel);
        const double * d_quats = c.st.in(quats, (size_t)(n_quat_rows * n_samp * 4));
        double * d_w = c.st.inout(weights, (size_t)(n_weight_rows * n_samp * 2));
        const double * d_hwp = optional_in(c.st, hwp, n_hwp, n_samp);
        c.check(toast_hip_stokes_weights_QU_dev(quat_index, n_det, d_quats, weight_index, d_w, n_samp,
                                                d_hwp, d_hwp ? n_samp : 0, intervals, n_view, epsilon,
                                                gamma, cal, iau, c.stream));
        c.st.finish();
    });
}

int toast_hip_stokes_weights_I(const int32_t * weight_index, int64_t n_det, double * weights,
                               int64_t n_weight_rows, int64_t n_samp,
                               const toast_hip_interval * intervals, int64_t n_view,
                               const double * cal, int use_accel) {
    return guarded([&] {
        Call c(use_accel);
        double * d_w = c.st.inout(weights, (size_t)(n_weight_rows * n_samp));
        c.check(toast_hip_stokes_weights_I_dev(weight_index, n_det, d_w, n_samp, intervals, n_view, cal,
                                               c.stream));
        c.st.finish();
    });
}

int toast_hip_scan_map(int map_dtype, const int64_t * global2local, int64_t n_submap,
                       int64_t n_pix_submap, const void * mapdata, int64_t n_local_submap,
                       int64_t nnz, double * det_data, int64_t n_data_rows,
                       const int32_t * data_index, const int64_t * pixels, int64_t n_pixel_rows,
                       const int32_t * pixel_index, const double * weights, int64_t n_weight_rows,
                       const int32_t * weight_index, int64_t n_det, int64_t n_samp,
                       const toast_hip_interval * intervals, int64_t n_view, double data_scale,
                       int should_zero, int should_subtract, int should_scale, int use_accel) {
    return guarded([&] {
        static const size_t esize[4] = {8, 4, 8, 4};
        if (map_dtype < 0 || map_dtype > 3) fail_arg("unknown map_dtype");
        Call c(use_accel);
        const int64_t * d_g2l = resolve_g2l(c.st, global2local, n_submap, use_accel != 0);
        const char * d_map = c.st.in(static_cast<const char *>(mapdata),
                                     (size_t)(n_local_submap * n_pix_submap * nnz) * esize[map_dtype]);
        double * d_tod = c.st.inout(det_data, (size_t)(n_data_rows * n_samp));
        const int64_t * d_pix = c.st.in(pixels, (size_t)(n_pixel_rows * n_samp));
        const double * d_w = c.st.in(weights, (size_t)(n_weight_rows * n_samp * nnz));
        c.check(toast_hip_scan_map_dev(map_dtype, d_g2l, n_pix_submap, d_map, nnz, d_tod, data_index,
                                       d_pix, pixel_index, d_w, weight_index, n_det, n_samp, intervals,
                                       n_view, data_scale, should_zero, should_subtract, should_scale,
                                       nullptr, c.stream));
        c.st.finish();
    });
}

int toast_hip_build_noise_weighted(
    const int64_t * global2local, int64_t n_submap, double * zmap, int64_t n_local_submap,
    int64_t n_pix_submap, int64_t nnz, const int32_t * pixel_index, const int64_t * pixels,
    int64_t n_pixel_rows, const int32_t * weight_index, const double * weights,
    int64_t n_weight_rows, const int32_t * data_index, const double * det_data, int64_t n_data_rows,
    const int32_t * flag_index, const uint8_t * det_flags, int64_t n_flag_rows, int64_t n_flag_samp,
    const double * det_scale, uint8_t det_flag_mask, int64_t n_det, int64_t n_samp,
    const toast_hip_interval * intervals, int64_t n_view, const uint8_t * shared_flags,
    int64_t n_shared_flags, uint8_t shared_flag_mask, int use_accel) {
    return guarded([&] {
        Call c(use_accel);
        const int64_t * d_g2l = resolve_g2l(c.st, global2local, n_submap, use_accel != 0);
        double * d_z = c.st.inout(zmap, (size_t)(n_local_submap * n_pix_submap * nnz));
        const int64_t * d_pix = c.st.in(pixels, (size_t)(n_pixel_rows * n_samp));
        const double * d_w = c.st.in(weights, (size_t)(n_weight_rows * n_samp * nnz));
        const double * d_tod = c.st.in(det_data, (size_t)(n_data_rows * n_samp));
        const uint8_t * d_df = optional_in(c.st, det_flags, n_flag_samp, n_samp, n_flag_rows);
        const uint8_t * d_sf = optional_in(c.st, shared_flags, n_shared_flags, n_samp);
        c.check(toast_hip_build_noise_weighted_dev(
            d_g2l, d_z, n_pix_submap, nnz, pixel_index, d_pix, weight_index, d_w, data_index, d_tod,
            flag_index, d_df, d_df ? n_samp : 0, det_scale, det_flag_mask, n_det, n_samp, intervals,
            n_view, d_sf, d_sf ? n_samp : 0, shared_flag_mask, c.stream));
        c.st.finish();
    });
}

int toast_hip_noise_weight(double * det_data, int64_t n_data_rows, int64_t n_samp,
                           const int32_t * data_index, int64_t n_det,
                           const toast_hip_interval * intervals, int64_t n_view,
                           const double * detector_weights, int use_accel) {
    return guarded([&] {
        Call c(use_accel);
        double * d_tod = c.st.inout(det_data, (size_t)(n_data_rows * n_samp));
        c.check(toast_hip_noise_weight_dev(d_tod, n_samp, data_index, n_det, intervals, n_view,
                                           detector_weights, c.stream));
        c.st.finish();
    });
}

int toast_hip_cov_apply_diag(int64_t n_sub, int64_t subsize, int64_t nnz, const double * mat,
                             double * vec, int use_accel) {
    return guarded([&] {
        Call c(use_accel);
        const int64_t n_px = n_sub * subsize;
        const double * d_m = c.st.in(mat, (size_t)(n_px * nnz * (nnz + 1) / 2));
        double * d_v = c.st.inout(vec, (size_t)(n_px * nnz));
        c.check(toast_hip_cov_apply_diag_dev(n_sub, subsize, nnz, d_m, d_v, c.stream));
        c.st.finish();
    });
}

int toast_hip_healpix_ang2vec(int64_t n, const double * theta, const double * phi, double * vec, int use_accel) {
    return guarded([&] {
        Call c(use_accel);
        const double * d_t = c.st.in(theta, (size_t)n);
        const double * d_p = c.st.in(phi, (size_t)n);
        double * d_v = c.st.out(vec, (size_t)(3 * n));
        c.check(toast_hip_healpix_ang2vec_dev(n, d_t, d_p, d_v, c.stream));
        c.st.finish();
    });
}

int toast_hip_healpix_vec2ang(int64_t n, const double * vec, double * theta, double * phi, int use_accel) {
    return guarded([&] {
        Call c(use_accel);
        const double * d_v = c.st.in(vec, (size_t)(3 * n));
        double * d_t = c.st.out(theta, (size_t)n);
        double * d_p = c.st.out(phi, (size_t)n);
        c.check(toast_hip_healpix_vec2ang_dev(n, d_v, d_t, d_p, c.stream));
        c.st.finish();
    });
}

int toast_hip_healpix_ang2pix(int64_t nside, int nest, int64_t n, const double * theta, const double * phi,
                              int64_t * pix, int use_accel) {
    return guarded([&] {
        Call c(use_accel);
        const double * d_t = c.st.in(theta, (size_t)n);
        const double * d_p = c.st.in(phi, (size_t)n);
        int64_t * d_x = c.st.out(pix, (size_t)n);
        c.check(toast_hip_healpix_ang2pix_dev(nside, nest, n, d_t, d_p, d_x, c.stream));
        c.st.finish();
    });
}

int toast_hip_healpix_convert(int op, int64_t nside, int64_t levels, int64_t n, const int64_t * in, int64_t * out,
                              int use_accel) {
    return guarded([&] {
        Call c(use_accel);
        const int64_t * d_i = c.st.in(in, (size_t)n);
        int64_t * d_o = c.st.out(out, (size_t)n);
        c.check(toast_hip_healpix_convert_dev(op, nside, levels, n, d_i, d_o, c.stream));
        c.st.finish();
    });
}

int toast_hip_healpix_vec2pix(int64_t nside, int nest, int64_t n, const double * vec, int64_t * pix, int use_accel) {
    return guarded([&] {
        Call c(use_accel);
        const double * d_v = c.st.in(vec, (size_t)(3 * n));
        int64_t * d_p = c.st.out(pix, (size_t)n);
        c.check(toast_hip_healpix_vec2pix_dev(nside, nest, n, d_v, d_p, c.stream));
        c.st.finish();
    });
}

int toast_hip_cov_accum_diag_hits(int64_t n_sub, int64_t subsize, int64_t nnz, int64_t n_samp, const int64_t * submap,
                                 const int64_t * subpix, int64_t * hits, int use_accel) {
    (void)nnz;
    return guarded([&] {
        Call c(use_accel);
        const int64_t * d_sm = c.st.in(submap, (size_t)n_samp);
        const int64_t * d_px = c.st.in(subpix, (size_t)n_samp);
        int64_t * d_h = c.st.inout(hits, (size_t)(n_sub * subsize));
        c.check(toast_hip_cov_accum_diag_hits_dev(n_sub, subsize, n_samp, d_sm, d_px, d_h, c.stream));
        c.st.finish();
    });
}

int toast_hip_cov_accum_diag_invnpp(int64_t n_sub, int64_t subsize, int64_t nnz, int64_t n_samp,
                                   const int64_t * submap, const int64_t * subpix, const double * weights,
                                   double scale, double * invnpp, int use_accel) {
    return guarded([&] {
        Call c(use_accel);
        const int64_t * d_sm = c.st.in(submap, (size_t)n_samp);
        const int64_t * d_px = c.st.in(subpix, (size_t)n_samp);
        const double * d_w = c.st.in(weights, (size_t)(n_samp * nnz));
        double * d_c = c.st.inout(invnpp, (size_t)(n_sub * subsize * nnz * (nnz + 1) / 2));
        c.check(toast_hip_cov_accum_diag_invnpp_dev(n_sub, subsize, nnz, n_samp, d_sm, d_px, d_w, scale, d_c,
                                                    c.stream));
        c.st.finish();
    });
}

int toast_hip_cov_accum_zmap(int64_t n_sub, int64_t subsize, int64_t nnz, int64_t n_samp, const int64_t * submap,
                            const int64_t * subpix, const double * weights, double scale, const double * tod,
                            double * zmap, int use_accel) {
    return guarded([&] {
        Call c(use_accel);
        const int64_t * d_sm = c.st.in(submap, (size_t)n_samp);
        const int64_t * d_px = c.st.in(subpix, (size_t)n_samp);
        const double * d_w = c.st.in(weights, (size_t)(n_samp * nnz));
        const double * d_t = c.st.in(tod, (size_t)n_samp);
        double * d_z = c.st.inout(zmap, (size_t)(n_sub * subsize * nnz));
        c.check(toast_hip_cov_accum_zmap_dev(n_sub, subsize, nnz, n_samp, d_sm, d_px, d_w, scale, d_t, d_z, c.stream));
        c.st.finish();
    });
}

int toast_hip_global_to_local(int64_t n, const int64_t * global_pixels, int64_t n_pix_submap, const int64_t * global2local,
                              int64_t n_submap, int64_t * local_submaps, int64_t * local_pixels, int use_accel) {
    return guarded([&] {
        Call c(use_accel);
        const int64_t * d_g = c.st.in(global_pixels, (size_t)n);
        const int64_t * d_t = c.st.in(global2local, (size_t)n_submap);
        int64_t * d_s = c.st.out(local_submaps, (size_t)n);
        int64_t * d_p = c.st.out(local_pixels, (size_t)n);
        c.check(toast_hip_global_to_local_dev(n, d_g, n_pix_submap, d_t, d_s, d_p, c.stream));
        c.st.finish();
    });
}

int toast_hip_cov_mult_diag(int64_t n_sub, int64_t subsize, int64_t nnz, double * data1, const double * data2,
                            int use_accel) {
    return guarded([&] {
        Call c(use_accel);
        const int64_t n_px = n_sub * subsize;
        const size_t n = (size_t)(n_px * nnz * (nnz + 1) / 2);
        double * d_1 = c.st.inout(data1, n);
        const double * d_2 = c.st.in(data2, n);
        c.check(toast_hip_cov_mult_diag_dev(n_sub, subsize, nnz, d_1, d_2, c.stream));
        c.st.finish();
    });
}

int toast_hip_build_cov(int mode, const int64_t * global2local, int64_t n_submap, void * out,
                        int64_t n_local_submap, int64_t n_pix_submap, int64_t nnz,
                        const int32_t * pixel_index, const int64_t * pixels, int64_t n_pixel_rows,
                        const int32_t * weight_index, const double * weights, int64_t n_weight_rows,
                        const int32_t * flag_index, const uint8_t * det_flags, int64_t n_flag_rows,
                        int64_t n_flag_samp, const double * det_scale, uint8_t det_flag_mask,
                        int64_t n_det, int64_t n_samp, const toast_hip_interval * intervals,
                        int64_t n_view, const uint8_t * shared_flags, int64_t n_shared_flags,
                        uint8_t shared_flag_mask, int use_accel) {
    return guarded([&] {
        Call c(use_accel);
        const int64_t * d_g2l = resolve_g2l(c.st, global2local, n_submap, use_accel != 0);
        const int64_t nv = (mode == 0) ? 1 : nnz * (nnz + 1) / 2;
        char * d_out = c.st.inout(static_cast<char *>(out), (size_t)(n_local_submap * n_pix_submap * nv) * 8);
        const int64_t * d_pix = c.st.in(pixels, (size_t)(n_pixel_rows * n_samp));
        const double * d_w = (mode == 1) ? c.st.in(weights, (size_t)(n_weight_rows * n_samp * nnz)) : nullptr;
        const uint8_t * d_df = optional_in(c.st, det_flags, n_flag_samp, n_samp, n_flag_rows);
        const uint8_t * d_sf = optional_in(c.st, shared_flags, n_shared_flags, n_samp);
        c.check(toast_hip_build_cov_dev(mode, d_g2l, d_out, n_pix_submap, nnz, pixel_index, d_pix,
                                        weight_index, d_w, flag_index, d_df, d_df ? n_samp : 0, det_scale,
                                        det_flag_mask, n_det, n_samp, intervals, n_view, d_sf,
                                        d_sf ? n_samp : 0, shared_flag_mask, c.stream));
        c.st.finish();
    });
}

int toast_hip_build_cov_hits(const int64_t * global2local, int64_t n_submap, double * invcov, int64_t * hits,
                             int64_t n_local_submap, int64_t n_pix_submap, int64_t nnz,
                             const int32_t * pixel_index, const int64_t * pixels, int64_t n_pixel_rows,
                             const int32_t * weight_index, const double * weights, int64_t n_weight_rows,
                             const int32_t * flag_index, const uint8_t * det_flags, int64_t n_flag_rows,
                             int64_t n_flag_samp, const double * det_scale, uint8_t det_flag_mask,
                             int64_t n_det, int64_t n_samp, const toast_hip_interval * intervals,
                             int64_t n_view, const uint8_t * shared_flags, int64_t n_shared_flags,
                             uint8_t shared_flag_mask, int use_accel) {
    return guarded([&] {
        Call c(use_accel);
        const int64_t * d_g2l = resolve_g2l(c.st, global2local, n_submap, use_accel != 0);
        const int64_t nv = nnz * (nnz + 1) / 2;
        double * d_cov = c.st.inout(invcov, (size_t)(n_local_submap * n_pix_submap * nv));
        int64_t * d_hits = c.st.inout(hits, (size_t)(n_local_submap * n_pix_submap));
        const int64_t * d_pix = c.st.in(pixels, (size_t)(n_pixel_rows * n_samp));
        const double * d_w = c.st.in(weights, (size_t)(n_weight_rows * n_samp * nnz));
        const uint8_t * d_df = optional_in(c.st, det_flags, n_flag_samp, n_samp, n_flag_rows);
        const uint8_t * d_sf = optional_in(c.st, shared_flags, n_shared_flags, n_samp);
        c.check(toast_hip_build_cov_hits_dev(d_g2l, d_cov, d_hits, n_pix_submap, nnz, pixel_index, d_pix, weight_index,
                                             d_w, flag_index, d_df, d_df ? n_samp : 0, det_scale, det_flag_mask,
                                             n_det, n_samp, intervals, n_view, d_sf, d_sf ? n_samp : 0,
                                             shared_flag_mask, c.stream));
        c.st.finish();
    });
}

int toast_hip_cov_eigendecompose_diag(int64_t n_sub, int64_t subsize, int64_t nnz, double * data,
                                      double * cond, double threshold, int invert, int use_accel) {
    return guarded([&] {
        Call c(use_accel);
        const int64_t n_px = n_sub * subsize;
        double * d_d = c.st.inout(data, (size_t)(n_px * nnz * (nnz + 1) / 2));
        double * d_c = cond ? c.st.inout(cond, (size_t)n_px) : nullptr;
        c.check(toast_hip_cov_eigendecompose_diag_dev(n_sub, subsize, nnz, d_d, d_c, threshold, invert,
                                                      c.stream));
        c.st.finish();
    });
}

int toast_hip_template_offset_add_to_signal(int64_t step_length, int64_t amp_offset,
                                            const int64_t * n_amp_views, const double * amplitudes,
                                            const uint8_t * amplitude_flags, int64_t n_amp,
                                            int32_t data_index, double * det_data,
                                            int64_t n_data_rows, int64_t n_samp,
                                            const toast_hip_interval * intervals, int64_t n_view,
                                            int use_accel) {
    return guarded([&] {
        Call c(use_accel);
        const double * d_a = c.st.in(amplitudes, (size_t)n_amp);
        const uint8_t * d_af = c.st.in(amplitude_flags, (size_t)n_amp);
        double * d_tod = c.st.inout(det_data, (size_t)(n_data_rows * n_samp));
        c.check(toast_hip_template_offset_add_to_signal_dev(step_length, amp_offset, n_amp_views, d_a,
                                                            d_af, data_index, d_tod, n_samp, intervals,
                                                            n_view, c.stream));
        c.st.finish();
    });
}

int toast_hip_template_offset_project_signal(
    int32_t data_index, const double * det_data, int64_t n_data_rows, int32_t flag_index,
    const uint8_t * flag_data, int64_t n_flag_rows, uint8_t flag_mask, int64_t step_length,
    int64_t amp_offset, const int64_t * n_amp_views, double * amplitudes,
    const uint8_t * amplitude_flags, int64_t n_amp, int64_t n_samp,
    const toast_hip_interval * intervals, int64_t n_view, int use_accel) {
    return guarded([&] {
        Call c(use_accel);
        const double * d_tod = c.st.in(det_data, (size_t)(n_data_rows * n_samp));
        const uint8_t * d_f = nullptr;
        if (flag_index >= 0) d_f = c.st.in(flag_data, (size_t)(n_flag_rows * n_samp));
        double * d_a = c.st.inout(amplitudes, (size_t)n_amp);
        const uint8_t * d_af = c.st.in(amplitude_flags, (size_t)n_amp);
        c.check(toast_hip_template_offset_project_signal_dev(
            data_index, d_tod, flag_index, d_f, flag_mask, step_length, amp_offset, n_amp_views, d_a,
            d_af, n_samp, intervals, n_view, c.stream));
        c.st.finish();
    });
}

int toast_hip_template_offset_apply_diag_precond(const double * offset_var, const double * amp_in,
                                                 const uint8_t * amplitude_flags, double * amp_out,
                                                 int64_t n_amp, int use_accel) {
    return guarded([&] {
        Call c(use_accel);
        const double * d_v = c.st.in(offset_var, (size_t)n_amp);
        const double * d_i = c.st.in(amp_in, (size_t)n_amp);
        const uint8_t * d_f = c.st.in(amplitude_flags, (size_t)n_amp);
        double * d_o = c.st.inout(amp_out, (size_t)n_amp);
        c.check(toast_hip_template_offset_apply_diag_precond_dev(d_v, d_i, d_f, d_o, n_amp, c.stream));
        c.st.finish();
    });
}

}  // extern "C"
