// pybind_module.cpp -- `toast_amd._libtoast_hip`: the pybind11 host layer over the C ABI.
//
// Exposes the map-making hot path under the SAME Python names and argument orders as the
// reference's `toast._libtoast`, so that a kernels.py wrapper can switch
// `from ..._libtoast import pixels_healpix` to this module unchanged.  Buffers are validated
// like the reference's extract_buffer<T> (src/toast/_libtoast/common.hpp:32-124) -- same
// checks, same messages -- and then handed to libtoast_hip.so as plain pointers + sizes.
// Built with g++ only (no HIP headers): everything device-side sits behind include/toast_hip.h.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>

#include <cstdint>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "toast_hip.h"

namespace py = pybind11;

namespace {

using Shape = std::vector<int64_t>;

// numpy int64 may be reported as 'q' or 'l' (common.hpp:57-63)
std::string norm_format(const std::string & f) {
    std::string s;
    for (char ch : f) {
        if (ch == '@' || ch == '=' || ch == '<' || ch == '>' || ch == '!') continue;
        s.push_back(ch);
    }
    if (s == "l") s = "q";
    if (s == "L") s = "Q";
    return s;
}

template <typename T>
std::string target_format() {
    return norm_format(py::format_descriptor<T>::format());
}

[[noreturn]] void raise(const std::ostringstream & o) { throw std::runtime_error(o.str()); }

// Generic extraction: dtype given by format string + item size so that the Interval struct
// (a structured dtype) can be checked too.
void * extract_raw(py::buffer data, const char * name, const std::string & want_format, size_t itemsize,
                   bool check_format, size_t assert_dims, Shape & shape, const Shape & assert_shape) {
    auto info = data.request();
    std::ostringstream o;
    if (check_format) {
        const std::string got = norm_format(info.format);
        if (got != want_format) {
            o << "Object " << name << " has format \"" << got << "\" instead of \"" << want_format << "\"";
            raise(o);
        }
    }
    if ((size_t)info.itemsize != itemsize) {
        o << "Object " << name << " has item size of " << info.itemsize << " instead of " << itemsize;
        raise(o);
    }
    if ((size_t)info.ndim != assert_dims) {
        o << "Object " << name << " has " << info.ndim << " dimensions instead of " << assert_dims;
        raise(o);
    }
    shape.assign(info.shape.begin(), info.shape.end());
    py::ssize_t stride = info.itemsize;
    for (int d = (int)info.ndim - 1; d >= 0; d--) {
        if (info.strides[d] != stride && info.shape[d] > 1) {
            o << "Object " << name << ": python buffers must be contiguous in memory.";
            raise(o);
        }
        stride *= info.shape[d];
    }
    for (size_t d = 0; d < assert_dims; d++) {
        if (assert_shape[d] >= 0 && assert_shape[d] != shape[d]) {
            o << "Object " << name << " dimension " << d << " has length " << shape[d] << " instead of "
              << assert_shape[d];
            raise(o);
        }
    }
    return info.ptr;
}

template <typename T>
T * extract(py::buffer data, const char * name, size_t dims, Shape & shape, const Shape & assert_shape) {
    return static_cast<T *>(
        extract_raw(data, name, target_format<T>(), sizeof(T), true, dims, shape, assert_shape));
}

toast_hip_interval * extract_intervals(py::buffer data, Shape & shape) {
    return static_cast<toast_hip_interval *>(
        extract_raw(data, "intervals", "", sizeof(toast_hip_interval), false, 1, shape, {-1}));
}

void check(int rc) {
    if (rc != TOAST_HIP_OK) throw std::runtime_error(toast_hip_last_error());
}

struct RawBuf {
    void * ptr;
    size_t nbytes;
};

// accel_* take any contiguous buffer; the key is its base pointer and byte size
// (accelerator.cpp:722-766 extract_accel_buffer)
RawBuf accel_buf(py::buffer data) {
    auto info = data.request();
    size_t n = (size_t)info.itemsize;
    for (auto s : info.shape) n *= (size_t)s;
    return RawBuf{info.ptr, n};
}

template <int DTYPE, typename T>
void scan_map_binding(py::buffer global2local, int64_t n_pix_submap, py::buffer mapdata, py::buffer det_data,
                      py::buffer data_index, py::buffer pixels, py::buffer pixel_index, py::buffer weights,
                      py::buffer weight_index, py::buffer intervals, double data_scale, bool should_zero,
                      bool should_subtract, bool should_scale, bool use_accel) {
    Shape s(3);
    int32_t * p_idx = extract<int32_t>(pixel_index, "pixel_index", 1, s, {-1});
    const int64_t n_det = s[0];
    int64_t * pix = extract<int64_t>(pixels, "pixels", 2, s, {-1, -1});
    const int64_t n_pix_rows = s[0], n_samp = s[1];
    int32_t * w_idx = extract<int32_t>(weight_index, "weight_index", 1, s, {n_det});
    // weights may be 2-D (nnz = 1) or 3-D (ops_scan_map.cpp:125-139)
    double * w;
    int64_t nnz, n_w_rows;
    if (weights.request().ndim == 2) {
        w = extract<double>(weights, "weights", 2, s, {-1, n_samp});
        nnz = 1;
        n_w_rows = s[0];
    } else {
        w = extract<double>(weights, "weights", 3, s, {-1, n_samp, -1});
        nnz = s[2];
        n_w_rows = s[0];
    }
    int32_t * d_idx = extract<int32_t>(data_index, "data_index", 1, s, {n_det});
    double * tod = extract<double>(det_data, "det_data", 2, s, {-1, n_samp});
    const int64_t n_d_rows = s[0];
    toast_hip_interval * ivl = extract_intervals(intervals, s);
    const int64_t n_view = s[0];
    int64_t * g2l = extract<int64_t>(global2local, "global2local", 1, s, {-1});
    const int64_t n_submap = s[0];
    T * map = extract<T>(mapdata, "mapdata", 3, s, {-1, n_pix_submap, nnz});
    const int64_t n_local = s[0];
    check(toast_hip_scan_map(DTYPE, g2l, n_submap, n_pix_submap, map, n_local, nnz, tod, n_d_rows, d_idx, pix,
                             n_pix_rows, p_idx, w, n_w_rows, w_idx, n_det, n_samp, ivl, n_view, data_scale,
                             should_zero, should_subtract, should_scale, use_accel));
}

}  // namespace

// ------------------------------------------------------------------------------------------
// FFTPlanReal1D / FFTPlanReal1DStore: the reference's batched 1-D real FFT plans
// (src/toast/_libtoast/math_fft.cpp:10-175, src/libtoast/include/toast/math_fft.hpp:24-82, FFTW
// implementation src/libtoast/src/toast_math_fft_fftw.cpp:26-128): `n` transforms of `length`,
// one host slab of 2 n length doubles (time-domain buffers first, Fourier-domain buffers after
// them, zero-initialised), FFTW half-complex layout, forward result * scale, backward result *
// scale / length.  exec() runs on the GPU (toast_hip_fft_r1d: staged through device memory).
// ------------------------------------------------------------------------------------------
enum class FFTPlanType { fast = 0, best = 1 };
enum class FFTDirection { forward = 0, backward = 1 };

class HipFFTPlanReal1D : public std::enable_shared_from_this<HipFFTPlanReal1D> {
public:
    typedef std::shared_ptr<HipFFTPlanReal1D> pshr;
    static pshr create(int64_t length, int64_t n, FFTPlanType type, FFTDirection dir, double scale) {
        if (length < 1 || n < 1) throw std::runtime_error("FFTPlanReal1D: length and n must be positive");
        return pshr(new HipFFTPlanReal1D(length, n, type, dir, scale));
    }
    void exec() {
        double * traw = data_.data();
        double * fraw = data_.data() + n_ * length_;
        if (dir_ == FFTDirection::forward) {
            check(toast_hip_fft_r1d(1, length_, n_, traw, fraw, scale_, 0));
        } else {
            check(toast_hip_fft_r1d(0, length_, n_, fraw, traw, scale_, 0));
        }
    }
    double * tdata(int64_t indx) { return data_.data() + checked(indx) * length_; }
    double * fdata(int64_t indx) { return data_.data() + (n_ + checked(indx)) * length_; }
    int64_t length() const { return length_; }
    int64_t count() const { return n_; }

private:
    HipFFTPlanReal1D(int64_t length, int64_t n, FFTPlanType type, FFTDirection dir, double scale)
        : length_(length), n_(n), scale_(scale), type_(type), dir_(dir), data_((size_t)(2 * n * length), 0.0) {}
    int64_t checked(int64_t indx) const {
        if (indx < 0 || indx >= n_) throw py::index_error("FFTPlanReal1D: buffer index out of range");
        return indx;
    }
    int64_t length_, n_;
    double scale_;
    FFTPlanType type_;
    FFTDirection dir_;
    std::vector<double> data_;
};

class HipFFTPlanReal1DStore {
public:
    static HipFFTPlanReal1DStore & get() {
        static HipFFTPlanReal1DStore instance;
        return instance;
    }
    void clear() {
        fplans_.clear();
        rplans_.clear();
    }
    void cache(int64_t len, int64_t n) {
        forward(len, n);
        backward(len, n);
    }
    HipFFTPlanReal1D::pshr forward(int64_t len, int64_t n) { return fetch(fplans_, len, n, FFTDirection::forward); }
    HipFFTPlanReal1D::pshr backward(int64_t len, int64_t n) { return fetch(rplans_, len, n, FFTDirection::backward); }

private:
    typedef std::map<std::pair<int64_t, int64_t>, HipFFTPlanReal1D::pshr> plan_map;
    static HipFFTPlanReal1D::pshr fetch(plan_map & plans, int64_t len, int64_t n, FFTDirection dir) {
        auto key = std::make_pair(len, n);
        auto it = plans.find(key);
        if (it == plans.end()) {
            it = plans.emplace(key, HipFFTPlanReal1D::create(len, n, FFTPlanType::fast, dir, 1.0)).first;
        }
        return it->second;
    }
    plan_map fplans_, rplans_;
};

PYBIND11_MODULE(_libtoast_hip, m) {
    m.doc() = "MI355X (gfx950) implementation of toast._libtoast's map-making hot path";

    // ---- Interval dtype (intervals.cpp:11-38)
    PYBIND11_NUMPY_DTYPE(toast_hip_interval, start, stop, first, last);
    py::class_<toast_hip_interval>(m, "Interval", "Numpy dtype for an interval")
        .def(py::init([]() { return toast_hip_interval(); }))
        .def_readwrite("start", &toast_hip_interval::start)
        .def_readwrite("stop", &toast_hip_interval::stop)
        .def_readwrite("first", &toast_hip_interval::first)
        .def_readwrite("last", &toast_hip_interval::last)
        .def("astuple", [](const toast_hip_interval & s) { return py::make_tuple(s.start, s.stop, s.first, s.last); });

    // ---- accelerator management (accelerator.cpp:768-1110)
    m.def("accel_enabled", []() { return toast_hip_accel_enabled() != 0; });
    m.def("accel_assign_device", [](int node_procs, int node_rank, float mem_gb, bool disabled) {
        check(toast_hip_accel_assign_device(node_procs, node_rank, mem_gb, disabled));
    });
    m.def("accel_get_device", []() {
        int d = -1;
        check(toast_hip_accel_get_device(&d));
        return d;
    });
    m.def("accel_present", [](py::buffer data, std::string name) {
        RawBuf b = accel_buf(data);
        int r = 0;
        check(toast_hip_accel_present(b.ptr, b.nbytes, &r));
        return r != 0;
    }, py::arg("data"), py::arg("name"));
    // kind (an extension of the reference's signature, accelerator.cpp:349-377): what the array is to the kernels, which
    // decides where in HBM the library puts it (include/toast_hip.h, toast_hip_accel_create_kind): 0 read-mostly, 1 a
    // timestream that sweeps read and write, 2 the target of a scatter.  Left out (-1), a two-dimensional float64 array
    // -- [detector][sample] -- counts as a timestream, everything else as read-mostly.
    m.def("accel_create", [](py::buffer data, std::string name, int kind) {
        if (kind < 0) {
            const py::buffer_info info = data.request();
            kind = (info.ndim == 2 && info.format == py::format_descriptor<double>::format()) ? 1 : 0;
        }
        RawBuf b = accel_buf(data);
        check(kind != 0 ? toast_hip_accel_create_kind(b.ptr, b.nbytes, name.c_str(), kind)
                        : toast_hip_accel_create(b.ptr, b.nbytes, name.c_str()));
    }, py::arg("data"), py::arg("name"), py::arg("kind") = -1);
    m.def("accel_reset", [](py::buffer data, std::string name) {
        RawBuf b = accel_buf(data);
        check(toast_hip_accel_reset(b.ptr, b.nbytes, name.c_str()));
    }, py::arg("data"), py::arg("name"));
    m.def("accel_update_device", [](py::buffer data, std::string name) {
        RawBuf b = accel_buf(data);
        check(toast_hip_accel_update_device(b.ptr, b.nbytes, name.c_str()));
    }, py::arg("data"), py::arg("name"));
    m.def("accel_update_host", [](py::buffer data, std::string name) {
        RawBuf b = accel_buf(data);
        check(toast_hip_accel_update_host(b.ptr, b.nbytes, name.c_str()));
    }, py::arg("data"), py::arg("name"));
    m.def("accel_delete", [](py::buffer data, std::string name) {
        RawBuf b = accel_buf(data);
        check(toast_hip_accel_delete(b.ptr, b.nbytes, name.c_str()));
    }, py::arg("data"), py::arg("name"));
    m.def("accel_dump", []() { check(toast_hip_accel_dump()); });
    m.def("accel_synchronize", []() { check(toast_hip_synchronize()); });

    // ---- pointing_detector (ops_pointing_detector.cpp:78-88)
    m.def("pointing_detector", [](py::buffer focalplane, py::buffer boresight, py::buffer quat_index,
                                  py::buffer quats, py::buffer intervals, py::buffer shared_flags,
                                  uint8_t shared_flag_mask, bool use_accel) {
        Shape s(3);
        int32_t * q_idx = extract<int32_t>(quat_index, "quat_index", 1, s, {-1});
        const int64_t n_det = s[0];
        double * fp = extract<double>(focalplane, "focalplane", 2, s, {n_det, 4});
        double * bore = extract<double>(boresight, "boresight", 2, s, {-1, 4});
        const int64_t n_samp = s[0];
        double * q = extract<double>(quats, "quats", 3, s, {-1, n_samp, 4});
        const int64_t n_q_rows = s[0];
        toast_hip_interval * ivl = extract_intervals(intervals, s);
        const int64_t n_view = s[0];
        uint8_t * fl = extract<uint8_t>(shared_flags, "flags", 1, s, {-1});
        check(toast_hip_pointing_detector(fp, bore, q_idx, n_det, q, n_q_rows, n_samp, ivl, n_view, fl, s[0],
                                          shared_flag_mask, use_accel));
    });

    // ---- pixels_healpix (ops_pixels_healpix.cpp:1153-1167)
    m.def("pixels_healpix", [](py::buffer quat_index, py::buffer quats, py::buffer shared_flags,
                               uint8_t shared_flag_mask, py::buffer pixel_index, py::buffer pixels,
                               py::buffer intervals, py::buffer hit_submaps, int64_t n_pix_submap, int64_t nside,
                               bool nest, bool use_accel) {
        Shape s(3);
        int32_t * q_idx = extract<int32_t>(quat_index, "quat_index", 1, s, {-1});
        const int64_t n_det = s[0];
        int32_t * p_idx = extract<int32_t>(pixel_index, "pixel_index", 1, s, {n_det});
        int64_t * pix = extract<int64_t>(pixels, "pixels", 2, s, {-1, -1});
        const int64_t n_p_rows = s[0], n_samp = s[1];
        double * q = extract<double>(quats, "quats", 3, s, {-1, n_samp, 4});
        const int64_t n_q_rows = s[0];
        toast_hip_interval * ivl = extract_intervals(intervals, s);
        const int64_t n_view = s[0];
        uint8_t * hs = extract<uint8_t>(hit_submaps, "hit_submaps", 1, s, {-1});
        const int64_t n_submap = s[0];
        uint8_t * fl = extract<uint8_t>(shared_flags, "flags", 1, s, {-1});
        check(toast_hip_pixels_healpix(q_idx, n_det, q, n_q_rows, fl, s[0], shared_flag_mask, p_idx, pix, n_p_rows,
                                       n_samp, ivl, n_view, hs, n_submap, n_pix_submap, nside, nest, use_accel));
    });

    // ---- stokes weights (ops_stokes_weights.cpp:151-163, :398-404)
    m.def("stokes_weights_IQU", [](py::buffer quat_index, py::buffer quats, py::buffer weight_index,
                                   py::buffer weights, py::buffer hwp, py::buffer intervals, py::buffer epsilon,
                                   py::buffer gamma, py::buffer cal, bool IAU, bool use_accel) {
        Shape s(3);
        int32_t * q_idx = extract<int32_t>(quat_index, "quat_index", 1, s, {-1});
        const int64_t n_det = s[0];
        int32_t * w_idx = extract<int32_t>(weight_index, "weight_index", 1, s, {n_det});
        double * w = extract<double>(weights, "weights", 3, s, {-1, -1, 3});
        const int64_t n_w_rows = s[0], n_samp = s[1];
        double * q = extract<double>(quats, "quats", 3, s, {-1, n_samp, 4});
        const int64_t n_q_rows = s[0];
        double * h = extract<double>(hwp, "hwp", 1, s, {-1});
        const int64_t n_hwp = s[0];
        toast_hip_interval * ivl = extract_intervals(intervals, s);
        const int64_t n_view = s[0];
        double * eps = extract<double>(epsilon, "epsilon", 1, s, {n_det});
        double * cl = extract<double>(cal, "cal", 1, s, {n_det});
        double * gm = extract<double>(gamma, "gamma", 1, s, {n_det});
        check(toast_hip_stokes_weights_IQU(q_idx, n_det, q, n_q_rows, w_idx, w, n_w_rows, n_samp, h, n_hwp, ivl,
                                           n_view, eps, gm, cl, IAU, use_accel));
    });
    m.def("stokes_weights_I", [](py::buffer weight_index, py::buffer weights, py::buffer intervals,
                                 py::buffer cal, bool use_accel) {
        Shape s(3);
        int32_t * w_idx = extract<int32_t>(weight_index, "weight_index", 1, s, {-1});
        const int64_t n_det = s[0];
        double * w = extract<double>(weights, "weights", 2, s, {n_det, -1});
        const int64_t n_samp = s[1];
        toast_hip_interval * ivl = extract_intervals(intervals, s);
        const int64_t n_view = s[0];
        double * cl = extract<double>(cal, "cal", 1, s, {n_det});
        check(toast_hip_stokes_weights_I(w_idx, n_det, w, n_det, n_samp, ivl, n_view, cl, use_accel));
    });

    // ---- scan_map, four map dtypes (ops_scan_map.cpp:84-102)
    m.def("ops_scan_map_float64", &scan_map_binding<TOAST_HIP_MAP_F64, double>);
    m.def("ops_scan_map_float32", &scan_map_binding<TOAST_HIP_MAP_F32, float>);
    m.def("ops_scan_map_int64", &scan_map_binding<TOAST_HIP_MAP_I64, int64_t>);
    m.def("ops_scan_map_int32", &scan_map_binding<TOAST_HIP_MAP_I32, int32_t>);

    // ---- build_noise_weighted (ops_mapmaker_utils.cpp:93-111)
    m.def("build_noise_weighted", [](py::buffer global2local, py::buffer zmap, py::buffer pixel_index,
                                     py::buffer pixels, py::buffer weight_index, py::buffer weights,
                                     py::buffer data_index, py::buffer det_data, py::buffer flag_index,
                                     py::buffer det_flags, py::buffer det_scale, uint8_t det_flag_mask,
                                     py::buffer intervals, py::buffer shared_flags, uint8_t shared_flag_mask,
                                     bool use_accel) {
        Shape s(3);
        int32_t * p_idx = extract<int32_t>(pixel_index, "pixel_index", 1, s, {-1});
        const int64_t n_det = s[0];
        int64_t * pix = extract<int64_t>(pixels, "pixels", 2, s, {-1, -1});
        const int64_t n_p_rows = s[0], n_samp = s[1];
        int32_t * w_idx = extract<int32_t>(weight_index, "weight_index", 1, s, {n_det});
        double * w;
        int64_t nnz, n_w_rows;
        if (weights.request().ndim == 2) {
            w = extract<double>(weights, "weights", 2, s, {-1, n_samp});
            nnz = 1;
            n_w_rows = s[0];
        } else {
            w = extract<double>(weights, "weights", 3, s, {-1, n_samp, -1});
            nnz = s[2];
            n_w_rows = s[0];
        }
        int32_t * d_idx = extract<int32_t>(data_index, "data_index", 1, s, {n_det});
        double * tod = extract<double>(det_data, "det_data", 2, s, {-1, n_samp});
        const int64_t n_d_rows = s[0];
        int32_t * f_idx = extract<int32_t>(flag_index, "flag_index", 1, s, {n_det});
        double * dscale = extract<double>(det_scale, "det_scale", 1, s, {n_det});
        toast_hip_interval * ivl = extract_intervals(intervals, s);
        const int64_t n_view = s[0];
        int64_t * g2l = extract<int64_t>(global2local, "global2local", 1, s, {-1});
        const int64_t n_submap = s[0];
        double * z = extract<double>(zmap, "zmap", 3, s, {-1, -1, nnz});
        const int64_t n_local = s[0], nps = s[1];
        uint8_t * sf = extract<uint8_t>(shared_flags, "flags", 1, s, {-1});
        const int64_t n_sf = s[0];
        uint8_t * df = extract<uint8_t>(det_flags, "det_flags", 2, s, {-1, -1});
        check(toast_hip_build_noise_weighted(g2l, n_submap, z, n_local, nps, nnz, p_idx, pix, n_p_rows, w_idx, w,
                                             n_w_rows, d_idx, tod, n_d_rows, f_idx, df, s[0], s[1], dscale,
                                             det_flag_mask, n_det, n_samp, ivl, n_view, sf, n_sf, shared_flag_mask,
                                             use_accel));
    });

    // ---- noise_weight (ops_noise_weight.cpp:12-19)
    m.def("noise_weight", [](py::buffer det_data, py::buffer data_index, py::buffer intervals,
                             py::buffer detector_weights, bool use_accel) {
        Shape s(3);
        int32_t * d_idx = extract<int32_t>(data_index, "data_index", 1, s, {-1});
        const int64_t n_det = s[0];
        double * tod = extract<double>(det_data, "det_data", 2, s, {-1, -1});
        const int64_t n_rows = s[0], n_samp = s[1];
        toast_hip_interval * ivl = extract_intervals(intervals, s);
        const int64_t n_view = s[0];
        double * dw = extract<double>(detector_weights, "detector_weights", 1, s, {n_det});
        check(toast_hip_noise_weight(tod, n_rows, n_samp, d_idx, n_det, ivl, n_view, dw, use_accel));
    });

    // ---- cov_apply_diag (map_cov.cpp:372-401; flat buffers)
    m.def("cov_apply_diag", [](int64_t nsub, int64_t nsubpix, int64_t nnz, py::buffer mat, py::buffer vec,
                               bool use_accel) {
        auto im = mat.request();
        auto iv = vec.request();
        if (norm_format(im.format) != "d" || norm_format(iv.format) != "d") {
            throw std::runtime_error("cov_apply_diag: buffers must be float64");
        }
        const int64_t block = nnz * (nnz + 1) / 2;
        const size_t nb = (size_t)(im.size / block), nv = (size_t)(iv.size / nnz);
        if (nb != nv) {
            std::ostringstream o;
            o << "Buffer sizes are not consistent. npix_matrix = " << nb << ", npix_map = " << nv
              << ", matrix_size = " << im.size << ", block = " << block << ", nnz = " << nnz;
            raise(o);
        }
        check(toast_hip_cov_apply_diag(nsub, nsubpix, nnz, static_cast<double *>(im.ptr),
                                       static_cast<double *>(iv.ptr), use_accel));
    }, py::arg("nsub"), py::arg("nsubpix"), py::arg("nnz"), py::arg("mat"), py::arg("vec"),
       py::arg("use_accel") = false);

    m.def("healpix_ang2vec", [](py::buffer theta, py::buffer phi, py::buffer vec) {
        Shape shape;
        double * raw_theta = extract<double>(theta, "theta", 1, shape, {-1});
        const int64_t n = shape[0];
        double * raw_phi = extract<double>(phi, "phi", 1, shape, {n});
        double * raw_vec = extract<double>(vec, "vec", 2, shape, {n, 3});
        check(toast_hip_healpix_ang2vec(n, raw_theta, raw_phi, raw_vec, 0));
    });
    m.def("healpix_vec2ang", [](py::buffer vec, py::buffer theta, py::buffer phi) {
        Shape shape;
        double * raw_vec = extract<double>(vec, "vec", 2, shape, {-1, 3});
        const int64_t n = shape[0];
        double * raw_theta = extract<double>(theta, "theta", 1, shape, {n});
        double * raw_phi = extract<double>(phi, "phi", 1, shape, {n});
        check(toast_hip_healpix_vec2ang(n, raw_vec, raw_theta, raw_phi, 0));
    });
    for (int nest = 1; nest >= 0; --nest) {
        m.def(nest ? "healpix_ang2nest" : "healpix_ang2ring", [nest](int64_t nside, py::buffer theta, py::buffer phi,
                                                                     py::buffer pix) {
            Shape shape;
            double * raw_theta = extract<double>(theta, "theta", 1, shape, {-1});
            const int64_t n = shape[0];
            double * raw_phi = extract<double>(phi, "phi", 1, shape, {n});
            int64_t * raw_pix = extract<int64_t>(pix, "pix", 1, shape, {n});
            check(toast_hip_healpix_ang2pix(nside, nest, n, raw_theta, raw_phi, raw_pix, 0));
        });
    }

    {
        struct Conv { const char * name; int op; const char * in; const char * out; };
        static const Conv two[] = {{"healpix_ring2nest", 0, "ring_pix", "nest_pix"},
                                   {"healpix_nest2ring", 1, "nest_pix", "ring_pix"}};
        for (const Conv & cv : two) {
            m.def(cv.name, [cv](int64_t nside, py::buffer in_pix, py::buffer out_pix) {
                Shape shape;
                int64_t * raw_in = extract<int64_t>(in_pix, cv.in, 1, shape, {-1});
                const int64_t n = shape[0];
                int64_t * raw_out = extract<int64_t>(out_pix, cv.out, 1, shape, {n});
                check(toast_hip_healpix_convert(cv.op, nside, 0, n, raw_in, raw_out, 0));
            });
        }
        static const Conv four[] = {{"healpix_degrade_nest", 2, "in_pix", "out_pix"},
                                    {"healpix_upgrade_nest", 3, "in_pix", "out_pix"},
                                    {"healpix_degrade_ring", 4, "in_pix", "out_pix"},
                                    {"healpix_upgrade_ring", 5, "in_pix", "out_pix"}};
        for (const Conv & cv : four) {
            m.def(cv.name, [cv](int64_t in_nside, int64_t factor, py::buffer in_pix, py::buffer out_pix) {
                Shape shape;
                int64_t * raw_in = extract<int64_t>(in_pix, cv.in, 1, shape, {-1});
                const int64_t n = shape[0];
                int64_t * raw_out = extract<int64_t>(out_pix, cv.out, 1, shape, {n});
                check(toast_hip_healpix_convert(cv.op, in_nside, factor, n, raw_in, raw_out, 0));
            });
        }
    }

    for (int nest = 1; nest >= 0; --nest) {
        m.def(nest ? "healpix_vec2nest" : "healpix_vec2ring", [nest](int64_t nside, py::buffer vec, py::buffer pix) {
            Shape shape;
            double * raw_vec = extract<double>(vec, "vec", 2, shape, {-1, 3});
            const int64_t n_samp = shape[0];
            int64_t * raw_pix = extract<int64_t>(pix, "pix", 1, shape, {n_samp});
            check(toast_hip_healpix_vec2pix(nside, nest, n_samp, raw_vec, raw_pix, 0));
        }, py::arg("nside"), py::arg("vec"), py::arg("pix"));
    }

    m.def("cov_accum_diag_hits", [](int64_t nsub, int64_t nsubpix, int64_t nnz, py::buffer submap, py::buffer subpix,
                                    py::buffer hits, bool use_accel) {
        auto ism = submap.request();
        auto ipx = subpix.request();
        auto ih = hits.request();
        if (!(norm_format(ism.format) == "q") || !(norm_format(ipx.format) == "q") || !(norm_format(ih.format) == "q")) {
            throw std::runtime_error("cov_accum_diag_hits: buffers must be int64");
        }
        if (ipx.size != ism.size) throw std::runtime_error("Buffer sizes are not consistent.");
        check(toast_hip_cov_accum_diag_hits(nsub, nsubpix, nnz, (int64_t)ism.size, static_cast<int64_t *>(ism.ptr),
                                            static_cast<int64_t *>(ipx.ptr), static_cast<int64_t *>(ih.ptr),
                                            use_accel));
    }, py::arg("nsub"), py::arg("nsubpix"), py::arg("nnz"), py::arg("submap"), py::arg("subpix"), py::arg("hits"),
       py::arg("use_accel") = false);

    m.def("cov_accum_diag_invnpp", [](int64_t nsub, int64_t nsubpix, int64_t nnz, py::buffer submap, py::buffer subpix,
                                      py::buffer weights, double scale, py::buffer invnpp, bool use_accel) {
        auto ism = submap.request();
        auto ipx = subpix.request();
        auto iw = weights.request();
        auto ic = invnpp.request();
        if (!(norm_format(ism.format) == "q") || !(norm_format(ipx.format) == "q")) {
            throw std::runtime_error("cov_accum_diag_invnpp: index buffers must be int64");
        }
        if (norm_format(iw.format) != "d" || norm_format(ic.format) != "d") {
            throw std::runtime_error("cov_accum_diag_invnpp: weights / invnpp must be float64");
        }
        if (ipx.size != ism.size || (size_t)(iw.size / nnz) != (size_t)ism.size) {
            throw std::runtime_error("Buffer sizes are not consistent.");
        }
        check(toast_hip_cov_accum_diag_invnpp(nsub, nsubpix, nnz, (int64_t)ism.size, static_cast<int64_t *>(ism.ptr),
                                              static_cast<int64_t *>(ipx.ptr), static_cast<double *>(iw.ptr), scale,
                                              static_cast<double *>(ic.ptr), use_accel));
    }, py::arg("nsub"), py::arg("nsubpix"), py::arg("nnz"), py::arg("submap"), py::arg("subpix"), py::arg("weights"),
       py::arg("scale"), py::arg("invnpp"), py::arg("use_accel") = false);

    // cov_accum_zmap and the all-in-one cov_accum_diag (map_cov.cpp:10-86, :199-250)
    auto accum_zmap = [](int64_t nsub, int64_t nsubpix, int64_t nnz, py::buffer submap, py::buffer subpix,
                         py::buffer weights, double scale, py::buffer tod, py::buffer zmap) {
        auto ism = submap.request();
        auto ipx = subpix.request();
        auto iw = weights.request();
        auto it = tod.request();
        auto iz = zmap.request();
        if (norm_format(ism.format) != "q" || norm_format(ipx.format) != "q") {
            throw std::runtime_error("cov_accum_zmap: index buffers must be int64");
        }
        if (norm_format(iw.format) != "d" || norm_format(it.format) != "d" || norm_format(iz.format) != "d") {
            throw std::runtime_error("cov_accum_zmap: weights / tod / zmap must be float64");
        }
        if (ipx.size != ism.size || it.size != ism.size || (size_t)(iw.size / nnz) != (size_t)ism.size) {
            throw std::runtime_error("Buffer sizes are not consistent.");
        }
        check(toast_hip_cov_accum_zmap(nsub, nsubpix, nnz, (int64_t)ism.size, static_cast<int64_t *>(ism.ptr),
                                       static_cast<int64_t *>(ipx.ptr), static_cast<double *>(iw.ptr), scale,
                                       static_cast<double *>(it.ptr), static_cast<double *>(iz.ptr), 0));
    };
    m.def("cov_accum_zmap", accum_zmap, py::arg("nsub"), py::arg("nsubpix"), py::arg("nnz"), py::arg("submap"),
          py::arg("subpix"), py::arg("weights"), py::arg("scale"), py::arg("tod"), py::arg("zmap"));
    m.def("cov_accum_diag", [accum_zmap](int64_t nsub, int64_t nsubpix, int64_t nnz, py::buffer submap, py::buffer subpix,
                                         py::buffer weights, double scale, py::buffer tod, py::buffer invnpp,
                                         py::buffer hits, py::buffer zmap) {
        auto ism = submap.request();
        auto ipx = subpix.request();
        auto iw = weights.request();
        auto ic = invnpp.request();
        auto ih = hits.request();
        if (norm_format(ism.format) != "q" || norm_format(ipx.format) != "q" || norm_format(ih.format) != "q") {
            throw std::runtime_error("cov_accum_diag: index / hit buffers must be int64");
        }
        if (norm_format(iw.format) != "d" || norm_format(ic.format) != "d") {
            throw std::runtime_error("cov_accum_diag: weights / invnpp must be float64");
        }
        if (ipx.size != ism.size || (size_t)(iw.size / nnz) != (size_t)ism.size) {
            throw std::runtime_error("Buffer sizes are not consistent.");
        }
        accum_zmap(nsub, nsubpix, nnz, submap, subpix, weights, scale, tod, zmap);
        check(toast_hip_cov_accum_diag_invnpp(nsub, nsubpix, nnz, (int64_t)ism.size, static_cast<int64_t *>(ism.ptr),
                                              static_cast<int64_t *>(ipx.ptr), static_cast<double *>(iw.ptr), scale,
                                              static_cast<double *>(ic.ptr), 0));
        check(toast_hip_cov_accum_diag_hits(nsub, nsubpix, nnz, (int64_t)ism.size, static_cast<int64_t *>(ism.ptr),
                                            static_cast<int64_t *>(ipx.ptr), static_cast<int64_t *>(ih.ptr), 0));
    }, py::arg("nsub"), py::arg("nsubpix"), py::arg("nnz"), py::arg("submap"), py::arg("subpix"), py::arg("weights"),
       py::arg("scale"), py::arg("tod"), py::arg("invnpp"), py::arg("hits"), py::arg("zmap"));

    m.def("global_to_local", [](py::array_t<int64_t, py::array::c_style | py::array::forcecast> global_pixels,
                                size_t npix_submap,
                                py::array_t<int64_t, py::array::c_style | py::array::forcecast> global2local) {
        auto ig = global_pixels.request();
        auto it = global2local.request();
        const int64_t n = (int64_t)ig.size;
        py::array_t<int64_t> local_submaps(n), local_pixels(n);
        if (n > 0) {
            check(toast_hip_global_to_local(n, static_cast<int64_t *>(ig.ptr), (int64_t)npix_submap,
                                            static_cast<int64_t *>(it.ptr), (int64_t)it.size,
                                            static_cast<int64_t *>(local_submaps.request().ptr),
                                            static_cast<int64_t *>(local_pixels.request().ptr), 0));
        }
        return py::make_tuple(local_submaps, local_pixels);
    }, py::arg("global_pixels"), py::arg("npix_submap"), py::arg("global2local"));

    m.def("cov_mult_diag", [](int64_t nsub, int64_t nsubpix, int64_t nnz, py::buffer data1, py::buffer data2,
                              bool use_accel) {
        auto i1 = data1.request();
        auto i2 = data2.request();
        if (norm_format(i1.format) != "d" || norm_format(i2.format) != "d") {
            throw std::runtime_error("cov_mult_diag: buffers must be float64");
        }
        if (i1.size != i2.size) {
            throw std::runtime_error("Buffer sizes are not consistent.");
        }
        check(toast_hip_cov_mult_diag(nsub, nsubpix, nnz, static_cast<double *>(i1.ptr),
                                      static_cast<double *>(i2.ptr), use_accel));
    }, py::arg("nsub"), py::arg("nsubpix"), py::arg("nnz"), py::arg("data1"), py::arg("data2"),
       py::arg("use_accel") = false);

    // ---- offset template (template_offset.cpp:16-25, :149-162, :334-340)
    m.def("template_offset_add_to_signal", [](int64_t step_length, int64_t amp_offset, py::buffer n_amp_views,
                                              py::buffer amplitudes, py::buffer amplitude_flags,
                                              int32_t data_index, py::buffer det_data, py::buffer intervals,
                                              bool use_accel) {
        Shape s(3);
        double * amps = extract<double>(amplitudes, "amplitudes", 1, s, {-1});
        const int64_t n_amp = s[0];
        uint8_t * af = extract<uint8_t>(amplitude_flags, "amplitude_flags", 1, s, {n_amp});
        double * tod = extract<double>(det_data, "det_data", 2, s, {-1, -1});
        const int64_t n_rows = s[0], n_samp = s[1];
        toast_hip_interval * ivl = extract_intervals(intervals, s);
        const int64_t n_view = s[0];
        int64_t * nav = extract<int64_t>(n_amp_views, "n_amp_views", 1, s, {n_view});
        check(toast_hip_template_offset_add_to_signal(step_length, amp_offset, nav, amps, af, n_amp, data_index, tod,
                                                      n_rows, n_samp, ivl, n_view, use_accel));
    });
    m.def("template_offset_project_signal", [](int32_t data_index, py::buffer det_data, int32_t flag_index,
                                               py::buffer flag_data, uint8_t flag_mask, int64_t step_length,
                                               int64_t amp_offset, py::buffer n_amp_views, py::buffer amplitudes,
                                               py::buffer amplitude_flags, py::buffer intervals, bool use_accel) {
        Shape s(3);
        double * amps = extract<double>(amplitudes, "amplitudes", 1, s, {-1});
        const int64_t n_amp = s[0];
        uint8_t * af = extract<uint8_t>(amplitude_flags, "amplitude_flags", 1, s, {n_amp});
        double * tod = extract<double>(det_data, "det_data", 2, s, {-1, -1});
        const int64_t n_rows = s[0], n_samp = s[1];
        toast_hip_interval * ivl = extract_intervals(intervals, s);
        const int64_t n_view = s[0];
        int64_t * nav = extract<int64_t>(n_amp_views, "n_amp_views", 1, s, {n_view});
        uint8_t * fd = nullptr;
        int64_t n_f_rows = 0;
        if (flag_index >= 0) {
            fd = extract<uint8_t>(flag_data, "flag_data", 2, s, {-1, n_samp});
            n_f_rows = s[0];
        }
        check(toast_hip_template_offset_project_signal(data_index, tod, n_rows, flag_index, fd, n_f_rows, flag_mask,
                                                       step_length, amp_offset, nav, amps, af, n_amp, n_samp, ivl,
                                                       n_view, use_accel));
    });
    m.def("template_offset_apply_diag_precond", [](py::buffer offset_var, py::buffer amplitudes_in,
                                                   py::buffer amplitude_flags, py::buffer amplitudes_out,
                                                   bool use_accel) {
        Shape s(3);
        double * in = extract<double>(amplitudes_in, "amplitudes_in", 1, s, {-1});
        const int64_t n_amp = s[0];
        double * out = extract<double>(amplitudes_out, "amplitudes_out", 1, s, {n_amp});
        double * var = extract<double>(offset_var, "offset_var", 1, s, {n_amp});
        uint8_t * af = extract<uint8_t>(amplitude_flags, "amplitude_flags", 1, s, {n_amp});
        check(toast_hip_template_offset_apply_diag_precond(var, in, af, out, n_amp, use_accel));
    });
    // ---- extensions without a toast._libtoast counterpart
    m.def("accel_device_ptr", [](py::buffer data) {
        RawBuf b = accel_buf(data);
        void * dev = nullptr;
        check(toast_hip_accel_device_ptr(b.ptr, &dev));
        return reinterpret_cast<uintptr_t>(dev);
    });
    // ---- the process' RCCL communicator on registered (device-resident) buffers: what PixelData.sync_allreduce /
    //      sync_alltoallv and covariance_*(use_alltoallv=True) call when their data is on the accelerator
    //      (INTEGRATION.md "Map reductions over several GPUs")
    m.def("comm_unique_id", []() {
        char id[TOAST_HIP_COMM_ID_BYTES];
        check(toast_hip_comm_unique_id(id));
        return py::bytes(id, sizeof(id));
    });
    m.def("comm_init", [](py::bytes id, int n_ranks, int rank) {
        const std::string s = id;
        if (s.size() != TOAST_HIP_COMM_ID_BYTES) throw std::runtime_error("comm_init: the RCCL unique id has 128 bytes");
        check(toast_hip_comm_init(s.data(), n_ranks, rank));
    });
    m.def("comm_destroy", []() { check(toast_hip_comm_destroy()); });
    m.def("comm_info", []() {
        int n = 0, r = -1, v = 0;
        check(toast_hip_comm_info(&n, &r, &v));
        return py::make_tuple(n, r, v);
    });
    m.def("comm_allreduce", [](py::buffer data, std::string op) {
        auto info = data.request();
        const std::string f = norm_format(info.format);
        int dt = -1;
        if (f == "d") dt = TOAST_HIP_COMM_F64;
        else if (f == "f") dt = TOAST_HIP_COMM_F32;
        else if (f == "q") dt = TOAST_HIP_COMM_I64;
        else if (f == "i") dt = TOAST_HIP_COMM_I32;
        else if (f == "B") dt = TOAST_HIP_COMM_U8;
        else throw std::runtime_error("comm_allreduce: unsupported element type " + info.format);
        const int o = (op == "sum") ? TOAST_HIP_COMM_SUM : (op == "max") ? TOAST_HIP_COMM_MAX
                      : (op == "min") ? TOAST_HIP_COMM_MIN : -1;
        if (o < 0) throw std::runtime_error("comm_allreduce: op must be sum, max or min");
        void * dev = nullptr;
        check(toast_hip_accel_device_ptr(info.ptr, &dev));
        check(toast_hip_comm_allreduce_dev(dev, (int64_t)info.size, dt, o, nullptr));
    }, py::arg("data"), py::arg("op") = "sum");
    m.def("comm_map_reduce_apply", [](py::object cov, py::buffer map, int64_t nnz, bool reduce) {
        auto im = map.request();
        if (norm_format(im.format) != "d") throw std::runtime_error("comm_map_reduce_apply: the map must be float64");
        void * d_map = nullptr;
        void * d_cov = nullptr;
        check(toast_hip_accel_device_ptr(im.ptr, &d_map));
        if (!cov.is_none()) {
            auto ic = cov.cast<py::buffer>().request();
            if (norm_format(ic.format) != "d" || ic.size / (nnz * (nnz + 1) / 2) != im.size / nnz) {
                throw std::runtime_error("comm_map_reduce_apply: covariance and map sizes are not consistent");
            }
            check(toast_hip_accel_device_ptr(ic.ptr, &d_cov));
        }
        check(toast_hip_comm_map_reduce_apply_dev((int64_t)(im.size / nnz), nnz, static_cast<const double *>(d_cov),
                                                  static_cast<double *>(d_map), reduce ? 1 : 0, nullptr));
    }, py::arg("cov"), py::arg("map"), py::arg("nnz"), py::arg("reduce") = true);

    m.def("accel_adopt", [](py::buffer data, uintptr_t device, std::string name) {
        RawBuf b = accel_buf(data);
        check(toast_hip_accel_adopt(b.ptr, b.nbytes, reinterpret_cast<void *>(device), name.c_str()));
    });

    // ---- hit map / inverse covariance accumulation (operator-level kernels of BuildHitMap and
    //      BuildInverseCovariance, src/toast/ops/mapmaker_utils/mapmaker_utils.py:100-210, :352-520)
    auto build_cov = [](int mode, py::buffer global2local, py::buffer out, py::buffer pixel_index,
                        py::buffer pixels, py::buffer weight_index, py::buffer weights, py::buffer flag_index,
                        py::buffer det_flags, py::buffer det_scale, uint8_t det_flag_mask, py::buffer intervals,
                        py::buffer shared_flags, uint8_t shared_flag_mask, bool use_accel) {
        Shape s(3);
        int32_t * p_idx = extract<int32_t>(pixel_index, "pixel_index", 1, s, {-1});
        const int64_t n_det = s[0];
        int64_t * pix = extract<int64_t>(pixels, "pixels", 2, s, {-1, -1});
        const int64_t n_p_rows = s[0], n_samp = s[1];
        int32_t * f_idx = extract<int32_t>(flag_index, "flag_index", 1, s, {n_det});
        toast_hip_interval * ivl = extract_intervals(intervals, s);
        const int64_t n_view = s[0];
        int64_t * g2l = extract<int64_t>(global2local, "global2local", 1, s, {-1});
        const int64_t n_submap = s[0];
        uint8_t * sf = extract<uint8_t>(shared_flags, "flags", 1, s, {-1});
        const int64_t n_sf = s[0];
        uint8_t * df = extract<uint8_t>(det_flags, "det_flags", 2, s, {-1, -1});
        const int64_t n_f_rows = s[0], n_f_samp = s[1];
        int32_t * w_idx = nullptr;
        double * w = nullptr;
        double * dscale = nullptr;
        int64_t nnz = 1, n_w_rows = 0, n_local, nps;
        void * outp;
        if (mode == 0) {
            outp = extract<int64_t>(out, "hits", 3, s, {-1, -1, 1});
            n_local = s[0];
            nps = s[1];
        } else {
            w_idx = extract<int32_t>(weight_index, "weight_index", 1, s, {n_det});
            if (weights.request().ndim == 2) {
                w = extract<double>(weights, "weights", 2, s, {-1, n_samp});
                nnz = 1;
            } else {
                w = extract<double>(weights, "weights", 3, s, {-1, n_samp, -1});
                nnz = s[2];
            }
            n_w_rows = s[0];
            dscale = extract<double>(det_scale, "det_scale", 1, s, {n_det});
            outp = extract<double>(out, "invcov", 3, s, {-1, -1, nnz * (nnz + 1) / 2});
            n_local = s[0];
            nps = s[1];
        }
        check(toast_hip_build_cov(mode, g2l, n_submap, outp, n_local, nps, nnz, p_idx, pix, n_p_rows, w_idx, w,
                                  n_w_rows, f_idx, df, n_f_rows, n_f_samp, dscale, det_flag_mask, n_det, n_samp,
                                  ivl, n_view, sf, n_sf, shared_flag_mask, use_accel));
    };
    m.def("build_hit_map", [build_cov](py::buffer global2local, py::buffer hits, py::buffer pixel_index,
                                       py::buffer pixels, py::buffer flag_index, py::buffer det_flags,
                                       uint8_t det_flag_mask, py::buffer intervals, py::buffer shared_flags,
                                       uint8_t shared_flag_mask, bool use_accel) {
        build_cov(0, global2local, hits, pixel_index, pixels, pixel_index, pixels, flag_index, det_flags,
                  pixels, det_flag_mask, intervals, shared_flags, shared_flag_mask, use_accel);
    });
    m.def("build_inverse_covariance", [build_cov](py::buffer global2local, py::buffer invcov,
                                                  py::buffer pixel_index, py::buffer pixels,
                                                  py::buffer weight_index, py::buffer weights,
                                                  py::buffer flag_index, py::buffer det_flags,
                                                  py::buffer det_scale, uint8_t det_flag_mask,
                                                  py::buffer intervals, py::buffer shared_flags,
                                                  uint8_t shared_flag_mask, bool use_accel) {
        build_cov(1, global2local, invcov, pixel_index, pixels, weight_index, weights, flag_index, det_flags,
                  det_scale, det_flag_mask, intervals, shared_flags, shared_flag_mask, use_accel);
    });
    // (not a reference binding) BuildInverseCovariance and BuildHitMap of one CovarianceAndHits pass in one call
    m.def("build_inverse_covariance_and_hits", [](py::buffer global2local, py::buffer invcov, py::buffer hits,
                                                  py::buffer pixel_index, py::buffer pixels,
                                                  py::buffer weight_index, py::buffer weights,
                                                  py::buffer flag_index, py::buffer det_flags,
                                                  py::buffer det_scale, uint8_t det_flag_mask,
                                                  py::buffer intervals, py::buffer shared_flags,
                                                  uint8_t shared_flag_mask, bool use_accel) {
        Shape s(3);
        int32_t * p_idx = extract<int32_t>(pixel_index, "pixel_index", 1, s, {-1});
        const int64_t n_det = s[0];
        int64_t * pix = extract<int64_t>(pixels, "pixels", 2, s, {-1, -1});
        const int64_t n_p_rows = s[0], n_samp = s[1];
        int32_t * f_idx = extract<int32_t>(flag_index, "flag_index", 1, s, {n_det});
        toast_hip_interval * ivl = extract_intervals(intervals, s);
        const int64_t n_view = s[0];
        int64_t * g2l = extract<int64_t>(global2local, "global2local", 1, s, {-1});
        const int64_t n_submap = s[0];
        uint8_t * sf = extract<uint8_t>(shared_flags, "flags", 1, s, {-1});
        const int64_t n_sf = s[0];
        uint8_t * df = extract<uint8_t>(det_flags, "det_flags", 2, s, {-1, -1});
        const int64_t n_f_rows = s[0], n_f_samp = s[1];
        int32_t * w_idx = extract<int32_t>(weight_index, "weight_index", 1, s, {n_det});
        double * w;
        int64_t nnz = 1;
        if (weights.request().ndim == 2) {
            w = extract<double>(weights, "weights", 2, s, {-1, n_samp});
        } else {
            w = extract<double>(weights, "weights", 3, s, {-1, n_samp, -1});
            nnz = s[2];
        }
        const int64_t n_w_rows = s[0];
        double * dscale = extract<double>(det_scale, "det_scale", 1, s, {n_det});
        double * cov = extract<double>(invcov, "invcov", 3, s, {-1, -1, nnz * (nnz + 1) / 2});
        const int64_t n_local = s[0], nps = s[1];
        int64_t * hp = extract<int64_t>(hits, "hits", 3, s, {n_local, nps, 1});
        check(toast_hip_build_cov_hits(g2l, n_submap, cov, hp, n_local, nps, nnz, p_idx, pix, n_p_rows, w_idx, w,
                                       n_w_rows, f_idx, df, n_f_rows, n_f_samp, dscale, det_flag_mask, n_det, n_samp,
                                       ivl, n_view, sf, n_sf, shared_flag_mask, use_accel));
    });
    // map_cov.cpp:269-325
    m.def("cov_eigendecompose_diag", [](int64_t nsub, int64_t nsubpix, int64_t nnz, py::buffer data,
                                        py::buffer cond, double threshold, bool invert, bool use_accel) {
        auto id = data.request();
        auto ic = cond.request();
        if (norm_format(id.format) != "d" || norm_format(ic.format) != "d") {
            throw std::runtime_error("cov_eigendecompose_diag: buffers must be float64");
        }
        const int64_t block = nnz * (nnz + 1) / 2;
        if ((int64_t)id.size != nsub * nsubpix * block || (int64_t)ic.size != nsub * nsubpix) {
            throw std::runtime_error("cov_eigendecompose_diag: buffer sizes are not consistent");
        }
        check(toast_hip_cov_eigendecompose_diag(nsub, nsubpix, nnz, static_cast<double *>(id.ptr),
                                                static_cast<double *>(ic.ptr), threshold, invert, use_accel));
    }, py::arg("nsub"), py::arg("nsubpix"), py::arg("nnz"), py::arg("data"), py::arg("cond"),
       py::arg("threshold"), py::arg("invert"), py::arg("use_accel") = false);

    // math_fft.cpp:8-175
    py::enum_<FFTPlanType>(m, "FFTPlanType", "FFT Plan Type")
        .value("fast", FFTPlanType::fast)
        .value("best", FFTPlanType::best);
    py::enum_<FFTDirection>(m, "FFTDirection", "FFT Direction")
        .value("forward", FFTDirection::forward)
        .value("backward", FFTDirection::backward);
    py::class_<HipFFTPlanReal1D, HipFFTPlanReal1D::pshr>(m, "FFTPlanReal1D")
        .def_static("create", &HipFFTPlanReal1D::create, py::arg("length"), py::arg("n"), py::arg("type"),
                    py::arg("dir"), py::arg("scale"))
        .def("exec", &HipFFTPlanReal1D::exec)
        .def("length", &HipFFTPlanReal1D::length)
        .def("count", &HipFFTPlanReal1D::count)
        .def("tdata", [](HipFFTPlanReal1D & self, int64_t indx) {
            return py::array_t<double>({self.length()}, {sizeof(double)}, self.tdata(indx), py::cast(self));
        }, py::return_value_policy::reference_internal)
        .def("fdata", [](HipFFTPlanReal1D & self, int64_t indx) {
            return py::array_t<double>({self.length()}, {sizeof(double)}, self.fdata(indx), py::cast(self));
        }, py::return_value_policy::reference_internal);
    py::class_<HipFFTPlanReal1DStore, std::unique_ptr<HipFFTPlanReal1DStore, py::nodelete>>(m, "FFTPlanReal1DStore")
        .def("get", []() {
            return std::unique_ptr<HipFFTPlanReal1DStore, py::nodelete>(&HipFFTPlanReal1DStore::get());
        })
        .def("clear", &HipFFTPlanReal1DStore::clear)
        .def("cache", &HipFFTPlanReal1DStore::cache, py::arg("length"), py::arg("n"))
        .def("forward", &HipFFTPlanReal1DStore::forward, py::arg("length"), py::arg("n"))
        .def("backward", &HipFFTPlanReal1DStore::backward, py::arg("length"), py::arg("n"));
}
