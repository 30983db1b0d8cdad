// Build glue for oracle/_ref: a pybind11 module that registers the reference's OWN
// binding initialisers for the map-making hot path.  No reference source is copied:
// the init_* functions are defined in the reference files compiled in place from
// /root/reference (see ref_build.sh).  Test infrastructure only.
#include <pybind11/pybind11.h>
namespace py = pybind11;

// Defined in /root/reference/src/toast/_libtoast/{intervals,accelerator,ops_*,template_offset}.cpp
void init_intervals(py::module &);
void init_accelerator(py::module &);
void init_ops_pointing_detector(py::module &);
void init_ops_stokes_weights(py::module &);
void init_ops_pixels_healpix(py::module &);
void init_ops_mapmaker_utils(py::module &);
void init_ops_noise_weight(py::module &);
void init_ops_scan_map(py::module &);
void init_template_offset(py::module &);
void init_tod_filter(py::module &);   // tod_filter.cpp: legendre_templates, bin_proj, bin_invcov, add_templates
void init_map_cov(py::module &);      // map_cov.cpp: cov_accum_*, cov_apply_diag (LAPACK-free parts)

PYBIND11_MODULE(_toast_ref, m) {
    m.doc() = "hpc4cmb/toast hot-path bindings compiled in place (parity oracle, tests only)";
    init_intervals(m);
    init_accelerator(m);
    init_ops_pointing_detector(m);
    init_ops_stokes_weights(m);
    init_ops_pixels_healpix(m);
    init_ops_mapmaker_utils(m);
    init_ops_noise_weight(m);
    init_ops_scan_map(m);
    init_template_offset(m);
    init_tod_filter(m);
    init_map_cov(m);
}
