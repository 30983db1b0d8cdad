"""GPU: the Operator mirror (toast_amd.ops) run the way the reference's own operator tests do
(src/toast/tests/ops_pointing_healpix.py, ops_mapmaker_utils.py, ops_mapmaker_binning.py,
ops_scan_map.py, ops_mapmaker_solve.py, ops_mapmaker.py): against pure-Python loops, against
each other (accelerator-resident vs host-staged), and against the CPU oracle."""
import os

import numpy as np
import pytest

from toast_amd import ops
from toast_amd.data import defaults
from toast_amd.pixels import PixelData, covariance_apply
from toast_amd.sim import create_satellite_data
from toast_amd.templates import Offset

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def device():
    from toast_amd import accel

    assert accel.accel_enabled()
    accel.accel_assign_device(1, 0, 1.0, False)


def pointing_ops(nside=64, nest=True, mode="IQU", hwp=True, create_dist=None):
    dp = ops.PointingDetectorSimple()
    pix = ops.PixelsHealpix(detector_pointing=dp, nside=nside, nest=nest, create_dist=create_dist)
    sw = ops.StokesWeights(detector_pointing=dp, mode=mode, hwp_angle=defaults.hwp_angle if hwp else None)
    return dp, pix, sw


def test_pixels_and_weights_operators_vs_oracle(oracle):
    data = create_satellite_data(n_det=6, n_samp=4000, flagged_pixels=True)
    dp, pix, sw = pointing_ops(nside=256, create_dist="dist")
    pix.apply(data)
    sw.apply(data)
    ob = data.obs[0]
    dets = ob.select_local_detectors(flagmask=dp.det_mask)
    assert len(dets) == 4  # pixel 1's two detectors are flagged at the detector level
    assert ob.detdata[defaults.pixels].detectors == dets
    fp = np.array([ob.telescope.focalplane[d]["quat"] for d in dets])
    n_samp = ob.n_local_samples
    idx = np.arange(len(dets), dtype=np.int32)
    quats = np.zeros((len(dets), n_samp, 4))
    ivl = ob.intervals[None].data
    sflags = ob.shared[defaults.shared_flags].data
    oracle.pointing_detector(fp, ob.shared[defaults.boresight_radec].data, idx, quats, ivl, sflags, 1)
    assert np.array_equal(ob.detdata[defaults.quats].data, quats)
    want = np.zeros((len(dets), n_samp), dtype=np.int64)
    hs = np.zeros(pix._n_submap, dtype=np.uint8)
    oracle.pixels_healpix(idx, quats, sflags, 1, idx, want, ivl, hs, 3072, 256, True)
    assert np.array_equal(ob.detdata[defaults.pixels].data, want)
    assert np.all(want[:, sflags != 0] == -1)
    assert list(data["dist"].local_submaps) == list(np.flatnonzero(hs))
    w = np.zeros((len(dets), n_samp, 3))
    gamma = np.array([ob.telescope.focalplane[d]["gamma"] for d in dets])
    oracle.stokes_weights_IQU(idx, quats, idx, w, ob.shared[defaults.hwp_angle].data, ivl, np.zeros(len(dets)), gamma,
                              np.ones(len(dets)), False)
    np.testing.assert_allclose(ob.detdata[defaults.weights].data, w, rtol=1e-12, atol=1e-14)
    # second exec: buffers exist -> kernels skipped, results unchanged (pixels_healpix.py:215-243)
    before = ob.detdata[defaults.pixels].data.copy()
    pix.apply(data)
    assert np.array_equal(ob.detdata[defaults.pixels].data, before)


def python_zmap(data, dist, det_mask=defaults.det_mask_nonscience, shared_mask=defaults.shared_mask_nonscience):
    """The pure-Python triple loop of the reference test (tests/ops_mapmaker_utils.py:300-355)."""
    z = np.zeros((dist.n_local_submap, dist.n_pix_submap, 3))
    for ob in data.obs:
        noise = ob[defaults.noise_model]
        sf = ob.shared[defaults.shared_flags].data
        for det in ob.select_local_detectors(flagmask=det_mask):
            wt = noise.detector_weight(det)
            pix = ob.detdata[defaults.pixels][det]
            w = ob.detdata[defaults.weights][det]
            sig = ob.detdata[defaults.det_data][det]
            df = ob.detdata[defaults.det_flags][det]
            for i in range(ob.n_local_samples):
                if pix[i] < 0 or (df[i] & det_mask) or (sf[i] & shared_mask):
                    continue
                sm = dist.global_submap_to_local[pix[i] // dist.n_pix_submap]
                z[sm, pix[i] % dist.n_pix_submap, :] += wt * sig[i] * w[i]
    return z


def fill_signal(data, seed=5):
    rng = np.random.default_rng(seed)
    for ob in data.obs:
        ob.detdata[defaults.det_data].data[:] = rng.standard_normal(ob.detdata[defaults.det_data].data.shape)


@pytest.mark.parametrize("use_accel", [None, False])
def test_build_noise_weighted_vs_python_loop(use_accel):
    """use_accel=None lets the Pipeline stage everything on the GPU (accelerator-resident
    kernels); False stages per call.  Both must equal the Python loop (atol 1e-6 in the
    reference; here 1e-12 relative)."""
    data = create_satellite_data(n_det=4, n_obs=2, n_samp=1500)
    fill_signal(data)
    dp, pix, sw = pointing_ops(nside=64, create_dist="dist")
    ops.Pipeline(operators=[pix, sw]).apply(data, use_accel=use_accel)
    build = ops.BuildNoiseWeighted(pixel_dist="dist", zmap="zmap")
    ops.Pipeline(operators=[build]).apply(data, use_accel=use_accel)
    want = python_zmap(data, data["dist"])
    got = data["zmap"].data
    assert not data["zmap"].accel_in_use()
    assert np.max(np.abs(got - want)) < 1e-12 * np.max(np.abs(want))
    # accumulate across calls: a second exec doubles the map (mapmaker_utils.py:706-773)
    ops.Pipeline(operators=[build]).apply(data, use_accel=use_accel)
    assert np.max(np.abs(data["zmap"].data - 2 * want)) < 1e-12 * np.max(np.abs(want))


def test_hits_covariance_and_binmap(monkeypatch):
    data = create_satellite_data(n_det=4, n_samp=3000)
    fill_signal(data)
    dp, pix, sw = pointing_ops(nside=32, create_dist=None)
    cov_op = ops.CovarianceAndHits(pixel_dist="dist", covariance="cov", hits="hits", rcond="rcond",
                                   inverse_covariance="invcov", pixel_pointing=pix, stokes_weights=sw,
                                   save_pointing=True, rcond_threshold=1e-6)
    cov_op.apply(data)
    dist = data["dist"]
    ob = data.obs[0]
    # hits / inverse covariance vs numpy (reference tests/ops_mapmaker_utils.py:44-208)
    hits = np.zeros((dist.n_local_submap, dist.n_pix_submap, 1), dtype=np.int64)
    inv = np.zeros((dist.n_local_submap, dist.n_pix_submap, 6))
    sf = ob.shared[defaults.shared_flags].data
    iu = np.triu_indices(3)
    for det in ob.select_local_detectors(flagmask=defaults.det_mask_nonscience):
        wt = ob[defaults.noise_model].detector_weight(det)
        p = ob.detdata[defaults.pixels][det]
        w = ob.detdata[defaults.weights][det]
        df = ob.detdata[defaults.det_flags][det]
        good = (p >= 0) & ((df & defaults.det_mask_nonscience) == 0) & ((sf & defaults.shared_mask_nonscience) == 0)
        sm = dist.global_submap_to_local[p[good] // dist.n_pix_submap]
        px = p[good] % dist.n_pix_submap
        np.add.at(hits[:, :, 0], (sm, px), 1)
        outer = wt * w[good][:, iu[0]] * w[good][:, iu[1]]
        np.add.at(inv, (sm, px), outer)
    assert np.array_equal(data["hits"].data, hits)
    assert np.max(np.abs(data["invcov"].data - inv)) < 1e-12 * np.max(np.abs(inv))
    # covariance = inverse where rcond >= threshold else 0 (toast_map_cov.cpp:246-396)
    cov = data["cov"].data.reshape(-1, 6)
    rc = data["rcond"].data.reshape(-1)
    invf = inv.reshape(-1, 6)
    n_good = 0
    for i in range(cov.shape[0]):
        m = np.zeros((3, 3))
        m[iu] = invf[i]
        m = m + m.T - np.diag(np.diag(m))
        ev = np.linalg.eigvalsh(m)
        r = ev[0] / ev[-1] if ev[-1] > 0 else 0.0
        if r >= 1e-6:
            n_good += 1
            # conditioning-aware bound: the inverse of a matrix with reciprocal condition number r is defined to
            # ~eps / r (LAPACK is absent from the reference build here -- it throws --, NumPy is the yardstick)
            want = np.linalg.inv(m)[iu]
            assert np.max(np.abs(cov[i] - want)) <= (64 * 2.2e-16 / r) * np.max(np.abs(want))
            assert rc[i] == pytest.approx(r, rel=64 * 2.2e-16 / r)
        else:
            assert np.all(cov[i] == 0) and rc[i] == 0
    assert n_good > 100
    # BinMap == manual BuildNoiseWeighted + covariance_apply (tests/ops_mapmaker_binning.py:27-127)
    binner = ops.BinMap(pixel_dist="dist", covariance="cov", binned="binned", pixel_pointing=pix, stokes_weights=sw,
                        full_pointing=True)
    binner.apply(data)
    ops.BuildNoiseWeighted(pixel_dist="dist", zmap="zcheck").apply(data)
    covariance_apply(data["cov"], data["zcheck"])
    assert np.max(np.abs(data["binned"].data - data["zcheck"].data)) < 1e-12 * np.max(np.abs(data["zcheck"].data))
    # full_pointing=False (SINGLE pipeline: pointing recomputed per detector) gives the same map
    for key in (defaults.pixels, defaults.weights, defaults.quats):
        if key in ob.detdata:   # no quaternion buffer when the expansion ran on the device
            del ob.detdata[key]
    binner2 = ops.BinMap(pixel_dist="dist", covariance="cov", binned="binned2", pixel_pointing=pix,
                         stokes_weights=sw, full_pointing=False)
    binner2.apply(data)
    assert np.max(np.abs(data["binned2"].data - data["binned"].data)) < 1e-12 * np.max(np.abs(data["binned"].data))
    # ... through the pointing-on-the-fly kernel: no pointing buffers at all
    assert defaults.pixels not in ob.detdata and defaults.weights not in ob.detdata
    # the reference's own sequence (scratch pointing recomputed per detector pass)
    monkeypatch.setenv("TOAST_HIP_POINTING_BATCH", "1")
    binner3 = ops.BinMap(pixel_dist="dist", covariance="cov", binned="binned3", pixel_pointing=pix,
                         stokes_weights=sw, full_pointing=False, on_the_fly=False)
    binner3.apply(data)
    assert np.max(np.abs(data["binned3"].data - data["binned"].data)) < 1e-12 * np.max(np.abs(data["binned"].data))
    assert ob.detdata[defaults.pixels].data.shape[0] == 1  # one-detector buffers were recycled


def test_scan_map_operator():
    """Scan, then subtract: exact zeros (tests/ops_scan_map.py:99-172); values vs Python loop."""
    data = create_satellite_data(n_det=2, n_samp=2000, flag_samples=False)
    dp, pix, sw = pointing_ops(nside=64, create_dist="dist")
    ops.Pipeline(operators=[pix, sw]).apply(data)
    dist = data["dist"]
    m = PixelData(dist, np.float64, n_value=3)
    m.raw[:] = np.random.default_rng(2).standard_normal(m.raw.size)
    data["sky"] = m
    scanner = ops.ScanMap(det_data=defaults.det_data, map_key="sky")
    scanner.apply(data)
    ob = data.obs[0]
    for det in ob.local_detectors:
        p = ob.detdata[defaults.pixels][det]
        w = ob.detdata[defaults.weights][det]
        want = np.zeros(ob.n_local_samples)
        for i in range(ob.n_local_samples):
            sm = dist.global_submap_to_local[p[i] // dist.n_pix_submap]
            want[i] = np.dot(w[i], m.data[sm, p[i] % dist.n_pix_submap])
        np.testing.assert_allclose(ob.detdata[defaults.det_data][det], want, rtol=1e-13, atol=1e-13)
    ops.ScanMap(det_data=defaults.det_data, map_key="sky", subtract=True).apply(data)
    assert np.all(ob.detdata[defaults.det_data].data == 0)


def make_solver_setup(n_det=4, n_samp=6000, step_time=20.0, seed=11, noise_rms=0.0):
    data = create_satellite_data(n_det=n_det, n_samp=n_samp, rate=10.0, spin_angle_deg=25.0, prec_angle_deg=35.0)
    dp, pix, sw = pointing_ops(nside=16, create_dist=None)
    pix.nside_submap = 4
    rng = np.random.default_rng(seed)
    # sky signal + per-detector baseline offsets
    ops.Pipeline(operators=[pix, sw]).apply(data)
    ob = data.obs[0]
    sky = rng.standard_normal((12 * 16 * 16, 3)) * np.array([1.0, 0.1, 0.1])
    truth = {}
    step = int(step_time * 10.0 + 0.5)
    for det in ob.local_detectors:
        p = ob.detdata[defaults.pixels][det]
        w = ob.detdata[defaults.weights][det]
        good = p >= 0
        sig = np.zeros(n_samp)
        sig[good] = np.einsum("ij,ij->i", w[good], sky[p[good]])
        n_amp = (n_samp + step - 1) // step
        offs = rng.standard_normal(n_amp) * 3.0
        truth[det] = offs
        sig += np.repeat(offs, step)[:n_samp]
        sig += noise_rms * rng.standard_normal(n_samp)
        ob.detdata[defaults.det_data][det] = sig
    for key in (defaults.pixels, defaults.weights, defaults.quats):
        if key in ob.detdata:
            del ob.detdata[key]
    return data, pix, sw, truth, sky


@pytest.mark.parametrize("full_pointing", [True, False])
def test_lhs_equals_rhs_of_projected_amplitudes(full_pointing):
    """LHS(a) == RHS(M a) without a prior (reference test_lhs, tests/ops_mapmaker_solve.py:151-265)."""
    data, pix, sw, truth, sky = make_solver_setup()
    ops.CovarianceAndHits(pixel_dist="dist", covariance="cov", pixel_pointing=pix, stokes_weights=sw,
                          save_pointing=full_pointing).apply(data)
    binner = ops.BinMap(pixel_dist="dist", covariance="cov", binned="solve_bin", pixel_pointing=pix,
                        stokes_weights=sw, full_pointing=full_pointing)
    tmpl = Offset(step_time=20.0, noise_model=defaults.noise_model, name="baselines")
    tmatrix = ops.TemplateMatrix(templates=[tmpl], amplitudes="amps_in", det_data="proj")
    # amplitudes a, and d = M a
    tmatrix.det_data = "proj"
    tmatrix.initialize(data)
    amps = tmpl.zeros()
    rng = np.random.default_rng(0)
    amps.local[:] = rng.standard_normal(amps.n_local)
    from toast_amd.templates import AmplitudesMap

    data["amps_in"] = AmplitudesMap(baselines=amps)
    tmatrix.transpose = False
    tmatrix.apply(data)  # proj = M a
    # RHS(M a)
    tm_rhs = tmatrix.duplicate()
    tm_rhs.amplitudes = "rhs_out"
    ops.SolverRHS(det_data="proj", binning=binner, template_matrix=tm_rhs).apply(data)
    # LHS(a)
    lhs_bin = ops.BinMap(pixel_dist="dist", covariance="cov", binned="lhs_bin", pixel_pointing=pix,
                         stokes_weights=sw, full_pointing=full_pointing)
    tm_lhs = tmatrix.duplicate()
    tm_lhs.amplitudes = "amps_in"
    data["lhs_out"] = data["amps_in"].duplicate()
    data["lhs_out"].reset()
    ops.SolverLHS(binning=lhs_bin, template_matrix=tm_lhs, out="lhs_out").apply(data)
    a = data["rhs_out"]["baselines"].local
    b = data["lhs_out"]["baselines"].local
    assert np.max(np.abs(a)) > 0
    np.testing.assert_allclose(b, a, rtol=1e-9, atol=1e-10 * np.max(np.abs(a)))


def test_mapmaker_eager_and_lazy_host_coherence_agree():
    """Data.lazy_host = False (the reference's copy-back / delete at the end of every Pipeline; also
    TOAST_HIP_LAZY_HOST=0) must give the products of the default lazy mode: solver flags, covariances, right-hand
    side, amplitudes, cleaned timestreams and maps agree to rounding."""
    def run(lazy):
        data, pix, sw, truth, sky = make_solver_setup(noise_rms=0.0)
        data.lazy_host = lazy
        binner = ops.BinMap(pixel_dist="dist", pixel_pointing=pix, stokes_weights=sw, full_pointing=True)
        tmpl = Offset(step_time=20.0, noise_model=defaults.noise_model, name="baselines", good_fraction=0.2)
        mapper = ops.MapMaker(name="mm", keep_solver_products=True, det_data=defaults.det_data, binning=binner,
                              template_matrix=ops.TemplateMatrix(templates=[tmpl]), iter_max=200, convergence=1e-20,
                              solve_rcond_threshold=1e-3, map_rcond_threshold=1e-3, save_cleaned=True)
        mapper.apply(data)
        out = {k: np.array(data[k].raw, dtype=np.float64)
               for k in ("mm_solve_hits", "mm_solve_cov", "mm_solve_rcond_mask", "mm_solve_bin", "mm_hits", "mm_cov",
                         "mm_map", "mm_rcond")}
        out["amps"] = np.array(data["mm_solve_amplitudes"]["baselines"].local)
        out["rhs"] = np.array(data["mm_solve_rhs"]["baselines"].local)
        ob = data.obs[0]
        for k in ("mm_solve_flags", "mm_cleaned", defaults.det_data):
            out["dd_" + k] = np.array(ob.detdata[k].data, dtype=np.float64)
        return out

    lazy, eager = run(True), run(False)
    for k, a in lazy.items():
        b = eager[k]
        assert a.shape == b.shape, k
        scale = max(np.max(np.abs(a)), 1e-30)
        assert np.max(np.abs(a - b)) <= 1e-12 * scale + 1e-15, k
    assert np.array_equal(lazy["dd_mm_solve_flags"], eager["dd_mm_solve_flags"])
    assert np.array_equal(lazy["mm_hits"], eager["mm_hits"])


@pytest.mark.parametrize("full_pointing", [True, False])
def test_mapmaker_with_cut_detectors(full_pointing):
    """Detectors cut by their per-detector flags (Observation.local_detector_flags & det_mask) take no part: hits,
    maps and the surviving detectors' amplitudes equal those of a run over data that holds only the surviving
    detectors (the reference selects detectors the same way: observation.py select_local_detectors).  A second
    observation in which every detector is cut contributes nothing."""
    def run(cut):
        data, pix, sw, truth, sky = make_solver_setup(n_det=6, noise_rms=0.3)
        ob = data.obs[0]
        keep = list(ob.local_detectors)
        if cut:
            ob.update_local_detector_flags({ob.local_detectors[1]: 1, ob.local_detectors[4]: 1})
            keep = [d for i, d in enumerate(keep) if i not in (1, 4)]
            # a whole observation without a valid detector
            extra, *_ = make_solver_setup(n_det=6, noise_rms=0.3, seed=5)
            ob2 = extra.obs[0]
            ob2.name = "all_cut"
            ob2.update_local_detector_flags({d: 1 for d in ob2.local_detectors})
            data.obs.append(ob2)
        binner = ops.BinMap(pixel_dist="dist", pixel_pointing=pix, stokes_weights=sw, full_pointing=full_pointing)
        tmpl = Offset(step_time=20.0, noise_model=defaults.noise_model, name="baselines", good_fraction=0.2)
        mapper = ops.MapMaker(name="mm", det_data=defaults.det_data, binning=binner, keep_solver_products=True,
                              template_matrix=ops.TemplateMatrix(templates=[tmpl]), iter_max=100, convergence=1e-18)
        dets = None if cut else [d for i, d in enumerate(ob.local_detectors) if i not in (1, 4)]
        mapper.apply(data, detectors=dets)
        amps = data["mm_solve_amplitudes"]["baselines"]
        per_det = {}
        for d in (keep if cut else dets):
            first = tmpl._det_start[d]
            per_det[d] = np.array(amps.local[first:first + int(np.sum(tmpl._obs_views[0]))])
        dist = data["dist"]
        full = {}
        for key in ("mm_hits", "mm_map"):
            m = np.zeros((dist.n_submap, dist.n_pix_submap, data[key].n_value))
            m[dist.local_submaps] = data[key].data
            full[key] = m
        return full, per_det

    cut_maps, cut_amps = run(True)
    sel_maps, sel_amps = run(False)
    assert np.array_equal(cut_maps["mm_hits"], sel_maps["mm_hits"]) and cut_maps["mm_hits"].sum() > 0
    scale = np.max(np.abs(sel_maps["mm_map"]))
    assert np.max(np.abs(cut_maps["mm_map"] - sel_maps["mm_map"])) < 1e-9 * scale
    assert set(cut_amps) == set(sel_amps) and len(cut_amps) == 4
    for d in sel_amps:
        assert np.max(np.abs(cut_amps[d] - sel_amps[d])) < 1e-8 * np.max(np.abs(sel_amps[d]))


@pytest.mark.parametrize("full_pointing", [True, False])
def test_mapmaker_with_an_empty_view(full_pointing):
    """An observation whose view has no interval at all contributes nothing: MapMaker over [observation,
    observation with an empty "scan" view] equals MapMaker over the first one alone (hits exactly, maps to rounding)."""
    from toast_amd.data import IntervalList
    from toast_amd.synth import interval_dtype

    def run(with_empty):
        data, pix, sw, truth, sky = make_solver_setup(n_det=4, noise_rms=0.3)
        ob = data.obs[0]
        ob.intervals["scan"] = IntervalList(data=ob.intervals[None].data.copy())
        if with_empty:
            extra, *_ = make_solver_setup(n_det=4, noise_rms=0.3, seed=5)
            ob2 = extra.obs[0]
            ob2.name = "empty_view"
            ob2.intervals["scan"] = IntervalList(data=np.zeros(0, dtype=interval_dtype))
            data.obs.append(ob2)
        pix.view = "scan"
        sw.view = "scan"
        pix.detector_pointing.view = "scan"
        binner = ops.BinMap(pixel_dist="dist", pixel_pointing=pix, stokes_weights=sw, full_pointing=full_pointing)
        tmpl = Offset(step_time=20.0, noise_model=defaults.noise_model, name="baselines", good_fraction=0.2,
                      view="scan")
        mapper = ops.MapMaker(name="mm", det_data=defaults.det_data, binning=binner, keep_solver_products=True,
                              template_matrix=ops.TemplateMatrix(templates=[tmpl], view="scan"), iter_max=100,
                              convergence=1e-18)
        mapper.apply(data)
        dist = data["dist"]
        full = {}
        for key in ("mm_hits", "mm_map"):
            m = np.zeros((dist.n_submap, dist.n_pix_submap, data[key].n_value))
            m[dist.local_submaps] = data[key].data
            full[key] = m
        return full

    both, alone = run(True), run(False)
    assert np.array_equal(both["mm_hits"], alone["mm_hits"]) and alone["mm_hits"].sum() > 0
    assert np.max(np.abs(both["mm_map"] - alone["mm_map"])) < 1e-9 * np.max(np.abs(alone["mm_map"]))


def test_repeated_runs_reuse_existing_products():
    """Operators applied a second time to the same Data (existing pixel distribution, covariances, maps, resident
    buffers) reproduce the first run; a binned map that the host scribbled over in between is reset, not accumulated
    into."""
    data, pix, sw, truth, sky = make_solver_setup(n_det=4, noise_rms=0.3)
    binner = ops.BinMap(pixel_dist="dist", pixel_pointing=pix, stokes_weights=sw, full_pointing=True)
    tmpl = Offset(step_time=20.0, noise_model=defaults.noise_model, name="baselines", good_fraction=0.2)
    mapper = ops.MapMaker(name="mm", det_data=defaults.det_data, binning=binner, keep_solver_products=True,
                          template_matrix=ops.TemplateMatrix(templates=[tmpl]), iter_max=100, convergence=1e-18)
    mapper.apply(data)
    first = {k: np.array(data[k].data) for k in ("mm_hits", "mm_cov", "mm_map", "mm_rcond")}
    amps1 = np.array(data["mm_solve_amplitudes"]["baselines"].local)
    mapper.apply(data)
    for k, v in first.items():
        got = np.array(data[k].data)
        if k == "mm_hits":
            assert np.array_equal(got, v)
        else:
            assert np.max(np.abs(got - v)) <= 1e-10 * np.max(np.abs(v)), k
    assert np.max(np.abs(np.array(data["mm_solve_amplitudes"]["baselines"].local) - amps1)) < 1e-8 * np.max(np.abs(amps1))
    # BinMap alone, three ways to find its output: resident and device-current, host-current after a host write,
    # freshly deleted device copy
    bm = ops.BinMap(pixel_dist="dist", covariance="mm_cov", binned="again", pixel_pointing=pix, stokes_weights=sw,
                    det_data=defaults.det_data, full_pointing=True)
    bm.apply(data)
    want = np.array(data["again"].data)
    assert np.max(np.abs(want)) > 0
    bm.apply(data)
    assert np.max(np.abs(np.array(data["again"].data) - want)) <= 1e-12 * np.max(np.abs(want))
    data["again"].data[:] = 1.0e6                      # host becomes the current side, full of garbage
    bm.apply(data)
    assert np.max(np.abs(np.array(data["again"].data) - want)) <= 1e-12 * np.max(np.abs(want))
    data["again"].data[:] = -3.0
    if data["again"].accel_exists():
        data["again"].accel_delete()
    bm.apply(data)
    assert np.max(np.abs(np.array(data["again"].data) - want)) <= 1e-12 * np.max(np.abs(want))


def test_solver_rhs_fused_equals_sequence_and_accumulates():
    """SolverRHS with the one-pass tail (toast_hip_offset_scan_project_signal_dev) against the reference's operator
    sequence: same amplitudes to rounding, the timestreams untouched, and -- like TemplateMatrix(transpose) -- a second
    application adds to the existing amplitudes."""
    out = {}
    for fused in (False, True):
        data, pix, sw, truth, sky = make_solver_setup(n_det=4, noise_rms=0.3)
        ops.CovarianceAndHits(pixel_dist="dist", covariance="cov", pixel_pointing=pix, stokes_weights=sw,
                              save_pointing=True).apply(data)
        binner = ops.BinMap(pixel_dist="dist", covariance="cov", binned="bin", pixel_pointing=pix, stokes_weights=sw,
                            full_pointing=True)
        tmpl = Offset(step_time=13.0, noise_model=defaults.noise_model, name="baselines", good_fraction=0.2)
        tm = ops.TemplateMatrix(templates=[tmpl], amplitudes="rhs")
        before = data.obs[0].detdata[defaults.det_data].data.copy()
        rhs = ops.SolverRHS(det_data=defaults.det_data, binning=binner, template_matrix=tm, fused=fused)
        rhs.apply(data)
        once = np.array(data["rhs"]["baselines"].local)
        rhs.apply(data)
        twice = np.array(data["rhs"]["baselines"].local)
        assert np.array_equal(data.obs[0].detdata[defaults.det_data].data, before)
        out[fused] = (once, twice)
    seq, fus = out[False], out[True]
    scale = np.max(np.abs(seq[0]))
    assert scale > 0
    assert np.max(np.abs(fus[0] - seq[0])) < 1e-12 * scale
    assert np.max(np.abs(seq[1] - 2 * seq[0])) < 1e-12 * scale
    assert np.max(np.abs(fus[1] - 2 * fus[0])) < 1e-12 * scale


def test_mapmaker_recovers_offsets_and_sky():
    """End to end (configs[0] shape: 4 detectors x 10 min @10 Hz, Nside 16): destriping removes
    the injected baselines; the binned map equals the input sky on well-conditioned pixels."""
    data, pix, sw, truth, sky = make_solver_setup(noise_rms=0.0)
    binner = ops.BinMap(pixel_dist="dist", pixel_pointing=pix, stokes_weights=sw, full_pointing=True)
    # good_fraction low enough that no baseline is flagged: samples under a flagged amplitude
    # stay in the map-making (in the reference too), which would break exact recovery
    tmpl = Offset(step_time=20.0, noise_model=defaults.noise_model, name="baselines", good_fraction=0.2)
    tmatrix = ops.TemplateMatrix(templates=[tmpl])
    mapper = ops.MapMaker(name="mm", keep_solver_products=True, det_data=defaults.det_data, binning=binner, template_matrix=tmatrix,
                          iter_max=200, convergence=1e-20, solve_rcond_threshold=1e-3, map_rcond_threshold=1e-3)
    mapper.apply(data)
    assert mapper.history[-1] < 1e-12 and len(mapper.history) < 200
    m = data["mm_map"]
    rc = data["mm_rcond"].data[:, :, 0]
    dist = data["dist"]
    good = rc > 1e-2
    assert np.count_nonzero(good) > 50
    got = m.data[good]
    gpix = (dist.local_submaps[:, None] * dist.n_pix_submap + np.arange(dist.n_pix_submap)[None, :])[good]
    resid = got - sky[gpix]
    # the map is determined up to a global offset in I (degenerate with the baselines)
    resid[:, 0] -= np.mean(resid[:, 0])
    assert np.max(np.abs(resid)) < 1e-6
    # the baselines themselves, up to the common offset
    amps = data["mm_solve_amplitudes"]["baselines"]
    off = 0
    errs = []
    for det in data.obs[0].local_detectors:
        t = truth[det]
        errs.append(amps.local[off:off + t.size] - t)
        off += t.size
    errs = np.concatenate(errs)
    assert np.max(np.abs(errs - errs.mean())) < 1e-6
    # without templates: plain binning
    data2, pix2, sw2, _, _ = make_solver_setup(noise_rms=0.0)
    binner2 = ops.BinMap(pixel_dist="dist", pixel_pointing=pix2, stokes_weights=sw2, full_pointing=False)
    ops.MapMaker(name="plain", binning=binner2).apply(data2)
    assert "plain_map" in data2 and "plain_hits" in data2
    assert data2["plain_hits"].data.sum() > 0


def test_noise_filter_operator():
    """NoiseFilter suppresses 1/f: the filtered low-frequency power drops (the reference checks
    the fitted knee frequency, tests/ops_noise_filter.py:158-165)."""
    from oracle import fft_oracle as fo

    data = create_satellite_data(n_det=3, n_samp=20000, rate=20.0, fknee=0.5, flag_samples=False)
    ob = data.obs[0]
    rng = np.random.default_rng(8)
    sig = rng.standard_normal((3, 20000)) + 0.2 * rng.standard_normal((3, 20000)).cumsum(axis=1)
    ob.detdata[defaults.det_data].data[:] = sig
    want = sig.copy()
    nse = ob[defaults.noise_model]
    kernels = []
    from toast_amd.ops.noise_filter import estimate_net

    for d in ob.local_detectors:
        kernels.append(fo.noise_filter_kernel(nse.psd(d), estimate_net(nse.freq(d), nse.psd(d))))
    # detector flags (a bit under the mask, one outside) and shared flags: the operator ORs mask * (shared & mask) into
    # the detector flags and extends them by the width of the impulse response (noise_filter.py:118-126, fft.py:935-945)
    dflags = ((rng.random((3, 20000)) < 0.001) * 1 + (rng.random((3, 20000)) < 0.01) * 2).astype(np.uint8)
    shared = ((rng.random(20000) < 0.0005) * 1).astype(np.uint8)
    ob.detdata[defaults.det_flags].data[:] = dflags
    ob.shared[defaults.shared_flags].data[:] = shared
    want_flags = dflags.copy()
    shflg = (defaults.det_mask_invalid * (shared & defaults.shared_mask_invalid)).astype(np.uint8)
    for row in want_flags:
        row |= shflg
    fo.convolve(want, 20.0, flags=list(want_flags), flag_mask=defaults.det_mask_invalid,
                kernel_freq=nse.freq(ob.local_detectors[0]), kernels=np.array(kernels))
    ops.NoiseFilter(noise_model=defaults.noise_model).apply(data)
    got = ob.detdata[defaults.det_data].data
    assert np.max(np.abs(got - want)) < 1e-11 * np.max(np.abs(want))
    assert np.array_equal(ob.detdata[defaults.det_flags].data, want_flags)
    assert np.count_nonzero(want_flags != dflags) > 100
    lo_before = np.abs(np.fft.rfft(sig[0]))[1:20].mean()
    lo_after = np.abs(np.fft.rfft(got[0]))[1:20].mean()
    assert lo_after < 0.1 * lo_before


@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("TOAST_TEST_FUZZ_SEEDS", "8")))))
def test_noise_filter_random_configurations(seed):
    """Randomly drawn NoiseFilter inputs (lengths, detector counts, knee frequencies, flag densities, with / without shared
    and detector flags, eager or lazy host coherence): timestreams within 1e-11 and flags identical to the oracle's
    convolve with the reference's flag bookkeeping."""
    from oracle import fft_oracle as fo
    from toast_amd.ops.noise_filter import estimate_net

    rng = np.random.default_rng(11000 + seed)
    n_det = int(rng.integers(1, 6))
    n_samp = int(rng.choice([rng.integers(300, 3000), rng.integers(3000, 40000)]))
    rate = float(rng.choice([10.0, 20.0, 50.0]))
    use_det_flags, use_shared = bool(rng.integers(0, 4) > 0), bool(rng.integers(0, 2))
    data = create_satellite_data(n_det=n_det, n_samp=n_samp, rate=rate, fknee=float(rng.choice([0.1, 0.5, 2.0])),
                                 flag_samples=False)
    data.lazy_host = bool(rng.integers(0, 2))
    ob = data.obs[0]
    sig = rng.standard_normal((n_det, n_samp)) + 0.2 * rng.standard_normal((n_det, n_samp)).cumsum(axis=1)
    ob.detdata[defaults.det_data].data[:] = sig
    dflags = ((rng.random((n_det, n_samp)) < rng.choice([0.0, 0.001, 0.02])) * 1
              + (rng.random((n_det, n_samp)) < 0.01) * 2).astype(np.uint8)
    shared = ((rng.random(n_samp) < rng.choice([0.0, 0.002])) * 1).astype(np.uint8)
    ob.detdata[defaults.det_flags].data[:] = dflags
    ob.shared[defaults.shared_flags].data[:] = shared
    nse = ob[defaults.noise_model]
    dets = ob.local_detectors
    kernels = np.array([fo.noise_filter_kernel(nse.psd(d), estimate_net(nse.freq(d), nse.psd(d))) for d in dets])
    want, want_flags = sig.copy(), dflags.copy()
    if use_det_flags and use_shared:
        for row in want_flags:
            row |= (defaults.det_mask_invalid * (shared & defaults.shared_mask_invalid)).astype(np.uint8)
    try:
        fo.convolve(want, rate, flags=list(want_flags) if use_det_flags else None,
                    flag_mask=defaults.det_mask_invalid if use_det_flags else None, kernel_freq=nse.freq(dets[0]),
                    kernels=kernels)
        failed = None
    except RuntimeError as err:
        failed = str(err)
    op = ops.NoiseFilter(noise_model=defaults.noise_model, det_flags=defaults.det_flags if use_det_flags else None,
                         shared_flags=defaults.shared_flags if use_shared else None)
    if failed is not None:
        # (a timestream shorter than the impulse response: the reference raises, so does the operator)
        with pytest.raises(RuntimeError):
            op.apply(data)
        return
    op.apply(data)
    got = ob.detdata[defaults.det_data].data
    assert np.max(np.abs(got - want)) < 1e-11 * np.max(np.abs(want)), (n_det, n_samp)
    assert np.array_equal(ob.detdata[defaults.det_flags].data, want_flags if use_det_flags else dflags)


def test_noise_filter_pipelined_upload():
    """A host-resident timestream buffer of 256 MB or more is uploaded in row blocks on the library's upload stream
    while the transform of the blocks that have arrived runs (toast_hip_accel_update_device_parts / _wait / _finish):
    same bits as one blocking upload followed by one transform, the oracle's numbers on sampled rows, and the
    detector subset / row order handled per block."""
    from oracle import fft_oracle as fo
    from toast_amd.ops.noise_filter import estimate_net

    n_det, n_samp, rate = 36, 1000000, 50.0           # 288 MB of timestreams
    out = {}
    for parts in (1, 5):
        data = create_satellite_data(n_det=n_det, n_samp=n_samp, rate=rate, fknee=0.2, flag_samples=False)
        ob = data.obs[0]
        rng = np.random.default_rng(77)
        sig = rng.standard_normal((n_det, n_samp))
        sig[:, ::1000] += 5.0
        ob.detdata[defaults.det_data].data[:] = sig
        dets = ob.local_detectors
        frng = np.random.default_rng(78)
        ob.detdata[defaults.det_flags].data[:] = (frng.random((n_det, n_samp)) < 1e-5).astype(np.uint8)
        ob.shared[defaults.shared_flags].data[:] = (frng.random(n_samp) < 1e-5).astype(np.uint8)
        op = ops.NoiseFilter(noise_model=defaults.noise_model, upload_parts=parts)   # flags on, as by default
        op.apply(data, detectors=[d for i, d in enumerate(dets) if i % 7 != 3])    # a subset: some rows stay untouched
        assert ob.detdata[defaults.det_data].accel_in_use()                         # left resident for the map-maker
        out[parts] = ob.detdata[defaults.det_data].data.copy()
        out[("flags", parts)] = ob.detdata[defaults.det_flags].data.copy()
    assert np.array_equal(out[1], out[5]) and np.array_equal(out[("flags", 1)], out[("flags", 5)])
    assert np.count_nonzero(out[("flags", 5)]) > 10 * n_det
    for i in (3, 10, 17):                                                           # rows outside the subset
        assert np.array_equal(out[5][i], sig[i])
    nse = ob[defaults.noise_model]
    rows = [0, 18, 35]
    kernels = np.array([fo.noise_filter_kernel(nse.psd(dets[i]), estimate_net(nse.freq(dets[i]), nse.psd(dets[i])))
                        for i in rows])
    want = sig[rows].copy()
    fo.convolve(want, rate, kernel_freq=nse.freq(dets[0]), kernels=kernels)
    assert np.max(np.abs(out[5][rows] - want)) < 1e-11 * np.max(np.abs(want))


def test_filters_skip_cut_detectors():
    """NoiseFilter and GroundFilter leave the timestreams and flags of detectors cut by their per-detector flags
    untouched and treat the others exactly as a run restricted to them with ``detectors=``; an observation without any
    valid detector is skipped."""
    from toast_amd.sim import create_ground_data

    def noise(cut):
        data = create_satellite_data(n_det=4, n_samp=20000, rate=20.0, fknee=0.5, flag_samples=True)
        ob = data.obs[0]
        rng = np.random.default_rng(3)
        ob.detdata[defaults.det_data].data[:] = rng.standard_normal((4, 20000)).cumsum(axis=1) * 0.1
        before = ob.detdata[defaults.det_data].data.copy(), ob.detdata[defaults.det_flags].data.copy()
        dets = list(ob.local_detectors)
        if cut:
            ob.update_local_detector_flags({dets[2]: 1})
            ops.NoiseFilter(noise_model=defaults.noise_model).apply(data)
        else:
            ops.NoiseFilter(noise_model=defaults.noise_model).apply(data, detectors=[d for d in dets if d != dets[2]])
        return before, ob.detdata[defaults.det_data].data.copy(), ob.detdata[defaults.det_flags].data.copy()

    (sig0, fl0), sig_c, fl_c = noise(True)
    _, sig_s, fl_s = noise(False)
    assert np.array_equal(sig_c, sig_s) and np.array_equal(fl_c, fl_s)
    assert np.array_equal(sig_c[2], sig0[2]) and np.array_equal(fl_c[2], fl0[2])
    assert not np.array_equal(sig_c[0], sig0[0])

    def ground(cut):
        data = create_ground_data(n_det=4, n_samp=24000, rate=50.0)
        ob = data.obs[0]
        rng = np.random.default_rng(4)
        ob.detdata[defaults.det_data].data[:] = rng.standard_normal((4, 24000))
        before = ob.detdata[defaults.det_data].data.copy()
        dets = list(ob.local_detectors)
        gf = ops.GroundFilter(filter_order=3, trend_order=2)
        if cut:
            ob.update_local_detector_flags({dets[1]: 1})
            extra = create_ground_data(n_det=4, n_samp=24000, rate=50.0)
            ob2 = extra.obs[0]
            ob2.name = "all_cut"
            ob2.update_local_detector_flags({d: 1 for d in ob2.local_detectors})
            data.obs.append(ob2)
            gf.apply(data)
        else:
            gf.apply(data, detectors=[d for d in dets if d != dets[1]])
        return before, ob.detdata[defaults.det_data].data.copy()

    g0, g_c = ground(True)
    _, g_s = ground(False)
    assert np.max(np.abs(g_c - g_s)) < 1e-12          # (the fit sums are accumulated with atomics: equal to rounding)
    assert np.array_equal(g_c[1], g0[1]) and not np.array_equal(g_c[0], g0[0])


def test_workflow_sim_satellite_simple(oracle):
    """BASELINE configs[0]: the simple satellite workflow (4 det x 10 min @100 Hz, Nside 64);
    its hit count must equal the number of unflagged samples and its binned map must equal
    the oracle's build_noise_weighted + cov_apply_diag on the same pointing."""
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "workflows",
                        "sim_satellite_simple.py")
    spec = importlib.util.spec_from_file_location("wf", path)
    wf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(wf)
    data = wf.main(["--ndet", "4", "--minutes", "10", "--rate", "100", "--nside", "64", "--full-pointing"])
    ob = data.obs[0]
    dist = data["pixel_dist"]
    sf = ob.shared[defaults.shared_flags].data
    n_good = 0
    for det in ob.local_detectors:
        p = ob.detdata[defaults.pixels][det]
        df = ob.detdata[defaults.det_flags][det]
        n_good += np.count_nonzero((p >= 0) & ((df & defaults.det_mask_nonscience) == 0)
                                   & ((sf & defaults.shared_mask_nonscience) == 0))
    assert int(data["mapmaker_hits"].data.sum()) == n_good
    idx = np.arange(len(ob.local_detectors), dtype=np.int32)
    z = np.zeros((dist.n_local_submap, dist.n_pix_submap, 3))
    detw = np.array([ob[defaults.noise_model].detector_weight(d) for d in ob.local_detectors])
    oracle.build_noise_weighted(dist.global_submap_to_local, z, idx, ob.detdata[defaults.pixels].data, idx,
                                ob.detdata[defaults.weights].data, idx, ob.detdata[defaults.det_data].data, idx,
                                ob.detdata[defaults.det_flags].data, detw, defaults.det_mask_nonscience,
                                ob.intervals[None].data, sf, defaults.shared_mask_nonscience)
    oracle.cov_apply_diag(dist.n_local_submap, dist.n_pix_submap, 3, data["mapmaker_cov"].raw, z)
    got = data["mapmaker_map"].data
    assert np.max(np.abs(got - z)) < 1e-10 * np.max(np.abs(z))


def test_fused_lhs_equals_operator_sequence():
    """The fused device-resident LHS (offset_accumulate + cov_apply + offset_scan_project, no
    timestream buffer) must reproduce the reference operator sequence (TemplateMatrix, BinMap,
    ScanMap, NoiseWeight, TemplateMatrix^T) on the same amplitudes."""
    from toast_amd.templates import AmplitudesMap

    results = {}
    for fused in (False, True):
        data, pix, sw, truth, sky = make_solver_setup(n_det=6, n_samp=9000)
        ops.CovarianceAndHits(pixel_dist="dist", covariance="cov", pixel_pointing=pix, stokes_weights=sw,
                              save_pointing=True).apply(data)
        lhs_bin = ops.BinMap(pixel_dist="dist", covariance="cov", binned="lhs_bin", pixel_pointing=pix,
                             stokes_weights=sw, full_pointing=True)
        tmpl = Offset(step_time=7.3, noise_model=defaults.noise_model, name="baselines", good_fraction=0.2)
        tmatrix = ops.TemplateMatrix(templates=[tmpl], amplitudes="amps_in", det_data="temp_LHS")
        tmatrix.initialize(data)
        amps = tmpl.zeros()
        amps.local[:] = np.random.default_rng(3).standard_normal(amps.n_local)
        data["amps_in"] = AmplitudesMap(baselines=amps)
        data["lhs_out"] = data["amps_in"].duplicate()
        data["lhs_out"].reset()
        lhs = ops.SolverLHS(binning=lhs_bin, template_matrix=tmatrix, out="lhs_out", fused=fused)
        assert lhs._can_fuse(data) == fused
        lhs.apply(data)
        first = data["lhs_out"]["baselines"].local.copy()
        lhs.apply(data)  # second application: buffers are reused / reset correctly
        assert np.array_equal(data["lhs_out"]["baselines"].local, first) or np.allclose(
            data["lhs_out"]["baselines"].local, first, rtol=1e-12, atol=1e-12 * np.max(np.abs(first)))
        results[fused] = first
        assert np.array_equal(data["amps_in"]["baselines"].local, amps.local)  # input untouched
    a, b = results[False], results[True]
    assert np.max(np.abs(a)) > 0
    assert np.max(np.abs(a - b)) < 1e-11 * np.max(np.abs(a))


def test_amplitudes_device_algebra_matches_host():
    """The resident Amplitudes arithmetic (toast_hip_vec_axpby_dev / vec_dot_dev) against the same
    operations on the host copies (reference templates/amplitudes.py:400-565)."""
    from toast_amd.templates import Amplitudes

    rng = np.random.default_rng(11)
    n = 100003
    host, devs = [], []
    for i in range(3):
        h = Amplitudes(None, n, n)
        h.local[:] = rng.standard_normal(n)
        h.local_flags[:] = rng.random(n) < 0.1
        d = h.duplicate()
        d.accel_resident(f"amp_alg_{i}")
        assert d.accel_in_use()
        host.append(h)
        devs.append(d)
    for vecs in (host, devs):
        x, y, z = vecs
        x.axpby(0.37, y)            # x += 0.37 y
        z.axpby(1.0, x, -1.25)      # z = -1.25 z + x
        y *= 3.0
        y -= z
        x += y
    for h, d in zip(host, devs):
        assert d.accel_in_use()     # nothing fell back to the host
    dots_h = [host[0].dot(host[1]), host[2].dot(host[2])]
    dots_d = [devs[0].dot(devs[1]), devs[2].dot(devs[2])]
    np.testing.assert_allclose(dots_d, dots_h, rtol=1e-12)
    dup = devs[2].duplicate()
    assert dup.accel_in_use()
    for h, d in zip(host + [host[2]], devs + [dup]):
        d.accel_update_host()
        assert np.array_equal(d.local, h.local)   # axpby is elementwise: bit-identical
        d.clear()
    # mixed residency: the host operand is uploaded, the result lives on the device
    a, b = host[0].duplicate(), host[1].duplicate()
    a.accel_resident("amp_mixed")
    a += b
    assert a.accel_in_use() and b.accel_in_use()
    a.accel_update_host()
    assert np.array_equal(a.local, host[0].local + host[1].local)
    a.clear()
    b.clear()


def test_solve_resident_equals_host_algebra():
    """PCG with device-resident vectors (fused LHS) and with the reference's host algebra
    (operator-sequence LHS) follow the same trajectory."""
    hist, amps = {}, {}
    for fused in (True, False):
        data, pix, sw, truth, sky = make_solver_setup(noise_rms=0.1)
        binner = ops.BinMap(pixel_dist="dist", pixel_pointing=pix, stokes_weights=sw, full_pointing=True)
        tmpl = Offset(step_time=20.0, noise_model=defaults.noise_model, name="baselines", good_fraction=0.2)
        tmatrix = ops.TemplateMatrix(templates=[tmpl])
        mapper = ops.MapMaker(name="mm", keep_solver_products=True, det_data=defaults.det_data, binning=binner, template_matrix=tmatrix,
                              iter_max=15, convergence=1e-30, solve_rcond_threshold=1e-3, map_rcond_threshold=1e-3,
                              fused_lhs=fused)
        mapper.apply(data)
        hist[fused] = np.array(mapper.history)
        amps[fused] = data["mm_solve_amplitudes"]["baselines"].local.copy()
    n = min(len(hist[True]), len(hist[False]))
    assert n >= 5
    np.testing.assert_allclose(hist[True][:5], hist[False][:5], rtol=1e-6)
    scale = np.max(np.abs(amps[False]))
    assert np.max(np.abs(amps[True] - amps[False])) < 1e-6 * scale


@pytest.mark.parametrize("case", ["converges", "iteration_limit", "stalls", "noise_prior"])
def test_pcg_scalars_on_the_device_follow_the_host_loop(monkeypatch, case):
    """solve() with alpha / beta / residual norms and the convergence logic on the device (csrc/pcg.hip: the host
    enqueues one iteration ahead and reads the status one iteration late) against the same solve with the scalars
    on the host (TOAST_HIP_PCG_SCALARS=host, the reference's loop structure): the same number of iterations, the same
    history and the same amplitudes -- whichever of the reference's exits ends the loop (convergence, the iteration
    limit, the stall test), i.e. the speculative iteration enqueued after the end changes nothing."""
    kw = dict(converges=dict(iter_max=60, convergence=1e-10), iteration_limit=dict(iter_max=7, convergence=1e-30),
              stalls=dict(iter_max=100, convergence=1e-30, iter_min=3), noise_prior=dict(iter_max=25, convergence=1e-12))[case]
    hist, amps, n_sync = {}, {}, {}
    from toast_amd.accel import native

    for mode in ("host", "device"):
        monkeypatch.setenv("TOAST_HIP_PCG_SCALARS", mode)
        data, pix, sw, truth, sky = make_solver_setup(noise_rms=0.1)
        binner = ops.BinMap(pixel_dist="dist", pixel_pointing=pix, stokes_weights=sw, full_pointing=True)
        tmpl = Offset(step_time=20.0, noise_model=defaults.noise_model, name="baselines", good_fraction=0.2,
                      use_noise_prior=(case == "noise_prior"), precond_width=10)
        mapper = ops.MapMaker(name="mm", keep_solver_products=True, det_data=defaults.det_data, binning=binner,
                              template_matrix=ops.TemplateMatrix(templates=[tmpl]), solve_rcond_threshold=1e-3,
                              map_rcond_threshold=1e-3, **kw)
        mapper.apply(data)
        hist[mode] = np.array(mapper.history)
        amps[mode] = data["mm_solve_amplitudes"]["baselines"].local.copy()
        assert len(mapper.iteration_seconds) == len(mapper.history)
    # (the two loops add the dot products in different orders: the trajectories agree to rounding amplified by the
    # iteration, and an exit test sitting on its threshold may fall one iteration apart)
    n_dev, n_host = len(hist["device"]), len(hist["host"])
    assert n_host >= 3 and abs(n_dev - n_host) <= (0 if case == "iteration_limit" else 1), (n_dev, n_host)
    if case == "iteration_limit":
        assert n_host == 7
    if case == "converges":
        assert hist["host"][-1] < 1e-10 and hist["device"][-1] < 1e-10 and n_host < 60
    if case == "stalls":
        assert n_host < 100 and n_dev < 100
    n = min(n_dev, n_host)
    # (towards the rounding floor the two summation orders drift apart: tight above 1e-12, a factor of two down to
    # 1e-18, below that only "small")
    h_dev, h_host = hist["device"][:n], hist["host"][:n]
    tight, loose = h_host > 1e-12, (h_host <= 1e-12) & (h_host > 1e-18)
    np.testing.assert_allclose(h_dev[tight], h_host[tight], rtol=1e-5, atol=0)
    assert np.all(h_dev[loose] < 2.0 * h_host[loose]) and np.all(h_dev[loose] > 0.5 * h_host[loose])
    assert np.all(h_dev[h_host <= 1e-18] < 1e-15)
    scale = np.max(np.abs(amps["host"]))
    # (an exit test sitting on its threshold: one more / one fewer step of the size of the converged residual)
    assert np.max(np.abs(amps["device"] - amps["host"])) < (1e-6 if n_dev == n_host else 1e-4) * scale


@pytest.mark.parametrize("prior", [False, True])
def test_pcg_fused_updates_give_the_bits_of_the_separate_launches(monkeypatch, prior):
    """The device-scalar loop with the result / residual update inside the r . r launch and the diagonal preconditioner
    inside the z . r launch (toast_hip_pcg_step_dot_dev, _precond_diag_dot_dev) against the same loop with separate
    launches (TOAST_HIP_PCG_FUSE=0): the same history and amplitudes to the run-to-run rounding of the left-hand side's
    atomic scatter (the fused kernels themselves give the bits of the separate launches:
    test_pcg_fused_kernels_against_numpy).  With the noise prior the preconditioner is banded: only the step is
    fused."""
    from toast_amd import capi

    hist, amps, calls = {}, {}, {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("TOAST_HIP_PCG_FUSE", fuse)
        monkeypatch.setenv("TOAST_HIP_PCG_SCALARS", "device")
        seen = {}
        for name in ("pcg_step_dot", "pcg_precond_diag_dot", "pcg_step", "pcg_dot"):
            real = getattr(capi.dev, name)

            def counted(*a, _real=real, _name=name, **k):
                seen[_name] = seen.get(_name, 0) + 1
                return _real(*a, **k)

            monkeypatch.setattr(capi.dev, name, counted)
        data, pix, sw, truth, sky = make_solver_setup(noise_rms=0.1)
        binner = ops.BinMap(pixel_dist="dist", pixel_pointing=pix, stokes_weights=sw, full_pointing=True)
        tmpl = Offset(step_time=20.0, noise_model=defaults.noise_model, name="baselines", good_fraction=0.2,
                      use_noise_prior=prior, precond_width=10)
        mapper = ops.MapMaker(name="mm", keep_solver_products=True, det_data=defaults.det_data, binning=binner,
                              template_matrix=ops.TemplateMatrix(templates=[tmpl]), solve_rcond_threshold=1e-3,
                              map_rcond_threshold=1e-3, iter_max=12, convergence=1e-30)
        mapper.apply(data)
        hist[fuse] = np.array(mapper.history)
        amps[fuse] = data["mm_solve_amplitudes"]["baselines"].local.copy()
        calls[fuse] = seen
        monkeypatch.undo()
    n = len(hist["1"])
    assert calls["1"].get("pcg_step_dot", 0) >= n and calls["1"].get("pcg_step", 0) == 0, calls
    assert (calls["1"].get("pcg_precond_diag_dot", 0) >= n) == (not prior), calls
    assert calls["0"].get("pcg_step_dot", 0) == 0 and calls["0"].get("pcg_step", 0) >= n, calls
    assert len(hist["0"]) == n
    np.testing.assert_allclose(hist["1"], hist["0"], rtol=1e-7)
    assert np.max(np.abs(amps["1"] - amps["0"])) < 1e-9 * np.max(np.abs(amps["0"]))


def test_pcg_fused_kernels_against_numpy():
    """toast_hip_pcg_step_dot_dev / _precond_diag_dot_dev on vectors of awkward lengths (one element, a partial last
    pass, more than kDotBlocks x 2048 elements): the updated vectors are exact, the sums agree with NumPy to rounding,
    flagged amplitudes are left out of the sums, alpha = 0 (solver finished) leaves result and residual untouched."""
    import torch

    from toast_amd import capi

    D = capi.dev
    rng = np.random.default_rng(77)
    for n in (1, 2047, 230400, 1024 * 2048 + 12345):
        p, res, ap, r, var = (rng.standard_normal(n) for _ in range(5))
        var = np.abs(var) + 0.1            # variances: z . r is a sum of positive terms
        flags = (rng.random(n) < 0.05).astype(np.uint8)
        dp, dres, dap, dr, dvar = (torch.from_numpy(x.copy()).cuda() for x in (p, res, ap, r, var))
        dfl = torch.from_numpy(flags).cuda()
        dz = torch.full((n,), 7.0, dtype=torch.float64, device="cuda")
        state = torch.zeros(D.pcg_state_bytes(4) // 8 + 1, dtype=torch.float64, device="cuda")
        delta, p_ap = 3.0, float(np.dot(p, ap))
        D.pcg_init(state.data_ptr(), 10.0, delta, 1e-30, 3, 4)      # (n_iter_min = 3: no stall test in iteration 0)
        D.pcg_dot(state.data_ptr(), n, dp.data_ptr(), dap.data_ptr(), 0, 0, accumulate=False, stage=1)
        # alpha as the device computed it (delta / (p . Ap) with ITS summation order): 0 + alpha * 1
        y0 = torch.zeros(1, dtype=torch.float64, device="cuda")
        one1 = torch.ones(1, dtype=torch.float64, device="cuda")
        D.pcg_axpby(state.data_ptr(), 1, D.PCG_ALPHA, one1.data_ptr(), D.PCG_ONE, y0.data_ptr())
        D.pcg_step_dot(state.data_ptr(), n, dp.data_ptr(), dres.data_ptr(), dap.data_ptr(), dr.data_ptr(),
                       dfl.data_ptr(), accumulate=False, stage=2)
        torch.cuda.synchronize()
        hist, st = D.pcg_history(state.data_ptr(), 4)
        alpha = float(y0.item())
        assert abs(alpha - delta / p_ap) < 1e-10 * abs(alpha)
        assert np.array_equal(dres.cpu().numpy(), res + alpha * p)
        r_new = r + (-alpha) * ap
        assert np.array_equal(dr.cpu().numpy(), r_new)
        want = float(np.sum(r_new[flags == 0] ** 2))
        assert abs(st.sqsum - want) <= 1e-12 * want + 1e-300, (n, st.sqsum, want)
        D.pcg_precond_diag_dot(state.data_ptr(), n, dvar.data_ptr(), dr.data_ptr(), dfl.data_ptr(), dz.data_ptr(),
                               dfl.data_ptr(), accumulate=False, stage=3)
        torch.cuda.synchronize()
        z = np.where(flags == 0, r_new * var, 0.0)
        assert np.array_equal(dz.cpu().numpy(), z)
        # the separate launches on copies of the same inputs: the same bits in every vector and every scalar
        sres, sr = torch.from_numpy(res.copy()).cuda(), torch.from_numpy(r.copy()).cuda()
        sz = torch.full((n,), 7.0, dtype=torch.float64, device="cuda")
        state2 = torch.zeros_like(state)
        D.pcg_init(state2.data_ptr(), 10.0, delta, 1e-30, 3, 4)
        D.pcg_dot(state2.data_ptr(), n, dp.data_ptr(), dap.data_ptr(), 0, 0, accumulate=False, stage=1)
        D.pcg_step(state2.data_ptr(), n, dp.data_ptr(), sres.data_ptr(), dap.data_ptr(), sr.data_ptr())
        D.pcg_dot(state2.data_ptr(), n, sr.data_ptr(), sr.data_ptr(), dfl.data_ptr(), dfl.data_ptr(), accumulate=False,
                  stage=2)
        D.template_offset_apply_diag_precond(dvar.data_ptr(), sr.data_ptr(), dfl.data_ptr(), sz.data_ptr(), n)
        D.pcg_dot(state2.data_ptr(), n, sz.data_ptr(), sr.data_ptr(), dfl.data_ptr(), dfl.data_ptr(), accumulate=False,
                  stage=3)
        torch.cuda.synchronize()
        assert torch.equal(sres, dres) and torch.equal(sr, dr) and torch.equal(sz, dz)
        assert torch.equal(state2, state), n       # alpha, sqsum, delta, beta, history: the whole state block
        # stage 3: delta <- z . r; read it back through beta = delta_new / delta_old on the next axpby
        y = torch.zeros(n, dtype=torch.float64, device="cuda")
        one = torch.ones(n, dtype=torch.float64, device="cuda")
        D.pcg_axpby(state.data_ptr(), n, D.PCG_BETA, one.data_ptr(), D.PCG_ONE, y.data_ptr())
        torch.cuda.synchronize()
        beta = float(y[0].item())
        want_beta = float(np.dot(z[flags == 0], r_new[flags == 0])) / delta
        assert abs(beta - want_beta) <= 1e-12 * abs(want_beta) + 1e-300
    # a finished solver: alpha = 0, nothing is written
    state = torch.zeros(D.pcg_state_bytes(0) // 8 + 2, dtype=torch.float64, device="cuda")
    D.pcg_init(state.data_ptr(), 10.0, 3.0, 1e-30, 0, 0)            # n_iter_max = 0: done from the start
    before = dres.clone(), dr.clone()
    D.pcg_dot(state.data_ptr(), n, dp.data_ptr(), dap.data_ptr(), 0, 0, accumulate=False, stage=1)
    D.pcg_step_dot(state.data_ptr(), n, dp.data_ptr(), dres.data_ptr(), dap.data_ptr(), dr.data_ptr(), dfl.data_ptr(),
                   accumulate=False, stage=2)
    torch.cuda.synchronize()
    assert torch.equal(dres, before[0]) and torch.equal(dr, before[1])


def test_pcg_status_of_a_second_state_block_is_read_synchronously():
    """The lagged status ring belongs to the state initialised last; another state block polled in between gets its own
    current status (not a slot of the other solver)."""
    import torch

    from toast_amd import capi

    D = capi.dev
    x = torch.ones(1000, dtype=torch.float64, device="cuda")
    a = torch.zeros(D.pcg_state_bytes(5) // 8 + 1, dtype=torch.float64, device="cuda")
    b = torch.zeros_like(a)
    D.pcg_init(a.data_ptr(), 100.0, 1.0, 1e-30, 3, 5)
    D.pcg_init(b.data_ptr(), 200.0, 1.0, 1e-30, 3, 5)          # b owns the ring
    for state, scale in ((a, 1.0), (b, 2.0)):
        y = x * scale
        D.pcg_dot(state.data_ptr(), 1000, y.data_ptr(), y.data_ptr(), 0, 0, accumulate=False, stage=2)
    sa = D.pcg_status(a.data_ptr(), lag=1)                      # not the owner: current, whatever the lag
    assert sa.n_history == 1 and sa.sqsum == 1000.0 and sa.relative == 10.0
    sb0 = D.pcg_status(b.data_ptr(), lag=1)                     # the owner's first lagged read: nothing yet
    assert sb0.n_history == 0
    sb1 = D.pcg_status(b.data_ptr(), lag=1)
    assert sb1.n_history == 1 and sb1.sqsum == 4000.0 and sb1.relative == 20.0


def test_lazy_host_coherence_and_eviction():
    """Pipelines leave detector data resident and device-current; the host copy is refreshed on
    access (DetectorData.data) or by eviction, and equals the eagerly copied result."""
    from toast_amd.ops.pipeline import Pipeline

    got = {}
    for lazy in (True, False):
        data = create_satellite_data(n_det=4, n_samp=3000)
        data.lazy_host = lazy
        dp, pix, sw = pointing_ops(nside=64, create_dist=None)
        Pipeline(operators=[dp, pix, sw]).apply(data)
        ob = data.obs[0]
        for key in (defaults.quats, defaults.pixels, defaults.weights):
            assert ob.detdata[key].accel_in_use() == lazy
            assert ob.detdata[key].accel_exists() == lazy
        if lazy:
            # stale host buffer until somebody looks
            assert not np.any(ob.detdata[defaults.pixels].buffer)
            px = ob.detdata[defaults.pixels].data          # copies back
            assert not ob.detdata[defaults.pixels].accel_in_use()
            freed = data.accel_evict()                      # quats + weights written back and freed
            assert freed >= ob.detdata[defaults.quats].buffer.nbytes + ob.detdata[defaults.weights].buffer.nbytes
            assert not ob.detdata[defaults.weights].accel_exists()
            assert data.accel_evict() == 0
        got[lazy] = {k: ob.detdata[k].data.copy() for k in (defaults.quats, defaults.pixels, defaults.weights)}
    for k in got[True]:
        assert np.any(got[False][k])
        assert np.array_equal(got[True][k], got[False][k])


@pytest.mark.parametrize("compact", [False, True])
def test_uncached_pointing_matches_full_pointing(monkeypatch, compact):
    """full_pointing=False (the reference default) runs through the pointing-on-the-fly kernels
    (BinMap, fused SolverLHS) and batched scratch passes; the products equal the cached-pointing
    run: maps, hits, covariance, amplitudes."""
    monkeypatch.setenv("TOAST_HIP_POINTING_BATCH", "4")   # two scratch passes: 4 + 2 detectors
    out = {}
    for full in (True, False):
        data, pix, sw, truth, sky = make_solver_setup(noise_rms=0.1, n_det=6, n_samp=9000)
        binner = ops.BinMap(pixel_dist="dist", pixel_pointing=pix, stokes_weights=sw, full_pointing=full,
                            compact_cache=compact)
        tmpl = Offset(step_time=20.0, noise_model=defaults.noise_model, name="baselines", good_fraction=0.2)
        tmatrix = ops.TemplateMatrix(templates=[tmpl])
        mapper = ops.MapMaker(name="mm", keep_solver_products=True, det_data=defaults.det_data, binning=binner, template_matrix=tmatrix,
                              iter_max=12, convergence=1e-30, solve_rcond_threshold=1e-3, map_rcond_threshold=1e-3)
        mapper.apply(data)
        ob = data.obs[0]
        if not full:
            # nothing of the 56 B/det-sample pointing was kept
            assert defaults.weights not in ob.detdata or ob.detdata[defaults.weights].buffer.shape[0] < 6
            cname = defaults.pixels + "_compact"
            assert (cname in ob.detdata) == compact
            if compact:
                assert ob.detdata[cname].dtype == np.int32 and ob.detdata[cname].accel_in_use()
        out[full] = dict(map=data["mm_map"].data.copy(), hits=data["mm_hits"].data.copy(),
                         cov=data["mm_cov"].data.copy(), amps=data["mm_solve_amplitudes"]["baselines"].local.copy(),
                         hist=np.array(mapper.history))
    assert np.array_equal(out[True]["hits"], out[False]["hits"])
    np.testing.assert_allclose(out[False]["cov"], out[True]["cov"], rtol=1e-10, atol=1e-14 * np.max(np.abs(out[True]["cov"])))
    np.testing.assert_allclose(out[False]["hist"][:5], out[True]["hist"][:5], rtol=1e-6)
    scale = np.max(np.abs(out[True]["amps"]))
    assert np.max(np.abs(out[False]["amps"] - out[True]["amps"])) < 1e-7 * scale
    mscale = np.max(np.abs(out[True]["map"]))
    assert np.max(np.abs(out[False]["map"] - out[True]["map"])) < 1e-7 * mscale


def test_binmap_on_the_fly_equals_cached():
    data, pix, sw, truth, sky = make_solver_setup(noise_rms=0.1, n_det=5, n_samp=6000)
    ops.CovarianceAndHits(pixel_dist="dist", covariance="cov", pixel_pointing=pix, stokes_weights=sw).apply(data)
    # the scratch pointing of the covariance pass happens to hold every detector here: cached
    # pointing would be preferred over recomputation, so drop it
    ops.Delete(detdata=[defaults.pixels, defaults.weights, defaults.quats]).apply(data)
    got = {}
    for key, kw in (("otf", dict(full_pointing=False)), ("batch", dict(full_pointing=False, on_the_fly=False)),
                    ("full", dict(full_pointing=True))):
        b = ops.BinMap(pixel_dist="dist", covariance="cov", binned=f"bin_{key}", pixel_pointing=pix,
                       stokes_weights=sw, **kw)
        if key == "otf":
            assert b._on_the_fly(data, None, None)
        b.apply(data)
        got[key] = data[f"bin_{key}"].data.copy()
    scale = np.max(np.abs(got["full"]))
    assert scale > 0
    assert np.max(np.abs(got["otf"] - got["full"])) < 1e-12 * scale
    assert np.max(np.abs(got["batch"] - got["full"])) < 1e-12 * scale


@pytest.mark.parametrize("save_pointing", [True, False])
def test_build_pixel_distribution(save_pointing, monkeypatch):
    """BuildPixelDistribution (reference ops/pointing.py:18-130): the distribution equals the one
    PixelsHealpix(create_dist=...) builds in a single pass; scratch passes leave no full pointing."""
    monkeypatch.setenv("TOAST_HIP_POINTING_BATCH", "2")
    data = create_satellite_data(n_det=6, n_samp=5000)
    dp, pix, sw = pointing_ops(nside=128, create_dist=None)
    ops.BuildPixelDistribution(pixel_dist="dist", pixel_pointing=pix, save_pointing=save_pointing).apply(data)
    with pytest.raises(RuntimeError, match="already exists"):
        ops.BuildPixelDistribution(pixel_dist="dist", pixel_pointing=pix).apply(data)
    ob = data.obs[0]
    n_dets = len(ob.select_local_detectors(flagmask=dp.det_mask))
    if save_pointing:
        assert ob.detdata[defaults.pixels].data.shape[0] == n_dets
    else:
        assert ob.detdata[defaults.pixels].data.shape[0] <= 2
    data2 = create_satellite_data(n_det=6, n_samp=5000)
    dp2, pix2, sw2 = pointing_ops(nside=128, create_dist="dist")
    pix2.apply(data2)
    assert list(data["dist"].local_submaps) == list(data2["dist"].local_submaps)
    assert data["dist"].n_pix == data2["dist"].n_pix and data["dist"].n_pix_submap == data2["dist"].n_pix_submap


def test_pointing_expansion_without_quaternions():
    """On the device the pointing operators write pixels / weights straight from the boresight
    (no [n_det, n_samp, 4] quaternion buffer); the results equal the three-kernel chain's."""
    from toast_amd.ops.pipeline import Pipeline

    out = {}
    for skip in (True, False):
        data = create_satellite_data(n_det=6, n_samp=5000, flagged_pixels=True)
        dp, pix, sw = pointing_ops(nside=256, create_dist="dist")
        pix.skip_quaternions = skip
        sw.skip_quaternions = skip
        Pipeline(operators=[pix, sw]).apply(data)
        ob = data.obs[0]
        assert (defaults.quats in ob.detdata) == (not skip)
        out[skip] = (ob.detdata[defaults.pixels].data.copy(), ob.detdata[defaults.weights].data.copy(),
                     list(data["dist"].local_submaps))
    assert np.array_equal(out[True][0], out[False][0])
    assert np.array_equal(out[True][1], out[False][1])
    assert out[True][2] == out[False][2]
    assert np.any(out[True][0] >= 0)


@pytest.mark.parametrize("step_time", [7.3, 1.5, 1.6, 1.7, 100.0])
def test_offset_initialisation_device_equals_host(step_time):
    """Offset amplitude flags / preconditioner variances from the device flag counts
    (toast_hip_offset_count_flagged_dev) equal the reference's per-detector host loop
    (offset.py:262-343), for several views, a ragged last baseline and a dead detector.  Baselines of 73 samples, of 15
    (one flag byte per lane), of 16 and 17 (round 6's sixteen bytes per lane, a boundary inside most lanes) and of 1000
    (longer than a view's tail)."""
    from toast_amd import synth

    data = create_satellite_data(n_det=6, n_samp=7013, flagged_pixels=True)
    ob = data.obs[0]
    from toast_amd.data import IntervalList

    ob.intervals["scan"] = IntervalList(data=synth.make_intervals(7013, n_split=3, rate=10.0, gap=17))
    rng = np.random.default_rng(5)
    fl = ob.detdata[defaults.det_flags].data
    fl[:] = (rng.random(fl.shape) < 0.3).astype(np.uint8) * 5
    fl[2, 1000:1800] = 1                      # baselines cut by the good fraction
    tmpl = Offset(step_time=step_time, noise_model=defaults.noise_model, name="baselines", good_fraction=0.6,
                  view="scan", det_flag_mask=1)
    tmpl.det_data = defaults.det_data
    tmpl.data = data                          # initialises on the device (an accelerator is in use)
    flags_dev, var_dev = tmpl._amp_flags.copy(), tmpl._offsetvar.copy()
    # (with 1000-sample baselines no baseline of a live detector falls below the good fraction: the comparison below stands)
    assert (0 < flags_dev.sum() or step_time > 50) and flags_dev.sum() < flags_dev.size, (int(flags_dev.sum()), flags_dev.size)
    tmpl._amp_flags[:] = 0
    tmpl._offsetvar[:] = 0
    tmpl._init_variances_host(data)
    assert np.array_equal(flags_dev, tmpl._amp_flags)
    assert np.array_equal(var_dev, tmpl._offsetvar)


def test_mapmaker_two_observations_fused_equals_operator_sequence():
    """Two observations of different length (amplitude blocks interleave per detector,
    offset.py:727-760): the fused device path and the reference operator sequence agree, cached
    and uncached."""
    res = {}
    for key, kw in (("seq", dict(fused_lhs=False, full=True)), ("fused", dict(fused_lhs=True, full=True)),
                    ("otf", dict(fused_lhs=True, full=False)), ("seq_uncached", dict(fused_lhs=False, full=False))):
        data, pix, sw, truth, sky = make_solver_setup(noise_rms=0.1, n_det=4, n_samp=5000, seed=3)
        data2, _, _, _, _ = make_solver_setup(noise_rms=0.1, n_det=4, n_samp=3100, seed=4)
        ob2 = data2.obs[0]
        ob2.name = "second"
        data.obs.append(ob2)
        binner = ops.BinMap(pixel_dist="dist", pixel_pointing=pix, stokes_weights=sw, full_pointing=kw["full"])
        tmpl = Offset(step_time=13.0, noise_model=defaults.noise_model, name="baselines", good_fraction=0.2)
        mapper = ops.MapMaker(name="mm", keep_solver_products=True, det_data=defaults.det_data, binning=binner,
                              template_matrix=ops.TemplateMatrix(templates=[tmpl]), iter_max=10, convergence=1e-30,
                              solve_rcond_threshold=1e-3, map_rcond_threshold=1e-3, fused_lhs=kw["fused_lhs"])
        mapper.apply(data)
        res[key] = (data["mm_solve_amplitudes"]["baselines"].local.copy(), data["mm_map"].data.copy(),
                    np.array(mapper.history))
    n_amp = res["seq"][0].size
    assert n_amp == 4 * (-(-5000 // 130) + -(-3100 // 130))
    for key in ("fused", "otf", "seq_uncached"):
        np.testing.assert_allclose(res[key][2][:5], res["seq"][2][:5], rtol=1e-6)
        assert np.max(np.abs(res[key][0] - res["seq"][0])) < 1e-7 * np.max(np.abs(res["seq"][0]))
        assert np.max(np.abs(res[key][1] - res["seq"][1])) < 1e-7 * np.max(np.abs(res["seq"][1]))


@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("TOAST_TEST_FUZZ_SEEDS", "16")))))
def test_mapmaker_random_configurations(seed):
    """Randomly drawn inputs -- 1 to 3 observations of different lengths, a view with gaps (or none), random sample
    flags, a detector cut in one observation, odd baseline lengths, I or IQU, NEST or RING, with or without HWP, compact
    pixel cache, Offset noise prior -- through the complete MapMaker: the fused
    device left-hand side (cached and on-the-fly pointing) and the reference's operator sequence agree on hits,
    amplitudes and maps."""
    from toast_amd import synth
    from toast_amd.data import IntervalList

    rng = np.random.default_rng(100 + seed)
    n_obs = int(rng.integers(1, 4))
    n_det = int(rng.choice([2, 3, 4, 6]))
    lengths = [int(rng.integers(2500, 6000)) for _ in range(n_obs)]
    step_time = float(rng.choice([7.3, 13.0, 20.0, 31.7]))
    use_view = bool(rng.integers(0, 2))
    mode = str(rng.choice(["I", "IQU"]))
    cut = (int(rng.integers(0, n_obs)), int(rng.integers(0, n_det))) if n_det > 2 and rng.integers(0, 2) else None
    flag_frac = float(rng.choice([0.0, 0.02, 0.2]))
    splits = [int(rng.integers(1, 5)) for _ in range(n_obs)]
    gaps = [int(rng.integers(0, 40)) for _ in range(n_obs)]
    flag_seeds = [int(rng.integers(0, 2**31)) for _ in range(n_obs)]
    nest, hwp, compact = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    prior = (not use_view) and bool(rng.integers(0, 2))

    def build():
        data = None
        for iob, n_samp in enumerate(lengths):
            d, pix, sw, _, _ = make_solver_setup(noise_rms=0.2, n_det=n_det, n_samp=n_samp, seed=10 * seed + iob)
            ob = d.obs[0]
            ob.name = f"obs{iob}"
            if use_view:
                ob.intervals["scan"] = IntervalList(data=synth.make_intervals(n_samp, n_split=splits[iob], rate=10.0,
                                                                               gap=gaps[iob]))
            if flag_frac > 0:
                fl = ob.detdata[defaults.det_flags].data
                fl[:] = (np.random.default_rng(flag_seeds[iob]).random(fl.shape) < flag_frac).astype(np.uint8)
            if cut is not None and cut[0] == iob:
                ob.update_local_detector_flags({ob.local_detectors[cut[1]]: 1})
            if data is None:
                data, pix0, sw0 = d, pix, sw
            else:
                data.obs.append(ob)
        sw0.mode = mode
        pix0.nest = nest
        if not hwp:
            sw0.hwp_angle = None
        if use_view:
            pix0.view = "scan"
            sw0.view = "scan"
            pix0.detector_pointing.view = "scan"
        return data, pix0, sw0

    res = {}
    for key, kw in (("seq", dict(fused=False, full=True)), ("fused", dict(fused=True, full=True)),
                    ("otf", dict(fused=True, full=False))):
        data, pix, sw = build()
        binner = ops.BinMap(pixel_dist="dist", pixel_pointing=pix, stokes_weights=sw, full_pointing=kw["full"],
                            compact_cache=compact and not kw["full"])
        tmpl = Offset(step_time=step_time, noise_model=defaults.noise_model, name="baselines", good_fraction=0.3,
                      view="scan" if use_view else None, use_noise_prior=prior, precond_width=5)
        mapper = ops.MapMaker(name="mm", keep_solver_products=True, det_data=defaults.det_data, binning=binner,
                              template_matrix=ops.TemplateMatrix(templates=[tmpl], view="scan" if use_view else None),
                              iter_max=8, convergence=1e-30, solve_rcond_threshold=1e-3, map_rcond_threshold=1e-3,
                              fused_lhs=kw["fused"])
        mapper.apply(data)
        dist = data["dist"]
        full = {}
        for name in ("mm_hits", "mm_map"):
            m = np.zeros((dist.n_submap, dist.n_pix_submap, data[name].n_value))
            m[dist.local_submaps] = data[name].data
            full[name] = m
        res[key] = (np.array(data["mm_solve_amplitudes"]["baselines"].local), full, np.array(mapper.history),
                    np.array(data["mm_solve_amplitudes"]["baselines"].local_flags))
    ref = res["seq"]
    assert ref[1]["mm_hits"].sum() > 0 and np.max(np.abs(ref[0])) > 0
    for key in ("fused", "otf"):
        got = res[key]
        assert np.array_equal(got[1]["mm_hits"], ref[1]["mm_hits"]), key
        assert np.array_equal(got[3], ref[3]), key
        np.testing.assert_allclose(got[2][:4], ref[2][:4], rtol=1e-6, err_msg=key)
        assert np.max(np.abs(got[0] - ref[0])) < 1e-7 * np.max(np.abs(ref[0])), key
        assert np.max(np.abs(got[1]["mm_map"] - ref[1]["mm_map"])) < 1e-7 * np.max(np.abs(ref[1]["mm_map"])), key


def test_fused_lhs_plan_replay_and_invalidation():
    """The recorded launch plan of the fused LHS is replayed while nothing changed and rebuilt
    when the memory manager's generation changes (here: cached pointing evicted in between)."""
    from toast_amd import capi
    from toast_amd.templates import AmplitudesMap

    data, pix, sw, truth, sky = make_solver_setup(n_det=6, n_samp=9000)
    ops.CovarianceAndHits(pixel_dist="dist", covariance="cov", pixel_pointing=pix, stokes_weights=sw,
                          save_pointing=True).apply(data)
    lhs_bin = ops.BinMap(pixel_dist="dist", covariance="cov", binned="lhs_bin", pixel_pointing=pix, stokes_weights=sw,
                         full_pointing=True)
    tmpl = Offset(step_time=7.3, noise_model=defaults.noise_model, name="baselines", good_fraction=0.2)
    tmatrix = ops.TemplateMatrix(templates=[tmpl], amplitudes="amps_in", det_data="temp_LHS")
    tmatrix.initialize(data)
    amps = tmpl.zeros()
    amps.local[:] = np.random.default_rng(3).standard_normal(amps.n_local)
    data["amps_in"] = AmplitudesMap(baselines=amps)
    data["lhs_out"] = data["amps_in"].duplicate()
    lhs = ops.SolverLHS(binning=lhs_bin, template_matrix=tmatrix, out="lhs_out", fused=True)
    lhs.keep_on_device = True
    def result():
        o = data["lhs_out"]["baselines"]
        o.accel_update_host()          # no device allocation: the manager's generation is unchanged
        return o.local.copy()

    outs = []
    lhs.apply(data)
    plan0 = lhs._fused_plan
    outs.append(result())
    lhs.apply(data)
    assert lhs._fused_plan is plan0                      # replayed
    outs.append(result())
    gen = capi.accel_generation()
    assert data.accel_evict() > 0                        # pixels / weights leave the device
    assert capi.accel_generation() != gen
    lhs.apply(data)
    assert lhs._fused_plan is not plan0                  # rebuilt with the new device pointers
    outs.append(result())
    assert np.max(np.abs(outs[0])) > 0
    for o in outs[1:]:
        assert np.max(np.abs(o - outs[0])) < 1e-12 * np.max(np.abs(outs[0]))


@pytest.mark.parametrize("op", ["add", "subtract", "multiply", "divide"])
@pytest.mark.parametrize("target", ["first", "second", "new"])
def test_combine_operator(op, target):
    """ops.Combine (reference src/toast/ops/arithmetic.py): host path and the device path for
    add / subtract of resident float64 buffers."""
    for resident in (False, True):
        data = create_satellite_data(n_det=3, n_samp=1000)
        ob = data.obs[0]
        rng = np.random.default_rng(2)
        a = rng.standard_normal((3, 1000))
        b = rng.standard_normal((3, 1000)) + 3.0
        ob.detdata.create("a", dtype=np.float64)
        ob.detdata.create("b", dtype=np.float64)
        ob.detdata["a"].data[:] = a
        ob.detdata["b"].data[:] = b
        if resident:
            for k in ("a", "b"):
                ob.detdata[k].accel_create(k)
                ob.detdata[k].accel_update_device()
        res = {"first": "a", "second": "b", "new": "c"}[target]
        ops.Combine(op=op, first="a", second="b", result=res).apply(data)
        want = {"add": a + b, "subtract": a - b, "multiply": a * b, "divide": a / b}[op]
        on_dev = ob.detdata[res].accel_in_use()
        assert on_dev == (resident and op in ("add", "subtract"))
        assert np.array_equal(ob.detdata[res].data, want)
        if target == "new":
            assert np.array_equal(ob.detdata["a"].data, a) and np.array_equal(ob.detdata["b"].data, b)
    with pytest.raises(RuntimeError):
        ops.Combine(op="power")


def test_mapmaker_products_and_solve_mask():
    """MapMaker products as in the reference (mapmaker.py:304-313): hits / cov / invcov / rcond /
    map / noiseweighted_map (+ binmap, cleaned on request); solver products only with
    keep_solver_products; a pixel mask for the solve cuts those samples from the template fit
    but not from the final map."""
    data, pix, sw, truth, sky = make_solver_setup(noise_rms=0.1, n_det=4, n_samp=6000)
    binner = ops.BinMap(pixel_dist="dist", pixel_pointing=pix, stokes_weights=sw, full_pointing=True)
    tmpl = Offset(step_time=20.0, noise_model=defaults.noise_model, name="baselines", good_fraction=0.2)
    mapper = ops.MapMaker(name="mm", det_data=defaults.det_data, binning=binner,
                          template_matrix=ops.TemplateMatrix(templates=[tmpl]), iter_max=10, convergence=1e-30,
                          solve_rcond_threshold=1e-3, map_rcond_threshold=1e-3, write_binmap=True, save_cleaned=True)
    mapper.apply(data)
    for key in ("mm_hits", "mm_cov", "mm_invcov", "mm_rcond", "mm_map", "mm_noiseweighted_map", "mm_binmap"):
        assert key in data, key
    for key in ("mm_solve_hits", "mm_solve_cov", "mm_solve_rcond", "mm_solve_rcond_mask", "mm_solve_rhs",
                "mm_solve_bin", "mm_solve_amplitudes"):
        assert key not in data, key
    ob = data.obs[0]
    assert "mm_cleaned" in ob.detdata and "mm_solve_flags" not in ob.detdata
    assert not np.array_equal(ob.detdata["mm_cleaned"].data, ob.detdata[defaults.det_data].data)
    # map = C * noiseweighted map
    z = data["mm_noiseweighted_map"].duplicate()
    covariance_apply(data["mm_cov"], z)
    assert np.max(np.abs(z.data - data["mm_map"].data)) < 1e-12 * np.max(np.abs(data["mm_map"].data))
    assert np.max(np.abs(data["mm_binmap"].data - data["mm_map"].data)) > 0     # destriping did something
    hits_all = int(data["mm_hits"].data.sum())

    # second run with half of the hit pixels masked for the solve
    data2, pix2, sw2, _, _ = make_solver_setup(noise_rms=0.1, n_det=4, n_samp=6000)
    binner2 = ops.BinMap(pixel_dist="dist", pixel_pointing=pix2, stokes_weights=sw2, full_pointing=True)
    ops.BuildPixelDistribution(pixel_dist="dist", pixel_pointing=pix2, save_pointing=True).apply(data2)
    mask = PixelData(data2["dist"], np.uint8, n_value=1)
    mask.data[::2] = 1
    data2["solve_mask"] = mask
    tmpl2 = Offset(step_time=20.0, noise_model=defaults.noise_model, name="baselines", good_fraction=0.2)
    mapper2 = ops.MapMaker(name="mm", det_data=defaults.det_data, binning=binner2, mask="solve_mask",
                           template_matrix=ops.TemplateMatrix(templates=[tmpl2]), iter_max=10, convergence=1e-30,
                           solve_rcond_threshold=1e-3, map_rcond_threshold=1e-3, keep_solver_products=True)
    mapper2.apply(data2)
    assert int(data2["mm_hits"].data.sum()) == hits_all                  # final map: all samples
    assert 0 < int(data2["mm_solve_hits"].data.sum()) < hits_all         # solve: masked samples cut
    sf = data2.obs[0].detdata["mm_solve_flags"].data
    assert np.any(sf & 2) and np.any(sf & 1) and not np.any(sf & ~np.uint8(7))
    amps2 = data2["mm_solve_amplitudes"]["baselines"].local
    assert np.all(np.isfinite(amps2)) and np.max(np.abs(amps2)) > 0


def test_mapmaker_mc_mode_reuses_flags_and_covariance():
    """mc_mode (mapmaker_templates.py:499-501, :702-712, :852-864): realisation k re-uses the solver
    flags and the covariances of the first pass; only rhs / bin / amplitudes / maps are per index."""
    data, pix, sw, truth, sky = make_solver_setup(noise_rms=0.1, n_det=4, n_samp=6000)
    binner = ops.BinMap(pixel_dist="dist", pixel_pointing=pix, stokes_weights=sw, full_pointing=True)
    tmpl = Offset(step_time=20.0, noise_model=defaults.noise_model, name="baselines", good_fraction=0.2)
    tm = ops.TemplateMatrix(templates=[tmpl])
    kw = dict(name="mm", det_data=defaults.det_data, binning=binner, template_matrix=tm, iter_max=8,
              convergence=1e-30, solve_rcond_threshold=1e-3, map_rcond_threshold=1e-3, keep_solver_products=True)
    with pytest.raises(RuntimeError, match="MC mode|In MC mode"):
        ops.MapMaker(mc_mode=True, mc_index=0, **kw).apply(data)      # nothing to re-use yet
    ops.MapMaker(**kw).apply(data)
    cov0, scov0, map0 = data["mm_cov"], data["mm_solve_cov"], data["mm_map"].data.copy()
    flags0 = data.obs[0].detdata["mm_solve_flags"]
    rng = np.random.default_rng(9)
    ob = data.obs[0]
    ob.detdata[defaults.det_data].data[:] += 0.5 * rng.standard_normal(ob.detdata[defaults.det_data].data.shape)
    ops.MapMaker(mc_mode=True, mc_index=3, **kw).apply(data)
    assert data["mm_cov"] is cov0 and data["mm_solve_cov"] is scov0          # not rebuilt
    assert data.obs[0].detdata["mm_solve_flags"] is flags0
    for key in ("mm_00003_map", "mm_00003_solve_amplitudes", "mm_00003_noiseweighted_map"):
        assert key in data, key
    assert np.array_equal(data["mm_map"].data, map0)                        # first realisation untouched
    assert np.max(np.abs(data["mm_00003_map"].data - map0)) > 0


def test_template_pattern_restricts_the_amplitudes():
    """Template.pattern (templates/template.py:45-49, offset.py:226-236): only matching detectors
    carry baselines; the fused and the operator-sequence left-hand sides agree on them."""
    from toast_amd.templates import AmplitudesMap

    results = {}
    for fused in (False, True):
        data, pix, sw, truth, sky = make_solver_setup(n_det=6, n_samp=6000)
        ops.CovarianceAndHits(pixel_dist="dist", covariance="cov", pixel_pointing=pix, stokes_weights=sw,
                              save_pointing=True).apply(data)
        lhs_bin = ops.BinMap(pixel_dist="dist", covariance="cov", binned="lhs_bin", pixel_pointing=pix,
                             stokes_weights=sw, full_pointing=True)
        tmpl = Offset(step_time=10.0, noise_model=defaults.noise_model, name="baselines", pattern=".*A")
        tmatrix = ops.TemplateMatrix(templates=[tmpl], amplitudes="amps_in", det_data="temp_LHS")
        tmatrix.initialize(data)
        assert tmpl.detectors() == [d for d in data.obs[0].local_detectors if d.endswith("A")]
        amps = tmpl.zeros()
        assert amps.n_local == 3 * 60
        amps.local[:] = np.random.default_rng(3).standard_normal(amps.n_local)
        data["amps_in"] = AmplitudesMap(baselines=amps)
        data["lhs_out"] = data["amps_in"].duplicate()
        data["lhs_out"].reset()
        ops.SolverLHS(binning=lhs_bin, template_matrix=tmatrix, out="lhs_out", fused=fused).apply(data)
        results[fused] = data["lhs_out"]["baselines"].local.copy()
    assert np.max(np.abs(results[False])) > 0
    assert np.max(np.abs(results[False] - results[True])) < 1e-11 * np.max(np.abs(results[False]))


@pytest.mark.parametrize("hwp", [True, False])
def test_stokes_weights_qu_mode(hwp):
    """mode="QU" (stokes_weights.py:251-279: the Q, U columns of the IQU weights), host-staged and
    accelerator-resident; and a QU-only binned map through nnz = 2 kernels."""
    got = {}
    for mode in ("IQU", "QU"):
        for use_accel in (False, True):
            data = create_satellite_data(n_det=4, n_samp=3000)
            dp, pix, sw = pointing_ops(nside=32, mode=mode, hwp=hwp)
            sw.apply(data, use_accel=use_accel) if use_accel else sw.apply(data)
            wd = data.obs[0].detdata[defaults.weights]
            got[(mode, use_accel)] = wd.data.copy()
            assert wd.data.shape == (4, 3000, len(mode))
    assert np.array_equal(got[("QU", False)], got[("IQU", False)][:, :, 1:])
    assert np.array_equal(got[("QU", True)], got[("IQU", False)][:, :, 1:])
    # QU map-making: covariance has 3 packed elements, the map 2 components
    data = create_satellite_data(n_det=4, n_samp=6000)
    dp, pix, sw = pointing_ops(nside=16, mode="QU", hwp=hwp)
    rng = np.random.default_rng(0)
    ob = data.obs[0]
    ops.Pipeline(operators=[pix, sw]).apply(data)
    q, u = 0.3, -0.2
    for det in ob.local_detectors:
        w = ob.detdata[defaults.weights][det]
        ob.detdata[defaults.det_data][det] = q * w[:, 0] + u * w[:, 1] + 1e-3 * rng.standard_normal(6000)
    binner = ops.BinMap(pixel_dist="dist", pixel_pointing=pix, stokes_weights=sw, full_pointing=True)
    ops.MapMaker(name="qu", det_data=defaults.det_data, binning=binner, map_rcond_threshold=1e-2).apply(data)
    m, rc = data["qu_map"].data, data["qu_rcond"].data[:, :, 0]
    assert m.shape[-1] == 2 and data["qu_cov"].data.shape[-1] == 3
    good = rc > 0.1
    assert np.count_nonzero(good) > 20
    assert np.max(np.abs(m[good] - np.array([q, u]))) < 5e-3


def test_mapmaker_focalplane_key_split():
    """Split map-making on a focalplane column (mapmaker.py:728-790): one set of products per
    value, named <name>_<value>; their hit maps add up to the unsplit run's."""
    data, pix, sw, truth, sky = make_solver_setup(n_det=6, n_samp=6000)
    fp = data.obs[0].telescope.focalplane
    for d in fp.detectors:
        fp[d]["pol"] = d[-1]          # "A" / "B"
    assert sorted(fp.detector_groups("pol")) == ["A", "B"]
    groups = data.all_detector_groups(column="pol")
    assert sorted(groups) == ["A", "B"] and all(len(v) == 3 for v in groups.values())
    assert list(data.all_detector_groups()) == ["ALL"]
    binner = ops.BinMap(pixel_dist="dist", pixel_pointing=pix, stokes_weights=sw, full_pointing=True)
    tmpl = Offset(step_time=20.0, noise_model=defaults.noise_model, name="baselines", good_fraction=0.2)
    mapper = ops.MapMaker(name="mm", det_data=defaults.det_data, binning=binner,
                          template_matrix=ops.TemplateMatrix(templates=[tmpl]), iter_max=10,
                          solve_rcond_threshold=1e-3, map_rcond_threshold=1e-3, focalplane_key="pol")
    mapper.apply(data)
    assert mapper.name == "mm" and "mm_hits" not in data
    hits = {k: data[f"mm_{k}_hits"].data.copy() for k in ("A", "B")}
    dists = {k: list(data["dist"].local_submaps) for k in ("B",)}
    mapper.focalplane_key = None
    mapper.reset_pix_dist = True
    mapper.apply(data)
    total = data["mm_hits"].data
    assert list(data["dist"].local_submaps) == dists["B"] or total.shape[0] >= hits["B"].shape[0]
    assert int(total.sum()) == int(hits["A"].sum() + hits["B"].sum()) and int(hits["A"].sum()) > 0
    assert np.all(np.isfinite(data["mm_A_map"].data)) and np.all(np.isfinite(data["mm_B_map"].data))


@pytest.mark.parametrize("use_accel", [False, True])
def test_hwp_deflection_of_the_boresight(oracle, use_accel):
    """hwp_deflection_radius (pointing_detector.py:236-276, host-only in the reference): detector
    quaternions = focalplane offsets applied to the boresight rotated by `radius` about an axis 90
    degrees from the HWP fast axis; the cached and the quaternion-free expansions agree."""
    data = create_satellite_data(n_det=3, n_samp=2000)
    ob = data.obs[0]
    radius, offset = np.radians(0.25), 0.3
    dp = ops.PointingDetectorSimple(hwp_angle=defaults.hwp_angle, hwp_deflection_radius=radius, hwp_angle_offset=offset)
    dp.apply(data, use_accel=use_accel) if use_accel else dp.apply(data)
    got = ob.detdata[defaults.quats].data.copy()
    # independent construction of the deflected boresight
    ang = ob.shared[defaults.hwp_angle].data + offset + np.pi / 2
    half = radius / 2
    defl = np.stack([np.cos(ang) * np.sin(half), np.sin(ang) * np.sin(half), np.zeros_like(ang),
                     np.full_like(ang, np.cos(half))], axis=1)
    p, q = ob.shared[defaults.boresight_radec].data, defl
    bore = np.stack([
        p[:, 0] * q[:, 3] + p[:, 1] * q[:, 2] - p[:, 2] * q[:, 1] + p[:, 3] * q[:, 0],
        -p[:, 0] * q[:, 2] + p[:, 1] * q[:, 3] + p[:, 2] * q[:, 0] + p[:, 3] * q[:, 1],
        p[:, 0] * q[:, 1] - p[:, 1] * q[:, 0] + p[:, 2] * q[:, 3] + p[:, 3] * q[:, 2],
        -p[:, 0] * q[:, 0] - p[:, 1] * q[:, 1] - p[:, 2] * q[:, 2] + p[:, 3] * q[:, 3]], axis=1)
    fp = np.array([ob.telescope.focalplane[d]["quat"] for d in ob.local_detectors])
    want = np.zeros_like(got)
    ivl = ob.intervals[None].data
    oracle.pointing_detector(fp, np.ascontiguousarray(bore), np.arange(3, dtype=np.int32), want, ivl,
                             ob.shared[defaults.shared_flags].data, defaults.shared_mask_invalid)
    assert np.array_equal(got, want)
    plain = create_satellite_data(n_det=3, n_samp=2000)
    ops.PointingDetectorSimple().apply(plain)
    assert np.max(np.abs(got - plain.obs[0].detdata[defaults.quats].data)) > 1e-4   # it does deflect
    if use_accel:
        # quaternion-free pixel expansion reads the same deflected boresight
        pix_a = ops.PixelsHealpix(detector_pointing=dp, nside=256)
        pix_a.apply(data, use_accel=True)
        d2 = create_satellite_data(n_det=3, n_samp=2000)
        dp2 = ops.PointingDetectorSimple(hwp_angle=defaults.hwp_angle, hwp_deflection_radius=radius,
                                         hwp_angle_offset=offset)
        ops.PixelsHealpix(detector_pointing=dp2, nside=256).apply(d2, use_accel=True)   # no cached quats here
        assert defaults.quats not in d2.obs[0].detdata
        assert np.array_equal(d2.obs[0].detdata[defaults.pixels].data, ob.detdata[defaults.pixels].data)


def test_boresight_coordinate_rotation():
    """coord_in / coord_out (pointing_detector.py:120-175): the galactic and ecliptic poles land on
    the pole of the output frame; a round trip is the identity; cached and quaternion-free
    expansions agree."""
    from toast_amd.ops.pointing import coordinate_rotation
    from toast_amd.synth import quat_mult

    def direction(q):
        x, y, z, w = q
        return np.array([2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)])

    def pointing_at(ra_deg, dec_deg):
        ra, th = np.radians(ra_deg), np.radians(90.0 - dec_deg)
        qz = np.array([0, 0, np.sin(ra / 2), np.cos(ra / 2)])
        qy = np.array([0, np.sin(th / 2), 0, np.cos(th / 2)])
        return quat_mult(qz, qy)

    for (cin, cout), (ra, dec) in {("C", "G"): (192.85948, 27.12825), ("C", "E"): (270.0, 66.56071)}.items():
        rot, suffix = coordinate_rotation(cin, cout)
        assert suffix == f"_{cin}2{cout}"
        v = direction(quat_mult(rot, pointing_at(ra, dec)))
        assert np.max(np.abs(v - np.array([0.0, 0.0, 1.0]))) < 2e-5
        back, _ = coordinate_rotation(cout, cin)
        ident = quat_mult(back, rot)
        assert np.max(np.abs(np.abs(ident) - np.array([0, 0, 0, 1.0]))) < 1e-10   # the matrices are rounded to 12 digits
    e2g = coordinate_rotation("E", "G")[0]
    c2g_via_e = quat_mult(e2g, coordinate_rotation("C", "E")[0])
    c2g = coordinate_rotation("C", "G")[0]
    assert min(np.max(np.abs(c2g_via_e - c2g)), np.max(np.abs(c2g_via_e + c2g))) < 1e-5
    assert coordinate_rotation(None, None) == (None, "") and coordinate_rotation("G", "G") == (None, "")
    # operator level: galactic pixels through cached quaternions == quaternion-free expansion
    results = []
    for cached in (True, False):
        data = create_satellite_data(n_det=3, n_samp=2500)
        dp = ops.PointingDetectorSimple(coord_in="C", coord_out="G")
        pix = ops.PixelsHealpix(detector_pointing=dp, nside=128)
        if cached:
            dp.apply(data)
            pix.apply(data)
        else:
            pix.apply(data, use_accel=True)
            assert defaults.quats not in data.obs[0].detdata
        assert defaults.boresight_radec + "_C2G" in data.obs[0].shared
        results.append(data.obs[0].detdata[defaults.pixels].data.copy())
    assert np.array_equal(results[0], results[1])
    plain = create_satellite_data(n_det=3, n_samp=2500)
    ops.PixelsHealpix(detector_pointing=ops.PointingDetectorSimple(), nside=128).apply(plain)
    assert np.count_nonzero(plain.obs[0].detdata[defaults.pixels].data != results[0]) > 2000


def test_covariance_multiply_inverse_is_identity():
    """covariance_multiply (cov_mult_diag, reference src/libtoast/src/toast_map_cov.cpp:398-469: packed upper triangle
    of the product of two symmetric blocks): against NumPy (the reference build here has no BLAS and throws), and
    C C^-1 = 1 as in the reference's covariance test (src/toast/tests/cov.py)."""
    from toast_amd.pixels import PixelData, PixelDistribution, covariance_invert, covariance_multiply

    rng = np.random.default_rng(11)
    for nnz in (1, 2, 3):
        blk = nnz * (nnz + 1) // 2
        dist = PixelDistribution(n_pix=8 * 50, n_submap=8, local_submaps=[1, 4, 6])
        a = PixelData(dist, np.float64, n_value=blk)
        b = PixelData(dist, np.float64, n_value=blk)
        iu = np.triu_indices(nnz)
        full_a = np.zeros((3 * 50, nnz, nnz))
        full_b = np.zeros_like(full_a)
        for full, pd in ((full_a, a), (full_b, b)):
            g = rng.standard_normal((3 * 50, nnz, nnz))
            full[:] = g @ np.swapaxes(g, 1, 2) + nnz * np.eye(nnz)
            pd.data.reshape(-1, blk)[:] = full[:, iu[0], iu[1]]
        want = np.einsum("pmj,pjk->pmk", full_a, full_b)      # entry (k, m >= k) <- product[m, k]
        for on_dev in (False, True):
            x = a.duplicate()
            y = b.duplicate()
            if on_dev:
                for v, nm in ((x, "cm_x"), (y, "cm_y")):
                    v.accel_create(nm)
                    v.accel_update_device()
            covariance_multiply(x, y)
            if on_dev:
                x.accel_update_host()
                x.accel_delete()
                y.accel_delete()
            got = x.data.reshape(-1, blk)
            assert np.max(np.abs(got - want[:, iu[1], iu[0]])) < 1e-13 * np.max(np.abs(want)), (nnz, on_dev)
        inv = a.duplicate()
        covariance_invert(inv, 1.0e-8)
        prod = a.duplicate()
        covariance_multiply(prod, inv)
        eye = np.eye(nnz)[iu]
        assert np.max(np.abs(prod.data.reshape(-1, blk) - eye)) < 1e-10


def test_cov_accum_ffi_kernels():
    """cov_accum_diag_hits / cov_accum_diag_invnpp, the reference's FFI-level kernels behind BuildHitMap and
    BuildInverseCovariance (src/toast/_libtoast/map_cov.cpp:87-197): one stream of (local submap, pixel) pairs,
    negative = skipped; against a NumPy scatter of the same definition."""
    import toast_amd

    m = toast_amd.load_native()
    rng = np.random.default_rng(21)
    nsub, nsubpix, n = 5, 64, 40000
    # scanning-like stream: runs of equal pixels, some skipped samples
    pix = np.repeat(rng.integers(0, nsub * nsubpix, n // 8), 8)[:n]
    submap = (pix // nsubpix).astype(np.int64)
    subpix = (pix % nsubpix).astype(np.int64)
    skip = rng.random(n) < 0.05
    submap[skip & (rng.random(n) < 0.5)] = -1
    subpix[skip & (submap >= 0)] = -1
    good = (submap >= 0) & (subpix >= 0)
    hits = np.zeros(nsub * nsubpix, dtype=np.int64)
    hits[7] = 3          # accumulates into what is there
    m.cov_accum_diag_hits(nsub, nsubpix, 3, submap, subpix, hits, False)
    want = np.zeros_like(hits)
    want[7] = 3
    np.add.at(want, (submap * nsubpix + subpix)[good], 1)
    assert np.array_equal(hits, want)
    for nnz in (1, 3):
        w = rng.standard_normal((n, nnz))
        blk = nnz * (nnz + 1) // 2
        inv = np.zeros(nsub * nsubpix * blk)
        m.cov_accum_diag_invnpp(nsub, nsubpix, nnz, submap, subpix, np.ascontiguousarray(w.reshape(-1)), 2.5, inv, False)
        iu = np.triu_indices(nnz)
        wanti = np.zeros((nsub * nsubpix, blk))
        np.add.at(wanti, (submap * nsubpix + subpix)[good], (2.5 * w[good][:, iu[0]]) * w[good][:, iu[1]])
        assert np.max(np.abs(inv.reshape(-1, blk) - wanti)) < 1e-12 * np.max(np.abs(wanti))
    with pytest.raises(RuntimeError):
        m.cov_accum_diag_hits(nsub, nsubpix, 3, submap, subpix[:-1], hits, False)


def test_cov_accum_zmap_diag_and_global_to_local():
    """cov_accum_zmap, the all-in-one cov_accum_diag and global_to_local of the native module (names and argument
    order of toast._libtoast: map_cov.cpp:10-86, :199-250, pixels.cpp:9-66) against NumPy."""
    import toast_amd

    m = toast_amd.load_native()
    rng = np.random.default_rng(31)
    n_submap, nsubpix, n, nnz = 9, 48, 30000, 3
    local = np.array([1, 4, 7])
    g2l = np.full(n_submap, -1, dtype=np.int64)
    g2l[local] = np.arange(local.size)
    gl = rng.integers(0, n_submap * nsubpix, n).astype(np.int64)
    gl[rng.random(n) < 0.03] = -1
    lsm, lpx = m.global_to_local(gl, nsubpix, g2l)
    want_sm = np.where(gl < 0, -1, g2l[np.maximum(gl, 0) // nsubpix])
    want_px = np.where(gl < 0, -1, gl % nsubpix)
    assert np.array_equal(lsm, want_sm) and np.array_equal(lpx, want_px) and lsm.dtype == np.int64
    assert m.global_to_local(np.zeros(0, dtype=np.int64), nsubpix, g2l)[0].size == 0
    nsub = local.size
    w = rng.standard_normal((n, nnz))
    tod = rng.standard_normal(n)
    good = (lsm >= 0) & (lpx >= 0)
    hpx = (lsm * nsubpix + lpx)[good]
    zmap = np.zeros(nsub * nsubpix * nnz)
    m.cov_accum_zmap(nsub, nsubpix, nnz, lsm, lpx, np.ascontiguousarray(w.reshape(-1)), 0.5, tod, zmap)
    wantz = np.zeros((nsub * nsubpix, nnz))
    np.add.at(wantz, hpx, (0.5 * tod[good])[:, None] * w[good])
    assert np.max(np.abs(zmap.reshape(-1, nnz) - wantz)) < 1e-12 * np.max(np.abs(wantz))
    blk = nnz * (nnz + 1) // 2
    inv = np.zeros(nsub * nsubpix * blk)
    hits = np.zeros(nsub * nsubpix, dtype=np.int64)
    z2 = np.zeros_like(zmap)
    m.cov_accum_diag(nsub, nsubpix, nnz, lsm, lpx, np.ascontiguousarray(w.reshape(-1)), 0.5, tod, inv, hits, z2)
    assert np.max(np.abs(z2 - zmap)) < 1e-12 * np.max(np.abs(zmap))
    iu = np.triu_indices(nnz)
    wanti = np.zeros((nsub * nsubpix, blk))
    np.add.at(wanti, hpx, (0.5 * w[good][:, iu[0]]) * w[good][:, iu[1]])
    assert np.max(np.abs(inv.reshape(-1, blk) - wanti)) < 1e-12 * np.max(np.abs(wanti))
    wanth = np.zeros_like(hits)
    np.add.at(wanth, hpx, 1)
    assert np.array_equal(hits, wanth)


class _DenseDeviceLHS(ops.Operator):
    """a' = A a for a dense SPD A with the amplitude vectors RESIDENT ON THE DEVICE: the matrix product itself is done on
    the host between a download and an upload, everything else of the solve (dot products, updates, the recurrence's
    scalars and exit tests of csrc/pcg.hip) runs on the device as in a map-making solve."""

    def __init__(self, A):
        super().__init__(name="dense")
        self.A = A
        self.out = None
        self.keep_on_device = False
        outer = self

        class _TM:
            amplitudes = None
            templates = []

            def apply_precond(self, amps_in, amps_out, **kw):
                outer._host_op(amps_in["t"], amps_out["t"], lambda v: v / np.diag(outer.A))

        self.template_matrix = _TM()

    @staticmethod
    def _host_op(a_in, a_out, fn):
        from toast_amd.accel import accel_data_update_device, accel_data_update_host

        # (``buffer``: the host arrays without the lazy coherence of ``local`` -- this stand-in moves the data itself and
        # leaves both vectors device-current, as a device operator would)
        assert a_in.accel_in_use()
        accel_data_update_host(a_in.buffer, a_in._accel_name)
        if not a_out.accel_exists():
            a_out.accel_create(a_out._accel_name)
        a_out.buffer[:] = fn(a_in.buffer)
        accel_data_update_device(a_out.buffer, a_out._accel_name)
        a_out.accel_used(True)

    def _can_fuse(self, data):
        return True

    def _exec(self, data, detectors=None, **kw):
        self._host_op(data[self.template_matrix.amplitudes]["t"], data[self.out]["t"], lambda v: self.A @ v)

    def _finalize(self, data, **kw):
        return


@pytest.mark.parametrize("case", ["converges", "iteration_limit", "stalls", "starting_guess"])
def test_pcg_device_scalars_against_the_reference_solve(monkeypatch, case):
    """The device-scalar PCG loop (csrc/pcg.hip, ops.mapmaker_solve._pcg_device_scalars) against the trajectory of the
    REFERENCE's own ``solve()`` on dense SPD systems (tests/golden/pcg_solve.npz, generated by running
    src/toast/ops/mapmaker_solve.py:524-755 itself: tests/golden/make_golden_pcg.py): the same exit -- convergence,
    iteration limit, stall test -- after the same number of iterations, the same residual history and solution.  The
    device sums its dot products in a different order (block partial sums), hence a tolerance that grows with the
    condition number of the case instead of bit identity."""
    from toast_amd.data import Comm, Data
    from toast_amd.templates import Amplitudes, AmplitudesMap

    monkeypatch.setenv("TOAST_HIP_PCG_SCALARS", "device")
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pcg_solve.npz"))
    A, b, x0 = g[f"{case}_A"], g[f"{case}_b"], g[f"{case}_x0"]
    n = b.size
    data = Data(comm=Comm(use_dist=False))
    rhs = Amplitudes(data.comm, n, n)
    rhs.local[:] = b
    data["rhs"] = AmplitudesMap(t=rhs)
    if np.any(x0 != 0):
        start = Amplitudes(data.comm, n, n)
        start.local[:] = x0
        data["result"] = AmplitudesMap(t=start)
    calls = {"n": 0}
    import toast_amd.ops.mapmaker_solve as ms

    real = ms._pcg_device_scalars

    def counted(*a, **k):
        calls["n"] += 1
        return real(*a, **k)

    monkeypatch.setattr(ms, "_pcg_device_scalars", counted)
    hist = ops.solve(data, None, _DenseDeviceLHS(A), "rhs", "result", convergence=float(g[f"{case}_convergence"]),
                     n_iter_min=int(g[f"{case}_n_iter_min"]), n_iter_max=int(g[f"{case}_n_iter_max"]))
    assert calls["n"] == 1                       # the device loop ran, not the host loop
    want = g[f"{case}_history"]
    sol = g[f"{case}_solution"]
    got = data["result"]["t"].local
    if case == "stalls":
        # condition number 1e14: the trajectory is chaotic at the level of rounding, the EXIT is what is pinned -- the
        # stall test of every tenth iteration ends the loop long before the iteration limit, as in the reference
        assert 11 <= len(hist) < 300 and (len(hist) - 1) % 10 == 0
        assert hist[-1] < 1e-6
        return
    # A conjugate gradient amplifies rounding from iteration to iteration: tight while the difference is still rounding
    # (the first dozen iterations), then the same decay within a small factor and the same exit within two iterations.
    hist = np.array(hist)
    if case == "iteration_limit":
        assert len(hist) == len(want)
    else:
        assert abs(len(hist) - len(want)) <= 2, (len(hist), len(want))
    m = min(len(hist), len(want))
    np.testing.assert_allclose(hist[:12], want[:12], rtol=1e-7, atol=0)
    floor = want[:m] > 1e-12 * want[0]
    ratio = hist[:m][floor] / want[:m][floor]
    assert np.all(ratio < 50.0) and np.all(ratio > 0.02)
    assert np.max(np.abs(got - sol)) <= 1e-6 * np.max(np.abs(sol))


def test_byte_mix_probe_reads_what_it_says_and_leaves_the_inputs_alone():
    """toast_hip_probe_byte_mix_dev (bench.py's roofline.stream_ceiling): the streams of scan_map / build_noise_weighted with no
    gather -- its output is a fixed function of the three inputs (so every byte was read), the inputs are unchanged, and
    without an output array nothing is written at all."""
    import torch

    from toast_amd import capi

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(3)
    n_det, n_samp = 5, 4098
    pix = torch.from_numpy(rng.integers(-1, 1000, (n_det, n_samp))).to(dev)
    w = torch.from_numpy(rng.standard_normal((n_det, n_samp, 3))).to(dev)
    tod = torch.from_numpy(rng.standard_normal((n_det, n_samp))).to(dev)
    out = torch.full((n_det, n_samp), -7.0, dtype=torch.float64, device=dev)
    before = (pix.clone(), w.clone(), tod.clone())
    capi.probe_byte_mix(pix.data_ptr(), w.data_ptr(), tod.data_ptr(), out.data_ptr(), n_det, n_samp)
    capi.probe_byte_mix(pix.data_ptr(), w.data_ptr(), tod.data_ptr(), 0, n_det, n_samp)
    torch.cuda.synchronize()
    want = tod * w.sum(dim=2) + (pix & 1).to(torch.float64)
    assert torch.allclose(out, want, rtol=0, atol=1e-12)
    assert torch.equal(pix, before[0]) and torch.equal(w, before[1]) and torch.equal(tod, before[2])
    with pytest.raises(RuntimeError):
        capi.probe_byte_mix(pix.data_ptr(), w.data_ptr(), tod.data_ptr(), 0, n_det, n_samp - 1)      # odd rows
