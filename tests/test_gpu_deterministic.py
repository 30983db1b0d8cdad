"""GPU: the two "identical to the reference, not only close to it" switches.

* Deterministic debug mode (TOAST_HIP_DETERMINISTIC=1 / capi.set_deterministic): build_noise_weighted
  and build_inverse_covariance sum every pixel in (detector, interval, sample) order like the
  reference's host path (ops_mapmaker_utils.cpp:294-378, toast_map_cov.cpp:96-153) -- the result is
  BIT-identical to the reference's own outputs (tests/golden/chain_*.npz, cov_filter.npz), to the
  oracle, and from run to run; the default atomic kernels agree with it to 1e-12.
* toast_hip_set_stokes_reference_nan: NaN Q / U weights exactly where the reference produces them (the default)."""
import numpy as np
import pytest

import cases
import golden_util as gu

pytestmark = pytest.mark.gpu


@pytest.fixture()
def hip():
    from toast_amd import capi

    assert capi.accel_enabled()
    capi.accel_assign_device(1, 0, 1.0, False)
    capi.set_deterministic(True)
    yield capi
    capi.set_deterministic(False)
    capi.set_stokes_reference_nan(True)      # the default


def _bnw(impl, c, g2l, zmap, pixels, weights, tail=()):
    impl.build_noise_weighted(g2l, zmap, c["pixel_index"], pixels, c["weight_index"], weights, c["data_index"],
                              c["tod"], c["flag_index"], c["det_flags"], c["det_scale"], 1, c["intervals"],
                              c["shared_flags"], 1, *tail)


@pytest.mark.parametrize("name", gu.CHAINS)
def test_deterministic_zmap_and_invcov_equal_reference_outputs_bit_for_bit(hip, name):
    import toast_amd

    assert hip.get_deterministic()
    case, want, nest, iau = gu.load_chain(name)
    pixels, weights, g2l = want["pixels"], want["weights"], want["g2l"]
    runs = []
    for _ in range(2):
        zmap = np.zeros_like(want["zmap"])
        _bnw(hip, case, g2l, zmap, pixels, weights, tail=(False,))
        runs.append(zmap)
    assert np.array_equal(runs[0], runs[1])                       # run to run
    assert np.array_equal(runs[0], want["zmap"])                  # == the reference's own host path
    # the default kernels (atomics) stay within the parity bar of the same numbers
    hip.set_deterministic(False)
    zmap = np.zeros_like(want["zmap"])
    _bnw(hip, case, g2l, zmap, pixels, weights, tail=(False,))
    hip.set_deterministic(True)
    assert np.max(np.abs(zmap - want["zmap"])) <= 1e-12 * np.max(np.abs(want["zmap"]))
    # inverse covariance against the reference's cov_accum_diag_invnpp outputs
    m = toast_amd.load_native()
    g = gu.load("cov_filter")
    want_invcov = g[name + "_invcov"]
    invcov = np.zeros_like(want_invcov)
    m.build_inverse_covariance(np.ascontiguousarray(g2l), invcov, case["pixel_index"], np.ascontiguousarray(pixels),
                               case["weight_index"], np.ascontiguousarray(weights), case["flag_index"],
                               case["det_flags"], case["det_scale"], 1, case["intervals"], case["shared_flags"], 1, False)
    assert np.array_equal(invcov, want_invcov)


CASES = {
    "default": dict(),
    "split_gap_extra": dict(n_split=3, gap=5, extra_rows=2),
    "ragged": dict(n_samp=1029, n_split=4, gap=1, n_det=3, nside=256),
    "no_flags": dict(with_det_flags=False, with_shared_flags=False, n_samp=4097, nside=512),
    "ground_nside2048": dict(ground=True, n_samp=72000, rate=100.0, nside=2048, n_det=6),
    "many_hits": dict(n_det=16, n_samp=60000, nside=16, nside_submap=4),
}


@pytest.mark.parametrize("name", list(CASES))
def test_deterministic_accumulate_equals_oracle_bit_for_bit(hip, oracle, name):
    """Same pointing into both: the oracle's chain provides pixels / weights, then A^T N^-1 d is
    accumulated ON TOP of existing map content in two calls (the reference accumulates across
    observations, mapmaker_utils.py:706-773) -- resident buffers (use_accel=True path)."""
    import torch

    c = cases.make_case(**CASES[name])
    ref = cases.run_chain(oracle, c)
    rng = np.random.default_rng(5)
    start = rng.standard_normal(ref["zmap"].shape)
    want = start.copy()
    _bnw(oracle, c, ref["g2l"], want, ref["pixels"], ref["weights"])
    c2 = dict(c)
    c2["tod"] = np.ascontiguousarray(c["tod"][:, ::-1])
    _bnw(oracle, c2, ref["g2l"], want, ref["pixels"], ref["weights"])
    dev = torch.device("cuda")
    D = hip.dev
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in
         dict(g2l=ref["g2l"], z=start, pix=ref["pixels"], w=ref["weights"], tod=c["tod"], tod2=c2["tod"],
              df=c["det_flags"], sf=c["shared_flags"]).items()}
    n_samp = c["n_samp"]
    nf = n_samp if c["det_flags"].shape[1] == n_samp else 0
    ns = n_samp if c["shared_flags"].size == n_samp else 0
    for tod in (t["tod"], t["tod2"]):
        D.build_noise_weighted(t["g2l"].data_ptr(), t["z"].data_ptr(), c["n_pix_submap"], 3, c["pixel_index"],
                               t["pix"].data_ptr(), c["weight_index"], t["w"].data_ptr(), c["data_index"],
                               tod.data_ptr(), c["flag_index"], t["df"].data_ptr(), nf, c["det_scale"], 1, n_samp,
                               c["intervals"], t["sf"].data_ptr(), ns, 1)
    torch.cuda.synchronize()
    assert np.array_equal(t["z"].cpu().numpy(), want)


def test_operators_are_reproducible_in_deterministic_mode(hip):
    """BinMap (hits, inverse covariance, noise-weighted map, binned map) twice from scratch, cached and
    uncached pointing: every product bit-identical between the runs."""
    from toast_amd import ops
    from toast_amd.data import defaults
    from toast_amd.sim import create_satellite_data

    outs = []
    for full_pointing in (True, False, True, False):
        data = create_satellite_data(n_det=8, n_samp=40000, rate=50.0, spin_angle_deg=30.0, prec_angle_deg=60.0)
        rng = np.random.default_rng(2)
        data.obs[0].detdata[defaults.det_data].data[:] = rng.standard_normal((8, 40000))
        dp = ops.PointingDetectorSimple()
        pix = ops.PixelsHealpix(detector_pointing=dp, nside=64)
        sw = ops.StokesWeights(detector_pointing=dp, mode="IQU", hwp_angle=defaults.hwp_angle)
        binner = ops.BinMap(pixel_dist="dist", pixel_pointing=pix, stokes_weights=sw, full_pointing=full_pointing)
        ops.MapMaker(name="mm", binning=binner, template_matrix=None, keep_final_products=True).apply(data)
        outs.append({k: data[k].data.copy() for k in ("mm_hits", "mm_cov", "mm_noiseweighted_map", "mm_map")})
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[2][k]), k       # cached pointing, run to run
        assert np.array_equal(outs[1][k], outs[3][k]), k       # uncached pointing, run to run
    assert np.array_equal(outs[0]["mm_hits"], outs[1]["mm_hits"])


@pytest.mark.parametrize("use_hwp", [False, True])
def test_stokes_reference_nan_switch(hip, oracle, use_hwp):
    """By default the Q / U weights are NaN at exactly the samples where the reference's -sqrt(1 - z^2) is
    (ops_stokes_weights.cpp:66-75), finite and equal to 1e-13 everywhere else; with the switch off everything is
    finite."""
    from test_gpu_pixels_adversarial import boundary_pointings, quats_pointing_at

    rng = np.random.default_rng(3)
    q = boundary_pointings(rng, n_each=2000)
    th = np.concatenate([np.zeros(1000), np.full(1000, np.pi), rng.uniform(0, 1e-12, 1000),
                         np.pi - rng.uniform(0, 1e-12, 1000), rng.uniform(0, 3e-8, 4000)])
    q = np.concatenate([q, quats_pointing_at(th, rng.uniform(0, 2 * np.pi, th.size), rng.uniform(0, 2 * np.pi, th.size))])
    n = q.shape[0]
    quats = np.ascontiguousarray(q.reshape(1, n, 4))
    iv = np.zeros(1, cases.interval_dtype)
    iv["last"] = n
    idx = np.zeros(1, np.int32)
    hwp = rng.uniform(0, 2 * np.pi, n) if use_hwp else np.zeros(1)
    eps, gamma, cal = np.array([0.1]), np.array([0.3]), np.array([1.7])
    want = np.zeros((1, n, 3))
    oracle.stokes_weights_IQU(idx, quats, idx, want, hwp, iv, eps, gamma, cal, False)
    want_nan = np.isnan(want[0, :, 1])
    assert want_nan.any() and np.array_equal(want_nan, np.isnan(want[0, :, 2]))
    got = np.zeros((1, n, 3))          # (no call of the switch: this is the default)
    hip.stokes_weights_IQU(idx, quats, idx, got, hwp, iv, eps, gamma, cal, False, False)
    assert np.array_equal(np.isnan(got[0, :, 1]), want_nan) and np.array_equal(np.isnan(got[0, :, 2]), want_nan)
    assert not np.isnan(got[0, :, 0]).any()
    ok = ~want_nan
    assert np.max(np.abs(got[0][ok] - want[0][ok])) < 1e-13
    hip.set_stokes_reference_nan(False)
    got2 = np.zeros((1, n, 3))
    hip.stokes_weights_IQU(idx, quats, idx, got2, hwp, iv, eps, gamma, cal, False, False)
    assert not np.isnan(got2).any()


def test_deterministic_detector_groups_at_shard_size(hip, oracle):
    """256 detectors x 720 000 samples (the configs[4] shard; 1.8e8 det-samples = two detector groups of
    the sorted reduction, which is bounded to 2^27 entries): zmap bit-identical to the oracle's host-path
    order, i.e. splitting into groups does not change any sum."""
    import torch

    from toast_amd import capi, synth

    n_det, n_samp, rate, nside, nps, nnz = 256, 720000, 200.0, 512, 3072, 3
    dev = torch.device("cuda")
    D = capi.dev
    st = torch.cuda.current_stream().cuda_stream
    fp, gamma = synth.hex_focalplane(n_det, fov_deg=10.0)
    bore = synth.satellite_boresight(n_samp, rate, 600.0, 30.0, 3000.0, 65.0)
    ivl = synth.make_intervals(n_samp, 3, rate, gap=7)
    idx = np.arange(n_det, dtype=np.int32)
    n_submap = 12 * nside * nside // nps
    sflags_h = synth.shared_flags_block(n_samp, 0.01, value=1)
    d_bore, d_sf = torch.from_numpy(bore).to(dev), torch.from_numpy(sflags_h).to(dev)
    d_hsub = torch.zeros(n_submap, dtype=torch.uint8, device=dev)
    d_pix = torch.full((n_det, n_samp), -1, dtype=torch.int64, device=dev)
    d_w = torch.zeros((n_det, n_samp, 3), dtype=torch.float64, device=dev)
    pt = capi.otf_pointing(d_bore.data_ptr(), fp, nside, True, nnz, d_shared_flags=d_sf.data_ptr(), n_shared_flags=n_samp,
                           shared_flag_mask=1, epsilon=np.zeros(n_det), gamma=gamma, cal=np.ones(n_det))
    D.otf_pixels_healpix(pt, idx, d_pix.data_ptr(), n_samp, ivl, d_hsub.data_ptr(), n_submap, nps, st)
    D.otf_stokes_weights(pt, idx, d_w.data_ptr(), n_samp, ivl, st)
    gen = torch.Generator(device=dev)
    gen.manual_seed(3)
    d_tod = torch.randn((n_det, n_samp), dtype=torch.float64, device=dev, generator=gen)
    d_df = (torch.rand((n_det, n_samp), device=dev, generator=gen) < 0.005).to(torch.uint8)
    g2l_h, hit = synth.global_to_local(d_hsub.cpu().numpy())
    d_g2l = torch.from_numpy(g2l_h).to(dev)
    det_scale = np.linspace(0.5, 1.5, n_det)
    d_z = torch.zeros((hit.size, nps, nnz), dtype=torch.float64, device=dev)
    assert hip.get_deterministic()
    D.build_noise_weighted(d_g2l.data_ptr(), d_z.data_ptr(), nps, nnz, idx, d_pix.data_ptr(), idx, d_w.data_ptr(), idx,
                           d_tod.data_ptr(), idx, d_df.data_ptr(), n_samp, det_scale, 1, n_samp, ivl, d_sf.data_ptr(), n_samp, 1, st)
    torch.cuda.synchronize()
    want = np.zeros((hit.size, nps, nnz))
    oracle.build_noise_weighted(g2l_h, want, idx, d_pix.cpu().numpy(), idx, d_w.cpu().numpy(), idx, d_tod.cpu().numpy(), idx,
                                d_df.cpu().numpy(), det_scale, 1, ivl, sflags_h, 1)
    assert np.array_equal(d_z.cpu().numpy(), want)


def test_deterministic_project_signal_and_dot(hip, oracle):
    """M^T in deterministic mode: every amplitude adds its samples in increasing order like the
    reference's host loop (template_offset.cpp:243-290) -- bit-identical to the oracle; the amplitude
    dot product (block partials summed in block order) is reproducible to the bit."""
    import torch

    rng = np.random.default_rng(8)
    for kw in (dict(n_split=3, gap=5, n_samp=7001), dict(n_samp=4096), dict(n_samp=131, n_split=2, gap=1)):
        c = cases.make_case(**kw)
        ivl = c["intervals"]
        for step in (1, 37, 1000):
            n_amp_views = np.array([(iv["last"] - iv["first"] + step - 1) // step for iv in ivl], dtype=np.int64)
            amp_offset = 5
            n_amp = int(amp_offset + n_amp_views.sum() + 3)
            amps = rng.standard_normal(n_amp)
            aflags = (rng.random(n_amp) < 0.1).astype(np.uint8)
            for fidx in (-1, 2):
                a_h, a_o = amps.copy(), amps.copy()
                hip.template_offset_project_signal(1, c["tod"], fidx, c["det_flags"], 1, step, amp_offset, n_amp_views,
                                                   a_h, aflags, ivl, False)
                oracle.template_offset_project_signal(1, c["tod"], fidx, c["det_flags"], 1, step, amp_offset,
                                                      n_amp_views, a_o, aflags, ivl)
                assert np.array_equal(a_h, a_o), (kw, step, fidx)
    dev = torch.device("cuda")
    x = torch.randn(3_000_001, dtype=torch.float64, device=dev)
    y = torch.randn(3_000_001, dtype=torch.float64, device=dev)
    f = (torch.rand(3_000_001, device=dev) < 0.01).to(torch.uint8)
    vals = {hip.dev.vec_dot(x.numel(), x.data_ptr(), y.data_ptr(), f.data_ptr(), 0) for _ in range(20)}
    assert len(vals) == 1
    want = float((x * y * (f == 0)).sum())
    assert abs(vals.pop() - want) < 1e-9 * abs(want) + 1e-6


def test_mapmaker_with_templates_is_reproducible_in_deterministic_mode(hip):
    """The complete destriping MapMaker (solver covariance, RHS, PCG with offset templates, final binning)
    twice from scratch: amplitudes, residual history and every map bit-identical between the runs."""
    from toast_amd import ops
    from toast_amd.data import defaults
    from toast_amd.sim import create_satellite_data
    from toast_amd.templates import Offset

    outs = []
    for _ in range(2):
        data = create_satellite_data(n_det=8, n_samp=40000, rate=50.0, spin_angle_deg=30.0, prec_angle_deg=60.0)
        rng = np.random.default_rng(2)
        sig = rng.standard_normal((8, 40000)) + np.repeat(3.0 * rng.standard_normal((8, 40)), 1000, axis=1)
        data.obs[0].detdata[defaults.det_data].data[:] = sig
        dp = ops.PointingDetectorSimple()
        pix = ops.PixelsHealpix(detector_pointing=dp, nside=64)
        sw = ops.StokesWeights(detector_pointing=dp, mode="IQU", hwp_angle=defaults.hwp_angle)
        binner = ops.BinMap(pixel_dist="dist", pixel_pointing=pix, stokes_weights=sw, full_pointing=True)
        tm = ops.TemplateMatrix(templates=[Offset(step_time=20.0, noise_model=defaults.noise_model, name="baselines")])
        mm = ops.MapMaker(name="mm", binning=binner, template_matrix=tm, iter_min=8, iter_max=8, convergence=1e-30,
                          keep_solver_products=True, keep_final_products=True)
        mm.apply(data)
        out = {k: data[k].data.copy() for k in ("mm_hits", "mm_cov", "mm_noiseweighted_map", "mm_map")}
        out["amps"] = data["mm_solve_amplitudes"]["baselines"].local.copy()
        out["history"] = np.array(mm.history)
        outs.append(out)
    assert len(outs[0]["history"]) >= 8 and outs[0]["history"][-1] < outs[0]["history"][0]
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), k
