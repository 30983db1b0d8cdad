"""GPU: the Offset template's amplitude-domain noise prior (toast_amd/templates/offset_prior.py,
csrc/offset_prior.hip) against the oracle restatement of offset.py:455-566, 884-1005
(oracle/offset_prior.py: scipy.signal.convolve, scipy.linalg.cho_solve_banded -- the calls the
reference makes on the host).  Tolerance: 1e-11 of the largest output (different summation order
than LAPACK / np.convolve; the inputs are identical)."""
import numpy as np
import pytest
import scipy.linalg
import scipy.signal

from toast_amd import ops
from toast_amd.data import defaults
from toast_amd.sim import create_satellite_data
from toast_amd.templates import AmplitudesMap, Offset

pytestmark = pytest.mark.gpu

TOL = 1.0e-11


@pytest.fixture(scope="module", autouse=True)
def device():
    from toast_amd import accel

    assert accel.accel_enabled()
    accel.accel_assign_device(1, 0, 1.0, False)


def rel(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


def dev_arrays(**arrs):
    import torch

    out = {}
    for k, v in arrs.items():
        out[k] = torch.from_numpy(np.ascontiguousarray(v)).cuda()
    return out


@pytest.mark.parametrize("accumulate", [False, True])
@pytest.mark.parametrize("long_filters", [False, True, None])
def test_convolve_kernel_vs_scipy(accumulate, long_filters):
    import torch

    from toast_amd import capi

    rng = np.random.default_rng(7)
    if long_filters:
        # > 512 taps: the LDS-tiled kernel (segments longer than one 1024-output tile, filters longer
        # than their segment, tap counts that are not multiples of the 256-tap window or of 4)
        seg_len = [1, 5, 1023, 1024, 1025, 3600, 0, 2500, 130]
        filt_len = [3, 4097, 87, 8191, 1, 2051, 3, 513, 257]
    elif long_filters is None:
        # at most 32 taps everywhere: the one-thread-per-amplitude kernel
        seg_len = [1, 2, 5, 63, 64, 65, 300, 1000, 7, 0, 129]
        filt_len = [3, 9, 15, 31, 1, 27, 5, 32, 5, 3, 17]
    else:
        seg_len = [1, 2, 5, 63, 64, 65, 300, 1000, 7, 0, 129]
        filt_len = [3, 9, 15, 87, 1, 131, 87, 511, 5, 3, 257]  # filters longer than their segment included
    seg_start = np.concatenate([[0], np.cumsum(seg_len)]).astype(np.int64)
    n_amp = int(seg_start[-1])
    filters = [rng.standard_normal(n) for n in filt_len]
    filt_start = np.concatenate([[0], np.cumsum(filt_len)[:-1]]).astype(np.int64)
    amp_in = rng.standard_normal(n_amp)
    flags = (rng.random(n_amp) < 0.05).astype(np.uint8)
    prev = rng.standard_normal(n_amp)
    expect = prev.copy() if accumulate else np.zeros(n_amp)
    for s, n in enumerate(seg_len):
        if n == 0:
            continue
        sl = slice(seg_start[s], seg_start[s + 1])
        conv = scipy.signal.convolve(amp_in[sl], filters[s], mode="same", method="direct")
        expect[sl] = expect[sl] + conv if accumulate else conv
    expect[flags != 0] = 0.0
    d = dev_arrays(seg_start=seg_start, filt_start=filt_start, filt_len=np.array(filt_len, dtype=np.int64),
                   filters=np.concatenate(filters), amp_in=amp_in, flags=flags, out=prev)
    capi.dev.offset_convolve(n_amp, len(seg_len), d["seg_start"].data_ptr(), max(seg_len), d["filt_start"].data_ptr(),
                             d["filt_len"].data_ptr(), max(filt_len), d["filters"].data_ptr(), d["amp_in"].data_ptr(),
                             d["flags"].data_ptr(), d["out"].data_ptr(), accumulate)
    torch.cuda.synchronize()
    got = d["out"].cpu().numpy()
    assert rel(got, expect) < TOL
    assert np.all(got[flags != 0] == 0.0)


def pack_factor(cb):
    w, n = cb.shape
    f = np.zeros((n, w))
    b = np.zeros((n, w))
    f[:, 0] = b[:, 0] = 1.0 / cb[0]
    for k in range(1, min(w, n)):
        f[:n - k, k] = cb[k, :n - k]
        b[k:, k] = cb[k, :n - k]
    return f.ravel(), b.ravel()


@pytest.mark.parametrize("width", [1, 2, 20, 64, 65, 128, 200])
def test_banded_solve_kernel_vs_scipy(width):
    import torch

    from toast_amd import capi

    rng = np.random.default_rng(width)
    seg_len = [1, 2, 19, 63, 64, 65, 500, 1000, 129]
    seg_start = np.concatenate([[0], np.cumsum(seg_len)]).astype(np.int64)
    n_amp = int(seg_start[-1])
    amp_in = rng.standard_normal(n_amp)
    flags = (rng.random(n_amp) < 0.05).astype(np.uint8)
    expect = np.zeros(n_amp)
    fwd, bwd, widths, starts = [], [], [], []
    cursor = 0
    for s, n in enumerate(seg_len):
        w = width if s % 2 == 0 else max(1, width // 2)  # ragged widths inside one launch
        # SPD banded matrix: decaying Toeplitz band plus a varying diagonal (offset.py:522-545)
        ab = np.zeros((w, n))
        ab[0] = 3.0 + rng.random(n) * 5.0
        ab += (0.9 ** np.arange(w))[:, None] * (1.0 / w)
        cb = scipy.linalg.cholesky_banded(ab, lower=True)
        sl = slice(seg_start[s], seg_start[s + 1])
        expect[sl] = scipy.linalg.cho_solve_banded((cb, True), amp_in[sl])
        f, b = pack_factor(cb)
        fwd.append(f)
        bwd.append(b)
        widths.append(w)
        starts.append(cursor)
        cursor += n * w
    expect[flags != 0] = 0.0
    d = dev_arrays(seg_start=seg_start, bw=np.array(widths, dtype=np.int32), bs=np.array(starts, dtype=np.int64),
                   fwd=np.concatenate(fwd), bwd=np.concatenate(bwd), amp_in=amp_in, flags=flags,
                   out=np.full(n_amp, np.nan))
    capi.dev.offset_banded_solve(len(seg_len), d["seg_start"].data_ptr(), d["bw"].data_ptr(), max(widths),
                                 d["bs"].data_ptr(), d["fwd"].data_ptr(), d["bwd"].data_ptr(), d["amp_in"].data_ptr(),
                                 d["flags"].data_ptr(), d["out"].data_ptr())
    torch.cuda.synchronize()
    got = d["out"].cpu().numpy()
    assert np.all(np.isfinite(got))
    assert rel(got, expect) < TOL
    assert np.all(got[flags != 0] == 0.0)


@pytest.mark.parametrize("width", [1, 5, 20, 32, 40, 64])
def test_banded_cholesky_kernel_vs_scipy(width):
    """toast_hip_template_offset_banded_cholesky_dev vs scipy.linalg.cholesky_banded, including a
    flagged amplitude (variance 0), a detector without a diagonal term and a matrix that is not
    positive definite (status)."""
    import torch

    from toast_amd import capi

    rng = np.random.default_rng(100 + width)
    seg_len = [1, 3, 19, 64, 65, 700, 129, 50]
    n_seg = len(seg_len)
    seg_start = np.concatenate([[0], np.cumsum(seg_len)]).astype(np.int64)
    n_amp = int(seg_start[-1])
    var = 1.0 / (3.0 + 5.0 * rng.random(n_amp))
    var[seg_start[5] + 11] = 0.0
    widths = np.array([width if s % 2 == 0 else max(1, width // 2) for s in range(n_seg)], dtype=np.int32)
    band_a = (0.9 ** np.arange(64)) / 4.0
    band_a[0] = 2.0                           # positive definite on its own
    band_b = np.concatenate([[0.1], np.full(63, 0.4)])   # needs the diagonal term
    band_bad = np.concatenate([[0.1], np.full(63, -3.0)])
    bands = [band_a, band_b, band_a, band_b, band_a, band_b, band_a, band_bad]
    dscale = np.array([1.0, 1.0, 0.0, 1.0, 1.0, 1.0, 1.0, 1.0])
    tlen = np.minimum(widths, [64, 64, 64, 3, 64, 64, 64, 64]).astype(np.int32)   # one band shorter than the width
    tstart = (np.arange(n_seg) * 64).astype(np.int64)
    start = np.concatenate([[0], np.cumsum(np.array(seg_len) * widths)[:-1]]).astype(np.int64)
    total = int(np.sum(np.array(seg_len) * widths))
    d = dev_arrays(seg_start=seg_start, bw=widths, bs=start, ts=tstart, tl=tlen, toep=np.concatenate(bands),
                   ds=dscale, var=var, fwd=np.full(total, np.nan), bwd=np.zeros(total),
                   status=np.full(n_seg, -1, dtype=np.int32))
    capi.dev.offset_banded_cholesky(n_seg, d["seg_start"].data_ptr(), d["bw"].data_ptr(), int(widths.max()),
                                    d["bs"].data_ptr(), d["ts"].data_ptr(), d["tl"].data_ptr(), d["toep"].data_ptr(),
                                    d["ds"].data_ptr(), d["var"].data_ptr(), d["fwd"].data_ptr(), d["bwd"].data_ptr(),
                                    d["status"].data_ptr())
    torch.cuda.synchronize()
    status = d["status"].cpu().numpy()
    fwd, bwd = d["fwd"].cpu().numpy(), d["bwd"].cpu().numpy()
    for s, n in enumerate(seg_len):
        w = int(widths[s])
        ab = np.zeros((w, n))
        v = var[seg_start[s]:seg_start[s + 1]]
        ab[0] = dscale[s] * np.where(v > 0, 1.0 / np.where(v > 0, v, 1.0), 0.0)
        ab[:tlen[s]] += bands[s][:tlen[s], None]
        try:
            cb = scipy.linalg.cholesky_banded(ab, lower=True)
        except np.linalg.LinAlgError:
            assert status[s] == 1
            continue
        assert status[s] == 0
        f, b = pack_factor(cb)
        assert rel(fwd[start[s]:start[s] + n * w], f) < TOL
        assert rel(bwd[start[s]:start[s] + n * w], b) < TOL
    assert status[-1] == (1 if width > 1 else 0)


def prior_setup(precond_width, n_det=3, n_samp=6000, step_time=5.0, n_intervals=1, view=None):
    data = create_satellite_data(n_det=n_det, n_samp=n_samp, rate=10.0, fknee=0.1, net=2.0, n_intervals=n_intervals,
                                 flag_samples=False)
    tmpl = Offset(step_time=step_time, noise_model=defaults.noise_model, name="baselines", use_noise_prior=True,
                  precond_width=precond_width)
    tmatrix = ops.TemplateMatrix(templates=[tmpl], amplitudes="amps", det_data=defaults.det_data, view=view)
    tmatrix.initialize(data)
    return data, tmpl, tmatrix


@pytest.mark.parametrize("precond_width", [1, 20])
def test_template_prior_vs_oracle(oracle, precond_width):
    """Filters, preconditioners and their application by the Offset template vs the oracle."""
    from oracle import offset_prior as OP

    data, tmpl, tmatrix = prior_setup(precond_width, n_intervals=2, view="scan")
    ob = data.obs[0]
    # baselines follow the whole observation, not the view (offset.py:135-140)
    step = int(5.0 * 10.0 + 0.5)
    n_amp_view = (ob.n_local_samples + step - 1) // step
    assert list(tmpl._obs_views[0]) == [n_amp_view]
    t = ob.shared[defaults.times].data
    freq = OP.prior_freq(float(t[-1] - t[0]), 5.0, tmpl._obs_rate[0])
    noise = ob[defaults.noise_model]
    segments, filters, preconds = [], [], []
    for idet, det in enumerate(tmpl._all_dets):
        opsd = OP.offset_psd(noise.freq(det), noise.psd(det), freq, 5.0)
        filt = OP.view_filter(freq, opsd, n_amp_view, 5.0)
        first = idet * n_amp_view
        segments.append((first, n_amp_view))
        filters.append(filt)
        detnoise = noise.detector_weight(det)
        if precond_width <= 1:
            preconds.append(OP.toeplitz_preconditioner(freq, opsd, n_amp_view, 5.0, detnoise))
        else:
            preconds.append(OP.banded_preconditioner(filt, tmpl._offsetvar[first:first + n_amp_view], precond_width,
                                                     detnoise))
        assert rel(tmpl._prior.filters[idet], filt) < 1e-12
        if precond_width <= 1:
            got_pre, want_pre = tmpl._prior.precond[idet], preconds[-1]
        else:
            # factorised on the device (k_offset_banded_cholesky); entries below the last row of the
            # matrix are unused padding in scipy's layout
            assert tmpl._prior.factor_on_device
            got_pre, want_pre = tmpl._prior.banded_factor(idet), preconds[-1][0].copy()
            for k in range(1, want_pre.shape[0]):
                want_pre[k, n_amp_view - k:] = 0.0
        assert got_pre.shape == want_pre.shape and rel(got_pre, want_pre) < TOL
    amps_in = tmpl.zeros()
    rng = np.random.default_rng(5)
    amps_in.local[:] = rng.standard_normal(amps_in.n_local)
    amps_in.local_flags[rng.random(amps_in.n_local) < 0.03] = 1
    amps_out = amps_in.duplicate()
    amps_out.local[:] = rng.standard_normal(amps_in.n_local)
    want = amps_out.local.copy()
    OP.add_prior(segments, filters, amps_in.local, amps_in.local_flags, want)
    tmpl.add_prior(amps_in, amps_out)
    assert not amps_out.accel_in_use()
    assert rel(amps_out.local, want) < TOL
    want = np.zeros_like(want)
    OP.apply_precond(segments, preconds, precond_width, amps_in.local, amps_in.local_flags, want)
    tmpl.apply_precond(amps_in, amps_out)
    assert rel(amps_out.local, want) < TOL
    # resident vectors stay on the device
    amps_in.accel_resident("t_in")
    amps_out.accel_resident("t_out")
    amps_out.reset()
    tmpl.apply_precond(amps_in, amps_out)
    assert amps_out.accel_in_use()
    amps_out.accel_update_host()
    assert rel(amps_out.local, want) < TOL
    tmatrix.reset()


def test_fused_lhs_with_prior_equals_operator_sequence_plus_oracle_prior(oracle):
    """SolverLHS with the noise prior: fused kernels == operator sequence == (LHS without prior)
    + oracle add_prior."""
    from oracle import offset_prior as OP
    from test_gpu_ops import make_solver_setup

    results = {}
    for label, fused, prior in (("ops", False, True), ("fused", True, True), ("noprior", True, False)):
        data, pix, sw, truth, sky = make_solver_setup(n_det=5, n_samp=9000)
        ops.CovarianceAndHits(pixel_dist="dist", covariance="cov", pixel_pointing=pix, stokes_weights=sw,
                              save_pointing=True).apply(data)
        lhs_bin = ops.BinMap(pixel_dist="dist", covariance="cov", binned="lhs_bin", pixel_pointing=pix,
                             stokes_weights=sw, full_pointing=True)
        tmpl = Offset(step_time=7.3, noise_model=defaults.noise_model, name="baselines", good_fraction=0.2,
                      use_noise_prior=prior)
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
        lhs.apply(data)
        results[label] = data["lhs_out"]["baselines"].local.copy()
        if label == "fused":
            segs = [(int(a), int(b - a)) for a, b in zip(tmpl._prior.seg_start[:-1], tmpl._prior.seg_start[1:])]
            extra = np.zeros(amps.n_local)
            OP.add_prior(segs, tmpl._prior.filters, amps.local, amps.local_flags, extra)
            assert np.max(np.abs(extra)) > 0
    scale = np.max(np.abs(results["ops"]))
    assert np.max(np.abs(results["ops"] - results["fused"])) < TOL * scale
    assert np.max(np.abs(results["noprior"] + extra - results["fused"])) < TOL * scale


def test_mapmaker_with_noise_prior_damps_the_baselines():
    """MapMaker with and without the prior on the same data: both solves converge; the prior
    (a finite baseline covariance) shrinks the solution towards smooth baselines but keeps it
    close to the unregularised one where the data constrain it."""
    from test_gpu_ops import make_solver_setup

    sols = {}
    for prior in (False, True):
        data, pix, sw, truth, sky = make_solver_setup(n_det=4, n_samp=12000, step_time=20.0, noise_rms=0.01)
        binner = ops.BinMap(pixel_dist="dist", pixel_pointing=pix, stokes_weights=sw, full_pointing=True)
        tmpl = Offset(step_time=20.0, noise_model=defaults.noise_model, name="baselines", good_fraction=0.2,
                      use_noise_prior=prior, precond_width=20)
        tmatrix = ops.TemplateMatrix(templates=[tmpl])
        mapper = ops.MapMaker(name="mm", keep_solver_products=True, det_data=defaults.det_data, binning=binner,
                              template_matrix=tmatrix, iter_max=200, convergence=1e-16, solve_rcond_threshold=1e-3,
                              map_rcond_threshold=1e-3)
        mapper.apply(data)
        assert mapper.history[-1] < 1e-10 and len(mapper.history) < 200
        amps = data["mm_solve_amplitudes"]["baselines"]
        if amps.accel_in_use():
            amps.accel_update_host()
        sols[prior] = amps.local.copy()
        assert np.all(np.isfinite(sols[prior]))
    # the offsets have an arbitrary common level without the prior; compare after removing means
    a = sols[False] - np.mean(sols[False])
    b = sols[True] - np.mean(sols[True])
    assert np.std(a) > 1.0  # offsets of rms 3 were injected
    assert np.corrcoef(a, b)[0, 1] > 0.9
    assert np.std(b) < 1.02 * np.std(a)
    assert 0.0 < np.std(a - b) < 0.6 * np.std(a)


def test_pattern_select_with_toeplitz_prior():
    """The reference's test_pattern_select (src/toast/tests/ops_mapmaker.py:668-808): maps of the
    A and B detectors and of all of them, Offset template with the noise prior and the Toeplitz
    preconditioner (precond_width=1); the hit maps must add up and the detector flags be restored."""
    from test_gpu_ops import make_solver_setup

    data, pix, sw, truth, sky = make_solver_setup(n_det=6, n_samp=9000, step_time=3.0, noise_rms=0.05)
    ob = data.obs[0]
    flags_before = dict(ob.local_detector_flags)
    binner = ops.BinMap(pixel_dist="dist", pixel_pointing=pix, stokes_weights=sw, noise_model=defaults.noise_model)
    tmpl = Offset(times=defaults.times, noise_model=defaults.noise_model, step_time=3.0, use_noise_prior=True,
                  precond_width=1, name="baselines")
    mapper = ops.MapMaker(name="mapmaker", det_data=defaults.det_data, binning=binner,
                          template_matrix=ops.TemplateMatrix(templates=[tmpl]), solve_rcond_threshold=1.0e-1,
                          map_rcond_threshold=1.0e-1, iter_max=5)
    hits = {}
    for name, pattern in (("map_A", ".*A"), ("map_B", ".*B"), ("map_total", None)):
        mapper.name = name
        mapper.pattern = pattern
        mapper.apply(data)
        hits[name] = data[f"{name}_hits"].data.copy()
        assert np.all(np.isfinite(data[f"{name}_map"].data))
        assert ob.local_detector_flags == flags_before
    good = (hits["map_A"] > 0) & (hits["map_B"] > 0)
    assert np.count_nonzero(good) > 100
    assert np.array_equal(hits["map_total"][good], hits["map_A"][good] + hits["map_B"][good])
    assert hits["map_A"].sum() > 0 and hits["map_A"].sum() + hits["map_B"].sum() == hits["map_total"].sum()


def test_many_short_segments():
    """More (detector, observation, view) segments than one grid dimension holds (65535): the
    tiled convolution, the factorisation and the solve index segments on grid.x."""
    import torch

    from toast_amd import capi

    rng = np.random.default_rng(1)
    n_seg, n, w = 70001, 8, 4
    n_amp = n_seg * n
    filt = np.exp(-np.abs(np.arange(41) - 20) / 3.0)
    x = rng.standard_normal(n_amp)
    seg_start = (np.arange(n_seg + 1) * n).astype(np.int64)
    d = dev_arrays(seg_start=seg_start, zero=np.zeros(n_seg, dtype=np.int64), flen=np.full(n_seg, filt.size, dtype=np.int64),
                   filt=filt, x=x, flags=np.zeros(n_amp, dtype=np.uint8), out=np.zeros(n_amp))
    capi.dev.offset_convolve(n_amp, n_seg, d["seg_start"].data_ptr(), n, d["zero"].data_ptr(), d["flen"].data_ptr(),
                             filt.size, d["filt"].data_ptr(), d["x"].data_ptr(), d["flags"].data_ptr(),
                             d["out"].data_ptr(), False)
    torch.cuda.synchronize()
    got = d["out"].cpu().numpy().reshape(n_seg, n)
    for s in (0, 1, 65535, 65536, n_seg - 1):
        want = scipy.signal.convolve(x[s * n:(s + 1) * n], filt, mode="same", method="direct")
        assert rel(got[s], want) < TOL
    band = np.array([2.0, 0.5, 0.2, 0.1])
    var = 1.0 / (1.0 + rng.random(n_amp))
    e = dev_arrays(bw=np.full(n_seg, w, dtype=np.int32), bs=(np.arange(n_seg) * n * w).astype(np.int64), band=band,
                   ones=np.ones(n_seg), var=var, fwd=np.zeros(n_amp * w), bwd=np.zeros(n_amp * w),
                   status=np.full(n_seg, -1, dtype=np.int32), sol=np.zeros(n_amp))
    capi.dev.offset_banded_cholesky(n_seg, d["seg_start"].data_ptr(), e["bw"].data_ptr(), w, e["bs"].data_ptr(),
                                    d["zero"].data_ptr(), e["bw"].data_ptr(), e["band"].data_ptr(), e["ones"].data_ptr(),
                                    e["var"].data_ptr(), e["fwd"].data_ptr(), e["bwd"].data_ptr(), e["status"].data_ptr())
    capi.dev.offset_banded_solve(n_seg, d["seg_start"].data_ptr(), e["bw"].data_ptr(), w, e["bs"].data_ptr(),
                                 e["fwd"].data_ptr(), e["bwd"].data_ptr(), d["x"].data_ptr(), d["flags"].data_ptr(),
                                 e["sol"].data_ptr())
    torch.cuda.synchronize()
    assert int(e["status"].abs().max()) == 0
    sol = e["sol"].cpu().numpy().reshape(n_seg, n)
    for s in (0, 65535, 65536, n_seg - 1):
        ab = np.zeros((w, n))
        ab[0] = 1.0 / var[s * n:(s + 1) * n]
        ab += band[:, None]
        want = scipy.linalg.solveh_banded(ab, x[s * n:(s + 1) * n], lower=True)
        assert rel(sol[s], want) < TOL


def test_prior_with_two_observations_fused_equals_operator_sequence(oracle):
    """Two observations: the amplitude blocks are ordered detector -> observation -> view
    (offset.py:240-251) and so are the prior's segments; fused and operator-sequence LHS agree,
    and the prior term equals the oracle's per-segment convolution."""
    from oracle import offset_prior as OP

    results = {}
    for fused in (False, True):
        data = create_satellite_data(n_det=4, n_obs=2, n_samp=4000, rate=10.0, fknee=0.1, net=2.0, flag_samples=True)
        rng = np.random.default_rng(4)
        for ob in data.obs:
            ob.detdata[defaults.det_data].data[:] = rng.standard_normal(ob.detdata[defaults.det_data].data.shape)
        dp = ops.PointingDetectorSimple()
        pix = ops.PixelsHealpix(detector_pointing=dp, nside=16, nside_submap=4)
        sw = ops.StokesWeights(detector_pointing=dp, mode="IQU", hwp_angle=defaults.hwp_angle)
        ops.CovarianceAndHits(pixel_dist="dist", covariance="cov", pixel_pointing=pix, stokes_weights=sw,
                              save_pointing=True).apply(data)
        lhs_bin = ops.BinMap(pixel_dist="dist", covariance="cov", binned="lhs_bin", pixel_pointing=pix,
                             stokes_weights=sw, full_pointing=True)
        tmpl = Offset(step_time=8.0, noise_model=defaults.noise_model, name="baselines", good_fraction=0.2,
                      use_noise_prior=True)
        tmatrix = ops.TemplateMatrix(templates=[tmpl], amplitudes="amps_in", det_data="temp_LHS")
        tmatrix.initialize(data)
        n_amp_obs = (4000 + 79) // 80
        assert tmpl._prior.seg_start.size - 1 == 4 * 2 and int(tmpl._prior.seg_start[-1]) == 8 * n_amp_obs
        amps = tmpl.zeros()
        amps.local[:] = np.random.default_rng(3).standard_normal(amps.n_local)
        data["amps_in"] = AmplitudesMap(baselines=amps)
        data["lhs_out"] = data["amps_in"].duplicate()
        data["lhs_out"].reset()
        lhs = ops.SolverLHS(binning=lhs_bin, template_matrix=tmatrix, out="lhs_out", fused=fused)
        assert lhs._can_fuse(data) == fused
        lhs.apply(data)
        results[fused] = data["lhs_out"]["baselines"].local.copy()
        if fused:
            segs = [(int(a), int(b - a)) for a, b in zip(tmpl._prior.seg_start[:-1], tmpl._prior.seg_start[1:])]
            prior_only = np.zeros(amps.n_local)
            OP.add_prior(segs, tmpl._prior.filters, amps.local, amps.local_flags, prior_only)
            got = np.zeros_like(prior_only)
            out = tmpl.zeros()
            tmpl.add_prior(amps, out)
            assert rel(out.local, prior_only) < TOL
    scale = np.max(np.abs(results[False]))
    assert scale > 0 and np.max(np.abs(results[False] - results[True])) < TOL * scale
