"""GPU: the ground-filter kernels (csrc/ground_filter.hip) and ops.GroundFilter against the
reference's own kernel outputs (tests/golden/cov_filter.npz) and the oracle restatement of
groundfilter.py (oracle/ground_filter.py); plus the reference's operator tests
(src/toast/tests/ops_groundfilter.py: residual rms after filtering an injected ground signal)."""
import numpy as np
import pytest

import golden_util as gu
from toast_amd import ops
from toast_amd.data import defaults
from toast_amd.sim import create_ground_data

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def device():
    from toast_amd import accel

    assert accel.accel_enabled()
    accel.accel_assign_device(1, 0, 1.0, False)


def dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_legendre_kernel_bit_exact_vs_reference_fixture():
    import torch

    from toast_amd import capi

    g = gu.load("cov_filter")
    for key, start, stop, want in (("gf_x", 1, 4, g["gf_trend"]), ("gf_phase", 0, 6, g["gf_poly"])):
        x = dev(g[key])
        out = torch.full((stop - start, x.numel()), float("nan"), dtype=torch.float64, device="cuda")
        capi.dev.legendre_templates(x.data_ptr(), x.numel(), start, stop, out.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), want)


def test_fit_and_subtract_kernels_vs_reference_fixture():
    import torch

    from toast_amd import capi

    g = gu.load("cov_filter")
    templates = np.vstack([g["gf_trend"], g["gf_poly"]])
    nt, n = templates.shape
    sig, good = g["gf_signal"], g["gf_good"]
    # express `good` as shared & detector flags: common = bad in both rows, rest per detector
    shared = ((good[0] == 0) & (good[1] == 0)).astype(np.uint8)
    dflags = ((good == 0) & (shared[None, :] == 0)).astype(np.uint8) * 2
    # rows in a larger buffer, permuted
    buf = np.zeros((4, n))
    buf[3], buf[1] = sig[0], sig[1]
    fbuf = np.zeros((3, n), dtype=np.uint8)
    fbuf[2], fbuf[0] = dflags[0], dflags[1]
    d_t, d_s, d_f, d_sh = dev(templates), dev(buf), dev(fbuf), dev(shared)
    proj = torch.full((2, nt), float("nan"), dtype=torch.float64, device="cuda")
    gram = torch.full((nt, nt), float("nan"), dtype=torch.float64, device="cuda")
    dgram = torch.full((2, nt, nt), float("nan"), dtype=torch.float64, device="cuda")
    nflag = torch.full((2,), -1, dtype=torch.int64, device="cuda")
    capi.dev.template_fit(d_t.data_ptr(), nt, n, [3, 1], d_s.data_ptr(), [2, 0], d_f.data_ptr(), 2, d_sh.data_ptr(), 1,
                          proj.data_ptr(), gram.data_ptr(), dgram.data_ptr(), nflag.data_ptr())
    torch.cuda.synchronize()
    proj, invcov = proj.cpu().numpy(), (gram[None] - dgram).cpu().numpy()
    assert list(nflag.cpu().numpy()) == [int(np.count_nonzero(dflags[0])), int(np.count_nonzero(dflags[1]))]
    for d in range(2):
        assert np.max(np.abs(proj[d] - g[f"gf_proj{d}"])) < 1e-12 * np.max(np.abs(g[f"gf_proj{d}"]))
        assert np.max(np.abs(invcov[d] - g[f"gf_invcov{d}"])) < 1e-12 * np.max(np.abs(g[f"gf_invcov{d}"]))
    # subtraction with the reference's coefficients: exactly signal - fit
    coeff = np.stack([g["gf_coeff0"], g["gf_coeff1"]])
    d_c = dev(coeff)
    capi.dev.template_subtract(d_t.data_ptr(), nt, 3, n, [3, 1], d_s.data_ptr(), d_c.data_ptr())
    torch.cuda.synchronize()
    res = d_s.cpu().numpy()
    assert np.array_equal(res[3], sig[0] - g["gf_fit0"])
    assert np.array_equal(res[1], sig[1] - g["gf_fit1"])
    assert np.array_equal(res[0], np.zeros(n)) and np.array_equal(res[2], np.zeros(n))


def test_fit_kernel_many_detectors_and_templates_vs_oracle(oracle):
    """More templates than one LDS group, more detectors than one block, ragged sizes."""
    import torch

    from oracle import ground_filter as GF
    from toast_amd import capi

    rng = np.random.default_rng(2)
    n, n_det, nt = 20011, 37, 41
    x = np.sort(rng.uniform(-1, 1, n))
    templates = GF.legendre_templates(x, 0, nt)
    sig = rng.standard_normal((n_det, n))
    shared = (rng.random(n) < 0.05).astype(np.uint8)
    dflags = (rng.random((n_det, n)) < 0.02).astype(np.uint8)
    dflags[5] = 1   # a detector without a single good sample
    d_t, d_s, d_f, d_sh = dev(templates), dev(sig), dev(dflags), dev(shared)
    proj = torch.zeros((n_det, nt), dtype=torch.float64, device="cuda")
    gram = torch.zeros((nt, nt), dtype=torch.float64, device="cuda")
    dgram = torch.zeros((n_det, nt, nt), dtype=torch.float64, device="cuda")
    nflag = torch.zeros(n_det, dtype=torch.int64, device="cuda")
    idx = np.arange(n_det, dtype=np.int32)
    capi.dev.template_fit(d_t.data_ptr(), nt, n, idx, d_s.data_ptr(), idx, d_f.data_ptr(), 1, d_sh.data_ptr(), 1,
                          proj.data_ptr(), gram.data_ptr(), dgram.data_ptr(), nflag.data_ptr())
    torch.cuda.synchronize()
    proj, invcov = proj.cpu().numpy(), (gram[None] - dgram).cpu().numpy()
    want_n = np.count_nonzero((dflags != 0) & (shared[None, :] == 0), axis=1)
    assert np.array_equal(nflag.cpu().numpy(), want_n)
    for d in (0, 5, 17, 36):
        good = ((shared == 0) & (dflags[d] == 0)).astype(np.uint8)
        want_p = GF.bin_proj(sig[d], templates, good)
        want_i = GF.bin_invcov(templates, good)
        scale = np.max(np.abs(GF.bin_invcov(templates, (shared == 0).astype(np.uint8))))
        assert np.max(np.abs(proj[d] - want_p)) < 1e-12 * max(np.max(np.abs(want_p)), 1.0)
        assert np.max(np.abs(invcov[d] - want_i)) < 1e-12 * scale
    # no flags at all
    capi.dev.template_fit(d_t.data_ptr(), nt, n, idx, d_s.data_ptr(), None, 0, 1, 0, 1, proj_t := torch.zeros(
        (n_det, nt), dtype=torch.float64, device="cuda").data_ptr(), gram.data_ptr(), dgram.data_ptr(), nflag.data_ptr())
    torch.cuda.synchronize()
    want = GF.bin_invcov(templates, np.ones(n, dtype=np.uint8))
    assert np.max(np.abs(gram.cpu().numpy() - want)) < 1e-12 * np.max(np.abs(want))
    assert float(dgram.abs().max()) == 0.0


def inject_ground(data, rng, amp=5.0, white=1.0):
    """signal = white noise + a smooth function of azimuth (different for the two directions)."""
    truth = {}
    for ob in data.obs:
        az = ob.shared[defaults.azimuth].data
        phase = (az - az.min()) / (az.max() - az.min()) * 2 - 1
        direction = np.zeros(ob.n_local_samples)
        for iv in ob.intervals[defaults.throw_leftright_interval]:
            direction[iv.first:iv.last] = 1
        for iv in ob.intervals[defaults.throw_rightleft_interval]:
            direction[iv.first:iv.last] = -1
        for det in ob.local_detectors:
            ground = amp * (np.sin(3 * phase) + 0.5 * phase ** 2) + 0.3 * amp * direction * np.cos(2 * phase)
            noise = white * rng.standard_normal(ob.n_local_samples)
            ob.detdata[defaults.det_data][det] = ground + noise
            truth[(ob.name, det)] = noise
    return truth


@pytest.mark.parametrize("split,bin_width,orders", [(False, None, (3, 6)), (True, None, (3, 6)),
                                                   (True, np.radians(1.0), (None, None))])
def test_operator_vs_oracle_and_rms(oracle, split, bin_width, orders):
    from oracle import ground_filter as GF

    rng = np.random.default_rng(31)
    data = create_ground_data(n_det=6, n_samp=24000, rate=20.0, n_obs=2)
    inject_ground(data, rng)
    before = {ob.name: ob.detdata[defaults.det_data].data.copy() for ob in data.obs}
    trend_order, filter_order = orders
    gf = ops.GroundFilter(trend_order=trend_order, filter_order=filter_order, bin_width=bin_width, split_template=split,
                          detrend=trend_order is None, name="gf")
    gf.apply(data)
    # (azimuth bins visited only during the flagged turnarounds have no good sample: pseudo-inverse)
    assert gf.ngood + gf.nsingular == 12 and (bin_width is not None or gf.nsingular == 0)
    for ob in data.obs:
        n = ob.n_local_samples
        lr = np.zeros(n, dtype=bool)
        rl = np.zeros(n, dtype=bool)
        for iv in ob.intervals[defaults.throw_leftright_interval]:
            lr[iv.first:iv.last] = True
        for iv in ob.intervals[defaults.throw_rightleft_interval]:
            rl[iv.first:iv.last] = True
        templates = GF.build_templates(n, ob.shared[defaults.azimuth].data, trend_order, filter_order,
                                       bin_width=bin_width, split=split, lr_mask=lr, rl_mask=rl)
        want = before[ob.name].copy()
        failed = GF.apply(want, ob.detdata[defaults.det_flags].data, 1, ob.shared[defaults.shared_flags].data, 1,
                          templates, trend_order, trend_order is None)
        assert failed == []
        got = ob.detdata[defaults.det_data].data
        assert np.max(np.abs(got - want)) < 1e-9 * np.max(np.abs(before[ob.name]))
        # the reference's own check (ops_groundfilter.py:140-149): the ground signal is gone
        good = (ob.shared[defaults.shared_flags].data & 1) == 0
        for i, det in enumerate(ob.local_detectors):
            g = good & ((ob.detdata[defaults.det_flags][det] & 1) == 0)
            old_rms, new_rms = np.std(before[ob.name][i][g]), np.std(got[i][g])
            if split:
                assert new_rms < 0.4 * old_rms and new_rms < 1.1
            else:
                assert new_rms < old_rms


@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("TOAST_TEST_FUZZ_SEEDS", "10")))))
def test_operator_random_configurations(oracle, seed):
    """Randomly drawn GroundFilter set-ups (detector and observation counts, lengths, scan rates, polynomial orders,
    split templates, detrending, random sample flags, a cut detector) against the oracle restatement of
    groundfilter.py."""
    from oracle import ground_filter as GF

    rng = np.random.default_rng(9000 + seed)
    n_det = int(rng.integers(1, 8))
    n_obs = int(rng.integers(1, 3))
    n_samp = int(rng.integers(6000, 30000))
    trend_order = [None, 0, 2, 5][int(rng.integers(0, 4))]
    filter_order = int(rng.integers(1, 9))
    split = bool(rng.integers(0, 2))
    data = create_ground_data(n_det=n_det, n_samp=n_samp, rate=float(rng.choice([20.0, 50.0])), n_obs=n_obs,
                              scan_rate_deg_s=float(rng.choice([1.0, 2.5])), seed=seed)
    inject_ground(data, rng)
    for ob in data.obs:
        fl = ob.detdata[defaults.det_flags].data
        fl |= (rng.random(fl.shape) < float(rng.choice([0.0, 0.01, 0.1]))).astype(np.uint8)
    cut = None
    if n_det > 1 and rng.integers(0, 2):
        cut = data.obs[0].local_detectors[int(rng.integers(0, n_det))]
        data.obs[0].update_local_detector_flags({cut: 1})
    before = {ob.name: ob.detdata[defaults.det_data].data.copy() for ob in data.obs}
    gf = ops.GroundFilter(trend_order=trend_order, filter_order=filter_order, split_template=split,
                          detrend=trend_order is None, name="gf")
    gf.apply(data)
    for iob, ob in enumerate(data.obs):
        n = ob.n_local_samples
        lr = np.zeros(n, dtype=bool)
        rl = np.zeros(n, dtype=bool)
        for iv in ob.intervals[defaults.throw_leftright_interval]:
            lr[iv.first:iv.last] = True
        for iv in ob.intervals[defaults.throw_rightleft_interval]:
            rl[iv.first:iv.last] = True
        templates = GF.build_templates(n, ob.shared[defaults.azimuth].data, trend_order, filter_order, bin_width=None,
                                       split=split, lr_mask=lr, rl_mask=rl)
        rows = [i for i, d in enumerate(ob.local_detectors) if not (iob == 0 and d == cut)]
        want = before[ob.name].copy()
        sub = np.ascontiguousarray(want[rows])
        GF.apply(sub, np.ascontiguousarray(ob.detdata[defaults.det_flags].data[rows]), 1,
                 ob.shared[defaults.shared_flags].data, 1, templates, trend_order, trend_order is None)
        want[rows] = sub
        got = ob.detdata[defaults.det_data].data
        assert np.max(np.abs(got - want)) < 1e-9 * np.max(np.abs(before[ob.name])), (seed, ob.name)


def test_operator_detrend_flags_and_resident_data():
    rng = np.random.default_rng(5)
    data = create_ground_data(n_det=4, n_samp=12000, rate=20.0)
    inject_ground(data, rng)
    ob = data.obs[0]
    ob.detdata[defaults.det_flags].data[1] = defaults.det_mask_invalid      # detector 1: nothing to fit
    sig = ob.detdata[defaults.det_data]
    t = np.arange(ob.n_local_samples) / ob.n_local_samples
    for det in ob.local_detectors:
        sig[det] = sig[det] + 40.0 * t      # a linear drift
    before = sig.data.copy()
    sig.accel_create(defaults.det_data)
    sig.accel_update_device()
    gf = ops.GroundFilter(trend_order=2, filter_order=5, detrend=True, split_template=True, name="gf")
    gf.apply(data)
    assert sig.accel_in_use()                      # stays where the caller keeps it
    after = sig.data                               # lazy host sync
    det1 = ob.local_detectors[1]
    assert ob.local_detector_flags[det1] & defaults.det_mask_invalid
    assert np.array_equal(after[1], before[1])
    good = (ob.shared[defaults.shared_flags].data & 1) == 0
    for i in (0, 2, 3):
        g = good & ((ob.detdata[defaults.det_flags].data[i] & 1) == 0)
        assert np.std(after[i][g]) < 1.1           # drift and ground template removed
        slope = np.polyfit(t[g], after[i][g], 1)[0]
        assert abs(slope) < 0.2
    assert set(gf.coefficients) == {ob.local_detectors[i] for i in (0, 2, 3)}


def test_azimuth_from_boresight_azel():
    """Without an azimuth key the azimuth comes from the Az/El boresight quaternions:
    az = 2 pi - phi with phi in [0, 2 pi) (groundfilter.py:285-289, qa.to_iso_angles)."""
    from toast_amd.synth import quat_mult, quat_rotation

    data = create_ground_data(n_det=2, n_samp=4000, rate=20.0)
    ob = data.obs[0]
    az = np.array(ob.shared[defaults.azimuth].data)
    el = np.radians(50.0)
    # Az/El boresight: azimuth is measured the other way round from the ISO phi
    q = quat_mult(quat_rotation(np.array([0.0, 0.0, 1.0]), -az), quat_rotation(np.array([0.0, 1.0, 0.0]), np.pi / 2 - el))
    ob.shared.create(defaults.boresight_azel, np.ascontiguousarray(q))
    gf = ops.GroundFilter(azimuth=None, name="gf")
    got = gf._azimuth(ob)
    assert np.max(np.abs(got - az)) < 1e-12 and got.min() > 0 and got.max() <= 2 * np.pi
    # crossing the meridian: azimuths just below 2 pi and just above 0 stay in (0, 2 pi]
    az2 = np.array([6.2, 6.28, 0.01, 0.1])
    q2 = quat_mult(quat_rotation(np.array([0.0, 0.0, 1.0]), -az2), quat_rotation(np.array([0.0, 1.0, 0.0]), np.pi / 2 - el))
    ob2 = create_ground_data(n_det=1, n_samp=4, rate=1.0).obs[0]
    ob2.shared.create(defaults.boresight_azel, np.ascontiguousarray(q2))
    assert np.max(np.abs(ops.GroundFilter(azimuth=None, name="g2")._azimuth(ob2) - az2)) < 1e-12
