"""GPU: rocFFT noise-weighting pipeline against the NumPy restatement of toast.fft.convolve
(oracle/fft_oracle.py), tolerance 1e-12 relative to the signal scale, plus the reference's
own two-tone low-pass criterion and the FFTW half-complex plan semantics."""
import numpy as np
import pytest

from test_fft_oracle import butter_kernel, check_lowpass, two_tone

pytestmark = pytest.mark.gpu

TOL = 1e-12


@pytest.fixture(scope="module")
def pf():
    from toast_amd import capi, fft

    assert capi.accel_enabled()
    capi.accel_assign_device(1, 0, 1.0, False)
    return fft


def test_two_tone_lowpass_like_reference(pf):
    from oracle import fft_oracle as fo

    rate, n_samp = 200.0, 12345
    times, orig, lowf = two_tone(rate, n_samp)
    kfreqs = np.fft.rfftfreq(fo.fft_length(n_samp), d=1.0 / rate)
    kvals, shift = butter_kernel(rate, 4, kfreqs, np.array([5.0]))
    assert pf.fft_length(n_samp) == fo.fft_length(n_samp) == 32768
    got = orig.copy()
    pf.convolve(got, rate, kernel_freq=kfreqs, kernels=kvals)
    check_lowpass(times, got, lowf, shift[0], rate)  # reference criterion, tests/fft.py:225-237
    want = orig.copy()
    fo.convolve(want, rate, kernel_freq=kfreqs, kernels=kvals)
    assert np.max(np.abs(got - want)) < TOL * np.max(np.abs(want))
    for algo in ("internal", None):
        g2 = orig.copy()
        pf.convolve(g2, rate, kernel_freq=kfreqs, kernels=kvals, algorithm=algo)
        assert np.array_equal(g2, got)


@pytest.mark.parametrize("n_samp", [1000, 4096, 50001])
def test_noise_filter_kernels_per_detector(pf, n_samp):
    """NoiseFilter-style real inverse-PSD kernels on a log frequency grid, one per detector,
    with detector row indirection (reference ops/noise_filter.py:130-188)."""
    from oracle import fft_oracle as fo

    rng = np.random.default_rng(n_samp)
    rate, n_det, rows = 100.0, 5, 7
    freq = np.concatenate([[0.0], np.geomspace(1e-5, rate / 2, 70)])
    kernels = []
    for d in range(n_det):
        net = 1.0 + 0.1 * d
        fknee = 0.05 * (d + 1)
        psd = net**2 * (freq + fknee) / np.maximum(freq + 1e-5, 1e-12)
        kernels.append(fo.noise_filter_kernel(psd, net))
    kernels = np.array(kernels)
    buf = rng.standard_normal((rows, n_samp)).cumsum(axis=1) * 0.01 + rng.standard_normal((rows, n_samp))
    idx = np.array([5, 0, 3, 6, 2], dtype=np.int32)
    want = buf.copy()
    sub = np.ascontiguousarray(buf[idx])
    fo.convolve(sub, rate, kernel_freq=freq, kernels=kernels)
    want[idx] = sub
    got = buf.copy()
    pf.convolve_buffer(got, idx, rate, freq, kernels)
    assert np.max(np.abs(got - want)) < TOL * np.max(np.abs(want))
    untouched = [r for r in range(rows) if r not in idx]
    assert np.array_equal(got[untouched], buf[untouched])


@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("TOAST_TEST_FUZZ_SEEDS", "12")))))
def test_convolve_random_cases(pf, seed):
    """Randomly drawn timestream lengths (1 .. 150 000: both the rocFFT pipeline and the fused three-pass kernels and
    every padding / reflection regime of toast.fft.convolve), detector counts, row indirection, knot counts, real or
    complex kernels, one kernel per detector or a common one, deconvolution -- against the oracle, 1e-12."""
    from oracle import fft_oracle as fo

    rng = np.random.default_rng(7000 + seed)
    n_samp = int(rng.choice([rng.integers(2, 200), rng.integers(200, 5000), rng.integers(5000, 150000)]))
    n_det = int(rng.integers(1, 7))
    rows = n_det + int(rng.integers(0, 3))
    rate = float(rng.choice([10.0, 37.5, 200.0]))
    n_knot = int(rng.integers(4, 90))
    freq = np.concatenate([[0.0], np.geomspace(1e-4, rate / 2, n_knot - 1)])
    common = bool(rng.integers(0, 2))
    cplx = bool(rng.integers(0, 2))
    deconvolve = bool(rng.integers(0, 2))
    nk = 1 if common else n_det
    mag = 0.2 + rng.random((nk, n_knot))
    kernels = mag * np.exp(1j * rng.uniform(-0.5, 0.5, (nk, n_knot))) if cplx else mag
    if cplx:
        kernels[:, 0] = kernels[:, 0].real
    kern = kernels[0] if common else kernels
    buf = rng.standard_normal((rows, n_samp)) + 0.01 * rng.standard_normal((rows, n_samp)).cumsum(axis=1)
    idx = rng.permutation(rows)[:n_det].astype(np.int32)
    want = buf.copy()
    sub = np.ascontiguousarray(buf[idx])
    fo.convolve(sub, rate, kernel_freq=freq, kernels=kern, deconvolve=deconvolve)
    want[idx] = sub
    got = buf.copy()
    pf.convolve_buffer(got, idx, rate, freq, kern, deconvolve=deconvolve)
    assert np.max(np.abs(got - want)) < TOL * np.max(np.abs(want)), (n_samp, n_det, common, cplx, deconvolve)
    untouched = [r for r in range(rows) if r not in idx]
    assert np.array_equal(got[untouched], buf[untouched])


def test_deconvolve_and_flags(pf):
    from oracle import fft_oracle as fo

    rng = np.random.default_rng(4)
    rate, n_samp, n_tod = 50.0, 6000, 3
    freq = np.linspace(0, rate / 2, 200)
    kern = (1.0 / (1.0 + (freq / 5.0) ** 2)) * np.exp(-1j * 0.02 * freq)
    data = rng.standard_normal((n_tod, n_samp))
    flags = (rng.random((n_tod, n_samp)) < 0.002).astype(np.uint8)
    got, gflags = data.copy(), flags.copy()
    want, wflags = data.copy(), flags.copy()
    pf.convolve(got, rate, flags=gflags, flag_mask=1, kernel_freq=freq, kernels=kern)
    fo.convolve(want, rate, flags=wflags, flag_mask=1, kernel_freq=freq, kernels=kern)
    assert np.max(np.abs(got - want)) < TOL * np.max(np.abs(want))
    assert np.array_equal(gflags, wflags)
    g2, w2 = data.copy(), data.copy()
    pf.convolve(g2, rate, kernel_freq=freq, kernels=kern, deconvolve=True)
    fo.convolve(w2, rate, kernel_freq=freq, kernels=kern, deconvolve=True)
    assert np.max(np.abs(g2 - w2)) < 1e-11 * np.max(np.abs(w2))


def test_r1d_half_complex_plans(pf):
    """FFTPlanReal1D semantics: forward unscaled r2hc, backward scaled by 1/length; round trip
    (reference tests src/toast/tests/fft.py:42-94)."""
    from oracle import fft_oracle as fo

    rng = np.random.default_rng(2)
    for n in (64, 1000, 65536):
        x = rng.standard_normal((4, n))
        hc = pf.r1d_forward(x)
        want = fo.r1d_forward(x)
        assert np.max(np.abs(hc - want)) < 1e-12 * np.max(np.abs(want))
        back = pf.r1d_backward(hc)
        assert np.max(np.abs(back - x)) < 1e-12 * np.max(np.abs(x))
    one = pf.r1d_forward(x[0])
    assert one.shape == (n,)


def test_fft_plan_classes_exec(pf):
    """FFTPlanReal1D.exec through the class surface of the reference binding (math_fft.cpp:19-130):
    forward = scale * r2hc, backward = scale / length * hc2r, buffers are views into the plan;
    the reference's round-trip test (src/toast/tests/fft.py:42-94) through the plan store."""
    import toast_amd
    from oracle import fft_oracle as fo

    m = toast_amd.load_native()
    rng = np.random.default_rng(5)
    for length, n, scale in ((65536, 3, 1.0), (1000, 2, 0.25), (15, 1, 3.0)):
        x = rng.standard_normal((n, length))
        fwd = m.FFTPlanReal1D.create(length, n, m.FFTPlanType.fast, m.FFTDirection.forward, scale)
        for i in range(n):
            fwd.tdata(i)[:] = x[i]
        fwd.exec()
        hc = np.array([fwd.fdata(i) for i in range(n)])
        want = scale * fo.r1d_forward(x)
        assert np.max(np.abs(hc - want)) < 1e-12 * np.max(np.abs(want))
        bwd = m.FFTPlanReal1D.create(length, n, m.FFTPlanType.best, m.FFTDirection.backward, 1.0 / scale)
        for i in range(n):
            bwd.fdata(i)[:] = hc[i]
        bwd.exec()
        back = np.array([bwd.tdata(i) for i in range(n)])
        assert np.max(np.abs(back - x)) < 1e-12 * np.max(np.abs(x))
    store = m.FFTPlanReal1DStore.get()
    store.clear()
    y = rng.standard_normal((4, 2048))
    np.testing.assert_array_almost_equal(pf.r1d_backward(pf.r1d_forward(y)), y)
    np.testing.assert_array_almost_equal(pf.r1d_backward(pf.r1d_forward(y[0])), y[0])
    assert store.forward(2048, 4).count() == 4
    store.clear()


def test_large_batch_chunking(pf):
    """More detectors than one work batch: exercises the batch loop and plan cache."""
    import torch
    from oracle import fft_oracle as fo

    rng = np.random.default_rng(9)
    rate, n_samp, n_det = 100.0, 3000, 37
    freq = np.concatenate([[0.0], np.geomspace(1e-3, 50, 40)])
    kern = 1.0 / (1.0 + 0.1 / np.maximum(freq, 1e-3))
    kern[0] = 0
    data = rng.standard_normal((n_det, n_samp))
    want = data.copy()
    fo.convolve(want, rate, kernel_freq=freq, kernels=kern)
    t = torch.from_numpy(data).cuda()
    pf.convolve_dev(t.data_ptr(), np.arange(n_det, dtype=np.int32), n_samp, rate, freq, kern, max_batch=8,
                    stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = t.cpu().numpy()
    assert np.max(np.abs(got - want)) < TOL * np.max(np.abs(want))


@pytest.mark.parametrize("name", ["a", "b", "c", "d"])
def test_rocfft_pipeline_vs_reference_convolve_fixture(pf, name):
    """The rocFFT pipeline against the outputs of the reference's own AlgorithmNumpy.convolve
    (tests/golden/fft_convolve.npz): complex per-detector kernels, deconvolution, a common real
    kernel; 1e-12 of the signal scale."""
    import os

    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fft_convolve.npz"))
    data = z[f"{name}_data"].copy()
    deconv = bool(z[f"{name}_deconvolve"]) if f"{name}_deconvolve" in z.files else False
    pf.convolve(data, float(z[f"{name}_rate"]), kernel_freq=z[f"{name}_kernel_freq"], kernels=z[f"{name}_kernels"],
                deconvolve=deconv)
    want = z[f"{name}_out"]
    assert np.max(np.abs(data - want)) < TOL * np.max(np.abs(want))


def _noise_kernels(freq, n_det, complex_phase=False):
    from oracle import fft_oracle as fo

    kernels = []
    for d in range(n_det):
        net = 1.0 + 0.1 * d
        fknee = 0.05 * (d + 1)
        psd = net**2 * (freq + fknee) / np.maximum(freq + 1e-5, 1e-12)
        k = fo.noise_filter_kernel(psd, net)
        if complex_phase:
            k = k * np.exp(-1j * 0.03 * (d + 1) * freq)
        kernels.append(k)
    return np.array(kernels)


@pytest.mark.parametrize("points", [(16, 16, 16), (8, 8, 8)])
@pytest.mark.parametrize("n_samp", [2049, 3000, 8192, 8193, 12345, 50001, 100000, 262145, 300001, 720000, 1100003,
                                    2200000, 4200001])
def test_fused_three_pass_all_column_lengths(pf, n_samp, points):
    """The fused three-pass pipeline (fft_fused.hip) for every column length N1 = 2 .. 4096 of its
    four-step factorisation (n_fft = 2^13 .. 2^24, N1 up to 4096, odd and even buffer offsets), against the NumPy
    restatement of toast.fft.convolve AND against the rocFFT pipeline: per-detector real kernels with
    row indirection; rows outside the index stay untouched."""
    from oracle import fft_oracle as fo

    assert pf.implementation(n_samp) == "fused-3pass"
    pf.set_points(*points)      # 256-thread radix-16 and 512-thread radix-8 variants of the three kernels
    rng = np.random.default_rng(n_samp)
    rate, n_det, rows = 200.0, 3, 4
    freq = np.concatenate([[0.0], np.geomspace(1e-5, rate / 2, 70)])
    kernels = _noise_kernels(freq, n_det)
    buf = rng.standard_normal((rows, n_samp)).cumsum(axis=1) * 0.01 + rng.standard_normal((rows, n_samp))
    idx = np.array([3, 0, 2], dtype=np.int32)
    want = np.ascontiguousarray(buf[idx])
    fo.convolve(want, rate, kernel_freq=freq, kernels=kernels)
    got = buf.copy()
    pf.convolve_buffer(got, idx, rate, freq, kernels)
    scale = np.max(np.abs(want))
    assert np.max(np.abs(got[idx] - want)) < TOL * scale
    assert np.array_equal(got[1], buf[1])
    pf.select(True)
    try:
        assert pf.implementation(n_samp) == "rocfft"
        lib = buf.copy()
        pf.convolve_buffer(lib, idx, rate, freq, kernels)
    finally:
        pf.select(False)
    assert np.max(np.abs(got - lib)) < 1e-13 * scale
    pf.set_points()             # back to the defaults


@pytest.mark.parametrize("deconvolve", [False, True])
def test_fused_complex_kernels_and_common_kernel(pf, deconvolve):
    """Complex (phase-carrying) kernels, deconvolution and one kernel shared by all detectors
    through the fused pipeline at a length that uses the register radix-16 stages (N1 = 256)."""
    from oracle import fft_oracle as fo

    rng = np.random.default_rng(77)
    rate, n_samp, n_det = 100.0, 270001, 2
    freq = np.linspace(0, rate / 2, 300)
    kern = (1.0 / (1.0 + (freq / 5.0) ** 2) + 0.05) * np.exp(-1j * 0.02 * freq)
    per_det = np.array([kern, kern * np.exp(-1j * 0.01 * freq)])
    for kernels in (per_det, kern):
        data = rng.standard_normal((n_det, n_samp))
        want = data.copy()
        fo.convolve(want, rate, kernel_freq=freq, kernels=kernels, deconvolve=deconvolve)
        got = data.copy()
        pf.convolve(got, rate, kernel_freq=freq, kernels=kernels, deconvolve=deconvolve)
        assert np.max(np.abs(got - want)) < (1e-11 if deconvolve else TOL) * np.max(np.abs(want))


@pytest.mark.parametrize("rows", ["reg", "reg32", "pair", "split"])
@pytest.mark.parametrize("tables", ["lds-complex", "lds-dense", "global-dense", "global-complex"])
def test_fused_kernel_table_paths(pf, rows, tables):
    """Row pass of the fused pipeline: kernel tables in LDS (few knots) or in global memory (many), interval search
    with several knots inside one block of N1 bins (dense knot vectors: the walk from the per-row-element hint), real
    and complex kernels; the three layouts of the row pass (one row per wave in registers / row pair per 64 KB LDS tile /
    one row per 32 KB LDS tile with the mirrored last butterfly)."""
    from oracle import fft_oracle as fo

    rate, n_det = 200.0, 3
    if tables == "lds-complex":
        freq = np.concatenate([[0.0], np.geomspace(1e-4, rate / 2, 90)])
        phase = True
    elif tables == "lds-dense":
        # knot spacing 0.0134 Hz below 2 Hz: 3 - 4 knots per block of N1 bins (rate / 4096 = 0.049 Hz)
        freq = np.concatenate([np.linspace(0.0, 2.0, 150), np.geomspace(2.1, rate / 2, 50)])
        phase = False
    elif tables == "global-dense":
        freq = np.linspace(0.0, rate / 2, 5000)
        phase = False
    else:
        freq = np.linspace(0.0, rate / 2, 400)
        phase = True
    kernels = _noise_kernels(freq, n_det, complex_phase=phase)
    pf.set_rows_mode(rows)
    try:
        for n_samp in (9000, 100001, 720000):
            rng = np.random.default_rng(n_samp)
            data = rng.standard_normal((n_det, n_samp)).cumsum(axis=1) * 0.01 + rng.standard_normal((n_det, n_samp))
            want = data.copy()
            fo.convolve(want, rate, kernel_freq=freq, kernels=kernels)
            got = data.copy()
            pf.convolve(got, rate, kernel_freq=freq, kernels=kernels)
            assert np.max(np.abs(got - want)) < TOL * np.max(np.abs(want)), (n_samp, tables, rows)
    finally:
        pf.set_rows_mode("reg")


@pytest.mark.parametrize("n_samp", [720000, 1100000, 1440000, 2200000, 2880000])
def test_register_tile_passes_against_lds_tile_passes(pf, n_samp):
    """csrc/fft_reg.hip (round 5): the row pass with the tile in registers (16 points per lane: "reg"; 32: "reg32") and the column passes with the tile in registers
    (n_fft 2^21 / 2^22 / 2^23: column tiles of 512 x 8, 1024 x 8, 2048 x 4) against the oracle and against the LDS-tile
    kernels of fft_fused.hip, per-detector kernels, rows through an index."""
    from oracle import fft_oracle as fo

    rng = np.random.default_rng(n_samp + 5)
    rate, n_det, rows = 200.0, 3, 4
    freq = np.concatenate([[0.0], np.geomspace(1e-5, rate / 2, 70)])
    kernels = _noise_kernels(freq, n_det)
    buf = rng.standard_normal((rows, n_samp)).cumsum(axis=1) * 0.01 + rng.standard_normal((rows, n_samp))
    idx = np.array([3, 0, 2], dtype=np.int32)
    want = np.ascontiguousarray(buf[idx])
    fo.convolve(want, rate, kernel_freq=freq, kernels=kernels)
    scale = np.max(np.abs(want))
    out = {}
    try:
        for cols, rowmode in (("reg9", "reg"), ("lds", "pair"), ("reg9", "pair"), ("lds", "reg"), ("reg9", "reg32")):
            pf.set_cols_mode(cols)
            pf.set_rows_mode(rowmode)
            got = buf.copy()
            pf.convolve_buffer(got, idx, rate, freq, kernels)
            assert np.max(np.abs(got[idx] - want)) < TOL * scale, (cols, rowmode)
            assert np.array_equal(got[1], buf[1])
            out[(cols, rowmode)] = got
    finally:
        pf.set_cols_mode("reg")
        pf.set_rows_mode("reg")
    for key, got in out.items():
        assert np.max(np.abs(got - out[("lds", "pair")])) < 1e-13 * scale, key


@pytest.mark.parametrize("tables", ["lds-real", "lds-complex", "global-dense"])
def test_fused_rows_of_1024(pf, tables):
    """The fused kernels with rows of N2 = 1024 (32 KB row-pair tiles, four workgroups per CU in the row pass; 64-byte
    pieces in the column passes) against the oracle and against the N2 = 2048 factorisation, for every N1 that the
    half-size rows allow (n_fft 2^13 .. 2^21) plus a length that falls back to N2 = 2048 (n_fft 2^22)."""
    from oracle import fft_oracle as fo

    rate, n_det = 200.0, 3
    if tables == "global-dense":
        freq, phase = np.linspace(0.0, rate / 2, 5000), False
    elif tables == "lds-complex":
        freq, phase = np.concatenate([[0.0], np.geomspace(1e-4, rate / 2, 90)]), True
    else:
        freq, phase = np.concatenate([[0.0], np.geomspace(1e-4, rate / 2, 70)]), False
    kernels = _noise_kernels(freq, n_det, complex_phase=phase)
    try:
        for n_samp in (2100, 9000, 33000, 100001, 400000, 720000, 1100000):
            rng = np.random.default_rng(n_samp)
            data = rng.standard_normal((n_det, n_samp)).cumsum(axis=1) * 0.01 + rng.standard_normal((n_det, n_samp))
            want = data.copy()
            fo.convolve(want, rate, kernel_freq=freq, kernels=kernels)
            out = {}
            for n2 in (2048, 1024):
                pf.set_rows_n2(n2)
                got = data.copy()
                pf.convolve(got, rate, kernel_freq=freq, kernels=kernels)
                assert np.max(np.abs(got - want)) < TOL * np.max(np.abs(want)), (n_samp, tables, n2)
                out[n2] = got
            assert np.max(np.abs(out[1024] - out[2048])) < 1e-13 * np.max(np.abs(want))
    finally:
        pf.set_rows_n2(2048)


@pytest.mark.parametrize("n_samp", [6000, 50001, 300000])
def test_impulse_extents_on_device(pf, n_samp):
    """toast_hip_fft_impulse_extents (impulses made, convolved and measured in HBM) against the reference procedure
    on the host: convolve temp[:, n_samp // 2] = 100 and walk from the peak (src/toast/fft.py:836-872)."""
    from oracle import fft_oracle as fo

    rate, n_det = 100.0, 5
    freq = np.concatenate([[0.0], np.geomspace(1e-4, rate / 2, 60)])
    for kernels in (_noise_kernels(freq, n_det), _noise_kernels(freq, n_det, complex_phase=True),
                    _noise_kernels(freq, 1)[0]):
        got = pf.impulse_extents(n_det, n_samp, rate, freq, kernels)
        temp = np.zeros((n_det, n_samp))
        temp[:, n_samp // 2] = 100.0
        pf.convolve_buffer(temp, np.arange(n_det, dtype=np.int32), rate, freq, kernels)
        want = np.array([pf.impulse_extent(np.absolute(temp[i])) for i in range(n_det)], dtype=np.int32)
        assert np.array_equal(got, want)
        # and the oracle's own impulse response gives the same widths
        ref = np.zeros((n_det, n_samp))
        ref[:, n_samp // 2] = 100.0
        fo.convolve(ref, rate, kernel_freq=freq, kernels=kernels)
        assert np.array_equal(want, [pf.impulse_extent(np.absolute(ref[i])) for i in range(n_det)])


def test_extend_flags_on_device(pf):
    """toast_hip_fft_extend_flags against the host restatement of the reference's extend_flags (pinned to the
    reference function's outputs in tests/test_fft_oracle.py) followed by the first / last samples of
    toast.fft.convolve: random flags, other flag bits that an assignment must clear, runs touching both ends,
    extents from 0 (Python's f[-0:] flags everything) to longer than the timestream, row indirection."""
    rng = np.random.default_rng(3)
    n_samp, rows = 50001, 9
    flags = np.zeros((rows, n_samp), dtype=np.uint8)
    flags[0] = (rng.random(n_samp) < 0.002) * 1
    flags[1] = (rng.random(n_samp) < 0.01) * 1 + (rng.random(n_samp) < 0.3) * 4
    flags[2, :3] = 1
    flags[2, -1] = 1
    flags[3, -2] = 1
    flags[3, 1000:1010] = 3
    flags[4] = (rng.random(n_samp) < 0.0005) * 1
    flags[5] = 4                      # nothing flagged under the mask
    flags[6, 25000] = 1
    flags[7] = (rng.random(n_samp) < 0.05) * 1
    flags[8, n_samp - 1] = 5
    idx = np.array([7, 0, 2, 3, 1, 5, 6, 4, 8], dtype=np.int32)
    extents = np.array([3, 250, 1, 17, 1000, 40, 60000, 0, 5], dtype=np.int32)
    want = flags.copy()
    for row, ext in zip(idx, extents):
        ext = int(ext)
        pf.extend_flags(want[row], 1, ext)
        want[row][:ext] |= 1
        want[row][-ext:] |= 1
    got = flags.copy()
    pf.extend_flags_buffer(got, idx, 1, extents)
    assert np.array_equal(got, want)
    # without the edges: extend_flags alone
    want2 = flags.copy()
    for row, ext in zip(idx, extents):
        pf.extend_flags(want2[row], 1, int(ext))
    got2 = flags.copy()
    pf.extend_flags_buffer(got2, idx, 1, extents, edges=False)
    assert np.array_equal(got2, want2)
    # a common row OR-ed in first (the shared flags NoiseFilter merges into the detector flags: value 3 has a bit under
    # the mask and one outside), only into the selected rows; host buffer and registered device copy
    from toast_amd import accel

    common = ((rng.random(n_samp) < 0.001) * 3).astype(np.uint8)
    sel = idx[:6]
    want3 = flags.copy()
    for row, ext in zip(sel, extents[:6]):
        ext = int(ext)
        want3[row] |= common
        pf.extend_flags(want3[row], 1, ext)
        want3[row][:ext] |= 1
        want3[row][-ext:] |= 1
    got3 = flags.copy()
    pf.extend_flags_buffer(got3, sel, 1, extents[:6], or_row=common)
    assert np.array_equal(got3, want3)
    got4 = flags.copy()
    accel.accel_data_create(got4, "flags")
    accel.accel_data_update_device(got4, "flags")
    pf.extend_flags_buffer(got4, sel, 1, extents[:6], or_row=common, use_accel=True)
    assert np.array_equal(got4, flags)            # host side untouched
    accel.accel_data_update_host(got4, "flags")
    accel.accel_data_delete(got4, "flags")
    assert np.array_equal(got4, want3)


@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("TOAST_TEST_FUZZ_SEEDS", "12")))))
def test_extend_flags_random_cases(pf, seed):
    """Randomly drawn flag buffers (lengths 1 .. 20 000, densities, masks with other bits set, extents 0 .. beyond the
    length, row subsets, with / without the edges and the common OR-ed row) against the host restatement of the
    reference's extend_flags."""
    rng = np.random.default_rng(13000 + seed)
    n_samp = int(rng.choice([rng.integers(1, 40), rng.integers(40, 20000)]))
    rows = int(rng.integers(1, 9))
    mask = int(rng.choice([1, 2, 3, 129]))
    flags = ((rng.random((rows, n_samp)) < rng.choice([0.0, 0.001, 0.05, 0.5])) * mask
             + (rng.random((rows, n_samp)) < 0.1) * 64).astype(np.uint8)
    # one completely flagged row carrying bits outside the mask: no edge, the reference leaves it alone
    flags[int(rng.integers(0, rows))] = (mask | 64 | 16)
    n_sel = int(rng.integers(1, rows + 1))
    idx = rng.permutation(rows)[:n_sel].astype(np.int32)
    extents = rng.choice([0, 1, 2, 7, 100, n_samp // 2, n_samp, n_samp + 5], size=n_sel).astype(np.int32)
    edges = bool(rng.integers(0, 2))
    common = ((rng.random(n_samp) < 0.01) * int(rng.choice([mask, 3, 64]))).astype(np.uint8) if rng.integers(0, 2) else None
    import oracle.fft_oracle as fo

    want = flags.copy()
    for row, ext in zip(idx, extents):
        ext = int(ext)
        if common is not None:
            want[row] |= common
        fo.extend_flags(want[row], mask, ext)     # the literal restatement of the reference (region loop)
        if edges:
            want[row][:ext] |= mask
            want[row][-ext:] |= mask
    got = flags.copy()
    pf.extend_flags_buffer(got, idx, mask, extents, edges=edges, or_row=common)
    assert np.array_equal(got, want), (n_samp, rows, mask, list(extents), edges, common is not None)
