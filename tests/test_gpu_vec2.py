"""GPU: the two-samples-per-lane kernels (16-byte lane accesses: k_scan_map_v2, k_build_noise_weighted_v2<1|2>,
k_noise_weight_v2) against the one-sample-per-lane kernels they replace and against the oracle.
scan_map and noise_weight are per-sample arithmetic: bit-identical between the two forms.  build_noise_weighted sums
in a different order: 1e-13 of the largest map value between the forms, the chain's 1e-12 against the oracle.  The
cases put chunk starts on odd samples (peeled head), odd chunk lengths (scalar tail), intervals shorter than a pair,
runs of equal pixels cut inside a lane, broken detector pairs and an odd detector count; an odd n_samp must fall back
to the one-sample kernels (rows are then not 16-byte aligned)."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from toast_amd import capi

    assert capi.accel_enabled(), "no HIP device visible"
    capi.accel_assign_device(1, 0, 1.0, False)
    yield capi
    capi.set_tuning("vec2", 1)
    capi.set_tuning("pair", 1)


CASES = {
    "even_whole": dict(n_samp=4096, n_det=4, nside=64),
    "odd_starts_odd_lengths": dict(n_samp=5000, n_split=7, gap=3, n_det=4, nside=128),
    "short_intervals": dict(n_samp=60, n_split=20, gap=1, n_det=2, nside=16),
    "fast_scan_short_runs": dict(n_samp=6000, n_det=6, nside=2048, spin_period_s=3.0, spin_angle_deg=40.0, n_split=3, gap=1),
    "broken_pairs_odd_dets": dict(n_samp=3000, n_det=5, nside=64, fp_roll=1, n_split=2, gap=5),
    "no_flags": dict(n_samp=2048, n_det=2, nside=32, with_det_flags=False, with_shared_flags=False),
    "row_indirection": dict(n_samp=2500, n_det=3, nside=64, extra_rows=2, n_split=3, gap=2),
    "random_pointing": dict(n_samp=3000, n_det=4, nside=256, random_pointing=True),
    "odd_n_samp_falls_back": dict(n_samp=3001, n_det=4, nside=64, n_split=3, gap=2),
    "ground": dict(ground=True, n_samp=36000, rate=100.0, nside=1024, n_det=4),
}


@pytest.mark.parametrize("pair", [1, 0])
@pytest.mark.parametrize("name", list(CASES))
def test_two_samples_per_lane_equals_one(hip, oracle, name, pair):
    c = cases.make_case(**CASES[name])
    hip.set_tuning("pair", pair)
    hip.set_tuning("vec2", 0)
    one = cases.run_chain(hip, c, tail=(False,))
    hip.set_tuning("vec2", 1)
    two = cases.run_chain(hip, c, tail=(False,))
    assert np.array_equal(one["pixels"], two["pixels"])
    assert np.array_equal(one["weights"], two["weights"])
    zs = max(np.max(np.abs(one["zmap"])), 1e-300)
    assert np.max(np.abs(one["zmap"] - two["zmap"])) / zs < 1e-13
    # scan_map + noise_weight are per-sample: same bits given the same map
    t1, t2 = c["tod"].copy(), c["tod"].copy()
    m = np.ascontiguousarray(one["zmap"])
    for vec2, t in ((0, t1), (1, t2)):
        hip.set_tuning("vec2", vec2)
        for zero, sub, mult in [(False, True, False), (False, False, False), (False, False, True), (True, False, False)]:
            hip.ops_scan_map_float64(one["g2l"], c["n_pix_submap"], m, t, c["data_index"], one["pixels"], c["pixel_index"],
                                     one["weights"], c["weight_index"], c["intervals"], 0.75, zero, sub, mult, False)
        hip.noise_weight(t, c["data_index"], c["intervals"], c["det_scale"], False)
    assert np.array_equal(t1, t2)
    want = cases.run_chain(oracle, c)
    assert np.array_equal(two["pixels"], want["pixels"])
    assert np.max(np.abs(two["zmap"] - want["zmap"])) / max(np.max(np.abs(want["zmap"])), 1e-300) < 1e-12
    assert np.max(np.abs(two["tod"] - want["tod"])) / max(np.max(np.abs(want["tod"])), 1e-300) < 1e-11


@pytest.mark.parametrize("map_dtype", [np.float32, np.int64, np.int32])
def test_two_samples_per_lane_map_dtypes(hip, oracle, map_dtype):
    c = cases.make_case(n_samp=3000, nside=128, n_split=3, gap=1)
    hip.set_tuning("vec2", 1)
    got = cases.run_chain(hip, c, map_dtype=map_dtype, scan_scale=0.37, tail=(False,))
    want = cases.run_chain(oracle, c, map_dtype=map_dtype, scan_scale=0.37)
    assert np.max(np.abs(got["tod"] - want["tod"])) / np.max(np.abs(want["tod"])) < 1e-11


def test_non_local_submap_is_skipped(hip):
    """A sample whose pixel lies in a submap that is not local (global2local = -1) is left alone by scan_map in both
    forms (the reference would read in front of the map)."""
    c = cases.make_case(n_samp=2000, n_det=2, nside=64)
    hip.set_tuning("vec2", 1)
    base = cases.run_chain(hip, c, tail=(False,))
    g2l = base["g2l"].copy()
    victim = int(np.flatnonzero(g2l >= 0)[0])
    g2l[victim] = -1
    m = np.ascontiguousarray(base["zmap"])
    outs = []
    for vec2 in (0, 1):
        hip.set_tuning("vec2", vec2)
        t = c["tod"].copy()
        hip.ops_scan_map_float64(g2l, c["n_pix_submap"], m, t, c["data_index"], base["pixels"], c["pixel_index"],
                                 base["weights"], c["weight_index"], c["intervals"], 1.0, False, True, False, False)
        outs.append(t)
    assert np.array_equal(outs[0], outs[1])
    hit_victim = (base["pixels"] // c["n_pix_submap"]) == victim
    assert hit_victim.any() and np.array_equal(outs[1][hit_victim], c["tod"][hit_victim])


@pytest.mark.parametrize("pair", [1, 0])
def test_run_reduction_on_adversarial_pixel_streams(hip, pair):
    """build_noise_weighted fed hand-made pixel streams that stress the two-samples-per-lane run reduction
    (scatter_runs2): runs that start on the second sample of a lane, runs of one and two samples, alternating pixels,
    every other sample flagged or without a pixel, one run over the whole chunk, runs crossing the 128-sample wave and
    the 1024-sample chunk boundaries -- against a plain NumPy scatter-add."""
    rng = np.random.default_rng(4242)
    n_samp, n_det, nps, n_sub = 4096, 4, 48, 40
    n_pix = nps * n_sub
    streams = []
    base = np.arange(n_samp)
    streams.append(np.full(n_samp, 77))                                  # one run
    streams.append((base // 2) % n_pix)                                  # runs of two, aligned with the lanes
    streams.append(((base + 1) // 2) % n_pix)                            # runs of two, straddling lanes
    streams.append(((base + 1) // 3) % n_pix)                            # runs of three
    streams.append(np.where(base % 2 == 0, 5, 9))                        # alternating
    streams.append(rng.integers(0, n_pix, n_samp))                       # runs of one
    streams.append(np.repeat(rng.integers(0, n_pix, n_samp // 127 + 1), 127)[:n_samp])   # runs across wave boundaries
    streams.append(np.repeat(rng.integers(0, n_pix, 5), 1000)[:n_samp])  # runs across chunk boundaries
    lengths = rng.integers(1, 9, n_samp)
    streams.append(np.repeat(rng.integers(0, n_pix, n_samp), lengths)[:n_samp])          # random short runs
    hip.set_tuning("pair", pair)
    hip.set_tuning("vec2", 1)
    g2l = np.arange(n_sub, dtype=np.int64)
    idx = np.arange(n_det, dtype=np.int32)
    for k, stream in enumerate(streams):
        for holes in (None, "no_pixel", "flagged"):
            pixels = np.tile(stream.astype(np.int64), (n_det, 1))
            pixels[1::2] = np.roll(pixels[1::2], 1, axis=1) if k % 2 else pixels[1::2]   # pair partner in or out of step
            dflags = np.zeros((n_det, n_samp), dtype=np.uint8)
            if holes == "no_pixel":
                pixels[:, rng.random(n_samp) < 0.3] = -1
                pixels[0, ::2] = -1
            elif holes == "flagged":
                dflags[:, 1::2] = 1
                dflags[2] = (rng.random(n_samp) < 0.5)
            weights = rng.standard_normal((n_det, n_samp, 3))
            tod = rng.standard_normal((n_det, n_samp))
            scale = rng.random(n_det) + 0.5
            sflags = (rng.random(n_samp) < 0.02).astype(np.uint8)
            ivl = np.zeros(3, dtype=cases.interval_dtype)
            for j, (a, b) in enumerate(((1, 1500), (1501, 1502), (1511, n_samp))):    # odd starts, a one-sample interval
                ivl[j]["first"], ivl[j]["last"] = a, b
            zmap = np.zeros((n_sub, nps, 3))
            hip.build_noise_weighted(g2l, zmap, idx, pixels, idx, weights, idx, tod, idx, dflags, scale, 1, ivl, sflags, 1,
                                     False)
            want = np.zeros((n_pix, 3))
            for d in range(n_det):
                for iv in ivl:
                    sl = slice(int(iv["first"]), int(iv["last"]))
                    good = (pixels[d, sl] >= 0) & (dflags[d, sl] == 0) & (sflags[sl] == 0)
                    np.add.at(want, pixels[d, sl][good], (tod[d, sl][good] * scale[d])[:, None] * weights[d, sl][good])
            err = np.max(np.abs(zmap.reshape(-1, 3) - want)) / max(np.max(np.abs(want)), 1e-300)
            assert err < 1e-12, (k, holes, pair, err)
