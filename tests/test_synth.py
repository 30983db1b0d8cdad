import numpy as np

from toast_amd import synth


def test_boresight_unit_and_smooth():
    q = synth.satellite_boresight(5000, 100.0)
    assert q.shape == (5000, 4)
    np.testing.assert_allclose(np.sum(q * q, axis=1), 1.0, rtol=0, atol=1e-14)
    # consecutive samples are close rotations (scan continuity)
    dots = np.abs(np.sum(q[1:] * q[:-1], axis=1))
    assert dots.min() > 0.999


def test_intervals_cover_and_gap():
    iv = synth.make_intervals(1000, 4, rate=10.0, gap=3)
    assert iv.dtype.itemsize == 32
    assert iv["first"][0] == 0 and iv["last"][-1] == 1000
    assert all(iv["first"][1:] - iv["last"][:-1] == 3)


def test_focalplane_pairs_orthogonal():
    fp, gamma = synth.hex_focalplane(8)
    np.testing.assert_allclose(np.sum(fp * fp, axis=1), 1.0, atol=1e-14)
    np.testing.assert_allclose(gamma[1::2] - gamma[0::2], np.pi / 2)


def test_global_to_local():
    hs = np.array([0, 1, 0, 1, 1], np.uint8)
    g2l, hit = synth.global_to_local(hs)
    assert list(g2l) == [-1, 0, -1, 1, 2] and list(hit) == [1, 3, 4]
