"""Per-operation checks of the device arithmetic the bit-exact pixel path relies on:
f64 sqrt and division must be correctly rounded on gfx950 (== x86-64), and the
double-double atan2 must agree with glibc's except for rare 1-ulp differences."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_dev():
    import torch

    assert torch.cuda.is_available()
    return torch


def _run(torch, op, a, b):
    from toast_amd import capi

    ta = torch.from_numpy(a).cuda()
    tb = torch.from_numpy(b).cuda()
    out = torch.empty_like(ta)
    capi.dev.test_math(op, a.size, ta.data_ptr(), tb.data_ptr(), out.data_ptr(),
                       torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return out.cpu().numpy()


def test_sqrt_div_correctly_rounded(torch_dev, oracle):
    rng = np.random.default_rng(11)
    n = 4_000_000
    a = np.abs(rng.standard_normal(n)) * 10.0 ** rng.integers(-8, 8, n)
    b = rng.standard_normal(n) + 1e-3
    assert np.array_equal(_run(torch_dev, 1, a, b), oracle.libm_sqrt(a))
    # the pixel path's sqrt argument: 3 (1 - |z|)
    z = 3.0 * (1.0 - rng.random(n))
    assert np.array_equal(_run(torch_dev, 1, z, b), oracle.libm_sqrt(z))
    assert np.array_equal(_run(torch_dev, 2, a, b), oracle.ieee_div(a, b))
    phi = (rng.random(n) - 0.5) * 2 * np.pi
    assert np.array_equal(_run(torch_dev, 2, phi, np.full(n, 2 * np.pi)), oracle.ieee_div(phi, np.full(n, 2 * np.pi)))


def test_atan2_matches_glibc(torch_dev, oracle):
    rng = np.random.default_rng(12)
    n = 4_000_000
    y = rng.standard_normal(n)
    x = rng.standard_normal(n)
    got = _run(torch_dev, 0, y, x)
    want = oracle.libm_atan2(y, x)
    d = np.abs(got.view(np.int64) - want.view(np.int64))
    assert d.max() <= 1
    # glibc 2.35 itself misrounds ~1e-3 of random arguments by one ulp; ours is correctly
    # rounded except ~1e-5; the disagreement rate must stay at that level.
    assert np.count_nonzero(d) < 4e-3 * n
    # special values
    sy = np.array([0.0, -0.0, 0.0, -0.0, 1.0, -1.0, 1.0, -1.0, 0.0, -0.0, 1.0, 1.0, -1.0, 2.5])
    sx = np.array([1.0, 1.0, -1.0, -1.0, 0.0, 0.0, -0.0, -0.0, 0.0, -0.0, 1.0, -1.0, -1.0, 2.5])
    got = _run(torch_dev, 0, sy, sx)
    want = oracle.libm_atan2(sy, sx)
    assert np.array_equal(got.view(np.int64), want.view(np.int64))
