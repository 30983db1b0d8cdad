"""Deterministic synthetic inputs for the map-making hot path (no TOAST needed).

These reproduce the *shape* of the reference's simulated satellite data so the kernels
are exercised with realistic pointing locality:

* boresight: the composition of rotations of ``satellite_scanning``
  (reference: src/toast/ops/sim_satellite.py:30-183),
* focalplane: detector quaternions filling a hexagon-ish field of view in orthogonal
  polarisation pairs (reference: src/toast/scripts/toast_benchmark_satellite.py:153-168),
* intervals / flags: the layouts of src/toast/intervals.py:26-45 and the ``uint8`` shared
  and detector flags.

Pure NumPy; the big per-detector arrays are produced on the GPU by the HIP kernels
themselves (pointing_detector -> pixels_healpix / stokes_weights), see bench.py.
"""

import numpy as np

#: same layout as ``toast.intervals.interval_dtype`` (src/toast/intervals.py:26-45)
interval_dtype = np.dtype(
    {
        "names": ["start", "stop", "first", "last"],
        "formats": ["d", "d", "q", "q"],
        "offsets": [0, 8, 16, 24],
    }
)


def quat_rotation(axis, angle):
    """Quaternion(s) [x, y, z, w] rotating by ``angle`` about unit ``axis``.

    Reference semantics: src/libtoast/src/toast_math_qarray.cpp:722-731
    (``[axis * sin(angle/2), cos(angle/2)]``).
    """
    axis = np.asarray(axis, dtype=np.float64)
    angle = np.asarray(angle, dtype=np.float64)
    half = 0.5 * angle
    s = np.sin(half)
    out = np.empty(angle.shape + (4,), dtype=np.float64)
    out[..., 0] = axis[0] * s
    out[..., 1] = axis[1] * s
    out[..., 2] = axis[2] * s
    out[..., 3] = np.cos(half)
    return out


def quat_mult(p, q):
    """Hamilton product p*q, scalar last, broadcasting over leading dims.

    Same term order as ops_pointing_detector.cpp:21-31.
    """
    p = np.asarray(p, dtype=np.float64)
    q = np.asarray(q, dtype=np.float64)
    r = np.empty(np.broadcast_shapes(p.shape, q.shape), dtype=np.float64)
    r[..., 0] = p[..., 0] * q[..., 3] + p[..., 1] * q[..., 2] - p[..., 2] * q[..., 1] + p[..., 3] * q[..., 0]
    r[..., 1] = -p[..., 0] * q[..., 2] + p[..., 1] * q[..., 3] + p[..., 2] * q[..., 0] + p[..., 3] * q[..., 1]
    r[..., 2] = p[..., 0] * q[..., 1] - p[..., 1] * q[..., 0] + p[..., 2] * q[..., 3] + p[..., 3] * q[..., 2]
    r[..., 3] = -p[..., 0] * q[..., 0] - p[..., 1] * q[..., 1] - p[..., 2] * q[..., 2] + p[..., 3] * q[..., 3]
    return r


def quat_normalize(q):
    q = np.asarray(q, dtype=np.float64)
    return q / np.sqrt(np.sum(q * q, axis=-1, keepdims=True))


_X = np.array([1.0, 0.0, 0.0])
_Y = np.array([0.0, 1.0, 0.0])
_Z = np.array([0.0, 0.0, 1.0])


def satellite_boresight(
    n_samp,
    rate,
    spin_period_s=600.0,
    spin_angle_deg=30.0,
    prec_period_s=3000.0,
    prec_angle_deg=65.0,
    sample_offset=0,
):
    """Boresight quaternions ``[n_samp, 4]`` of a spinning, precessing satellite.

    ``q = satrot * Rz(prec) * Rx(prec_angle) * Rz(spin) * Rx(spin_angle) * Rz(pi/2)`` with
    the precession axis fixed along ecliptic X (src/toast/ops/sim_satellite.py:118-176).
    Defaults are the benchmark values (toast_benchmark_satellite.py:164-165; opening
    angles sim_satellite.py:226-233).
    """
    idx = np.arange(n_samp, dtype=np.float64) + float(sample_offset)
    satrot = quat_rotation(_Y, np.pi / 2)
    frac = idx * ((1.0 / prec_period_s) / rate)
    precrot = quat_rotation(_Z, 2.0 * np.pi * (frac - np.floor(frac)))
    precopen = quat_rotation(_X, np.radians(prec_angle_deg))
    frac = idx * ((1.0 / spin_period_s) / rate)
    spinrot = quat_rotation(_Z, 2.0 * np.pi * (frac - np.floor(frac)))
    spinopen = quat_rotation(_X, np.radians(spin_angle_deg))
    fprot = quat_rotation(_Z, 0.5 * np.pi)
    q = quat_mult(satrot, quat_mult(precrot, quat_mult(precopen, quat_mult(spinrot, quat_mult(spinopen, fprot)))))
    return np.ascontiguousarray(quat_normalize(q))


def ground_scan(n_samp, rate, az_min_deg=40.0, az_max_deg=110.0, el_deg=50.0, scan_rate_deg_s=1.0,
                turnaround_s=2.0, site_lat_deg=-22.96, lst0_deg=30.0, with_azimuth=False):
    """Constant-elevation scan (CES) of a ground telescope: boresight quaternions ``[n_samp, 4]``
    in equatorial coordinates, the half-open sample intervals of the constant-velocity sweeps and
    the uint8 shared flags (1 during turnarounds) -- the structure of BASELINE configs[4] inputs
    (reference: src/toast/ops/sim_ground.py builds az/el from a schedule, flags turnarounds and
    stores throw intervals; atmosphere and ground templates are outside the hot path).

    The azimuth is a triangle wave at ``scan_rate_deg_s`` with ``turnaround_s`` of flagged samples
    at every reversal; ``q = Rz(LST) Ry(pi/2 - lat) Rz(-az) Ry(pi/2 - el)``, LST advancing at
    the sidereal rate.
    """
    t = np.arange(n_samp, dtype=np.float64) / rate
    throw = float(az_max_deg - az_min_deg)
    sweep_s = throw / scan_rate_deg_s
    phase = t / sweep_s
    k = np.floor(phase).astype(np.int64)
    frac = phase - k
    az = np.where(k % 2 == 0, az_min_deg + throw * frac, az_max_deg - throw * frac)
    lst = np.radians(lst0_deg) + 2.0 * np.pi * t / 86164.0905
    q_lst = quat_rotation(_Z, lst)
    q_lat = quat_rotation(_Y, np.pi / 2 - np.radians(site_lat_deg))
    q_az = quat_rotation(_Z, -np.radians(az))
    q_el = quat_rotation(_Y, np.pi / 2 - np.radians(el_deg))
    q = quat_mult(q_lst, quat_mult(q_lat, quat_mult(q_az, q_el)))
    bore = np.ascontiguousarray(quat_normalize(q))
    # sweeps: samples farther than turnaround/2 from a reversal
    half = 0.5 * turnaround_s / sweep_s
    turning = (frac < half) | (frac > 1.0 - half)
    flags = turning.astype(np.uint8)
    edges = np.diff(np.concatenate([[1], flags, [1]]).astype(np.int8))
    first = np.flatnonzero(edges == -1)
    last = np.flatnonzero(edges == 1)
    ivl = np.zeros(first.size, dtype=interval_dtype)
    ivl["first"] = first
    ivl["last"] = last
    ivl["start"] = first / rate
    ivl["stop"] = (last - 1) / rate
    if with_azimuth:
        # also: azimuth [rad] and the direction of every sweep (+1 left to right, -1 right to left)
        return bore, ivl, flags, np.radians(az), np.where(k[first] % 2 == 0, 1, -1)
    return bore, ivl, flags


def hex_focalplane(n_det, fov_deg=10.0):
    """Detector quaternions ``[n_det, 4]`` plus polarisation angles ``gamma[n_det]``.

    Pixels sit on a sunflower spiral inside a ``fov_deg`` wide field; each pixel carries two
    detectors with polarisation angles psi and psi+90 deg (pairs "A"/"B" of the reference's
    fake hexagon focalplane, src/toast/instrument_sim.py).  Detector d's rotation is
    ``Rz(phi) Ry(theta) Rz(-phi) Rz(psi)``: offset the line of sight by theta towards
    azimuth phi, then rotate the polarisation axis by psi.
    """
    n_pix = (n_det + 1) // 2
    k = np.arange(n_pix, dtype=np.float64)
    golden = np.pi * (3.0 - np.sqrt(5.0))
    theta = np.radians(0.5 * fov_deg) * np.sqrt((k + 0.5) / n_pix)
    phi = golden * k
    pol_base = (np.pi / 4.0) * (np.arange(n_pix) % 2)
    quats = np.empty((n_det, 4), dtype=np.float64)
    gamma = np.empty(n_det, dtype=np.float64)
    for d in range(n_det):
        ip = d // 2
        psi = pol_base[ip] + (np.pi / 2.0) * (d % 2)
        q = quat_mult(
            quat_rotation(_Z, phi[ip]),
            quat_mult(quat_rotation(_Y, theta[ip]), quat_mult(quat_rotation(_Z, -phi[ip]), quat_rotation(_Z, psi))),
        )
        quats[d] = q
        gamma[d] = psi
    return np.ascontiguousarray(quat_normalize(quats)), gamma


def make_intervals(n_samp, n_split=1, rate=1.0, gap=0):
    """``n_split`` half-open sample intervals covering ``[0, n_samp)`` with ``gap``-sample
    holes between them (the kernels use ``first <= s < last``, e.g. ops_scan_map.cpp:253-255)."""
    edges = np.linspace(0, n_samp, n_split + 1).astype(np.int64)
    ivl = np.zeros(n_split, dtype=interval_dtype)
    for i in range(n_split):
        first = int(edges[i]) + (gap if i > 0 else 0)
        last = int(edges[i + 1])
        first = min(first, last)
        ivl[i]["first"] = first
        ivl[i]["last"] = last
        ivl[i]["start"] = first / rate
        ivl[i]["stop"] = last / rate
    return ivl


def shared_flags_block(n_samp, frac=0.01, value=1, where=0.37):
    """uint8 shared flags, zero except one contiguous block of ``frac * n_samp`` samples."""
    f = np.zeros(n_samp, dtype=np.uint8)
    n = int(frac * n_samp)
    start = int(where * n_samp)
    f[start : start + n] = value
    return f


def det_flags_random(n_det, n_samp, frac=0.005, value=1, seed=1234):
    rng = np.random.default_rng(seed)
    f = np.zeros((n_det, n_samp), dtype=np.uint8)
    f[rng.random((n_det, n_samp)) < frac] = value
    return f


def white_noise_tod(n_det, n_samp, rate, net=50.0e-6, seed=20261001):
    """Gaussian white TOD with sigma = NET * sqrt(rate), one RNG stream per detector."""
    out = np.empty((n_det, n_samp), dtype=np.float64)
    sigma = net * np.sqrt(rate)
    for d in range(n_det):
        out[d] = np.random.default_rng(seed + d).standard_normal(n_samp) * sigma
    return out


def global_to_local(hit_submaps):
    """``global2local`` (int64, -1 where not hit) and the sorted list of local submaps;
    reference: PixelDistribution._glob2loc, src/toast/pixels.py:216-241."""
    hit = np.flatnonzero(np.asarray(hit_submaps) != 0).astype(np.int64)
    g2l = np.full(len(hit_submaps), -1, dtype=np.int64)
    g2l[hit] = np.arange(hit.size, dtype=np.int64)
    return g2l, hit
