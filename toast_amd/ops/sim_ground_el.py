"""Elevation motions of a ground telescope around a constant-elevation scan: el-nods before / after the scan, a
periodic elevation modulation during it, elevation steps after every scan pair (inputs of
BASELINE configs[4]; host-side NumPy in the reference too).

Reference: src/toast/ops/sim_ground_utils.py:16-122 (``scan_time``, ``scan_profile``, ``scan_between``), :124-203
(``simulate_elnod``), :205-358 (``oscillate_el``), :360-432 (``step_el``).  Every motion is a move from rest to rest
under a rate and an acceleration limit -- accelerate, coast, decelerate -- tabulated at high resolution and
interpolated linearly to the sample times.  The functions here are written around that one idea (``_Move``); what is
kept from the reference exactly is the arithmetic -- operand order, the number of table points, the seeds of the random
phases -- because the outputs are pinned BIT FOR BIT to those of the reference's own functions
(tests/golden/make_golden_sim_ground_el.py -> tests/golden/sim_ground_el.npz, tests/test_sim_ground.py), including two
slips of the reference that change values: ``simulate_elnod`` takes the MINIMUM for the upper ends of its ranges, and
``oscillate_el`` starts its constant-rate wave in the middle of a sweep "without a proper accelerating phase"."""

import numpy as np


class _Move:
    """From rest at one coordinate to rest at another: acceleration ``accel`` up to ``rate``, coasting, the same
    deceleration.  A move too short to reach ``rate`` turns around at half way."""

    def __init__(self, distance, rate, accel):
        self.t_accel = rate / accel
        d_accel = 0.5 * accel * self.t_accel**2
        if 2 * d_accel > distance:
            d_accel = distance / 2
            self.t_accel = np.sqrt(2 * d_accel / accel)
            self.t_coast = 0
            self.short = True
        else:
            self.t_coast = (distance - 2 * d_accel) / rate
            self.short = False

    @property
    def duration(self):
        return 2 * self.t_accel + self.t_coast


def scan_time(coord_in, coord_out, scanrate, scan_accel):
    """Seconds a move between the two coordinates takes (sim_ground_utils.py:16-35)."""
    return _Move(np.abs(coord_in - coord_out), scanrate, scan_accel).duration


def scan_profile(coord_in, coord_out, scanrate, scan_accel, times, nstep=10000):
    """The coordinate at ``times`` of a move that starts at times[0]; once there it waits (sim_ground_utils.py:38-88)."""
    if np.abs(coord_in - coord_out) < 1e-6:
        return np.zeros(len(times)) + coord_out
    move = _Move(np.abs(coord_in - coord_out), scanrate, scan_accel)
    if move.short:
        scanrate = move.t_accel * scan_accel        # the rate reached at half way
    if coord_in > coord_out:
        scanrate, scan_accel = -scanrate, -scan_accel
    # legs of the knot table: (duration, points, coordinate as a function of the time since the leg began and of the
    # coordinate where it began)
    legs = [(move.t_accel, nstep, lambda tau, x0: x0 + 0.5 * scan_accel * tau**2)]
    if move.t_coast > 0:
        legs.append((move.t_coast, 3, lambda tau, x0: x0 + scanrate * tau))
    legs.append((move.t_accel, nstep, lambda tau, x0: x0 + scanrate * tau - 0.5 * scan_accel * tau**2))
    t_knot, x_knot = [], []
    t_now, x_now = times[0], coord_in
    for duration, points, shape in legs:
        t = np.linspace(t_now, t_now + duration, points)
        x = shape(t - t[0], x_now)
        t_knot.append(t)
        x_knot.append(x)
        t_now, x_now = t[-1], x[-1]
    if t_now < times[-1]:
        t_knot.append(np.linspace(t_now, times[-1], 3))
        x_knot.append(np.zeros(3) + coord_out)
    return np.interp(times, np.hstack(t_knot), np.hstack(x_knot))


def scan_between(time_start, az1, el1, az2, el2, az_rate, az_accel, el_rate, el_accel, nstep=10000):
    """Both axes move at once, each from rest to rest; the faster one waits (sim_ground_utils.py:91-122).
    Returns (times, az, el) on ``nstep`` points."""
    time_tot = max(scan_time(az1, az2, az_rate, az_accel), scan_time(el1, el2, el_rate, el_accel))
    times = np.linspace(0, time_tot, nstep)
    az = scan_profile(az1, az2, az_rate, az_accel, times, nstep=nstep)
    el = scan_profile(el1, el2, el_rate, el_accel, times, nstep=nstep)
    return times + time_start, az, el


def simulate_elnod(t_start, rate, az_start, el_start, az_rate, az_accel, el_rate, el_accel, elnod_el, elnod_az,
                   scan_min_az, scan_max_az, scan_min_el, scan_max_el):
    """An el-nod: the mount visits the (az, el) stations one after the other, coming to rest at each
    (sim_ground_utils.py:124-203).  Returns (times, az, el) sampled at ``rate`` from t_start and the updated
    (min_az, max_az, min_el, max_el) -- the upper ends with the reference's ``min`` (module docstring)."""
    t_parts, az_parts, el_parts = [], [], []
    t_now, az_now, el_now = t_start, az_start, el_start
    for az_to, el_to in zip(elnod_az, elnod_el):
        if np.abs(az_now - az_to) > 1e-3 or np.abs(el_now - el_to) > 1e-3:
            t, a, e = scan_between(t_now, az_now, el_now, az_to, el_to, az_rate, az_accel, el_rate, el_accel)
            t_parts.append(t)
            az_parts.append(a)
            el_parts.append(e)
            t_now = t[-1]
        az_now, el_now = az_to, el_to
    t, az, el = np.hstack(t_parts), np.hstack(az_parts), np.hstack(el_parts)
    scan_min_az = min(scan_min_az, np.min(az))
    scan_max_az = min(scan_max_az, np.max(az))
    scan_min_el = min(scan_min_el, np.min(el))
    scan_max_el = min(scan_max_el, np.max(el))
    n_sample = int((t[-1] - t[0]) * rate)
    t_sample = np.arange(n_sample) / rate + t_start
    return (t_sample, np.interp(t_sample, t, az), np.interp(t_sample, t, el), scan_min_az, scan_max_az, scan_min_el,
            scan_max_el)


def _triangle_wave(el_rate, el_accel, amplitude, period, n=1000):
    """One period of an elevation wave of peak-to-peak 2 * amplitude that moves at a constant rate between rounded
    reversals: knot table (t, el) starting at the lower reversal, and (t_accel, t_scan) of its pieces.  The rate
    follows from amplitude, period and acceleration (a quadratic in the acceleration time)."""
    a, b, c = el_accel, -0.5 * el_accel * period, 2 * amplitude
    if b**2 - 4 * a * c < 0:
        raise RuntimeError("Cannot perform {:.2f} deg elevation oscillation in {:.2f} s with {:.2f} deg/s^2 acceleration"
                           .format(np.degrees(amplitude * 2), period, np.degrees(el_accel)))
    root1 = (-b - np.sqrt(b**2 - 4 * a * c)) / (2 * a)
    root2 = (-b + np.sqrt(b**2 - 4 * a * c)) / (2 * a)
    t_accel = root1 if root1 > 0 else root2
    t_scan = 0.5 * period - 2 * t_accel
    scanrate = t_accel * el_accel
    if scanrate > el_rate:
        raise RuntimeError("Elevation oscillation requires {:.2f} > {:.2f} deg/s scan rate"
                           .format(np.degrees(scanrate), np.degrees(el_rate)))
    # pieces: (duration, points, elevation as a function of the time since the piece began and of where it began)
    pieces = [(t_accel, n, lambda t, e0: 0.5 * el_accel * t**2),
              (t_scan, 2, lambda t, e0: e0 + t * scanrate),
              (2 * t_accel, n, lambda t, e0: e0 + scanrate * t - 0.5 * el_accel * t**2),
              (t_scan, 2, lambda t, e0: e0 - t * scanrate),
              (t_accel, n, lambda t, e0: e0 - scanrate * t + 0.5 * el_accel * t**2)]
    t_knot, el_knot = [], []
    t_last, el_last = 0.0, 0.0
    for k, (duration, points, shape) in enumerate(pieces):
        t = np.linspace(0, duration, points)
        t_knot.append(t if k == 0 else t_last + t)
        el_knot.append(shape(t, el_last))
        t_last, el_last = t_knot[-1][-1], el_knot[-1][-1]
    return np.hstack(t_knot), np.hstack(el_knot), t_accel, t_scan


def oscillate_el(times, el, el_rate, el_accel, scan_min_el, scan_max_el, el_mod_amplitude, el_mod_rate,
                 ival_scan_leftright, ival_scan_rightleft, el_mod_sine=False, el_mod_sine_phase=None):
    """Modulate ``el`` IN PLACE with a wave of frequency ``el_mod_rate`` (sim_ground_utils.py:205-358): a constant-rate
    wave with rounded reversals over the whole scan, its phase drawn from the scan's start time; or
    (``el_mod_sine``) a sine per sweep, starting at the sweep's start, optionally advanced by a fixed phase per sweep
    (>= 0) or a random one (< 0: seeded by the phase value and the sweep's number).  Returns the new (min, max) of el."""
    tt = times - times[0]
    tt += np.random.RandomState(int(times[0] % 2**32)).rand() / el_mod_rate       # (the reference seeds the global generator)
    if el_mod_sine:
        angular_rate = 2 * np.pi * el_mod_rate
        sweeps = [None] * (len(ival_scan_leftright) + len(ival_scan_rightleft))    # left-right and right-left alternate
        sweeps[::2] = ival_scan_leftright
        sweeps[1::2] = ival_scan_rightleft
        for i, (t0, t1) in enumerate(sweeps):
            first = (np.abs(times - t0)).argmin()
            last = (np.abs(times - t1)).argmin()
            sweep_tt = times[first : last + 1] - t0
            if el_mod_sine_phase is not None and el_mod_sine_phase >= 0:
                sweep_tt += i * el_mod_sine_phase / el_mod_rate
            elif el_mod_sine_phase is not None and el_mod_sine_phase < 0:
                sweep_tt += np.random.RandomState(int(-1000 * el_mod_sine_phase + i)).rand() / el_mod_rate
            el[first : last + 1] += el_mod_amplitude * np.sin(sweep_tt * angular_rate)
        # what the mount must be able to do: the derivatives of a harmonic motion
        el_rate_max = angular_rate * el_mod_amplitude
        el_accel_max = angular_rate**2 * el_mod_amplitude
        if el_rate_max > el_rate:
            raise RuntimeError("Elevation oscillation requires {:.2f} deg/s but mount only allows {:.2f} deg/s"
                               .format(np.degrees(el_rate_max), np.degrees(el_rate)))
        if el_accel_max > el_accel:
            raise RuntimeError("Elevation oscillation requires {:.2f} deg/s^2 but mount only allows {:.2f} deg/s^2"
                               .format(np.degrees(el_accel_max), np.degrees(el_accel)))
    else:
        t_mod = 1 / el_mod_rate
        t_knot, el_knot, t_accel, t_scan = _triangle_wave(el_rate, el_accel, el_mod_amplitude, t_mod)
        tt += t_accel + 0.5 * t_scan          # the scan starts in the middle of the wave's first constant-rate piece
        el += np.interp(tt % t_mod, t_knot, el_knot) - el_mod_amplitude
    return min(scan_min_el, np.min(el)), max(scan_max_el, np.max(el))


def step_el(times, az, el, el_rate, el_accel, scan_min_el, scan_max_el, el_mod_step, n=1000):
    """Step ``el`` IN PLACE by ``el_mod_step`` after every scan pair -- at every second reversal of the azimuth --,
    each step a move from rest to rest centred on the reversal (sim_ground_utils.py:360-432).  Returns the new
    (min, max) of el."""
    direction = np.sign(el_mod_step)
    height = np.abs(el_mod_step)
    # rest -> full rate -> rest: a ramp of `ramp_s` seconds covers `ramp_rise`; a step lower than two ramps never reaches
    # the full rate and has no coasting piece (the operation order is the reference's: the knots are compared bit for bit)
    ramp_s = el_rate / el_accel
    ramp_rise = 0.5 * el_accel * ramp_s**2
    if height > 2 * ramp_rise:
        coast_s = (height - 2 * ramp_rise) / el_rate
    else:
        ramp_rise = np.abs(el_mod_step) / 2
        ramp_s = np.sqrt(2 * ramp_rise / el_accel)
        coast_s = 0
    rate_reached = el_accel * ramp_s
    pieces = [(ramp_s, lambda t, e0: 0.5 * el_accel * t**2)]
    if coast_s > 0:
        pieces.append((coast_s, lambda t, e0: e0 + t * el_rate))
    pieces.append((ramp_s, lambda t, e0: e0 + rate_reached * t - 0.5 * el_accel * t**2))
    t_knot, el_knot = [], []
    t_last, el_last = 0.0, 0.0
    for k, (duration, shape) in enumerate(pieces):
        t = np.linspace(0, duration, n)
        t_knot.append(t if k == 0 else t_last + t)
        el_knot.append(shape(t, el_last))
        t_last, el_last = t_knot[-1][-1], el_knot[-1][-1]
    t_knot = np.hstack(t_knot)
    t_knot -= t_knot[t_knot.size // 2]
    el_knot = direction * np.hstack(el_knot)
    daz = np.diff(az)
    reversals = np.where(daz[1:] * daz[:-1] < 0)[0] + 1
    for istep in reversals[1::2]:
        el += np.interp(times - times[istep], t_knot, el_knot)
    return min(scan_min_el, np.min(el)), max(scan_max_el, np.max(el))
