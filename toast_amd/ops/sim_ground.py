"""SimGround: schedule-driven constant-elevation scans of a ground telescope -- the generator of
BASELINE configs[4]'s pointing inputs.

Reference: src/toast/ops/sim_ground.py:61-1270 (operator), src/toast/ops/sim_ground_utils.py:458-753
(``simulate_ces_scan``).  This is input generation and runs on the host in the reference too
(NumPy interpolation of one high-resolution scan / turnaround cycle); nothing here is on the
A / A^T path.  Reproduced: the azimuth profile with finite-acceleration turnarounds (``CesCycle``: the
scan cycle written down from its geometry; outputs pinned bit for bit to the reference function's in
tests/golden/sim_ground.npz), optional cosecant modulation and randomised phase, the seven
interval lists (scan / turn / throw, left-right and right-left) plus ``scanning`` / ``turnaround`` /
``throw``, turnaround bits in the shared flags (``FlagIntervals`` with ``turnaround_mask``), the
horizontal boresight ``qa.from_lonlat_angles(-az, el, 0) * Rz(boresight_angle)``, HWP angle,
detector data / flag buffers.

El-nods before / after the scan, the periodic elevation modulation and the elevation steps after every scan pair
(``elnod_start`` / ``elnod_end`` / ``elnods``, ``el_mod_rate`` / ``el_mod_amplitude`` / ``el_mod_sine``, ``el_mod_step``;
sim_ground.py:905-1130) are in ``sim_ground_el.py``, pinned bit for bit to the reference's functions as well
(tests/golden/sim_ground_el.npz).

Not reproduced (their dependencies -- astropy, ephem, qpoint -- are absent, and none of it changes
the structure of the map-making inputs): Sun intervals, weather, ``track_azimuth`` (its drift rate comes from
ephem, sim_ground_utils.py:522-548), site position / velocity, and the astrometric part of horizontal -> equatorial.
The equatorial boresight is the rigid rotation
``Rz(ERA(t) + lon) Ry(pi/2 - lat) Rz(pi) q_azel`` with the Earth rotation angle of the UTC stamp
(no precession, nutation, aberration or refraction: arcminute-level differences from the
reference's ephem / astropy conversion)."""

from datetime import timezone

import numpy as np

from .. import synth
from ..data import Data, IntervalList, Observation, Telescope, defaults
from ..schedule import GroundSchedule
from ..traits import Bool, Float, Instance, Int, List, Unicode
from .sim_ground_el import oscillate_el, simulate_elnod, step_el
from .operator import Operator


class CesCycle:
    """One period of a constant-elevation scan as a closed piece of geometry.

    The mount sweeps the throw [az_lo, az_hi] at a constant azimuth rate (or, with
    ``cosecant`` set, at a constant rate of cos(az), which keeps the integration depth per
    declination stripe flat), reverses with the constant acceleration ``accel`` over
    2 * rate / accel seconds -- a parabola leaving and re-entering the throw edge at the sweep
    rate -- sweeps back and reverses again: period = 2 * (sweep + turn).

    The cycle is sampled as a knot table (linear interpolation between knots happens in
    ``simulate_ces_scan``): a constant-rate sweep is exact with its two end knots, curved pieces get
    ``nstep`` knots.  Bit-identity with the reference's outputs (tests/golden/sim_ground.npz) fixes
    the rounding order of the two curve expressions below and nothing else."""

    def __init__(self, el, az_lo, az_hi, az_rate, on_sky, accel, cosecant, nstep):
        self.shift = 0.0
        if cosecant:
            # constant d(cos az)/dt only makes sense inside one half of the circle: fold the throw into
            # [0, pi) and put the western half back afterwards
            if az_lo > np.pi:
                self.shift = np.pi
            az_lo, az_hi = az_lo % np.pi, az_hi % np.pi
            if az_lo > az_hi:
                raise RuntimeError("Cannot scan across zero meridian with cosecant-modulated scan")
        elif az_hi < az_lo:
            az_hi += 2 * np.pi          # throw across north
        self.az_lo, self.az_hi, self.cosecant, self.nstep, self.accel = az_lo, az_hi, cosecant, nstep, accel
        self.rate = az_rate / np.cos(el) if on_sky else az_rate
        if cosecant:
            self.sweep = (np.cos(az_lo) - np.cos(az_hi)) / self.rate
            self.edge_rate = self.rate / np.abs(np.sin(az_lo))   # both turnarounds use the rate at the low edge
        else:
            self.sweep = (az_hi - az_lo) / self.rate
            self.edge_rate = self.rate
        self.turn = 2 * self.edge_rate / accel
        self.period = 2 * self.sweep + 2 * self.turn

    def segments(self, t0):
        """Start / stop times of the four pieces of the cycle that starts at t0, shape (4, 2), in the
        order sweep up, turn at the high edge, sweep down, turn at the low edge."""
        edges = np.cumsum([t0, self.sweep, self.turn, self.sweep, self.turn])   # each piece starts where the last ended
        return np.column_stack((edges[:-1], edges[1:]))

    def _sweep_knots(self, span, az_from, az_to, direction):
        if not self.cosecant:
            return np.array(span), np.array([az_from, az_to])
        t = np.linspace(span[0], span[1], self.nstep, endpoint=True)
        return t, np.arccos((np.cos(az_from) + direction * (self.rate * span[0])) - direction * (self.rate * t))

    def _turn_knots(self, span, az_edge, direction):
        # interior knots only: the end points are the neighbouring sweeps' knots
        t = np.linspace(span[0], span[1], self.nstep, endpoint=True)[1:-1]
        tau = t - span[0]
        return t, (az_edge + direction * (tau * self.edge_rate)) - direction * (0.5 * self.accel * tau**2)

    def knots(self, t0):
        """(t, az) knot table of one period; the closing knot at t0 + period is left to the next period."""
        up, top, down, bottom = self.segments(t0)
        pieces = (self._sweep_knots(up, self.az_lo, self.az_hi, +1.0), self._turn_knots(top, self.az_hi, +1.0),
                  self._sweep_knots(down, self.az_hi, self.az_lo, -1.0), self._turn_knots(bottom, self.az_lo, -1.0))
        t = np.concatenate([p[0] for p in pieces])
        az = np.concatenate([p[1] for p in pieces])
        if az.min() < -2 * np.pi:
            az += 2 * np.pi
        if az.max() > 2 * np.pi:
            az -= 2 * np.pi
        return t, az + self.shift if self.shift else az


def _clip_tail(spans, t_last):
    """The reference trims only the very last span of each list against the last time stamp (and, through
    a slip in its head test, never the first): later consumers intersect with the sample range anyway
    (``timespans_to_samples``).  Same lists here, so that they compare equal span for span."""
    if spans[-1, 0] > t_last:
        spans = spans[:-1]
    elif spans[-1, 1] > t_last:
        spans[-1, 1] = t_last
    return [(float(a), float(b)) for a, b in spans]


def simulate_ces_scan(t_start, t_stop, rate, el, az_min, az_max, az_start, az_rate, fix_rate_on_sky, az_accel,
                      scan_min_az, scan_max_az, cosecant_modulation=False, nstep=10000, randomize_phase=False):
    """Azimuth / elevation of one constant-elevation scan sampled at ``rate`` Hz from t_start to t_stop
    (radians, rad/s, seconds), plus the time spans of its sweeps, turnarounds and throws (sweep extended by
    half a turnaround on both sides).  Same outputs as src/toast/ops/sim_ground_utils.py:458-753 without
    ``track_azimuth``: (times, az, el, min_az, max_az, scan_leftright, turn_leftright, scan_rightleft,
    turn_rightleft, throw_leftright, throw_rightleft).  A zero-width throw is a stare: constant
    azimuth, no spans (:435-455)."""
    n_samp = int((t_stop - t_start) * rate)
    times = t_start + np.arange(n_samp) / rate
    el_samp = np.full(n_samp, 0.0) + el
    if np.abs(az_min - az_max) < 1e-10:
        return times, np.full(n_samp, 0.0) + az_min, el_samp, az_min, az_min, [], [], [], [], [], []
    cyc = CesCycle(el, az_min, az_max, az_rate, fix_rate_on_sky, az_accel, cosecant_modulation, nstep)
    n_cycle = int((t_stop - t_start) / cyc.period) + 2      # covers the phase offset and the tail
    t1, az1 = cyc.knots(t_start)
    lag = np.arange(n_cycle)[:, None] * cyc.period
    t_knot = (t1[None, :] + lag).ravel()
    az_knot = np.tile(az1, n_cycle)
    # where in the cycle the scan begins: its start, or a point drawn from the start time's own seed
    phase = cyc.period * np.random.RandomState(int(t_start % 2**32)).random_sample() if randomize_phase else 0.0
    az_samp = np.interp(times + phase, t_knot, az_knot)
    # spans of cycle k = spans of the first cycle moved by k periods (summed up one by one), minus the phase
    moves = np.cumsum(np.concatenate(([-phase], np.full(n_cycle - 1, cyc.period))))[:, None]
    up, top, down, bottom = cyc.segments(t_start)
    half_top, half_bottom = 0.5 * (top[1] - top[0]), 0.5 * (bottom[1] - bottom[0])
    throw_up = ((up + moves) + np.array([-half_bottom, half_top]))
    throw_down = ((down + moves) + np.array([-half_top, half_bottom]))
    lists = [_clip_tail(x, times[-1]) for x in (up + moves, top + moves, down + moves, bottom + moves, throw_up,
                                                throw_down)]
    return (times, az_samp, el_samp, min(scan_min_az, az_knot.min()), max(scan_max_az, az_knot.max()), *lists)


def timespans_to_samples(times, timespans):
    """Half-open sample spans of time spans (src/toast/intervals.py:150-175): samples with
    start <= t < stop, the last sample included when the span ends on the last time stamp."""
    if len(timespans) == 0:
        return []
    spans = np.vstack(timespans).astype(np.float64)
    for i in range(len(spans) - 1):
        if np.isclose(spans[i][1], spans[i + 1][0], rtol=1e-12):
            spans[i][1] = spans[i + 1][0]
        if spans[i][1] > spans[i + 1][0]:
            raise RuntimeError("Timespans must be sorted and disjoint")
    start, stop = spans.T
    good = np.logical_and(start < times[-1], stop > times[0])
    first = np.searchsorted(times, start[good], side="left")
    last = np.searchsorted(times, stop[good], side="left")
    last[last == len(times) - 1] = len(times)
    return [(int(a), int(b)) for a, b in zip(first, last)]


def union_spans(n_samp, *span_lists):
    """Sample spans of the union of several interval lists (IntervalList.__or__)."""
    mask = np.zeros(n_samp + 2, dtype=np.int8)
    for spans in span_lists:
        for a, b in spans:
            mask[a + 1:b + 1] = 1
    edges = np.diff(mask)
    return list(zip(np.flatnonzero(edges == 1).tolist(), np.flatnonzero(edges == -1).tolist()))


def earth_rotation_angle(unix_seconds):
    """ERA of a UTC time stamp (IERS 2003; UT1 - UTC neglected): 2 pi (0.7790572732640 +
    1.00273781191135448 (JD - 2451545.0)) modulo 2 pi."""
    days = np.asarray(unix_seconds, dtype=np.float64) / 86400.0 + (2440587.5 - 2451545.0)
    frac = (0.7790572732640 + 0.00273781191135448 * days + (days % 1.0)) % 1.0
    return 2.0 * np.pi * frac


def azel_to_radec(times, bore_azel, site_lat_rad, site_lon_rad):
    """Horizontal -> equatorial boresight quaternions by the rigid rotation of the local frame
    (X north, Y west, Z zenith) at local sidereal angle ERA + longitude."""
    lst = earth_rotation_angle(times) + site_lon_rad
    q_frame = synth.quat_mult(
        synth.quat_rotation(np.array([0.0, 0.0, 1.0]), lst),
        synth.quat_mult(synth.quat_rotation(np.array([0.0, 1.0, 0.0]), np.pi / 2 - site_lat_rad),
                        synth.quat_rotation(np.array([0.0, 0.0, 1.0]), np.pi)))
    return np.ascontiguousarray(synth.quat_normalize(synth.quat_mult(q_frame, bore_azel)))


def from_lonlat_angles(lon, lat, psi):
    """qa.from_lonlat_angles: Rz(lon) Ry(pi/2 - lat) Rz(psi) (src/toast/qarray.py:454-484,
    src/toast/_libtoast/math_qarray.cpp:608-670)."""
    z, y = np.array([0.0, 0.0, 1.0]), np.array([0.0, 1.0, 0.0])
    q = synth.quat_mult(synth.quat_rotation(z, lon),
                        synth.quat_mult(synth.quat_rotation(y, 0.5 * np.pi - lat), synth.quat_rotation(z, psi)))
    return synth.quat_normalize(q)


class SimGround(Operator):
    """Simulate a generic ground-based telescope scanning (reference sim_ground.py:61)."""

    API = Int(0, help="Internal interface version for this operator")
    telescope = Instance(klass=Telescope, allow_none=True, help="This must be an instance of a Telescope")
    schedule = Instance(klass=GroundSchedule, allow_none=True, help="Instance of a GroundSchedule")
    schedule_file = Unicode(None, allow_none=True, help="Ground-based observing schedule file")
    sort_schedule_file = Bool(False, help="If True, sort schedule loaded from a file by name")
    randomize_phase = Bool(False, help="If True, the Constant Elevation Scan will begin at a randomized phase.")
    track_azimuth = Bool(False, help="If True, the azimuth throw is continually adjusted to center the field.")
    scan_rate_az = Float(1.0, help="The sky or mount azimuth scanning rate [deg / s].  See `fix_rate_on_sky`")
    fix_rate_on_sky = Bool(True, help="If True, `scan_rate_az` is given in sky coordinates and azimuthal rate on "
                                      "mount will be adjusted to meet it.  If False, `scan_rate_az` is used as the "
                                      "mount azimuthal rate.")
    scan_accel_az = Float(1.0, help="Mount scanning rate acceleration for turnarounds [deg / s^2]")
    scan_cosecant_modulation = Bool(False, help="Modulate the scan rate according to 1/sin(az) for uniform depth")
    scan_rate_el = Float(1.0, allow_none=True, help="The sky elevation scanning rate [deg / s]")
    scan_accel_el = Float(1.0, allow_none=True, help="Mount elevation rate acceleration [deg / s^2]")
    el_mod_step = Float(0.0, help="Amount to step elevation after each left-right scan pair [deg]")
    el_mod_rate = Float(0.0, help="Modulate elevation continuously at this rate [Hz]")
    el_mod_amplitude = Float(1.0, help="Range of elevation modulation [deg]")
    el_mod_sine = Bool(False, help="Modulate elevation with a sine wave instead of a triangle wave")
    el_mod_sine_phase = Float(0.0, allow_none=True, help="Add a per subscan extra phase to the sine modulation [deg]. "
                                                         "If negative adds random phase.")
    elnod_start = Bool(False, help="Perform an el-nod before the scan")
    elnod_end = Bool(False, help="Perform an el-nod after the scan")
    elnods = List([], help="List of relative el_nods [deg]")
    elnod_every_scan = Bool(False, help="Perform el nods every scan")
    detset_key = Unicode(None, allow_none=True, help="If specified, use this column of the focalplane detector_data "
                                                     "to group detectors")
    times = Unicode(defaults.times, help="Observation shared key for timestamps")
    shared_flags = Unicode(defaults.shared_flags, allow_none=True, help="Observation shared key for common flags")
    det_data = Unicode(defaults.det_data, allow_none=True, help="Observation detdata key to initialize")
    det_data_units = Unicode(defaults.det_data_units, help="Output units if creating detector data")
    det_flags = Unicode(defaults.det_flags, allow_none=True, help="Observation detdata key for flags to initialize")
    hwp_angle = Unicode(None, allow_none=True, help="Observation shared key for HWP angle")
    azimuth = Unicode(defaults.azimuth, help="Observation shared key for Azimuth")
    elevation = Unicode(defaults.elevation, help="Observation shared key for Elevation")
    boresight_azel = Unicode(defaults.boresight_azel, help="Observation shared key for boresight AZ/EL")
    boresight_radec = Unicode(defaults.boresight_radec, help="Observation shared key for boresight RA/DEC")
    hwp_rpm = Float(None, allow_none=True, help="The rate (in RPM) of the HWP rotation")
    scanning_interval = Unicode(defaults.scanning_interval, help="Interval name for scanning")
    turnaround_interval = Unicode(defaults.turnaround_interval, help="Interval name for turnarounds")
    throw_leftright_interval = Unicode(defaults.throw_leftright_interval,
                                       help="Interval name for left to right scans + turnarounds")
    throw_rightleft_interval = Unicode(defaults.throw_rightleft_interval,
                                       help="Interval name for right to left scans + turnarounds")
    throw_interval = Unicode("throw", help="Interval name for scan + turnaround intervals")
    scan_leftright_interval = Unicode("scan_leftright", help="Interval name for left to right scans")
    turn_leftright_interval = Unicode("turn_leftright", help="Interval name for turnarounds after left to right scans")
    scan_rightleft_interval = Unicode("scan_rightleft", help="Interval name for right to left scans")
    turn_rightleft_interval = Unicode("turn_rightleft", help="Interval name for turnarounds after right to left scans")
    turnaround_mask = Int(defaults.shared_mask_unstable_scanrate, help="Bit mask to raise turnaround flags with")
    elnod_interval = Unicode(defaults.elnod_interval, help="Interval name for elnods")
    elnod_mask = Int(defaults.shared_mask_irregular, help="Bit mask to raise elevation nod flags with")

    def _exec(self, data, detectors=None, **kwargs):
        if self.schedule is None and self.schedule_file is not None:
            sch = GroundSchedule()
            sch.read(self.schedule_file, sort=self.sort_schedule_file)
            self.schedule = sch
        if self.telescope is None:
            raise RuntimeError("The telescope attribute must be set before calling exec()")
        if self.schedule is None:
            raise RuntimeError("The schedule attribute must be set before calling exec()")
        if self.track_azimuth:
            raise NotImplementedError("track_azimuth needs the ephem package (sim_ground_utils.py:524-548)")
        if self.hwp_angle is not None and self.hwp_rpm is None:
            raise RuntimeError("Cannot simulate HWP without parameters")
        if (self.elnod_start or self.elnod_end) and len(self.elnods) == 0:
            raise RuntimeError("If simulating elnods, you must specify the list of offsets")
        focalplane = self.telescope.focalplane
        rate = focalplane.sample_rate
        lat, lon = np.radians(self.schedule.site_lat), np.radians(self.schedule.site_lon)
        comm = data.comm
        # scans are distributed round-robin over the process groups (sim_ground.py:497-520 uses
        # distribute_discrete over scan durations; one group per process here)
        mission_start = self.schedule.scans[0].start if self.schedule.scans else None
        incr = 1.0 / rate
        for iscan, scan in enumerate(self.schedule.scans):
            if comm.ngroups > 1 and iscan % comm.ngroups != comm.group:
                continue
            if np.abs((scan.stop - scan.start).total_seconds()) < incr:
                continue
            # sample indices relative to the global start time (sim_ground.py:526-540)
            ffirst = rate * (scan.start - mission_start).total_seconds()
            first = int(ffirst)
            if ffirst - first > 1.0e-3 * incr:
                first += 1
            t_start = first * incr + mission_start.timestamp()
            n_samples = 1 + int(rate * (scan.stop.timestamp() - t_start))
            if n_samples <= 1:
                continue
            stop_time = t_start + float(n_samples - 1) / rate        # sim_ground.py:920-921
            az_min, az_max, el = np.radians(scan.az_min), np.radians(scan.az_max), np.radians(scan.el)
            # sim_ground.py:905-1130 (_simulate_scanning): el-nod, the scan with its elevation modulations, el-nod
            rate_az, accel_az = np.radians(self.scan_rate_az), np.radians(self.scan_accel_az)
            min_el = max_el = el
            min_az, max_az = az_min, az_max
            parts, ival_elnod = [], []
            nod_el = nod_az = None
            if len(self.elnods) > 0:
                nod_el = np.array([el + np.radians(x) for x in self.elnods])
                nod_az = np.zeros_like(nod_el) + az_min
            nod_rates = (rate_az, accel_az, np.radians(self.scan_rate_el or 0.0), np.radians(self.scan_accel_el or 0.0))
            if self.elnod_start:
                nt, na, ne, min_az, max_az, min_el, max_el = simulate_elnod(t_start, rate, az_min, el, *nod_rates, nod_el,
                                                                            nod_az, min_az, max_az, min_el, max_el)
                if len(nt) > 0:
                    nt -= (nt[-1] - nt[0]) + incr       # ends one sample before the scan starts
                    parts.append((nt, na, ne))
                    ival_elnod.append((nt[0], nt[-1]))
            (times, az, elv, min_az, max_az, scan_lr, turn_lr, scan_rl, turn_rl, throw_lr, throw_rl) = simulate_ces_scan(
                t_start, stop_time, rate, el, az_min, az_max, az_min, rate_az, self.fix_rate_on_sky, accel_az, min_az, max_az,
                cosecant_modulation=self.scan_cosecant_modulation, randomize_phase=self.randomize_phase)
            if self.el_mod_rate > 0:
                min_el, max_el = oscillate_el(times, elv, nod_rates[2], nod_rates[3], min_el, max_el,
                                              np.radians(self.el_mod_amplitude), self.el_mod_rate, scan_lr, scan_rl,
                                              el_mod_sine=self.el_mod_sine,
                                              el_mod_sine_phase=None if self.el_mod_sine_phase is None
                                              else np.radians(self.el_mod_sine_phase))
            if np.radians(self.el_mod_step) > 0:
                min_el, max_el = step_el(times, az, elv, nod_rates[2], nod_rates[3], min_el, max_el,
                                         np.radians(self.el_mod_step))
            parts.append((times, az, elv))
            if self.elnod_end:
                nt, na, ne, min_az, max_az, min_el, max_el = simulate_elnod(times[-1] + incr, rate, az[-1], elv[-1],
                                                                            *nod_rates, nod_el, nod_az, min_az, max_az,
                                                                            min_el, max_el)
                if len(nt) > 0:
                    parts.append((nt, na, ne))
                    ival_elnod.append((nt[0], nt[-1]))
            if len(parts) > 1:
                times, az, elv = (np.hstack([p[k] for p in parts]) for k in range(3))
            name = f"{scan.name}-{scan.scan_indx}-{scan.subscan_indx}"
            ob = Observation(comm, Telescope(self.telescope.name, focalplane), len(times), name=name)
            ob["scan_el"] = scan.el
            ob["scan_min_az"], ob["scan_max_az"] = float(min_az), float(max_az)
            ob["scan_min_el"], ob["scan_max_el"] = float(min_el), float(max_el)
            ob["site"] = dict(name=self.schedule.site_name, lat=self.schedule.site_lat, lon=self.schedule.site_lon,
                              alt=self.schedule.site_alt)
            ob.set_times(times)
            ob.shared.create(self.azimuth, np.ascontiguousarray(az))
            ob.shared.create(self.elevation, np.ascontiguousarray(elv))
            # azimuth is measured clockwise, longitude counter-clockwise; focalplane X towards
            # decreasing elevation (sim_ground.py:765-775)
            bore_azel = from_lonlat_angles(-az, elv, np.zeros_like(elv))
            if scan.boresight_angle != 0:
                rot = synth.quat_rotation(np.array([0.0, 0.0, 1.0]), np.radians(scan.boresight_angle))
                bore_azel = synth.quat_normalize(synth.quat_mult(bore_azel, rot))
            ob.shared.create(self.boresight_azel, np.ascontiguousarray(bore_azel))
            ob.shared.create(self.boresight_radec, azel_to_radec(times, bore_azel, lat, lon))
            if self.hwp_angle is not None:
                # simulate_hwp_response (sim_hwp.py): constant rotation from the scan start
                ob.shared.create(self.hwp_angle, 2 * np.pi * (((times - times[0]) * self.hwp_rpm / 60.0) % 1.0))
            spans = {k: timespans_to_samples(times, v) for k, v in
                     dict(scan_lr=scan_lr, turn_lr=turn_lr, scan_rl=scan_rl, turn_rl=turn_rl, throw_lr=throw_lr,
                          throw_rl=throw_rl).items()}
            n = len(times)
            ob.intervals[self.throw_leftright_interval] = IntervalList(times, samplespans=spans["throw_lr"])
            ob.intervals[self.throw_rightleft_interval] = IntervalList(times, samplespans=spans["throw_rl"])
            ob.intervals[self.throw_interval] = IntervalList(times, samplespans=union_spans(n, spans["throw_lr"], spans["throw_rl"]))
            ob.intervals[self.scan_leftright_interval] = IntervalList(times, samplespans=spans["scan_lr"])
            ob.intervals[self.turn_leftright_interval] = IntervalList(times, samplespans=spans["turn_lr"])
            ob.intervals[self.scan_rightleft_interval] = IntervalList(times, samplespans=spans["scan_rl"])
            ob.intervals[self.turn_rightleft_interval] = IntervalList(times, samplespans=spans["turn_rl"])
            ob.intervals[self.scanning_interval] = IntervalList(times, samplespans=union_spans(n, spans["scan_lr"], spans["scan_rl"]))
            turn = union_spans(n, spans["turn_lr"], spans["turn_rl"])
            ob.intervals[self.turnaround_interval] = IntervalList(times, samplespans=turn)
            nod = timespans_to_samples(times, ival_elnod)
            ob.intervals[self.elnod_interval] = IntervalList(times, samplespans=nod)
            if self.shared_flags is not None:
                flags = np.zeros(n, dtype=np.uint8)
                for a, b in turn:      # FlagIntervals(view_mask=[(turnaround, turnaround_mask), (elnod, elnod_mask)])
                    flags[a:b] |= np.uint8(self.turnaround_mask)
                for a, b in nod:
                    flags[a:b] |= np.uint8(self.elnod_mask)
                ob.shared.create(self.shared_flags, flags)
            if self.det_data is not None:
                ob.detdata.create(self.det_data, dtype=np.float64, units=self.det_data_units)
            if self.det_flags is not None:
                ob.detdata.create(self.det_flags, dtype=np.uint8)
            data.obs.append(ob)

    def _finalize(self, data, **kwargs):
        return

    def _requires(self):
        return dict()

    def _provides(self):
        prov = {"shared": [self.times, self.azimuth, self.elevation, self.boresight_azel, self.boresight_radec],
                "detdata": [], "intervals": [self.scanning_interval, self.turnaround_interval,
                                             self.throw_leftright_interval, self.throw_rightleft_interval,
                                             self.elnod_interval]}
        if self.shared_flags is not None:
            prov["shared"].append(self.shared_flags)
        if self.det_data is not None:
            prov["detdata"].append(self.det_data)
        if self.det_flags is not None:
            prov["detdata"].append(self.det_flags)
        return prov


def create_ground_data_from_schedule(schedule, n_det=4, rate=20.0, fov_deg=4.0, net=1.0, fknee=0.05, fmin=1.0e-5,
                                     alpha=1.0, comm=None, **sim_ground_traits):
    """``Data`` with one observation per scheduled scan (the counterpart of the reference's test
    fixture ``create_ground_data``, src/toast/tests/helpers/ground.py), plus an analytic noise model."""
    from ..data import Comm, Focalplane
    from ..noise import AnalyticNoise

    comm = Comm() if comm is None else comm
    fp_q, gamma = synth.hex_focalplane(n_det, fov_deg=fov_deg)
    names = ["D%04d%s" % (i // 2, "AB"[i % 2]) for i in range(n_det)]
    fp = Focalplane(names, fp_q, gamma=gamma, epsilon=np.zeros(n_det), sample_rate=rate)
    data = Data(comm=comm)
    SimGround(telescope=Telescope("ground", fp), schedule=schedule, **sim_ground_traits).apply(data)
    for ob in data.obs:
        ob[defaults.noise_model] = AnalyticNoise(
            rate={d: rate for d in names}, fmin={d: fmin for d in names}, detectors=names,
            fknee={d: fknee for d in names}, alpha={d: alpha for d in names}, NET={d: net for d in names})
    return data
