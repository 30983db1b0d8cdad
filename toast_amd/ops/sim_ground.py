"""SimGround: schedule-driven constant-elevation scans of a ground telescope -- the generator of
BASELINE configs[4]'s pointing inputs.

Reference: src/toast/ops/sim_ground.py:61-1270 (operator), src/toast/ops/sim_ground_utils.py:458-753
(``simulate_ces_scan``).  This is input generation and runs on the host in the reference too
(NumPy interpolation of one high-resolution scan / turnaround cycle); nothing here is on the
A / A^T path.  Reproduced: the azimuth profile with finite-acceleration turnarounds (bit-identical
restatement of ``simulate_ces_scan``, pinned to the reference function's own outputs in
tests/golden/sim_ground.npz), optional cosecant modulation and randomised phase, the seven
interval lists (scan / turn / throw, left-right and right-left) plus ``scanning`` / ``turnaround`` /
``throw``, turnaround bits in the shared flags (``FlagIntervals`` with ``turnaround_mask``), the
horizontal boresight ``qa.from_lonlat_angles(-az, el, 0) * Rz(boresight_angle)``, HWP angle,
detector data / flag buffers.

Not reproduced (their dependencies -- astropy, ephem, qpoint -- are absent, and none of it changes
the structure of the map-making inputs): el-nods and el modulation, Sun intervals, weather,
``track_azimuth``, site position / velocity, and the astrometric part of horizontal -> equatorial.
The equatorial boresight is the rigid rotation
``Rz(ERA(t) + lon) Ry(pi/2 - lat) Rz(pi) q_azel`` with the Earth rotation angle of the UTC stamp
(no precession, nutation, aberration or refraction: arcminute-level differences from the
reference's ephem / astropy conversion)."""

from datetime import timezone

import numpy as np

from .. import synth
from ..data import Data, IntervalList, Observation, Telescope, defaults
from ..schedule import GroundSchedule
from ..traits import Bool, Float, Instance, Int, Unicode
from .operator import Operator


def simulate_stare(t_start, t_stop, rate, el, az):
    """sim_ground_utils.py:435-455: no motion, no intervals."""
    samples = int((t_stop - t_start) * rate)
    times = t_start + np.arange(samples) / rate
    az_sample = np.zeros(samples) + az
    el_sample = np.zeros(samples) + el
    return times, az_sample, el_sample, az, az, [], [], [], [], [], []


def simulate_ces_scan(t_start, t_stop, rate, el, az_min, az_max, az_start, az_rate, fix_rate_on_sky, az_accel,
                      scan_min_az, scan_max_az, cosecant_modulation=False, nstep=10000, randomize_phase=False):
    """One constant-elevation scan (angles in radians, rates in rad/s, times in s): a full
    left-right / turnaround / right-left / turnaround cycle is built at high resolution and
    interpolated to the sample times (sim_ground_utils.py:458-753, without ``track_azimuth``).
    Returns (times, az, el, min_az, max_az, scan_leftright, turn_leftright, scan_rightleft,
    turn_rightleft, throw_leftright, throw_rightleft) with the interval lists as time spans."""
    if np.abs(az_min - az_max) < 1e-10:
        return simulate_stare(t_start, t_stop, rate, el, az_min)
    mirror_cosecant = False
    if cosecant_modulation:
        if az_min > np.pi:
            mirror_cosecant = True
        az_min %= np.pi
        az_max %= np.pi
        if az_min > az_max:
            raise RuntimeError("Cannot scan across zero meridian with cosecant-modulated scan")
    elif az_max < az_min:
        az_max += 2 * np.pi
    base_rate = az_rate / np.cos(el) if fix_rate_on_sky else az_rate
    scan_accel = az_accel
    if cosecant_modulation:
        scan_time = (np.cos(az_min) - np.cos(az_max)) / base_rate
        dazdt = base_rate / np.abs(np.sin(az_min))
    else:
        scan_time = (az_max - az_min) / base_rate
        dazdt = base_rate
    turnaround_time = 2 * dazdt / scan_accel
    scan_pair_time = 2 * scan_time + 2 * turnaround_time
    az_drift = 0
    drift_time = 0
    all_t, all_az = [], []
    # left-to-right
    t0 = t_start
    t1 = t0 + scan_time + drift_time
    if cosecant_modulation:
        tvec = np.linspace(t0, t1, nstep, endpoint=True)
        azvec = np.arccos(np.cos(az_min) + base_rate * t0 - base_rate * tvec)
    else:
        tvec = np.array([t0, t1 + drift_time])
        azvec = np.array([az_min, az_max + az_drift])
    all_t.append(np.array(tvec))
    all_az.append(np.array(azvec))
    range_scan_leftright = (t0, t1)
    # turnaround
    t0 = t1
    az0 = az_max + az_drift
    t1 = t0 + turnaround_time
    tvec = np.linspace(t0, t1, nstep, endpoint=True)[1:]
    azvec = az0 + (tvec - t0) * dazdt - 0.5 * scan_accel * (tvec - t0) ** 2
    all_t.append(np.array(tvec[:-1]))
    all_az.append(np.array(azvec[:-1]))
    range_turn_leftright = (t0, t1)
    # right-to-left
    t0 = t1
    t1 = t0 + scan_time - drift_time
    if cosecant_modulation:
        tvec = np.linspace(t0, t1, nstep, endpoint=True)
        azvec = np.arccos(np.cos(az_max) - base_rate * t0 + base_rate * tvec)
    else:
        tvec = np.array([t0, t1])
        azvec = np.array([az_max + az_drift, az_min + 2 * az_drift])
    all_t.append(np.array(tvec))
    all_az.append(np.array(azvec))
    range_scan_rightleft = (t0, t1)
    # turnaround
    t0 = t1
    az0 = az_min + 2 * az_drift
    t1 = t0 + turnaround_time
    tvec = np.linspace(t0, t1, nstep, endpoint=True)[1:]
    azvec = az0 - (tvec - t0) * dazdt + 0.5 * scan_accel * (tvec - t0) ** 2
    all_t.append(np.array(tvec))
    all_az.append(np.array(azvec))
    range_turn_rightleft = (t0, t1)
    tvec = np.hstack(all_t)
    azvec = np.hstack(all_az)
    if np.amin(azvec) < -2 * np.pi:
        azvec += 2 * np.pi
    if np.amax(azvec) > 2 * np.pi:
        azvec -= 2 * np.pi
    if mirror_cosecant:
        azvec += np.pi
    n_repeat = int((t_stop - t_start) / scan_pair_time)
    n_repeat += 2
    tvec = tvec[:-1]
    azvec = azvec[:-1]
    t, az = [], []
    for i in range(n_repeat):
        t.append(tvec + i * scan_pair_time)
        az.append(azvec + i * 2 * az_drift)
    tvec = np.hstack(t)
    azvec = np.hstack(az)
    new_min_az = min(scan_min_az, np.min(azvec))
    new_max_az = max(scan_max_az, np.max(azvec))
    samples = int((t_stop - t_start) * rate)
    times = t_start + np.arange(samples) / rate
    if randomize_phase:
        np.random.seed(int(t_start % 2**32))
        t_off = scan_pair_time * np.random.rand()
    else:
        t_off = 0
    az_sample = np.interp(times + t_off, tvec, azvec)
    el_sample = np.zeros_like(az_sample) + el
    ival = {k: [] for k in ("scan_lr", "scan_rl", "turn_lr", "turn_rl", "throw_lr", "throw_rl")}
    t_off = -t_off
    for _ in range(n_repeat):
        ival["scan_lr"].append((range_scan_leftright[0] + t_off, range_scan_leftright[1] + t_off))
        ival["turn_lr"].append((range_turn_leftright[0] + t_off, range_turn_leftright[1] + t_off))
        ival["scan_rl"].append((range_scan_rightleft[0] + t_off, range_scan_rightleft[1] + t_off))
        ival["turn_rl"].append((range_turn_rightleft[0] + t_off, range_turn_rightleft[1] + t_off))
        half_turn_lr = 0.5 * (range_turn_leftright[1] - range_turn_leftright[0])
        half_turn_rl = 0.5 * (range_turn_rightleft[1] - range_turn_rightleft[0])
        ival["throw_lr"].append((range_scan_leftright[0] + t_off - half_turn_rl,
                                 range_scan_leftright[1] + t_off + half_turn_lr))
        ival["throw_rl"].append((range_scan_rightleft[0] + t_off - half_turn_lr,
                                 range_scan_rightleft[1] + t_off + half_turn_rl))
        t_off += scan_pair_time
    # trim to the time stamps (the reference tests ival[-1] where it means ival[0]; same outcome
    # for the first element whenever the list is longer than one entry -- kept literally)
    for key in ("scan_lr", "scan_rl", "turn_lr", "turn_rl", "throw_lr", "throw_rl"):
        lst = ival[key]
        first = tuple(lst[-1])
        if first[1] < times[0]:
            del lst[0]
        elif first[0] < times[0]:
            lst[0] = (times[0], first[1])
        last = tuple(lst[-1])
        if last[0] > times[-1]:
            del lst[-1]
        elif last[1] > times[-1]:
            lst[-1] = (last[0], times[-1])
    return (times, az_sample, el_sample, new_min_az, new_max_az, ival["scan_lr"], ival["turn_lr"], ival["scan_rl"],
            ival["turn_rl"], ival["throw_lr"], ival["throw_rl"])


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
            (times, az, elv, min_az, max_az, scan_lr, turn_lr, scan_rl, turn_rl, throw_lr, throw_rl) = simulate_ces_scan(
                t_start, stop_time, rate, el, az_min, az_max, az_min, np.radians(self.scan_rate_az),
                self.fix_rate_on_sky, np.radians(self.scan_accel_az), az_min, az_max,
                cosecant_modulation=self.scan_cosecant_modulation, randomize_phase=self.randomize_phase)
            name = f"{scan.name}-{scan.scan_indx}-{scan.subscan_indx}"
            ob = Observation(comm, Telescope(self.telescope.name, focalplane), len(times), name=name)
            ob["scan_el"] = scan.el
            ob["scan_min_az"], ob["scan_max_az"] = float(min_az), float(max_az)
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
            if self.shared_flags is not None:
                flags = np.zeros(n, dtype=np.uint8)
                for a, b in turn:      # FlagIntervals(view_mask=[(turnaround, turnaround_mask)])
                    flags[a:b] |= np.uint8(self.turnaround_mask)
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
                                             self.throw_leftright_interval, self.throw_rightleft_interval]}
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
