"""Ground observing schedules: the inputs of ``ops.SimGround`` (reference: src/toast/schedule.py
GroundScan :45-97, GroundSchedule text formats :386-660).  Only what the constant-elevation-scan
simulation reads: a header (site name, telescope name, latitude, longitude, altitude) and, per
scan, start / stop time, boresight angle, name, azimuth range and elevation."""

from datetime import datetime, timezone


class GroundScan:
    """One constant-elevation scan (angles in degrees; reference schedule.py:45-97)."""

    def __init__(self, name=None, start=None, stop=None, boresight_angle=0.0, az_min=0.0, az_max=0.0, el=0.0,
                 scan_indx=0, subscan_indx=0):
        self.name = name
        self.start = start
        self.stop = stop
        self.boresight_angle = float(boresight_angle)
        self.az_min = float(az_min)
        self.az_max = float(az_max)
        self.el = float(el)
        self.rising = (self.az_min % 360.0) < 180.0
        self.scan_indx = scan_indx
        self.subscan_indx = subscan_indx

    def __repr__(self):
        return (f"<GroundScan '{self.name}' at {self.start.isoformat(timespec='seconds')} with El = {self.el} deg, "
                f"Az {self.az_min} deg -- {self.az_max} deg>")


def _parse_time(text):
    import dateutil.parser

    try:
        return dateutil.parser.parse(text + " +0000")
    except Exception:
        t = dateutil.parser.parse(text)
        return t if t.tzinfo is not None else t.replace(tzinfo=timezone.utc)


class GroundSchedule:
    """A list of ``GroundScan`` plus the site (reference schedule.py:312-938)."""

    def __init__(self, scans=None, site_name="Unknown", telescope_name="Unknown", site_lat=0.0, site_lon=0.0,
                 site_alt=0.0):
        self.scans = [] if scans is None else list(scans)
        self.site_name = site_name
        self.telescope_name = telescope_name
        self.site_lat = float(site_lat)     # degrees
        self.site_lon = float(site_lon)     # degrees
        self.site_alt = float(site_alt)     # metres

    # text formats (schedule.py:386-520): fields per version
    _N_FIELDS = {4: 9, 3: 11, 2: 22, 1: 24}

    @staticmethod
    def _split(line, sep):
        fields = line.split(sep)
        return line.split() if len(fields) == 1 else [f.strip() for f in fields]

    def _parse_line(self, line, version, sep):
        f = self._split(line, sep)
        if len(f) != self._N_FIELDS[version]:
            raise RuntimeError(f"Version {version} schedule line does not have {self._N_FIELDS[version]} fields")
        if version == 4:
            start, stop, bangle, name, azmin, azmax, el, scan, subscan = f
        elif version == 3:
            start, stop = f[0] + " " + f[1], f[2] + " " + f[3]
            bangle, name, azmin, azmax, el, scan, subscan = f[4:]
        elif version == 2:
            start, stop, bangle, name, azmin, azmax, el = f[0], f[1], f[4], f[5], f[6], f[7], f[8]
            scan, subscan = f[19], f[20]
        else:
            start, stop = f[0] + " " + f[1], f[2] + " " + f[3]
            bangle, name, azmin, azmax, el = f[6], f[7], f[8], f[9], f[10]
            scan, subscan = f[21], f[22]
        return GroundScan(name, _parse_time(start), _parse_time(stop), float(bangle), float(azmin), float(azmax),
                          float(el), scan, subscan)

    def read(self, schedule_file, file_split=None, sort=False, field_separator="|"):
        """Load a text schedule (versions 4, 3, 2, 1 are tried in that order, like the reference).
        ``file_split = (isplit, nsplit)`` keeps every nsplit-th rising / setting pass of a patch."""
        with open(schedule_file, "r") as fh:
            lines = [ln.rstrip("\n") for ln in fh if not ln.startswith("#") and "SPECIAL" not in ln and ln.strip()]
        if not lines:
            raise RuntimeError("Schedule file does not have a recognized format")
        last_err = None
        for version in (4, 3, 2, 1):
            try:
                head = self._split(lines[0], field_separator)
                if len(head) != 5:
                    raise RuntimeError("schedule header must have 5 fields")
                scans = [self._parse_line(ln, version, field_separator) for ln in lines[1:]]
            except Exception as err:  # try the next format
                last_err = err
                continue
            self.site_name, self.telescope_name = head[0], head[1]
            self.site_lat, self.site_lon, self.site_alt = float(head[2]), float(head[3]), float(head[4])
            if file_split is not None:
                isplit, nsplit = file_split
                kept, counters, last_name, iscan = [], {}, None, 0
                for s in scans:
                    if s.name != last_name:
                        c = counters.setdefault(s.name, {})
                        key = "R" if s.rising else "S"
                        c[key] = c[key] + 1 if key in c else 0
                        iscan = c[key]
                    last_name = s.name
                    if iscan % nsplit == isplit:
                        kept.append(s)
                scans = kept
            self.scans = scans
            if sort:
                self.sort_by_name()
            return
        raise RuntimeError(f"Schedule file does not have a recognized format ({last_err})")

    def write(self, schedule_file):
        """Concise text format (version 4 layout of the reference's writer, schedule.py:745-800)."""
        with open(schedule_file, "w") as f:
            f.write("# Site | Telescope | Latitude [deg] | Longitude [deg] | Elevation [m]\n")
            f.write(f"{self.site_name:15} | {self.telescope_name:15} | {self.site_lat:15.3f} |"
                    f"{self.site_lon:15.3f} |{self.site_alt:15.1f}\n")
            f.write("# Start time UTC | Stop time UTC | Rotation | Patch name | Az min | Az max | El | Pass | Sub\n")
            for s in self.scans:
                f.write(f"{s.start.astimezone(timezone.utc).strftime('%Y-%m-%d %H:%M:%S')} | "
                        f"{s.stop.astimezone(timezone.utc).strftime('%Y-%m-%d %H:%M:%S')} | "
                        f"{s.boresight_angle:8.2f} | {s.name:35} | {s.az_min:8.2f} | {s.az_max:8.2f} | {s.el:8.2f} | "
                        f"{s.scan_indx:5} | {s.subscan_indx:3}\n")

    def sort_by_name(self):
        self.scans = sorted(self.scans, key=lambda s: s.name)


def make_ces_schedule(n_scan, t_start="2027-01-01T00:00:00", scan_seconds=3600.0, gap_seconds=60.0, az_min=40.0,
                      az_max=110.0, el=50.0, site_lat=-22.958, site_lon=-67.786, site_alt=5200.0, name="patch"):
    """A regular schedule of ``n_scan`` constant-elevation scans (alternating rising / setting
    azimuth ranges): what BASELINE configs[4] needs as an input when no scheduler output is at hand."""
    from datetime import timedelta

    t0 = datetime.fromisoformat(t_start).replace(tzinfo=timezone.utc)
    scans = []
    for i in range(n_scan):
        start = t0 + timedelta(seconds=i * (scan_seconds + gap_seconds))
        lo, hi = (az_min, az_max) if i % 2 == 0 else (360.0 - az_max, 360.0 - az_min)
        scans.append(GroundScan(f"{name}", start, start + timedelta(seconds=scan_seconds), 0.0, lo, hi, el, i, 0))
    return GroundSchedule(scans, "ATACAMA", "LAT", site_lat, site_lon, site_alt)
