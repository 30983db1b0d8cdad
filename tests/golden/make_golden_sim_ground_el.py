#!/usr/bin/env python3
"""Generate tests/golden/sim_ground_el.npz by RUNNING THE REFERENCE'S OWN elevation-motion functions.

The `toast` package cannot be imported here (astropy, ephem ... absent), but ``scan_time``, ``scan_profile``,
``scan_between``, ``simulate_elnod``, ``oscillate_el`` and ``step_el`` of src/toast/ops/sim_ground_utils.py are plain
NumPy.  This script parses that file where it lies under /root/reference, compiles ONLY those function definitions (plus
``simulate_stare`` / ``simulate_ces_scan``, which make the scans the modulations act on) from its syntax tree -- nothing is
copied into the repository; the decorators are dropped -- and calls them.  Build container only; the fixture (inputs and
outputs) is committed.

    python tests/golden/make_golden_sim_ground_el.py
"""
import ast
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src/toast/ops/sim_ground_utils.py"
WANTED = ("scan_time", "scan_profile", "scan_between", "simulate_elnod", "oscillate_el", "step_el", "simulate_stare",
          "simulate_ces_scan")
D = np.radians


def load_reference():
    tree = ast.parse(open(REF).read(), REF)
    funcs = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in WANTED]
    assert sorted(f.name for f in funcs) == sorted(WANTED)
    for f in funcs:
        f.decorator_list = []
    mod = ast.Module(body=funcs, type_ignores=[])
    ast.fix_missing_locations(mod)
    ns = {"np": np}
    exec(compile(mod, REF, "exec"), ns)
    return ns


#: scan_between: (time_start, az1, el1, az2, el2, az_rate, az_accel, el_rate, el_accel)
BETWEEN = {
    "both_axes": (1.7e9, D(40.0), D(50.0), D(47.0), D(53.0), D(1.0), D(1.0), D(1.0), D(1.0)),
    "el_only_short": (1.7e9 + 5.5, D(120.0), D(35.0), D(120.0), D(35.4), D(1.5), D(0.7), D(1.2), D(2.0)),
    "downwards": (1.68e9, D(200.0), D(60.0), D(190.0), D(57.5), D(2.0), D(0.5), D(0.8), D(0.4)),
}
#: simulate_elnod: (t_start, rate, az_start, el_start, az_rate, az_accel, el_rate, el_accel, offsets in degrees)
ELNOD = {
    "three_stations": (1.7e9, 20.0, D(40.0), D(50.0), D(1.0), D(1.0), D(1.0), D(1.0), (1.0, -1.0, 0.0)),
    "five_stations_fast": (1.71e9 + 0.25, 37.0, D(120.0), D(35.0), D(1.5), D(0.7), D(2.0), D(3.0), (2.0, 0.0, -2.0, 0.5, 0.0)),
    "starts_on_station": (1.69e9, 10.0, D(250.0), D(45.0), D(1.0), D(1.0), D(0.5), D(0.25), (0.0, 1.5, 0.0)),
}
#: the scans the modulations act on: simulate_ces_scan(t_start, t_stop, rate, el, az_min, az_max, az_rate, on_sky, accel)
SCANS = {
    "scan_a": (1.7e9, 1.7e9 + 300.0, 20.0, D(50.0), D(40.0), D(75.0), D(1.0), True, D(1.0)),
    "scan_b": (1.71e9 + 3.25, 1.71e9 + 503.25, 25.0, D(45.0), D(200.0), D(230.0), D(1.2), False, D(0.8)),
}
#: oscillate_el: (scan, el_rate, el_accel, amplitude, rate_hz, sine, sine_phase)
OSCILLATE = {
    "constant_rate": ("scan_a", D(1.0), D(1.0), D(1.0), 0.02, False, None),
    "constant_rate_fast": ("scan_b", D(2.0), D(4.0), D(0.3), 0.1, False, None),
    "sine": ("scan_a", D(1.0), D(1.0), D(0.5), 0.05, True, None),
    "sine_fixed_phase": ("scan_b", D(2.0), D(4.0), D(0.25), 0.04, True, 0.35),
    "sine_random_phase": ("scan_a", D(1.0), D(1.0), D(0.5), 0.05, True, -0.7),
}
#: step_el: (scan, el_rate, el_accel, step)
STEP = {
    "step_up": ("scan_a", D(1.0), D(1.0), D(0.2)),
    "step_down_reaches_rate": ("scan_b", D(0.5), D(2.0), D(-1.0)),
}


def main():
    ref = load_reference()
    out = {}
    for name, args in BETWEEN.items():
        t, az, el = ref["scan_between"](*args)
        out[f"between_{name}_args"] = np.array(args)
        out[f"between_{name}_t"], out[f"between_{name}_az"], out[f"between_{name}_el"] = t[::97], az[::97], el[::97]
        out[f"between_{name}_times"] = np.array([ref["scan_time"](args[1], args[3], args[5], args[6]),
                                                 ref["scan_time"](args[2], args[4], args[7], args[8])])
    for name, (t0, rate, az0, el0, azr, aza, elr, ela, offsets) in ELNOD.items():
        elnod_el = np.array([el0 + D(x) for x in offsets])
        elnod_az = np.zeros_like(elnod_el) + az0
        res = ref["simulate_elnod"](t0, rate, az0, el0, azr, aza, elr, ela, elnod_el, elnod_az, az0 - 0.1, az0 + 0.3, el0,
                                    el0)
        out[f"elnod_{name}_args"] = np.array([t0, rate, az0, el0, azr, aza, elr, ela])
        out[f"elnod_{name}_offsets"] = np.array(offsets)
        out[f"elnod_{name}_t"], out[f"elnod_{name}_az"], out[f"elnod_{name}_el"] = res[0], res[1], res[2]
        out[f"elnod_{name}_range"] = np.array(res[3:], dtype=np.float64)
    scans = {}
    for name, (t0, t1, rate, el, azmin, azmax, azrate, fix, accel) in SCANS.items():
        res = ref["simulate_ces_scan"](None, t0, t1, rate, el, azmin, azmax, azmin, azrate, fix, accel, azmin, azmax)
        scans[name] = res
        out[f"{name}_args"] = np.array([t0, t1, rate, el, azmin, azmax, azrate, float(fix), accel])
    for name, (scan, elr, ela, amp, hz, sine, phase) in OSCILLATE.items():
        res = scans[scan]
        el = res[2].copy()
        rng = ref["oscillate_el"](res[0], el, elr, ela, float(res[2][0]), float(res[2][0]), amp, hz, res[5], res[7],
                                  el_mod_sine=sine, el_mod_sine_phase=phase)
        out[f"oscillate_{name}_args"] = np.array([elr, ela, amp, hz, float(sine), np.nan if phase is None else phase])
        out[f"oscillate_{name}_scan"] = np.array([list(SCANS).index(scan)])
        out[f"oscillate_{name}_el"] = el
        out[f"oscillate_{name}_range"] = np.array(rng, dtype=np.float64)
    for name, (scan, elr, ela, step) in STEP.items():
        res = scans[scan]
        el = res[2].copy()
        rng = ref["step_el"](res[0], res[1], el, elr, ela, float(res[2][0]), float(res[2][0]), step)
        out[f"step_{name}_args"] = np.array([elr, ela, step])
        out[f"step_{name}_scan"] = np.array([list(SCANS).index(scan)])
        out[f"step_{name}_el"] = el
        out[f"step_{name}_range"] = np.array(rng, dtype=np.float64)
    path = os.path.join(HERE, "sim_ground_el.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
