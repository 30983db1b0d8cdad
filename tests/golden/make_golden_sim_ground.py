#!/usr/bin/env python3
"""Generate tests/golden/sim_ground.npz by RUNNING THE REFERENCE'S OWN ``simulate_ces_scan``.

The `toast` package cannot be imported here (astropy, ephem ... absent), but the constant-elevation
scan simulator is plain NumPy unless ``track_azimuth`` is requested.  This script parses
src/toast/ops/sim_ground_utils.py where it lies under /root/reference, compiles ONLY the function
definitions `simulate_stare` and `simulate_ces_scan` from its syntax tree (nothing is copied into
the repository; the decorators are dropped) and calls them.  Build container only; the fixture
(inputs + outputs) is committed.

    python tests/golden/make_golden_sim_ground.py
"""
import ast
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src/toast/ops/sim_ground_utils.py"
WANTED = ("simulate_stare", "simulate_ces_scan")

#: name -> (t_start, t_stop, rate, el, az_min, az_max, az_rate, fix_rate_on_sky, az_accel, cosecant, randomize)
CASES = {
    "ces_basic": (1.7e9, 1.7e9 + 300.0, 20.0, np.radians(50.0), np.radians(40.0), np.radians(75.0), np.radians(1.0), True,
                  np.radians(1.0), False, False),
    "ces_mount_rate": (1.7e9 + 17.0, 1.7e9 + 417.0, 37.0, np.radians(35.0), np.radians(120.0), np.radians(190.0),
                       np.radians(1.5), False, np.radians(0.7), False, False),
    "ces_wrap": (1.68e9, 1.68e9 + 500.0, 10.0, np.radians(60.0), np.radians(350.0), np.radians(20.0), np.radians(0.8), True,
                 np.radians(2.0), False, False),
    "ces_random_phase": (1.71e9 + 3.25, 1.71e9 + 303.25, 25.0, np.radians(45.0), np.radians(200.0), np.radians(260.0),
                         np.radians(1.0), True, np.radians(1.0), False, True),
    "ces_cosecant": (1.7e9, 1.7e9 + 400.0, 20.0, np.radians(55.0), np.radians(30.0), np.radians(100.0), np.radians(0.5), True,
                     np.radians(1.0), True, False),
    "ces_cosecant_setting": (1.7e9, 1.7e9 + 400.0, 20.0, np.radians(55.0), np.radians(230.0), np.radians(310.0),
                             np.radians(0.5), True, np.radians(1.0), True, False),
    "stare": (1.7e9, 1.7e9 + 60.0, 20.0, np.radians(50.0), np.radians(40.0), np.radians(40.0), np.radians(1.0), True,
              np.radians(1.0), False, False),
    "configs4_one_hour": (1.8e9, 1.8e9 + 3600.0, 200.0, np.radians(50.0), np.radians(40.0), np.radians(110.0),
                          np.radians(1.0), False, np.radians(1.0), False, False),
}
NAMES = ("times", "az", "el", "min_az", "max_az", "scan_leftright", "turn_leftright", "scan_rightleft",
         "turn_rightleft", "throw_leftright", "throw_rightleft")


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
    return ns["simulate_ces_scan"]


def main():
    ces = load_reference()
    out = {}
    for name, (t0, t1, rate, el, azmin, azmax, azrate, fix, accel, cosec, rand) in CASES.items():
        res = ces(None, t0, t1, rate, el, azmin, azmax, azmin, azrate, fix, accel, azmin, azmax,
                  cosecant_modulation=cosec, randomize_phase=rand, track_azimuth=False)
        out[name + "_args"] = np.array([t0, t1, rate, el, azmin, azmax, azrate, float(fix), accel, float(cosec), float(rand)])
        for key, val in zip(NAMES, res):
            arr = np.asarray(val, dtype=np.float64)
            if key in ("times", "el"):
                # fully determined by (first, last, count) resp. constant: keep the fixture small
                assert key == "times" or np.all(arr == arr[0])
                arr = np.array([arr[0], arr[-1], arr.size])
            if key == "az" and arr.size > 8000:
                arr = arr[::53]
            out[f"{name}_{key}"] = arr.reshape(-1, 2) if key.endswith(("leftright", "rightleft")) else arr
    path = os.path.join(HERE, "sim_ground.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
