#!/usr/bin/env python3
"""Generate tests/golden/offset_prior.npz by RUNNING THE REFERENCE'S OWN HELPERS.

The `toast` package cannot be imported in this container (astropy and the compiled _libtoast
are absent), but the four helpers of the Offset template's noise prior are pure NumPy / SciPy
methods.  This script parses src/toast/templates/offset/offset.py where it lies under
/root/reference, compiles ONLY the method definitions `_interpolate_psd`, `_truncate`,
`_remove_white_noise` and `_get_offset_psd` from its syntax tree (nothing is copied into the
repository) and calls them.  `_get_offset_psd` reads its inputs through astropy quantities
(`noise.freq(det).to_value(u.Hz)`): the inputs handed in here are plain arrays whose `to_value`
returns the array itself, i.e. every unit is 1 -- the arithmetic that is pinned is the
reference's.  Build container only; the fixture (inputs + outputs) is committed.

    python tests/golden/make_golden_offset_prior.py
"""
import ast
import os
import sys
import types

import numpy as np
import scipy
import scipy.optimize

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src/toast/templates/offset/offset.py"
WANTED = ("_interpolate_psd", "_truncate", "_remove_white_noise", "_get_offset_psd")


def load_reference_helpers():
    tree = ast.parse(open(REF).read(), REF)
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "Offset"][0]
    funcs = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in WANTED]
    assert sorted(f.name for f in funcs) == sorted(WANTED)
    for f in funcs:
        f.decorator_list = []
    mod = ast.Module(body=funcs, type_ignores=[])
    ast.fix_missing_locations(mod)
    unit = types.SimpleNamespace(Hz=1.0, second=1.0)
    ns = {"np": np, "scipy": scipy, "u": unit}
    exec(compile(mod, REF, "exec"), ns)
    holder = types.SimpleNamespace(det_data_units=1.0)
    for name in WANTED:
        setattr(holder, name, types.MethodType(ns[name], holder))
    return holder


class Plain(np.ndarray):
    def to_value(self, unit):
        return np.asarray(self)


class FakeNoise:
    def __init__(self, freq, psd):
        self._f, self._p = freq.view(Plain), psd.view(Plain)

    def freq(self, det):
        return self._f

    def psd(self, det):
        return self._p


def analytic_psd(rate, fmin, fknee, alpha, net):
    # the reference's analytic noise model grid and shape (src/toast/noise_sim.py:88-112)
    nyq = rate / 2.0
    f = []
    cur = 1.0e-9
    while cur < nyq:
        f.append(cur)
        cur *= 1.4
    f.append(nyq)
    f = np.array(f)
    t = np.power(f, alpha)
    return f, (t + fknee ** alpha) / (t + fmin ** alpha) * net ** 2


def main():
    ref = load_reference_helpers()
    out = {}
    cases = [
        # rate, fmin, fknee, alpha, NET, obstime, step_time, n_amp_view
        (200.0, 1.0e-5, 0.05, 1.0, 50.0e-6, 3600.0, 1.0, 3600),
        (100.0, 1.0e-5, 0.10, 1.5, 1.0, 1800.0, 2.5, 720),
        (37.0, 1.0e-4, 1.00, 2.0, 3.0e-3, 600.0, 10.0, 60),
    ]
    out["cases"] = np.array(cases)
    for ic, (rate, fmin, fknee, alpha, net, obstime, step, n_amp) in enumerate(cases):
        f, p = analytic_psd(rate, fmin, fknee, alpha, net)
        out[f"c{ic}_psdfreq"], out[f"c{ic}_psd"] = f, p
        out[f"c{ic}_corrpsd"] = ref._remove_white_noise(f, p)
        powmin = np.floor(np.log10(1 / obstime)) - 1
        powmax = min(np.ceil(np.log10(1 / step)) + 2, np.log10(rate))
        freq = np.logspace(powmin, powmax, 1000)
        out[f"c{ic}_freq"] = freq
        opsd = ref._get_offset_psd(FakeNoise(f, p), freq, step, "d")
        out[f"c{ic}_offset_psd"] = opsd
        filterlen = 2
        while filterlen < 2 * n_amp:
            filterlen *= 2
        ff = np.fft.rfftfreq(filterlen, step)
        four = ref._interpolate_psd(ff, np.log(freq), np.log(1.0 / opsd))
        out[f"c{ic}_fourierfilter"] = four
        out[f"c{ic}_noisefilter"] = ref._truncate(np.fft.irfft(four))
        out[f"c{ic}_toeplitz"] = ref._truncate(np.fft.irfft(ref._interpolate_psd(ff, np.log(freq), np.log(opsd))))
    # _interpolate_psd on negative / zero arguments
    x = np.array([-3.0, -1.0e-11, 0.0, 1.0e-12, 1.0e-10, 0.5, 7.0, 1.0e3])
    lf = np.log(np.array([1.0e-3, 1.0e-1, 1.0, 10.0]))
    lp = np.log(np.array([9.0, 4.0, 1.0, 0.5]))
    out["interp_x"], out["interp_lf"], out["interp_lp"] = x, lf, lp
    out["interp_out"] = ref._interpolate_psd(x, lf, lp)
    np.savez_compressed(os.path.join(HERE, "offset_prior.npz"), **out)
    print("wrote offset_prior.npz:", {k: v.shape for k, v in out.items() if k.startswith("c0")})


if __name__ == "__main__":
    sys.exit(main())
