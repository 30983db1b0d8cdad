#!/usr/bin/env python3
"""Generate tests/golden/fft_convolve.npz by RUNNING THE REFERENCE'S OWN convolution classes.

`toast.fft` cannot be imported here (the compiled _libtoast and its FFT plan classes are absent),
but `AlgorithmBase` and `AlgorithmNumpy` (src/toast/fft.py:121-350) are pure NumPy / SciPy.  This
script parses the reference file where it lies, compiles ONLY those two class definitions from
its syntax tree (decorators dropped, nothing copied into the repository) and runs
`AlgorithmNumpy(...).convolve(data)` -- the algorithm `toast.fft.convolve(..., algorithm="numpy")`
dispatches to and the one NoiseFilter uses.  Build container only; the fixture is committed.

    python tests/golden/make_golden_fft.py
"""
import ast
import os

import numpy as np
from scipy.interpolate import PchipInterpolator
from scipy.signal import windows

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src/toast/fft.py"


def load_reference_classes():
    tree = ast.parse(open(REF).read(), REF)
    classes = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name in ("AlgorithmBase", "AlgorithmNumpy")]
    assert [c.name for c in classes] == ["AlgorithmBase", "AlgorithmNumpy"]
    for c in classes:
        for f in c.body:
            if isinstance(f, ast.FunctionDef):
                f.decorator_list = []   # @function_timer
    mod = ast.Module(body=classes, type_ignores=[])
    ast.fix_missing_locations(mod)
    ns = {"np": np, "PchipInterpolator": PchipInterpolator, "windows": windows}
    exec(compile(mod, REF, "exec"), ns)
    return ns["AlgorithmNumpy"]


def main():
    AlgorithmNumpy = load_reference_classes()
    rng = np.random.default_rng(2024)
    out = {}
    cases = {"a": (3, 3000, 37.0, False), "b": (2, 4096, 200.0, False), "c": (2, 1500, 10.0, True)}
    for name, (n_tod, n_samp, rate, deconvolve) in cases.items():
        kfreq = np.concatenate([[0.0], np.geomspace(1e-4, rate / 2, 60)])
        mag = 1.0 / (1.0 + (0.2 / np.maximum(kfreq, 1e-6)) ** 1.3)
        mag[0] = 0.0
        if deconvolve:
            mag = 0.2 + mag   # away from zero
        phase = 0.3 * np.sin(kfreq / (rate / 2) * np.pi)
        kernels = np.array([(1.0 + 0.1 * i) * mag * np.exp(1j * phase) for i in range(n_tod)])
        data = rng.standard_normal((n_tod, n_samp)) + np.linspace(-1, 2, n_samp)[None, :]
        result = data.copy()
        alg = AlgorithmNumpy(n_tod, n_samp, rate, None, None, kfreq, kernels, None, deconvolve)
        alg.convolve(result)
        out.update({f"{name}_rate": np.array(rate), f"{name}_deconvolve": np.array(deconvolve), f"{name}_kernel_freq": kfreq,
                    f"{name}_kernels": kernels, f"{name}_data": data, f"{name}_out": result,
                    f"{name}_n_fft": np.array(alg.n_fft), f"{name}_apodize": alg.apodize})
    # a common (1-D) real kernel, the shape NoiseFilter passes per detector
    n_samp, rate = 2500, 50.0
    kfreq = np.concatenate([[0.0], np.geomspace(1e-3, rate / 2, 40)])
    kern = 1.0 / (1.0 + (0.5 / np.maximum(kfreq, 1e-6)) ** 2)
    kern[0] = 0
    data = rng.standard_normal((2, n_samp))
    result = data.copy()
    AlgorithmNumpy(2, n_samp, rate, None, None, kfreq, kern, None, False).convolve(result)
    out.update(d_rate=np.array(rate), d_kernel_freq=kfreq, d_kernels=kern, d_data=data, d_out=result)
    # extend_flags (src/toast/utils.py:1055-1113, pure NumPy): the function definition compiled from
    # the reference file in place, run on flag patterns covering every branch
    utils = "/root/reference/src/toast/utils.py"
    utree = ast.parse(open(utils).read(), utils)
    fn = [n for n in utree.body if isinstance(n, ast.FunctionDef) and n.name == "extend_flags"]
    umod = ast.Module(body=fn, type_ignores=[])
    ast.fix_missing_locations(umod)
    uns = {"np": np}
    exec(compile(umod, utils, "exec"), uns)
    patterns = {
        "none": np.zeros(40, np.uint8),
        "middle": np.array([0] * 10 + [1] * 3 + [0] * 12 + [3] * 2 + [0] * 13, np.uint8),
        "start": np.array([1] * 4 + [0] * 20 + [1] * 2 + [0] * 14, np.uint8),
        "end": np.array([0] * 25 + [1] * 15, np.uint8),
        "both": np.array([1] * 3 + [0] * 30 + [1] * 7, np.uint8),
        "only_start": np.array([1] * 9 + [0] * 31, np.uint8),
        "other_bits": np.array([2] * 5 + [0] * 10 + [3] * 2 + [0] * 10 + [2] * 13, np.uint8),
        "random": (rng.random(300) < 0.05).astype(np.uint8),
    }
    for pname, flags in patterns.items():
        for buf in (0, 1, 4, 50):
            got = flags.copy()
            uns["extend_flags"](got, 1, buf)
            out[f"ef_{pname}_{buf}"] = got
        out[f"ef_{pname}_in"] = flags
    # estimate_net (src/toast/ops/noise_model.py:108-170, pure NumPy / SciPy): white-noise level from
    # the high-frequency end of a PSD -- the normalisation of NoiseFilter's kernels
    from scipy.optimize import curve_fit

    nm = "/root/reference/src/toast/ops/noise_model.py"
    ntree = ast.parse(open(nm).read(), nm)
    fn = [n for n in ntree.body if isinstance(n, ast.FunctionDef) and n.name == "estimate_net"]
    nmod = ast.Module(body=fn, type_ignores=[])
    ast.fix_missing_locations(nmod)
    nns = {"np": np, "curve_fit": curve_fit}
    exec(compile(nmod, nm, "exec"), nns)
    for iname, (n_freq, rate, fknee, alpha, net) in enumerate([(77, 200.0, 0.05, 1.0, 50e-6), (60, 10.0, 0.1, 1.5, 1.0),
                                                             (300, 100.0, 1.0, 2.0, 3e-3), (12, 37.0, 0.5, 1.0, 2.0),
                                                             (8, 20.0, 0.2, 1.0, 0.7)]):
        f = np.geomspace(1e-5, rate / 2, n_freq)
        psd = net ** 2 * (f ** alpha + fknee ** alpha) / (f ** alpha + 1e-5 ** alpha)
        psd *= np.exp(0.02 * rng.standard_normal(n_freq))     # estimation noise
        out[f"net{iname}_freq"], out[f"net{iname}_psd"] = f, psd
        out[f"net{iname}_out"] = np.array(nns["estimate_net"](f, psd))
    np.savez_compressed(os.path.join(HERE, "fft_convolve.npz"), **out)
    print("wrote fft_convolve.npz", {k: v.shape for k, v in out.items() if k.endswith("_out")})


if __name__ == "__main__":
    main()
