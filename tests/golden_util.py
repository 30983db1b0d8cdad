"""Load the committed golden vectors (tests/golden/*.npz, produced by the reference itself:
tests/golden/make_golden.py) and check an implementation against them."""
import os

import numpy as np

import cases

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CHAINS = ("chain_a", "chain_b", "chain_c")


def load_chain(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    case = {k[3:]: np.ascontiguousarray(z[k]) for k in z.files if k.startswith("in_")}
    case["intervals"] = case["intervals"].astype(cases.interval_dtype)
    for k in ("n_det", "n_samp", "nside", "rows", "n_pix_submap", "n_submap"):
        case[k] = int(z["meta_" + k])
    want = {k[4:]: z[k] for k in z.files if k.startswith("out_")}
    return case, want, bool(z["meta_nest"]), bool(z["meta_iau"])


def check_chain(impl, name, tail=(), weights_rtol=1e-13, ztol=1e-12):
    case, want, nest, iau = load_chain(name)
    got = cases.run_chain(impl, case, nest=nest, iau=iau, tail=tail)
    assert np.array_equal(got["quats"], want["quats"]), "quats differ from the reference"
    nbad = np.count_nonzero(got["pixels"] != want["pixels"])
    assert nbad == 0, "%d pixel indices differ from the reference" % nbad
    assert np.array_equal(got["hsub"], want["hsub"])
    assert np.array_equal(got["g2l"], want["g2l"])
    np.testing.assert_allclose(got["weights"], want["weights"], rtol=weights_rtol, atol=1e-15)
    zs = np.max(np.abs(want["zmap"]))
    assert np.max(np.abs(got["zmap"] - want["zmap"])) <= ztol * zs
    ts = np.max(np.abs(want["tod"]))
    assert np.max(np.abs(got["tod"] - want["tod"])) <= 10 * ztol * ts
    return got, want


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))
