"""GPU: the destriping ``ops.MapMaker`` END TO END against tests/golden/mapmaker_e2e.npz -- amplitudes, residual history,
hits and maps produced in the build container by the reference's own compiled kernels (oracle/_ref) driven in the order of
the reference's operators through the reference's own ``solve()`` (tests/golden/make_golden_mapmaker.py; VERDICT round 5,
item 2).  Inputs are rebuilt from seeds on both sides (tests/mapmaker_case.py).

Reference bars: src/toast/tests/ops_mapmaker.py (final products of the complete operator), ops_mapmaker_solve.py:151-265.

* ``TOAST_HIP_DETERMINISTIC`` mode (the operator sequence with the order-exact scatter, deterministic.hip): residual
  history to 1e-12, amplitudes and maps below 1e-10 of the fixture, hit counts exact;
* default mode, all three routes of the solver's left-hand side -- sweeps over the packed pointing cache, the fused
  sweeps over the cached pixels / weights, pointing on the fly (``full_pointing=False``) --: within 10 x the run-to-run
  floor of the atomic scatter measured here (printed) or 1e-11, whichever is larger (the sums are taken in another ORDER
  than the reference's, which two runs of one route do not show: 1.3e-12 after ten iterations on the first box), and
  always below the north star's 1e-10.
"""
import os

import numpy as np
import pytest

import mapmaker_case

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "mapmaker_e2e.npz")


def _run(case, full_pointing=True, packed=True, deterministic=False, share=True, fused_final=True, cov_rhs=True):
    from toast_amd import capi, ops
    from toast_amd.data import defaults
    from toast_amd.templates import Offset

    data, cfg = mapmaker_case.build(case)
    old = os.environ.get("TOAST_HIP_PACKED_POINTING")
    os.environ["TOAST_HIP_PACKED_POINTING"] = "1" if packed else "0"
    os.environ["TOAST_HIP_SHARE_SOLVER_COV"] = "1" if share else "0"
    os.environ["TOAST_HIP_FUSED_FINAL"] = "1" if fused_final else "0"
    os.environ["TOAST_HIP_FUSED_COV_RHS"] = "1" if cov_rhs else "0"
    was = capi.get_deterministic()
    capi.set_deterministic(deterministic)
    try:
        dp = ops.PointingDetectorSimple()
        pix = ops.PixelsHealpix(detector_pointing=dp, nside=cfg["nside"], nest=True)
        sw = ops.StokesWeights(detector_pointing=dp, mode="IQU", hwp_angle=defaults.hwp_angle)
        binner = ops.BinMap(pixel_dist="pixel_dist", pixel_pointing=pix, stokes_weights=sw, full_pointing=full_pointing)
        tm = ops.TemplateMatrix(templates=[Offset(step_time=cfg["step_time"], noise_model=defaults.noise_model, name="baselines")])
        mm = ops.MapMaker(name="mm", det_data=defaults.det_data, binning=binner, template_matrix=tm, iter_min=cfg["iters"],
                          iter_max=cfg["iters"], convergence=1e-30, keep_solver_products=True, keep_final_products=True,
                          write_binmap=True)
        mm.apply(data)
    finally:
        capi.set_deterministic(was)
        os.environ.pop("TOAST_HIP_SHARE_SOLVER_COV", None)
        os.environ.pop("TOAST_HIP_FUSED_FINAL", None)
        os.environ.pop("TOAST_HIP_FUSED_COV_RHS", None)
        if old is None:
            os.environ.pop("TOAST_HIP_PACKED_POINTING", None)
        else:
            os.environ["TOAST_HIP_PACKED_POINTING"] = old
    dist = data["pixel_dist"]
    ob = data.obs[0]
    sflags, pix = ob.detdata["mm_solve_flags"].data, ob.detdata[defaults.pixels].data if defaults.pixels in ob.detdata else None
    flag_counts = None
    if pix is not None and pix.shape == sflags.shape:
        flag_counts = [int(np.count_nonzero(((sflags & b) != 0) & (pix >= 0))) for b in (1, 4)]
    out = dict(flag_counts=flag_counts, amplitudes=data["mm_solve_amplitudes"]["baselines"].local.copy(), history=np.array(mm.history),
               hits=data["mm_hits"].data.reshape(-1).copy(), map=data["mm_map"].data.reshape(-1, 3).copy(),
               binmap=data["mm_binmap"].data.reshape(-1, 3).copy(),
               noiseweighted=data["mm_noiseweighted_map"].data.reshape(-1, 3).copy(),
               cov=data["mm_cov"].data.reshape(-1, 6).copy(), local_submaps=np.array(dist.local_submaps),
               route=tuple(getattr(mm, "lhs_route", ())), shared=bool(getattr(mm, "shared_solver_covariance", False)),
               fused_final=bool(getattr(mm, "fused_final_binning", False)),
               rhs_with_cov=getattr(mm, "rhs_map_with_covariance", None))
    return out


def _fixture(case):
    z = np.load(FIXTURE)
    return {k[len(case) + 1:]: z[k] for k in z.files if k.startswith(case + "_")}


def _rel(a, b):
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


def _compare(got, want, tol_hist, tol, label):
    sel = want["pix_index"]
    assert np.array_equal(got["local_submaps"], want["local_submaps"]), label
    assert np.array_equal(got["hits"][sel], want["hits"]) and int(got["hits"].sum()) == int(want["hits_total"]), label
    n = len(want["history"])
    assert len(got["history"]) == n, (label, len(got["history"]), n)
    errs = dict(history=float(np.max(np.abs(got["history"] - want["history"]) / want["history"])),
                amplitudes=_rel(got["amplitudes"], want["amplitudes"]), map=_rel(got["map"][sel], want["map"]),
                binmap=_rel(got["binmap"][sel], want["binmap"]), noiseweighted=_rel(got["noiseweighted"][sel], want["noiseweighted"]),
                cov=_rel(got["cov"][sel], want["cov"]))
    # whole-map sums of the destriped map: the pixels the fixture does not list are covered too
    sums = np.array([got["map"][:, k].sum() for k in range(3)] + [np.abs(got["map"]).sum()])
    errs["map_sums"] = float(np.max(np.abs(sums - want["map_sums"])) / want["map_sums"][3])
    print(f"E2E {label}: " + "  ".join(f"{k} {v:.2e}" for k, v in errs.items()))
    assert errs["history"] <= tol_hist, (label, errs)
    for k in ("amplitudes", "map", "binmap", "noiseweighted", "cov", "map_sums"):
        assert errs[k] <= tol, (label, k, errs)
    # flagged amplitudes stay exactly zero, as in the reference
    assert not np.any(got["amplitudes"][want["amp_flags"] != 0]), label
    return errs


@pytest.mark.parametrize("case", ["small", "cfg3cut"])
def test_mapmaker_deterministic_mode_equals_reference_chain(case):
    """The operator sequence with the order-exact scatter: every sum is taken in the reference's order, so the only
    differences left are the 3 x 3 inversions (device code against numpy.linalg.eigh in the generator)."""
    want = _fixture(case)
    got = _run(case, deterministic=True)
    _compare(got, want, 1e-12, 1e-10, f"{case} deterministic")
    # the solver's flags -- bit 1 the binning's own cut, bit 4 the samples in poorly conditioned pixels (ScanMask of the
    # rcond mask) -- are the reference's, sample for sample in number
    assert got["flag_counts"] == [int(x) for x in want["solver_flag_counts_with_pixel"]], (got["flag_counts"], want["solver_flag_counts_with_pixel"])
    # the final hits / covariance / rcond were the solver's own arrays (same samples, same cut: accumulated once) ...
    assert got["shared"]
    # the right-hand side's noise-weighted map was asked of the covariance pass; in this mode the library runs the separate
    # order-exact sweeps behind that call (False per observation) -- and the results are the ones of the reference's order
    assert got["rhs_with_cov"] == (False,), got["rhs_with_cov"]
    if case == "small":
        # ... and accumulating them a second time, as the reference does, gives the same products bit for bit
        twice = _run(case, deterministic=True, share=False)
        assert not twice["shared"]
        _compare(twice, want, 1e-12, 1e-10, f"{case} deterministic, covariance accumulated twice")
        for k in ("hits", "cov", "map", "binmap", "noiseweighted", "amplitudes"):
            assert np.array_equal(twice[k], got[k]), k


@pytest.mark.parametrize("case", ["small", "cfg3cut"])
def test_mapmaker_default_routes_within_the_scatter_floor(case):
    """Default mode: hardware atomics, order not fixed; ten CG iterations amplify that rounding.  The yardstick is what
    the SAME route shows from one run to the next; every route stays within 10 x that floor of the reference's result."""
    want = _fixture(case)
    a = _run(case)                      # packed route (the default when the cache fits its form), else fused
    b = _run(case)
    floor = max(_rel(a["amplitudes"], b["amplitudes"]), _rel(a["map"], b["map"]),
                float(np.max(np.abs(a["history"] - b["history"]) / b["history"])), 1e-13)
    print(f"E2E {case}: run-to-run floor of the default route {floor:.2e} (route {a['route']})")
    assert floor < 1e-10, floor
    tol = min(max(10.0 * floor, 1e-11), 1e-10)
    routes = {}
    routes["default " + "/".join(a["route"])] = a
    routes["fused (TOAST_HIP_PACKED_POINTING=0)"] = _run(case, packed=False)
    routes["full_pointing=False"] = _run(case, full_pointing=False)
    # the last two steps (subtract the templates, bin) as one sweep -- the default with cached pointing -- and as the
    # reference's two operators
    # (cached pointing: k_offset_accumulate_v2<E, true>; full_pointing=False: k_otf_accumulate<.., SIG = 2, ..>)
    assert a["fused_final"] and routes["full_pointing=False"]["fused_final"]
    # the right-hand side's A^T N^-1 d in the covariance pass's sweep (the default with cached pointing:
    # k_build_cov_pair_v2<true, true>) and as the reference's separate operator; on the fly the right-hand side bins itself
    # (one kernel with the detector-pair, two-samples-per-lane form; the separate sweeps behind the same call otherwise)
    one_kernel = os.environ.get("TOAST_HIP_PAIR", "1") != "0" and os.environ.get("TOAST_HIP_VEC2", "1") != "0"
    assert a["rhs_with_cov"] == (one_kernel,) and routes["full_pointing=False"]["rhs_with_cov"] is None
    routes["right-hand side binned on its own (TOAST_HIP_FUSED_COV_RHS=0)"] = _run(case, cov_rhs=False)
    assert routes["right-hand side binned on its own (TOAST_HIP_FUSED_COV_RHS=0)"]["rhs_with_cov"] is None
    routes["two-operator final binning (TOAST_HIP_FUSED_FINAL=0)"] = _run(case, fused_final=False)
    routes["full_pointing=False, two-operator final binning"] = _run(case, full_pointing=False, fused_final=False)
    for key in ("two-operator final binning (TOAST_HIP_FUSED_FINAL=0)", "full_pointing=False, two-operator final binning"):
        assert not routes[key]["fused_final"]
    seen = set()
    for label, got in routes.items():
        seen.add(got["route"])
        _compare(got, want, tol, tol, f"{case} {label}")
    assert len(seen) >= 2, seen          # the routes really were different code paths
