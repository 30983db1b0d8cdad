#!/usr/bin/env python3
"""Generate tests/golden/mapmaker_e2e.npz: the destriping map-maker END TO END, produced by the reference's own pieces.

VERDICT round 5, item 2: every kernel of the path is pinned, ``solve()`` is pinned on dense systems and the binned map is
pinned -- this fixture pins the operator CHAIN (solver flags, solver covariance, condition-number cut, right-hand side,
PCG over the real left-hand side, amplitude subtraction, final covariance and binning, in the reference's order with the
reference's flags).  Nothing here is the repository's own arithmetic:

* every per-sample step is a call of the reference's compiled kernel in ``oracle/_ref`` (built in place from
  /root/reference by oracle/ref_build.sh): ``pointing_detector``, ``pixels_healpix``, ``stokes_weights_IQU``,
  ``cov_accum_diag_hits``, ``cov_accum_diag_invnpp``, ``build_noise_weighted``, ``cov_apply_diag``,
  ``ops_scan_map_float64``, ``noise_weight``, ``template_offset_add_to_signal`` / ``_project_signal`` /
  ``_apply_diag_precond``;
* the conjugate-gradient loop is the reference's ``solve()`` (src/toast/ops/mapmaker_solve.py:524-755), compiled from its
  syntax tree where it lies (as tests/golden/make_golden_pcg.py does), driving a left-hand side object whose ``apply`` runs
  the kernel sequence of ``SolverLHS._exec`` (:342-506);
* the order of the steps is that of ``MapMaker._exec`` (src/toast/ops/mapmaker.py:719-787), ``SolveAmplitudes._exec``
  (src/toast/ops/mapmaker_templates.py:1082-1125: flags :696-810, covariance :843-893, rcond mask :895-940, RHS :942-991
  with ``SolverRHS._exec`` mapmaker_solve.py:103-231), ``Offset._initialize`` (src/toast/templates/offset/offset.py:
  250-330: amplitude variances and flags), ``ApplyAmplitudes`` (:1205-1261) and the final binning (mapmaker.py:438-608);
* ONE step has no compiled reference here: ``cov_eigendecompose_diag`` (src/libtoast/src/toast_map_cov.cpp:246-396) needs
  LAPACK, which this image lacks (oracle/ref_build.sh builds the reference's no-LAPACK configuration).  It is restated
  below with ``numpy.linalg.eigh`` following those lines (lower triangle, rcond = emin / emax, V diag(1/e) V^T, zero below
  the threshold), like every other use of that function in this repository.

Inputs are rebuilt from seeds by tests/mapmaker_case.py on both sides; the fixture stores only outputs: amplitudes, their
flags and variances, the sequence of dot products of the solve (= the residual history), hit counts, and the binned /
destriped maps (the large case: every 8th hit pixel + whole-map sums).

Build container only.      python tests/golden/make_golden_mapmaker.py
"""
import ast
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))

import mapmaker_case  # noqa: E402
import oracle  # noqa: E402
from toast_amd import synth  # noqa: E402
from toast_amd.data import defaults  # noqa: E402

REF_SOLVE = "/root/reference/src/toast/ops/mapmaker_solve.py"
ref = oracle.load_ref()
assert ref is not None, "oracle/_ref missing: run oracle/ref_build.sh where /root/reference exists"


# ------------------------------------------------------------------ amplitude containers (what solve() touches)
class Amp:
    """src/toast/templates/amplitudes.py: local values, local flags, dot over the unflagged entries (:550-558)."""

    def __init__(self, n, flags):
        self.local = np.zeros(n)
        self.local_flags = flags
        self.n_global = n
        self.n_local = n

    def duplicate(self):
        out = Amp(self.n_local, self.local_flags)
        out.local[:] = self.local
        return out

    def reset(self):
        self.local[:] = 0.0

    def clear(self):
        pass

    def dot(self, other):
        return float(np.dot(np.where(self.local_flags == 0, self.local, 0), np.where(other.local_flags == 0, other.local, 0)))


class AmpMap(dict):
    dots = None

    def duplicate(self):
        out = AmpMap()
        for k, v in self.items():
            out[k] = v.duplicate()
        return out

    def reset(self):
        for v in self.values():
            v.reset()

    def clear(self):
        super().clear()

    def __isub__(self, other):
        for k, v in self.items():
            v.local[:] -= other[k].local
        return self

    def __iadd__(self, other):
        for k, v in self.items():
            v.local[:] += other[k].local
        return self

    def __imul__(self, scalar):
        for v in self.values():
            v.local[:] *= scalar
        return self

    def dot(self, other):
        val = 0.0
        for k, v in self.items():
            val += v.dot(other[k])
        AmpMap.dots.append(val)
        return val


class Data(dict):
    class _Comm:
        comm_world = None
        world_rank = 0

    comm = _Comm()


class _Quiet:
    @staticmethod
    def get():
        return _Quiet()

    def __getattr__(self, name):
        return lambda *a, **k: None


def load_reference_solve():
    tree = ast.parse(open(REF_SOLVE).read(), REF_SOLVE)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "solve"]
    assert len(fn) == 1
    fn[0].decorator_list = []
    mod = ast.Module(body=fn, type_ignores=[])
    ast.fix_missing_locations(mod)
    ns = {"np": np, "Logger": _Quiet, "Timer": _Quiet, "AmplitudesMap": AmpMap}
    exec(compile(mod, REF_SOLVE, "exec"), ns)
    return ns["solve"]


# ------------------------------------------------------------------ the one LAPACK step
def cov_eigendecompose_diag(nsub, nps, nnz, data, cond, threshold):
    """toast_map_cov.cpp:246-396 with invert = true, nnz > 1 (numpy.linalg.eigh in place of LAPACK dsyev)."""
    block = nnz * (nnz + 1) // 2
    d = data.reshape(nsub * nps, block)
    full = np.zeros((nsub * nps, nnz, nnz))
    off = 0
    for k in range(nnz):               # :297-311 row k holds elements (k, k..nnz-1); the LOWER triangle is what dsyev reads
        for m in range(k, nnz):        # of the column-major matrix, i.e. these entries
            full[:, k, m] = d[:, off]
            full[:, m, k] = d[:, off]
            off += 1
    evals, evecs = np.linalg.eigh(full)
    emin, emax = evals.min(axis=1), evals.max(axis=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        rc = np.where(emax > 0.0, emin / emax, 0.0)             # :342
        inv = np.einsum("bik,bk,bjk->bij", evecs, 1.0 / evals, evecs)     # :345-353 (V / e) V^T
    ok = rc >= threshold                                        # :360-361
    off = 0
    for k in range(nnz):
        for m in range(k, nnz):
            d[:, off] = np.where(ok, inv[:, k, m], 0.0)
            off += 1
    cond.reshape(-1)[:] = np.where(ok, rc, 0.0)                 # :389-391


def global_pixel_to_submap(pix, g2l, nps):
    """src/toast/pixels.py:300-330."""
    good = pix >= 0
    sm = np.where(good, g2l[np.where(good, pix // nps, 0)], -1).astype(np.int64)
    lp = np.where(good, pix % nps, -1).astype(np.int64)
    return sm, lp


def run_case(name, solve):
    t_start = time.time()
    data, cfg = mapmaker_case.build(name)
    ob = data.obs[0]
    n_det, n_samp, rate, nside = cfg["n_det"], cfg["n_samp"], cfg["rate"], cfg["nside"]
    dets = list(ob.local_detectors)
    idx = np.arange(n_det, dtype=np.int32)
    fpl = ob.telescope.focalplane
    fp = np.ascontiguousarray(np.array([fpl[d]["quat"] for d in dets]))
    gamma = np.array([float(fpl[d]["gamma"]) for d in dets])
    eps = np.array([float(fpl[d]["pol_leakage"]) for d in dets])
    cal = np.array([float(fpl[d]["cal"]) for d in dets])
    bore = ob.shared[defaults.boresight_radec].data
    sflags = ob.shared[defaults.shared_flags].data
    hwp = np.ascontiguousarray(ob.shared[defaults.hwp_angle].data)
    dflags = ob.detdata[defaults.det_flags].data
    signal = ob.detdata[defaults.det_data].data
    ivl = ob.intervals[None].data
    detw = np.array([float(ob[defaults.noise_model].detector_weight(d)) for d in dets])
    nps = 3072 if nside >= 16 else 12 * nside * nside
    n_submap = 12 * nside * nside // nps
    nnz = 3

    # -- pointing expansion (PointingDetectorSimple, PixelsHealpix, StokesWeights with their default masks)
    quats = np.zeros((n_det, n_samp, 4))
    ref.pointing_detector(fp, bore, idx, quats, ivl, sflags, defaults.shared_mask_invalid, False)
    pixels = np.zeros((n_det, n_samp), dtype=np.int64)
    hsub = np.zeros(n_submap, dtype=np.uint8)
    ref.pixels_healpix(idx, quats, sflags, defaults.shared_mask_invalid, idx, pixels, ivl, hsub, nps, nside, True, False)
    weights = np.zeros((n_det, n_samp, nnz))
    ref.stokes_weights_IQU(idx, quats, idx, weights, hwp, ivl, eps, gamma, cal, False, False)
    del quats
    g2l, hit = synth.global_to_local(hsub)
    n_local = int(hit.size)

    def accumulate_cov(flags, flag_mask, shared_mask, threshold):
        """CovarianceAndHits: BuildHitMap, BuildInverseCovariance (mapmaker_utils.py:178-204, 464-513), covariance_invert."""
        hits = np.zeros(n_local * nps, dtype=np.int64)
        invcov = np.zeros(n_local * nps * 6)
        for d in range(n_det):
            sm, lp = global_pixel_to_submap(pixels[d], g2l, nps)
            lp[(flags[d] & flag_mask) != 0] = -1
            lp[(sflags & shared_mask) != 0] = -1
            ref.cov_accum_diag_hits(n_local, nps, 1, sm, lp, hits, False)
            ref.cov_accum_diag_invnpp(n_local, nps, nnz, sm, lp, np.ascontiguousarray(weights[d]).reshape(-1), float(detw[d]),
                                      invcov, False)
        cov = invcov.copy()
        rcond = np.zeros(n_local * nps)
        cov_eigendecompose_diag(n_local, nps, nnz, cov, rcond, threshold)
        return hits, invcov, cov, rcond

    # -- SolveAmplitudes: solver flags (bit 1), solver covariance, rcond cut (bit 4)
    sflag1 = ((sflags & defaults.shared_mask_nonscience) > 0).astype(np.uint8)
    solver_flags = np.empty((n_det, n_samp), dtype=np.uint8)
    for d in range(n_det):
        solver_flags[d] = sflag1
        solver_flags[d] |= ((dflags[d] & defaults.det_mask_nonscience) > 0).astype(np.uint8)
    s_hits, _, s_cov, s_rcond = accumulate_cov(solver_flags, 255, defaults.shared_mask_nonscience, 1.0e-8)
    rcond_mask = (s_rcond < 1.0e-8).astype(np.uint8)
    for d in range(n_det):
        sm, lp = global_pixel_to_submap(pixels[d], g2l, nps)
        masked = (rcond_mask.reshape(n_local, nps)[sm, lp] & 255) != 0        # (ScanMask: negative indices wrap, as there)
        solver_flags[d][masked] |= 4

    # -- Offset template: amplitude layout, variances, flags (offset.py:250-330; view None, good_fraction 0.5)
    step = int(np.rint(cfg["step_time"] * rate))
    n_amp_views = np.array([(int(v["last"] - v["first"]) + step - 1) // step for v in ivl], dtype=np.int64)
    per_det = int(n_amp_views.sum())
    n_amp = n_det * per_det
    amp_flags = np.zeros(n_amp, dtype=np.uint8)
    offset_var = np.zeros(n_amp)
    off = 0
    for d in range(n_det):
        for ivw, vw in enumerate(ivl):
            first, last = int(vw["first"]), int(vw["last"])
            fl = (solver_flags[d, first:last] & 255).astype(np.uint8)
            voff = 0
            for amp in range(int(n_amp_views[ivw])):
                amplen = step if amp < n_amp_views[ivw] - 1 else (last - first) - voff
                n_good = amplen - int(np.count_nonzero(fl[voff:voff + amplen]))
                if (n_good / amplen) <= 0.5:
                    amp_flags[off + amp] = 1
                else:
                    offset_var[off + amp] = 1.0 / (detw[d] * n_good)
                voff += step
            off += int(n_amp_views[ivw])

    def template_add(tod, amps):
        for d in range(n_det):
            ref.template_offset_add_to_signal(step, d * per_det, n_amp_views, amps.local, amps.local_flags, d, tod, ivl, False)

    def template_project(tod, amps):
        for d in range(n_det):
            ref.template_offset_project_signal(d, tod, d, solver_flags, 255, step, d * per_det, n_amp_views, amps.local,
                                               amps.local_flags, ivl, False)

    def bin_map(tod, flags, flag_mask, shared_mask, cov, keep_noiseweighted=False):
        z = np.zeros((n_local, nps, nnz))
        ref.build_noise_weighted(g2l, z, idx, pixels, idx, weights, idx, tod, idx, flags, detw, flag_mask, ivl, sflags, shared_mask,
                                 False)
        nw = z.copy() if keep_noiseweighted else None
        ref.cov_apply_diag(n_local, nps, nnz, cov, z.reshape(-1))
        return z, nw

    def scan_subtract_weight(binned, tod):
        ref.ops_scan_map_float64(g2l, nps, binned, tod, idx, pixels, idx, weights, idx, ivl, 1.0, False, True, False, False)
        ref.noise_weight(tod, idx, ivl, detw, False)

    # -- right-hand side (SolverRHS._exec): bin, copy, scan - subtract, noise weight, project
    binned, _ = bin_map(signal, solver_flags, 255, 0, s_cov)
    temp = signal.copy()
    scan_subtract_weight(binned, temp)
    rhs = AmpMap(baselines=Amp(n_amp, amp_flags))
    template_project(temp, rhs["baselines"])

    # -- PCG with the reference's solve(); the left-hand side of SolverLHS._exec
    class TemplateMatrix:
        amplitudes = None

        def apply_precond(self, amps_in, amps_out):
            ref.template_offset_apply_diag_precond(offset_var, amps_in["baselines"].local, amps_in["baselines"].local_flags,
                                                   amps_out["baselines"].local, False)

    class LHS:
        name = "mm_lhs"
        out = None
        template_matrix = TemplateMatrix()
        calls = 0

        def apply(self, d, detectors=None):
            a_in = d[self.template_matrix.amplitudes]["baselines"]
            tmp = np.zeros((n_det, n_samp))
            template_add(tmp, a_in)                                   # binning.pre_process = template_matrix
            b, _ = bin_map(tmp, solver_flags, 255, 0, s_cov)
            d[self.out].reset()                                       # (+ add_prior: nothing without a noise prior)
            tmp[:] = 0.0
            template_add(tmp, a_in)
            scan_subtract_weight(b, tmp)
            template_project(tmp, d[self.out]["baselines"])
            LHS.calls += 1

    store = Data()
    store["rhs"] = rhs
    AmpMap.dots = []
    solve(store, None, LHS(), "rhs", "amplitudes", convergence=1.0e-30, n_iter_max=cfg["iters"], n_iter_min=cfg["iters"])
    dots = np.array(AmpMap.dots)
    amps = store["amplitudes"]["baselines"]
    history = dots[3::3] / dots[0]

    # -- MapMaker: final covariance with the binning's own flags, raw binned map, cleaned signal, destriped map
    f_hits, f_invcov, f_cov, f_rcond = accumulate_cov(dflags, defaults.det_mask_nonscience, defaults.shared_mask_nonscience, 1.0e-8)
    binmap, _ = bin_map(signal, dflags, defaults.det_mask_nonscience, defaults.shared_mask_nonscience, f_cov)
    tmpl = np.zeros((n_det, n_samp))
    template_add(tmpl, amps)
    cleaned = signal - tmpl                                           # ApplyAmplitudes(op="subtract") via Combine
    destriped, nw = bin_map(cleaned, dflags, defaults.det_mask_nonscience, defaults.shared_mask_nonscience, f_cov, True)

    out = {"amplitudes": amps.local.copy(), "amp_flags": amp_flags, "offset_var": offset_var, "rhs": rhs["baselines"].local.copy(),
           "dots": dots, "history": history, "lhs_calls": np.array(LHS.calls), "n_local_submap": np.array(n_local),
           "local_submaps": hit.astype(np.int64), "solver_hits_total": np.array(int(s_hits.sum())),
           "solver_flag_counts": np.array([int(np.count_nonzero(solver_flags & b)) for b in (1, 4)]),
           # ... and over the samples that have a pixel (for the others the reference's ScanMask reads mask[-1, -1]: the
           # samples are flagged through bit 1 either way)
           "solver_flag_counts_with_pixel": np.array([int(np.count_nonzero((solver_flags & b) != 0) - np.count_nonzero(((solver_flags & b) != 0) & (pixels < 0))) for b in (1, 4)]),
           "hits_total": np.array(int(f_hits.sum())),
           "map_sums": np.array([destriped[..., k].sum() for k in range(3)] + [np.abs(destriped).sum()]),
           "binmap_sums": np.array([binmap[..., k].sum() for k in range(3)] + [np.abs(binmap).sum()]),
           "noiseweighted_sums": np.array([nw[..., k].sum() for k in range(3)] + [np.abs(nw).sum()]),
           "cleaned_sums": np.array([cleaned.sum(), np.abs(cleaned).sum()])}
    hitpix = np.flatnonzero(f_hits)
    stride = 1 if name == "small" else 8
    sel = hitpix[::stride]
    out["pix_index"] = sel.astype(np.int64)                           # flat index into [n_local_submap * n_pix_submap]
    out["hits"] = f_hits[sel]
    out["solver_hits"] = s_hits[sel]
    out["rcond"] = f_rcond[sel]
    out["map"] = destriped.reshape(-1, 3)[sel]
    out["binmap"] = binmap.reshape(-1, 3)[sel]
    out["noiseweighted"] = nw.reshape(-1, 3)[sel]
    out["cov"] = f_cov.reshape(-1, 6)[sel]
    print(f"{name:8s} {n_det} x {n_samp}, nside {nside}: {n_amp} amplitudes ({int(amp_flags.sum())} flagged), "
          f"{LHS.calls} LHS applications, residual {history[0]:.3e} -> {history[-1]:.3e}, hit pixels {hitpix.size} "
          f"(stored {sel.size}), rcond-cut samples {int(np.count_nonzero(solver_flags & 4))}, {time.time() - t_start:.1f} s")
    return {f"{name}_{k}": v for k, v in out.items()}


def main():
    solve = load_reference_solve()
    blob = {}
    for name in mapmaker_case.CASES:
        blob.update(run_case(name, solve))
    path = os.path.join(HERE, "mapmaker_e2e.npz")
    np.savez_compressed(path, **blob)
    print("mapmaker_e2e.npz: %.2f MB" % (os.path.getsize(path) / 1e6))


if __name__ == "__main__":
    main()
