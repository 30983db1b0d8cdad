"""NumPy restatement of toast.ops.GroundFilter and of the four libtoast kernels it calls
(TEST INFRASTRUCTURE ONLY: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg are
the only importers).

* legendre_templates   src/libtoast/src/toast_tod_filter.cpp:269-331
* bin_proj             :160-177
* bin_invcov           :179-215
* add_templates        :333-355
* build_templates      src/toast/ops/groundfilter.py:208-331
* fit_templates        :334-381
* subtract_templates   :384-393
* apply (per detector) :443-500

Parity pinned: the kernels are run from the reference's own sources (oracle/_ref builds
toast_tod_filter.cpp in place) by tests/golden/make_golden_cov.py; tests/test_oracle_ground_filter.py
checks this restatement against those outputs (Legendre templates and the subtracted fit bit for
bit, the projections / Gram matrices to 1e-13: NumPy sums pairwise, the reference sequentially).
"""

import numpy as np


def legendre_templates(x, start_order, stop_order):
    x = np.asarray(x, dtype=np.float64)
    out = np.zeros((stop_order - start_order, x.size))
    if start_order == 0 and stop_order > 0:
        out[0] = 1.0 / np.sqrt(2.0)  # :273-276
    if start_order <= 1 and stop_order > 1:
        out[1 - start_order] = (1.0 / np.sqrt(2.0 / 3.0)) * x  # :279-283
    val = x.copy()
    prev = np.ones_like(x)
    for order in range(2, stop_order):
        orderinv = 1.0 / order
        nxt = ((2 * order - 1) * x * val - (order - 1) * prev) * orderinv  # :308-313
        prev = val
        val = nxt
        if order >= start_order:
            out[order - start_order] = val * (1.0 / np.sqrt(2.0 / (2.0 * order + 1.0)))  # :320-324
    return out


def bin_proj(signal, templates, good):
    return np.array([np.sum(t * signal * good) for t in templates])  # :168-175


def bin_invcov(templates, good):
    nt = templates.shape[0]
    invcov = np.zeros((nt, nt))
    for r in range(nt):
        for c in range(r, nt):
            invcov[r, c] = invcov[c, r] = np.sum(templates[r] * templates[c] * good)  # :198-209
    return invcov


def add_templates(signal, templates, coeff):
    for t, c in zip(templates, coeff):
        signal += c * t  # :343-349, template after template


def split_templates(templates, lr_mask, rl_mask):
    """groundfilter.py:208-224: per template one copy without the left-right samples and one
    without the right-left samples."""
    out = []
    for t in templates:
        for mask in (lr_mask, rl_mask):
            s = t.copy()
            s[mask] = 0
            out.append(s)
    return np.vstack(out)


def azimuth_phase(az):
    """groundfilter.py:296-312 (modifies a copy of az when it wraps)."""
    az = np.array(az, dtype=np.float64)
    azmin, azmax = np.amin(az), np.amax(az)
    while azmin < 0:
        azmin += 2 * np.pi
        azmax += 2 * np.pi
    if azmax - azmin > 2 * np.pi:
        azmin, azmax = 0, 2 * np.pi
        az %= 2 * np.pi
    return (az - azmin) / (azmax - azmin) * 2 - 1, az


def bin_templates(az, bin_width, drop_most_hit):
    """groundfilter.py:235-257.  With a polynomial filter the most-hit bin is dropped to break the
    degeneracy (the reference refers to `counts` without computing them, :246 -- the evident intent,
    np.unique(..., return_counts=True), is what is restated here)."""
    ibin = (az // bin_width).astype(int)
    bins, counts = np.unique(ibin, return_counts=True)
    if drop_most_hit:
        keep = np.ones(len(counts), dtype=bool)
        keep[np.argmax(counts)] = False
        bins = bins[keep]
    return [(ibin == b).astype(float) for b in bins]


def build_templates(n_samp, az, trend_order, filter_order, bin_width=None, split=False, lr_mask=None, rl_mask=None):
    x = np.arange(n_samp) / n_samp * 2 - 1  # :267
    blocks = []
    if trend_order is not None:
        blocks.append(legendre_templates(x, 1, trend_order + 1))  # :275-277
    phase, az = azimuth_phase(az)
    if filter_order is not None:
        lt = legendre_templates(phase, 0, filter_order + 1)
        if split:
            lt = split_templates(lt, lr_mask, rl_mask)
        blocks.append(lt)
    if bin_width is not None:
        bt = bin_templates(az, bin_width, filter_order is not None)
        if split:
            bt = split_templates(bt, lr_mask, rl_mask)
        blocks.append(np.vstack(bt))
    return np.vstack(blocks)


def fit_templates(templates, ref, good):
    """groundfilter.py:334-381: returns (coeff, rcond); None when every sample is flagged."""
    if np.sum(good) == 0:
        return None, None
    g = good.astype(np.uint8)
    proj = bin_proj(ref.astype(np.float64), templates, g)
    invcov = bin_invcov(templates, g)
    rcond = 1 / np.linalg.cond(invcov)
    if rcond > 1e-6:
        cov = np.linalg.inv(invcov)
    else:
        cov = np.linalg.pinv(invcov, rcond=1e-12, hermitian=True)
    return np.dot(cov, proj), rcond


def apply(signal, det_flags, det_flag_mask, shared_flags, shared_flag_mask, templates, trend_order, detrend):
    """Filter every row of ``signal`` in place (groundfilter.py:443-500).  Returns the list of rows
    that could not be fitted (all samples flagged)."""
    failed = []
    common = (shared_flags & shared_flag_mask) if shared_flags is not None else np.zeros(signal.shape[1], np.uint8)
    offset = 0 if detrend else (trend_order or 0)
    for d in range(signal.shape[0]):
        good = common == 0
        if det_flags is not None:
            good = np.logical_and(good, (det_flags[d] & det_flag_mask) == 0)
        coeff, _ = fit_templates(templates, signal[d], good)
        if coeff is None:
            failed.append(d)
            continue
        fit = np.zeros(signal.shape[1])
        add_templates(fit, templates[offset:], coeff[offset:])
        signal[d] -= fit
    return failed
