"""Shared helpers for the tests (may import oracle/: tests are the only place allowed to)."""
import ctypes

import numpy as np

from oracle import oracle_np as onp


def masks_ends_full(n, k, interior=1):
    h = n // 2
    return [(1 << h) - 1] + [interior] * (k - 1) + [(1 << h) - 1]


def n_fixed(masks):
    return sum(bin(m).count("1") for m in masks)


def reference_batch(bsz, k, n, dim, seed0, masks=None, box=10.0, v_max=3.0, a_max=5.0):
    """Bit-exact reference generators (vertex.cpp:27-82, :255-272), trajectory b uses seed0 + b.
    Fixed slots beyond the generator's (interior velocity etc.) get seeded uniform values."""
    h = n // 2
    if masks is None:
        masks = masks_ends_full(n, k)
    nf = n_fixed(masks)
    times = np.zeros((bsz, k))
    d_fixed = np.zeros((bsz, dim, nf))
    rng = np.random.default_rng(seed0)
    for b in range(bsz):
        vs = onp.create_random_vertices(max(h - 1, 1), k, [-box] * dim, [box] * dim, seed0 + b)
        times[b] = onp.estimate_segment_times(vs, v_max, a_max)
        col = 0
        for vi in range(k + 1):
            for p in range(h):
                if (masks[vi] >> p) & 1:
                    c = vs[vi].get_constraint(p)
                    if c is None:
                        c = rng.uniform(-1.0, 1.0, dim)
                    d_fixed[b, :, col] = c
                    col += 1
    return masks, times, d_fixed


def poly_relerr(c, cref):
    """max over polynomials of ||c - cref||_inf / ||cref||_inf  (SURVEY.md 8(d) parity metric)."""
    num = np.abs(c - cref).max(axis=-1)
    den = np.abs(cref).max(axis=-1)
    den = np.where(den == 0, 1.0, den)
    return float((num / den).max())


def emu_run(lib, n, dim, k, deriv, masks, times, d_fixed, mode=0, want_cost=True, d_free_in=None):
    dp = ctypes.POINTER(ctypes.c_double)
    ip = ctypes.POINTER(ctypes.c_int)
    bsz = times.shape[0]
    h = n // 2
    nfree = (k + 1) * h - n_fixed(masks)
    co = np.zeros((bsz, k, dim, n))
    dfr = np.zeros((bsz, dim, max(nfree, 1))) if d_free_in is None else np.ascontiguousarray(d_free_in, dtype=np.float64)
    cost = np.zeros(bsz)
    st = ctypes.c_int(0)
    m = np.array(masks, dtype=np.int32)
    times = np.ascontiguousarray(times)
    d_fixed = np.ascontiguousarray(d_fixed)
    rc = lib.mtg_emu_run(n, dim, k, deriv, m.ctypes.data_as(ip), bsz, times.ctypes.data_as(dp),
                         d_fixed.ctypes.data_as(dp), co.ctypes.data_as(dp), dfr.ctypes.data_as(dp),
                         cost.ctypes.data_as(dp) if want_cost else None, mode, ctypes.byref(st))
    return rc, co, dfr[:, :, :nfree], cost, st.value


def evaluate(coeffs, t, derivative):
    """Polynomial::evaluate (polynomial.h:118-149) for coeffs [..., N] at scalar/array t."""
    n = coeffs.shape[-1]
    out = np.zeros(coeffs.shape[:-1])
    for j in range(derivative, n):
        f = 1.0
        for i in range(derivative):
            f *= (j - i)
        out = out + f * coeffs[..., j] * np.asarray(t) ** (j - derivative)
    return out


def check_path(masks, times, d_fixed, coeffs, tol=1e-6):
    """checkPath (test_polynomial_optimization.cpp:113-174): fixed constraints met at both segment ends,
    C^0..C^(h-1) continuity at interior vertices.  Returns the max violation (scaled like the reference: abs)."""
    bsz, k, dim, n = coeffs.shape
    h = n // 2
    worst = 0.0
    col = 0
    vals = {}
    for v in range(k + 1):
        for p in range(h):
            if (masks[v] >> p) & 1:
                vals[(v, p)] = d_fixed[:, :, col]
                col += 1
    for i in range(k):
        for p in range(h):
            if (i, p) in vals:
                worst = max(worst, np.abs(evaluate(coeffs[:, i], 0.0, p) - vals[(i, p)]).max())
            if (i + 1, p) in vals:
                worst = max(worst, np.abs(evaluate(coeffs[:, i], times[:, i, None], p) - vals[(i + 1, p)]).max())
            if i > 0:
                a = evaluate(coeffs[:, i - 1], times[:, i - 1, None], p)
                b = evaluate(coeffs[:, i], 0.0, p)
                worst = max(worst, np.abs(a - b).max())
    return worst


def assert_extrema_close(ref_per, got_per, derivative, ctx=None):
    """Per-segment (t_min, v_min, t_max, v_max) tables, reference-restatement vs ours.

    Maxima of velocity / acceleration (the quantities scaleSegmentTimesToMeetConstraints consumes) sit at simple
    roots: 1e-9 relative, both sides.  Minima and position extrema often sit at the trajectory ends, where the zero
    end derivatives give the magnitude derivative a root of multiplicity >= 4: the reference's Jenkins-Traub scatters
    such a cluster by ~eps^(1/m) (observed +-8e-2 s) and evaluates its candidates THERE, so its value is only good
    to ~1e-8 (e.g. a 1-D position polynomial crossing zero inside the segment: neither side looks at the zero
    crossing, segment.cpp:124-131, and the reference's "minimum" is |p| at a scattered pseudo-root).  There the
    maxima are checked one-sided tight (ours is at least as large) and everything else to 1e-6."""
    scale = np.abs(ref_per[:, 3]).max()
    tight, loose = 1e-9 * scale, 1e-6 * scale
    d_max = got_per[:, 3] - ref_per[:, 3]
    d_min = got_per[:, 1] - ref_per[:, 1]
    if derivative in (1, 2):
        assert np.max(np.abs(d_max)) <= tight, (ctx, d_max)
    assert np.all(d_max >= -tight) and np.all(d_max <= loose), (ctx, d_max)
    assert np.all(np.abs(d_min) <= loose), (ctx, d_min)
