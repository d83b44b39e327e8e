"""Numpy model of the HIP kernel's algorithm (development aid, NOT product and NOT oracle):
unit-time constant tables + time scaling, per-vertex masked blocks, block-tridiagonal
LDL^T elimination along the vertex chain (optionally 'twisted': two half-chains meeting at a
middle vertex), coefficient recovery.  Vectorised over the batch so it doubles as a fast
CPU cross-check at bench sizes.  See DESIGN.md section 3.
"""
from __future__ import annotations

import math
from fractions import Fraction

import numpy as np

import importlib.util
import os

_spec = importlib.util.spec_from_file_location(
    "gen_tables", os.path.join(os.path.dirname(__file__), "..", "mav_trajectory_generation_amd", "csrc", "gen_tables.py"))
_gt = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_gt)

_cache = {}


def tables(n, d):
    if (n, d) not in _cache:
        ainv, hm, q = _gt.tables(n, d)
        _cache[(n, d)] = (np.array([[float(x) for x in r] for r in ainv]),
                          np.array([[float(x) for x in r] for r in hm]))
    return _cache[(n, d)]


def _ldl_solve(dmat, rhs_list, free):
    """dmat: [B,h,h] SPD on the 'free' index set; rhs_list: list of [B,h,(cols)] arrays.
    Plain LDL^T without pivoting restricted to free indices; returns solutions (zeros elsewhere)."""
    bsz, h, _ = dmat.shape
    l = np.zeros_like(dmat)
    dinv = np.zeros((bsz, h))
    for j in free:
        v = dmat[:, j, j].copy()
        for m in free:
            if m >= j:
                break
            v -= l[:, j, m] ** 2 / dinv[:, m]
        dinv[:, j] = 1.0 / v
        for i in free:
            if i <= j:
                continue
            w = dmat[:, i, j].copy()
            for m in free:
                if m >= j:
                    break
                w -= l[:, i, m] * l[:, j, m] / dinv[:, m]
            l[:, i, j] = w * dinv[:, j]
    outs = []
    for rhs in rhs_list:
        y = rhs.copy()
        for i in free:                       # forward: L y = rhs
            for m in free:
                if m >= i:
                    break
                y[:, i] -= l[:, i, m, None] * y[:, m]
        for i in free:
            y[:, i] *= dinv[:, i, None]
        for i in reversed(free):             # backward: L^T x = y
            for m in free:
                if m <= i:
                    continue
                y[:, i] -= l[:, m, i, None] * y[:, m]
        outs.append(y)
    return outs


def solve_fast(n, d, masks, times, d_fixed, twisted=False):
    """times [B,K], d_fixed [B,D,n_fixed] -> coeffs [B,K,D,N], d_free [B,D,n_free], cost [B]."""
    times = np.asarray(times, dtype=np.float64)
    d_fixed = np.asarray(d_fixed, dtype=np.float64)
    bsz, k = times.shape
    dim = d_fixed.shape[1]
    h = n // 2
    ainv, h1 = tables(n, d)
    # per-vertex value vectors (fixed values, 0 at free slots) and free index lists
    val = np.zeros((bsz, k + 1, h, dim))
    col = 0
    free = []
    for v in range(k + 1):
        fr = []
        for p in range(h):
            if (masks[v] >> p) & 1:
                val[:, v, p, :] = d_fixed[:, :, col]
                col += 1
            else:
                fr.append(p)
        free.append(fr)
    assert col == d_fixed.shape[2]
    # per-segment scaled blocks
    s = np.stack([times ** p for p in range(h)], axis=-1)          # [B,K,h]
    base = times ** (1 - 2 * d)                                      # [B,K]
    ss = base[..., None, None] * s[..., :, None] * s[..., None, :] * h1[:h, :h]
    se = base[..., None, None] * s[..., :, None] * s[..., None, :] * h1[:h, h:]
    ee = base[..., None, None] * s[..., :, None] * s[..., None, :] * h1[h:, h:]
    x = val.copy()                                                   # solution incl. fixed values

    def chain(order_segments, direction):
        """Eliminate along a chain of segments.  direction=+1: segment j goes from vertex j to j+1.
        direction=-1: reversed (vertex j+1 is 'left').  Returns Schur contribution on the last vertex and
        stored (G, g) per eliminated vertex for back-substitution."""
        sc = np.zeros((bsz, h, h))
        rc = np.zeros((bsz, h, dim))
        store = []
        for j in order_segments:
            if direction > 0:
                vl, vr = j, j + 1
                a_ll, a_lr, a_rr = ss[:, j], se[:, j], ee[:, j]
            else:
                vl, vr = j + 1, j
                a_ll, a_lr, a_rr = ee[:, j], np.swapaxes(se[:, j], 1, 2), ss[:, j]
            fl, fr = free[vl], free[vr]
            dv = sc + a_ll
            rv = rc - np.einsum('bpq,bqd->bpd', a_ll, val[:, vl]) - np.einsum('bpq,bqd->bpd', a_lr, val[:, vr])
            u = np.zeros((bsz, h, h))
            for p in fl:
                for q in fr:
                    u[:, p, q] = a_lr[:, p, q]
            if fl:
                gmat, gvec = _ldl_solve(dv, [u, rv], fl)
            else:
                gmat, gvec = np.zeros((bsz, h, h)), np.zeros((bsz, h, dim))
            for p in range(h):
                if p not in fl:
                    gmat[:, p, :] = 0
                    gvec[:, p, :] = 0
            store.append((vl, vr, gmat, gvec))
            # accumulators for the right vertex
            sc = a_rr - np.einsum('bpq,bpr->bqr', u, gmat)
            rc = (-np.einsum('bqp,bqd->bpd', a_lr, val[:, vl])
                  - np.einsum('bqp,bqd->bpd', u, gvec))
            # note: -a_rr val_r term is added when vr is processed as a left vertex (a_ll of the next
            # segment does not contain a_rr) -> add it here explicitly
            rc -= np.einsum('bpq,bqd->bpd', a_rr, val[:, vr])
        return sc, rc, store

    if twisted and k >= 2:
        ka = (k + 1) // 2
        sc_a, rc_a, st_a = chain(range(0, ka), +1)
        sc_b, rc_b, st_b = chain(range(k - 1, ka - 1, -1), -1)
        mid = ka
        dm = sc_a + sc_b
        rm = rc_a + rc_b
    else:
        sc_a, rc_a, st_a = chain(range(0, k), +1)
        st_b = []
        mid = k
        dm, rm = sc_a, rc_a
    fm = free[mid]
    if fm:
        (xm,) = _ldl_solve(dm, [rm], fm)
        for p in fm:
            x[:, mid, p, :] = xm[:, p, :]
    for st in (st_a, st_b):
        for (vl, vr, gmat, gvec) in reversed(st):
            xr = np.zeros((bsz, h, dim))
            for q in free[vr]:
                xr[:, q] = x[:, vr, q]
            xl = gvec - np.einsum('bpq,bqd->bpd', gmat, xr)
            for p in free[vl]:
                x[:, vl, p, :] = xl[:, p, :]
    # recovery
    coeffs = np.zeros((bsz, k, dim, n))
    fact = np.array([math.factorial(j) for j in range(h)], dtype=np.float64)
    tinv = 1.0 / times
    cost = np.zeros(bsz)
    for i in range(k):
        delta = np.concatenate([s[:, i, :, None] * x[:, i], s[:, i, :, None] * x[:, i + 1]], axis=1)  # [B,N,D]
        coeffs[:, i, :, :h] = np.swapaxes(x[:, i] / fact[None, :, None], 1, 2)
        hi = np.einsum('jk,bkd->bjd', ainv[h:, :], delta)                                        # [B,h,D]
        tp = np.stack([tinv[:, i] ** j for j in range(h, n)], axis=-1)                           # [B,h]
        coeffs[:, i, :, h:] = np.swapaxes(hi * tp[:, :, None], 1, 2)
        cost += 0.5 * base[:, i] * np.einsum('bkd,kl,bld->b', delta, h1, delta)
    # d_free in reference order
    cols = []
    for v in range(k + 1):
        for p in free[v]:
            cols.append(x[:, v, p, :])
    d_free = np.stack(cols, axis=-1) if cols else np.zeros((bsz, dim, 0))
    return coeffs, d_free, cost


def poly_relerr(c, cref):
    """max over polynomials of ||c - cref||_inf / ||cref||_inf (SURVEY.md section 8d parity metric)."""
    num = np.abs(c - cref).max(axis=-1)
    den = np.abs(cref).max(axis=-1)
    den = np.where(den == 0, 1.0, den)
    return float((num / den).max())
