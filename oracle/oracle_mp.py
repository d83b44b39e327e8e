"""GROUND-TRUTH ORACLE (test infrastructure, NOT product code): the same linear-algebra
problem as PolynomialOptimization<N>::solveLinear() (LIN:339-379) solved with mpmath at
50 significant digits.  It does NOT mimic the reference's floating-point evaluation order;
it exists to arbitrate between two float64 implementations (the literal restatement in
oracle_np.py standing in for Eigen, and the HIP kernels) because float64 evaluation of the
reference's own formulas is only ~1e-11 (N=10) .. ~4e-9 (N=12) norm-wise accurate.

Problem solved (LIN:308-379, notation of [1] Richter/Bry/Roy ISRR'13):
    minimise  sum_seg  c^T Q(T) c      s.t.  A(T) c = M [d_F; d_P]
    => R = M^T blkdiag(A^-T Q A^-1) M ;  R_PP d_P = -R_PF d_F ;  c = A^-1 M [d_F; d_P]

Only tests/ and offline fixture generation may import this.
"""
from __future__ import annotations

from typing import Sequence

import mpmath as mp
import numpy as np

mp.mp.dps = 50


def _base(n: int, i: int) -> int:
    """POLYC:145-160: i*(i-1)*...*(i-n+1) (0 for i<n)."""
    out = 1
    for k in range(n):
        out *= (i - k)
    return out if i >= n else 0


def mapping_matrix(n: int, t) -> mp.matrix:
    """LIN:112-121."""
    h = n // 2
    a = mp.zeros(n, n)
    t = mp.mpf(t)
    for k in range(h):
        a[k, k] = _base(k, k)
        for j in range(k, n):
            a[h + k, j] = _base(k, j) * t ** (j - k)
    return a


def cost_matrix(n: int, d: int, t) -> mp.matrix:
    """LIN:568-583."""
    q = mp.zeros(n, n)
    t = mp.mpf(t)
    for r in range(d, n):
        for c in range(d, n):
            e = r + c - 2 * d + 1
            q[r, c] = mp.mpf(_base(d, r) * _base(d, c)) * t ** e * 2 / e
    return q


def solve(n_coeffs: int, derivative: int, fixed_mask: Sequence[int], times, d_fixed):
    """One trajectory.  times [K] (float64 values taken exactly), d_fixed [D][n_fixed]
    ordered by (vertex, derivative) over fixed slots (LINH:288-295).
    Returns coeffs [K][D][N], d_free [D][n_free], cost as float64 arrays (rounded from mp)."""
    n = n_coeffs
    h = n // 2
    k = len(times)
    d_fixed = np.asarray(d_fixed, dtype=np.float64)
    dim = d_fixed.shape[0]
    fixed_keys = [(v, p) for v in range(k + 1) for p in range(h) if (fixed_mask[v] >> p) & 1]
    free_keys = [(v, p) for v in range(k + 1) for p in range(h) if not (fixed_mask[v] >> p) & 1]
    nf, npf = len(fixed_keys), len(free_keys)
    col_of = {key: i for i, key in enumerate(fixed_keys)}
    col_of.update({key: nf + i for i, key in enumerate(free_keys)})
    # M (LIN:182-260): row i*N + p -> (vertex i, p); row i*N + h + p -> (vertex i+1, p)
    m = mp.zeros(n * k, nf + npf)
    for i in range(k):
        for p in range(h):
            m[i * n + p, col_of[(i, p)]] = 1
            m[i * n + h + p, col_of[(i + 1, p)]] = 1
    ainvs, qs = [], []
    big = mp.zeros(n * k, n * k)
    for i in range(k):
        a = mapping_matrix(n, float(times[i]))
        ai = a ** -1
        q = cost_matrix(n, derivative, float(times[i]))
        hm = ai.T * q * ai
        ainvs.append(ai)
        qs.append(q)
        for r in range(n):
            for c in range(n):
                big[i * n + r, i * n + c] = hm[r, c]
    r_full = m.T * big * m
    coeffs = np.zeros((k, dim, n))
    d_free = np.zeros((dim, npf))
    cost = mp.mpf(0)
    if npf:
        rpf = r_full[nf:, :nf]
        rpp = r_full[nf:, nf:]
    for d in range(dim):
        df = mp.matrix([mp.mpf(float(x)) for x in d_fixed[d]])
        if npf:
            dp = mp.lu_solve(rpp, -(rpf * df))
            d_all = mp.matrix(list(df) + list(dp))
            d_free[d] = [float(x) for x in dp]
        else:
            d_all = df
        for i in range(k):
            new_d = m[i * n:(i + 1) * n, :] * d_all
            c = ainvs[i] * new_d
            coeffs[i, d] = [float(x) for x in c]
            cost += (c.T * qs[i] * c)[0, 0]
    return coeffs, d_free, float(cost / 2)


def solve_batch(n_coeffs, derivative, fixed_mask, times, d_fixed):
    """times [B][K], d_fixed [B][D][n_fixed] -> coeffs [B][K][D][N], d_free, cost."""
    times = np.asarray(times, dtype=np.float64)
    d_fixed = np.asarray(d_fixed, dtype=np.float64)
    outs = [solve(n_coeffs, derivative, fixed_mask, times[b], d_fixed[b])
            for b in range(times.shape[0])]
    return (np.stack([o[0] for o in outs]), np.stack([o[1] for o in outs]),
            np.array([o[2] for o in outs]))
