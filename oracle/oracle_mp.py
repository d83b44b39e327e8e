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


def _banded_ldl(a, n):
    """LDL^T of a symmetric positive definite banded mp.matrix without pivoting (exact enough at 50 digits: the pivots of an SPD
    matrix are bounded below by its smallest eigenvalue).  Returns (l rows as dicts, d, bandwidth) or None when a pivot is not
    positive (rank-deficient system: the caller falls back to the dense pivoted LU)."""
    bw = 0
    for i in range(n):
        for j in range(max(0, i - 64), i):
            if a[i, j] != 0:
                bw = max(bw, i - j)
        for j in range(0, max(0, i - 64)):
            if a[i, j] != 0:
                return None          # not banded the way a chain is
    low = [[a[i, j] for j in range(max(0, i - bw), i + 1)] for i in range(n)]    # low[i][j - (i - bw)] for j in [i - bw, i]
    dd = [mp.mpf(0)] * n
    for j in range(n):
        j0 = max(0, j - bw)
        acc = low[j][j - j0]
        for k in range(j0, j):
            acc -= low[j][k - j0] ** 2 * dd[k]
        if not acc > 0:
            return None
        dd[j] = acc
        for i in range(j + 1, min(n, j + bw + 1)):
            i0 = max(0, i - bw)
            v = low[i][j - i0]
            for k in range(max(i0, j0), j):
                v -= low[i][k - i0] * low[j][k - j0] * dd[k]
            low[i][j - i0] = v / acc
    return low, dd, bw


def _banded_solve(fac, rhs, n):
    low, dd, bw = fac
    y = [rhs[i] for i in range(n)]
    for i in range(n):
        i0 = max(0, i - bw)
        for k in range(i0, i):
            y[i] -= low[i][k - i0] * y[k]
    for i in range(n):
        y[i] /= dd[i]
    for i in range(n - 1, -1, -1):
        for k in range(i + 1, min(n, i + bw + 1)):
            y[i] -= low[k][i - max(0, k - bw)] * y[k]
    return mp.matrix(y)


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
    # M (LIN:182-260) is a 0/1 selection matrix: row i*N + p -> column of (vertex i, p); row i*N + h + p -> column of
    # (vertex i+1, p).  R = M^T blkdiag(H_i) M (LIN:334-335) is therefore the scatter-add of the segments' H_i onto the
    # columns of their two vertices, and M[iN:(i+1)N, :] d_all (LIN:274-275) a gather -- the same numbers as the dense
    # products (exact arithmetic on 0/1 entries), without two (N K)^2 x (n_f + n_p) multiplications at 50 digits.
    cols = [[col_of[(i, p)] for p in range(h)] + [col_of[(i + 1, p)] for p in range(h)] for i in range(k)]
    ainvs, qs = [], []
    r_full = mp.zeros(nf + npf, nf + npf)
    for i in range(k):
        a = mapping_matrix(n, float(times[i]))
        ai = a ** -1
        q = cost_matrix(n, derivative, float(times[i]))
        hm = ai.T * q * ai
        ainvs.append(ai)
        qs.append(q)
        for r in range(n):
            for c in range(n):
                r_full[cols[i][r], cols[i][c]] += hm[r, c]
    coeffs = np.zeros((k, dim, n))
    d_free = np.zeros((dim, npf))
    cost = mp.mpf(0)
    banded = None
    if npf:
        rpf = r_full[nf:, :nf]
        rpp = r_full[nf:, nf:]
        if npf > 48:
            banded = _banded_ldl(rpp, npf)      # long chains: R_PP is block tridiagonal (bandwidth < 2h), dense LU is O(n^3) at 50 digits
        if banded is None:
            lu = mp.matrix(rpp)     # one factorisation for all dimensions (as the reference shares its QR, LIN:365-375)
            lu_a, lu_p = mp.mp.LU_decomp(lu)
    for d in range(dim):
        df = mp.matrix([mp.mpf(float(x)) for x in d_fixed[d]])
        if npf:
            rhs = -(rpf * df)
            if banded is not None:
                dp = _banded_solve(banded, rhs, npf)
            else:
                # forward / backward substitution with the shared LU factors (mpmath's own helpers)
                y = mp.mp.L_solve(lu_a, rhs, lu_p)
                dp = mp.mp.U_solve(lu_a, y)
            d_all = list(df) + list(dp)
            d_free[d] = [float(x) for x in dp]
        else:
            d_all = list(df)
        for i in range(k):
            new_d = mp.matrix([d_all[cidx] for cidx in cols[i]])
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
