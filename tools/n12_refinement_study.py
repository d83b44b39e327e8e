"""Would iterative refinement bring the N = 12 results to 1e-8 of the 50-digit solution?  Evidence for
DESIGN.md; test infrastructure (imports oracle/).   python tests/n12_refinement_study.py > profiles/r05_n12_refinement_study.txt

For the six trajectories with the largest segment-time ratio of a 400-trajectory N = 12 / K = 16 random-waypoint batch:
  * error of the lane code's d_P (host emulation of the kernels) against the 50-digit solve;
  * the same system assembled the way the kernels do (H(T) = T^(1-2d) S H(1) S from the once-rounded H(1) table, float64
    products), solved with LAPACK in float64;
  * that solution after two steps of iterative refinement whose residual R_PP d_P + R_PF d_F is accumulated in extended
    precision (x87 long double, 64-bit mantissa: what a compensated two-prod / two-sum accumulation gives).
Finding: on the ill-conditioned trajectory the refined solution is still 6e-8 ... 8e-8 off (from 2e-7): the error is in the
float64 ENTRIES of R (each a product of three rounded numbers) times the problem's componentwise condition number, not in the
solve -- refinement with a better residual converges to the solution of the perturbed matrix.  Reaching 1e-8 needs the
assembly itself (table, powers of T, products) in double-double: ~4x the arithmetic of the whole solve for trajectories on
which the reference is 1e-6 off.  Not built.
"""
import ctypes
import os
import sys

import mpmath as mp
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))   # (helpers.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import helpers  # noqa: E402
from oracle import oracle_mp  # noqa: E402


def main():
    mp.mp.dps = 50
    lane = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmtg_host_emu.so"))
    dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
    lane.mtg_emu_run.argtypes = [ctypes.c_int] * 4 + [ip, ctypes.c_longlong, dp, dp, dp, dp, dp, ctypes.c_int, ip]
    n, d, k, dim = 12, 5, 16, 3
    h = n // 2
    masks = helpers.masks_ends_full(n, k, 1)
    masks, times, dfx = helpers.reference_batch(400, k, n, dim, 31415 + 7 * k + n, masks)
    _, _, fr, _, _ = helpers.emu_run(lane, n, dim, k, d, masks, times, dfx, want_cost=False)
    ratio = times.max(axis=1) / times.min(axis=1)
    a1, q1 = oracle_mp.mapping_matrix(n, 1.0), oracle_mp.cost_matrix(n, d, 1.0)
    ai = a1 ** -1
    h1mp = ai.T * q1 * ai
    h1 = np.array([[float(h1mp[i, j]) for j in range(n)] for i in range(n)])      # the kernels' table: exact H(1), rounded once
    fixed = [(v, p) for v in range(k + 1) for p in range(h) if (masks[v] >> p) & 1]
    free = [(v, p) for v in range(k + 1) for p in range(h) if not (masks[v] >> p) & 1]
    nf = len(fixed)
    col = {key: i for i, key in enumerate(fixed)}
    col.update({key: nf + i for i, key in enumerate(free)})
    print("# trajectory, max/min segment time | lane code d_P error | per dimension: float64 solve error -> after 2 refinement steps (extended-precision residual)")
    for b in np.argsort(ratio)[-6:]:
        _, truth_f, _ = oracle_mp.solve(n, d, masks, times[b], dfx[b])
        e_emu = np.abs(fr[b] - truth_f).max() / np.abs(truth_f).max()
        r = np.zeros((nf + len(free),) * 2)
        for i in range(k):
            t = times[b, i]
            s = np.array([t ** p for p in range(h)] * 2)
            hs = t ** (1 - 2 * d) * np.outer(s, s) * h1
            cols = [col[(i, p)] for p in range(h)] + [col[(i + 1, p)] for p in range(h)]
            for a in range(n):
                for c in range(n):
                    r[cols[a], cols[c]] += hs[a, c]
        rpp, rpf = r[nf:, nf:], r[nf:, :nf]
        rl = r.astype(np.longdouble)
        out = []
        for dm in range(dim):
            x = np.linalg.solve(rpp, -(rpf @ dfx[b, dm]))
            e0 = np.abs(x - truth_f[dm]).max() / np.abs(truth_f[dm]).max()
            dall = np.concatenate([dfx[b, dm], x]).astype(np.longdouble)
            for _ in range(2):
                dall[nf:] += np.linalg.solve(rpp, (-(rl[nf:, :] @ dall)).astype(np.float64))
            e1 = np.abs(dall[nf:].astype(np.float64) - truth_f[dm]).max() / np.abs(truth_f[dm]).max()
            out.append(f"{e0:.1e} -> {e1:.1e}")
        print(f"b={int(b):3d} ratio {ratio[b]:5.1f} | {e_emu:.1e} | " + "   ".join(out))


if __name__ == "__main__":
    main()
