"""Evidence behind the structural rank decision (tests/test_pivot_threshold.py, csrc/mtg_abi.hip structural_null_dim): how large
does the pivot that is ZERO in exact arithmetic come out of a float64 LDL^T of R_PP (the reference's own R = M^T H M,
LIN:308-336, from oracle/_ref), relative to the variable's own diagonal entry and to the largest one -- against the smallest
legitimate pivot of regular ill-conditioned problems.  Test infrastructure (imports oracle/); output committed as
profiles/r05_pivot_ratio_study.txt.   python tests/pivot_ratio_study.py > profiles/r05_pivot_ratio_study.txt"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))   # (helpers.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import helpers  # noqa: E402
from oracle import ref_linear  # noqa: E402


def ldl(rpp):
    a = rpp.copy()
    n = a.shape[0]
    orig = np.diag(rpp).copy()
    d = np.zeros(n)
    for j in range(n):
        d[j] = a[j, j] if a[j, j] != 0 else 1e-300
        l = a[j + 1:, j] / d[j]
        a[j + 1:, j + 1:] -= np.outer(l, a[j + 1:, j])
    return d, orig


def case(rng, n, d, k, masks, tlo, thi, alternate=False):
    nf = sum(bin(m).count("1") for m in masks)
    times = rng.uniform(tlo, thi, k)
    if alternate:
        times[::2] = tlo
    _, r, nfix, _ = ref_linear.m_and_r(n, d, masks, times, rng.uniform(-2, 2, (1, nf)))
    dd, orig = ldl(r[nfix:, nfix:])
    j = int(np.argmin(dd / orig))
    return dd[j] / orig[j], dd[j] / orig.max(), orig.max() / orig.min()


def main():
    rng = np.random.default_rng(0)
    print("# smallest pivot of a natural-order float64 LDL^T of R_PP: d_j / R_PP[j][j], d_j / max diag, max diag / min diag")
    print("# A. structurally rank-deficient (ends position only / start position + velocity only, every other slot free)")
    for n in (8, 10, 12):
        for k in (1, 2, 3, 8, 16, 32, 50):
            for masks in ([1] + [0] * (k - 1) + [1], [3] + [0] * k):
                print(f"N={n:2d} K={k:2d} ends={masks[0]},{masks[-1]}  %+.1e  %+.1e  %.1e" % case(rng, n, n // 2 - 1, k, masks, 0.5, 3.0))
    print("# B. regular (ends fully fixed, interior position): d < h - 1, long chains, segment times alternating 0.05 / U(0.05, 20)")
    for (n, d, k, masks, tlo, thi, alt) in [(12, 2, 5, [3, 1, 1, 1, 1, 3], 0.5, 3, False), (12, 5, 32, None, 0.05, 20, True),
                                            (12, 5, 32, None, 0.2, 5, True), (10, 2, 5, None, .5, 3, False), (10, 3, 9, None, 0.05, 20, True),
                                            (10, 4, 100, None, .5, 3, False), (10, 4, 16, None, 0.05, 20, True), (8, 3, 100, None, 0.05, 20, True)]:
        if masks is None:
            masks = helpers.masks_ends_full(n, k, 1)
        for _ in range(3):
            print(f"N={n:2d} d={d} K={k:3d} T in [{tlo}, {thi}]{' alternating' if alt else ''}  %+.1e  %+.1e  %.1e" % case(rng, n, d, k, masks, tlo, thi, alt))


if __name__ == "__main__":
    main()
