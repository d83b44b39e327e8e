"""The range arithmetic of the slab output (csrc/mtg_slab.h: range_of / range_of_phase, the head / tail pass), restated in
Python and checked exhaustively: for every standard shape, chain length and row phase the ranges a direction emits cover
its half of the trajectory's piece exactly once, every boundary that is not an end of the half lies on a 64-byte boundary
IN MEMORY, and no range is wider than the row width the kernel uses.  (The device code is exercised by the GPU tests; this
pins the formulas they implement.)"""
import pytest


def up64(x):
    return (x + 63) & ~63


def dn64(x):
    return x & ~63


def ranges(S, K, direction, phi, split_ends):
    """[(lo, hi)] in arrival order of the segments; split_ends: the misaligned head / tail go out in their own pass."""
    KA = (K + 1) // 2
    out = []
    segs = range(KA - 1, -1, -1) if direction > 0 else range(KA, K)
    for seg in segs:
        if direction > 0:
            lo = 0 if (seg == 0 and not split_ends) else up64(seg * S + phi) - phi
            hi = KA * S if seg == KA - 1 else up64((seg + 1) * S + phi) - phi
        else:
            lo = KA * S if seg == KA else dn64(seg * S + phi) - phi
            hi = K * S if (seg == K - 1 and not split_ends) else dn64((seg + 1) * S + phi) - phi
        hi = max(hi, lo)
        out.append((seg, lo, hi))
    extra = []
    if split_ends:
        if direction > 0:
            extra.append((0, (64 - phi) & 63))
        else:
            extra.append((dn64(K * S + phi) - phi, K * S))
    return out, extra


@pytest.mark.parametrize("S", [192, 240, 288, 320])        # N = 8 / 10 / 12 with D = 3; N = 10 with D = 4
@pytest.mark.parametrize("K", list(range(2, 33)) + [50])
def test_ranges_cover_each_half_once_and_are_aligned_in_memory(S, K):
    KA = (K + 1) // 2
    piece = K * S
    pmod = piece & 63
    phases = [0] if pmod == 0 else sorted({(r * pmod) & 63 for r in range(4)})
    row_width = ((S + 63) // 64) * 4 if pmod else None        # chunks per row of the phase mapping
    for phi in phases:
        for direction in (1, -1):
            split_ends = pmod != 0 and row_width is not None and (64 - (64 // row_width) * row_width <= 8) and row_width <= 32
            rs, extra = ranges(S, K, direction, phi, split_ends)
            lo_half, hi_half = (0, KA * S) if direction > 0 else (KA * S, piece)
            covered = []
            for seg, lo, hi in rs:
                assert lo_half <= lo <= hi <= hi_half
                # data dependencies: a range only needs its own segment and the neighbour recovered just before it
                if hi > lo:
                    first_seg, last_seg = lo // S, (hi - 1) // S
                    assert {first_seg, last_seg} <= ({seg, seg + 1} if direction > 0 else {seg, seg - 1})
                    if split_ends:
                        assert (hi - lo) // 16 <= row_width
                covered.append((lo, hi))
                for edge in (lo, hi):
                    if edge not in (lo_half, hi_half):
                        assert (edge + phi) % 64 == 0, (S, K, phi, direction, seg, edge)
            for lo, hi in extra:
                assert 0 <= hi - lo <= 48 and (hi - lo) % 16 == 0
                if hi > lo:
                    covered.append((lo, hi))
            covered = sorted(c for c in covered if c[1] > c[0])
            assert covered[0][0] == lo_half and covered[-1][1] == hi_half
            for (a0, a1), (b0, b1) in zip(covered, covered[1:]):
                assert a1 == b0                                   # no gap, no overlap
