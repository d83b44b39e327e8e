"""GPU tests of mtg_solve_linear_sequence: a queue of independent batches of one plan in ONE persistent launch
(csrc/mtg_kernels.h: mtg_solve_slab_queue_kernel, csrc/mtg_dimlane.h: the queue form of mtg_solve_dl_kernel).  Called
through the C ABI; every batch of the queue must equal, bit for bit, the same batch solved by its own launch (same
per-lane arithmetic), and the oracle within the north-star tolerance (1e-9 norm-wise per polynomial)."""
import numpy as np
import pytest

import helpers
from oracle import oracle_np as onp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    import mav_trajectory_generation_amd as m
    c = m.Context(0)
    yield c
    c.close()


def make_sets(m, plan, n, bsz, masks, layout, seed, yaw=False):
    import torch
    sets = []
    for s in range(n):
        t, f = m.random_waypoint_batch(bsz, plan.K, plan.D, plan.N, masks, seed=seed + 17 * s, device="cuda", layout=layout,
                                       yaw_dim=yaw)
        sets.append((t, f, torch.full((bsz, plan.K, plan.D, plan.N), float("nan"), dtype=torch.float64, device="cuda")))
    return sets


# (N, K, D, derivative, interior mask): shapes with a queue form (slab-output fused kernels; config 5 and other
# dimension-in-lane shapes go through the dimension-in-lane queue)
QUEUE_SHAPES = [(10, 8, 3, 4, 1), (8, 8, 3, 3, 1), (10, 4, 3, 4, 1), (8, 4, 3, 3, 1), (12, 4, 3, 5, 1),
                (10, 16, 4, 4, 7), (12, 8, 3, 5, 1), (10, 16, 3, 4, 1)]


@pytest.mark.parametrize("shape", QUEUE_SHAPES)
@pytest.mark.parametrize("n,bsz", [(2, 1), (3, 63), (2, 64), (5, 65), (7, 1000), (4, 2113), (100, 130), (97, 21)])
@pytest.mark.parametrize("layout", ["soa", "aos"])
def test_queue_equals_single_launches(ctx, shape, n, bsz, layout):
    """Queue lengths around the per-launch limit (96 batches), batch sizes around the tile widths (64 and 64 / D)."""
    import torch
    import mav_trajectory_generation_amd as m
    N, K, D, d, interior = shape
    masks = m.ends_full_masks(N, K, interior)
    plan = m.Plan(ctx, N, D, K, d, masks)
    sets = make_sets(m, plan, n, bsz, masks, layout, seed=1000 + bsz, yaw=(D == 4))
    plan.solve_sequence(sets, layout=layout)
    ctx.sync()
    singles = [torch.full_like(s[2], float("nan")) for s in sets]
    plan.solve_sequence([(t, f, c) for (t, f, _), c in zip(sets, singles)], layout=layout, one_launch_per_batch=True)
    ctx.sync()
    for i, ((_, _, co), ref) in enumerate(zip(sets, singles)):
        assert torch.isfinite(co).all(), f"batch {i}"
        assert torch.equal(co, ref), f"batch {i} of {n} x {bsz}"
    # first, middle and last batch against the oracle
    tol = 1e-9 if N <= 10 else 5e-7
    for i in sorted({0, n // 2, n - 1}):
        t, f, co = sets[i]
        nb = min(bsz, 40)
        if layout == "soa":
            th, fh = t.t()[:nb].contiguous().cpu().numpy(), f.permute(2, 0, 1)[:nb].contiguous().cpu().numpy()
        else:
            th, fh = t[:nb].cpu().numpy(), f[:nb].cpu().numpy()
        c_lit, _, _ = onp.solve_batch(N, d, masks, th, fh)
        assert helpers.poly_relerr(co[:nb].cpu().numpy(), c_lit) < tol
    plan.close()


def test_queue_status_flags(ctx):
    """A non-positive segment time in one batch of the queue raises the batch-wide status; the other batches are solved."""
    import torch
    import mav_trajectory_generation_amd as m
    masks = m.ends_full_masks(10, 8)
    plan = m.Plan(ctx, 10, 3, 8, 4, masks)
    sets = make_sets(m, plan, 4, 500, masks, "soa", seed=5)
    good = [torch.empty_like(s[2]) for s in sets]
    plan.solve_sequence([(t, f, c) for (t, f, _), c in zip(sets, good)], layout="soa")
    ctx.sync()
    sets[2][0][3, 77] = -1.0
    plan.solve_sequence(sets, layout="soa")
    with pytest.raises(m.MtgError) as e:
        ctx.sync()
    assert e.value.code == -2
    for i in (0, 1, 3):
        assert torch.equal(sets[i][2], good[i])
    ctx.sync()    # the flag was cleared
    plan.close()


def test_queue_large_total(ctx):
    """BASELINE config 2 as the bench runs it: 20 batches of 10 000 in one launch; device-side comparison with the single
    launches (norm-wise), and the events bracket the launch."""
    import torch
    import mav_trajectory_generation_amd as m
    masks = m.ends_full_masks(10, 8)
    plan = m.Plan(ctx, 10, 3, 8, 4, masks)
    sets = [plan.generate_waypoints(10_000, seed=300 + s, layout="soa") for s in range(20)]
    sets = [(t, f, torch.zeros((10_000, 8, 3, 10), dtype=torch.float64, device="cuda")) for (t, f) in sets]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(ctx.stream):
        e0.record(ctx.stream)
        e1.record(ctx.stream)
    torch.cuda.synchronize()
    plan.solve_sequence(sets, layout="soa", start_event=e0, stop_event=e1)
    ctx.sync()
    assert 0.0 < e0.elapsed_time(e1) < 50.0
    for i in (0, 7, 19):
        t, f, co = sets[i]
        ref, _, _ = plan.solve(t, f, layout="soa", dims="fused")
        ctx.sync()
        assert torch.equal(co, ref)
        dl, _, _ = plan.solve(t, f, layout="soa", dims="dimlane")
        rel, _ = ctx.compare_coefficients(co, dl)
        assert rel < 1e-12
    plan.close()
