"""Mixed requests that are REBUILT for every set of buffers (round 5: the library caches the cross-structure launch's unit schedule
per structure, shares the workspace, pools the item tables and uploads them asynchronously): results bit-identical to per-bucket
launches for every rebuilt request, whatever was built before it; the packed-item path of the Python layer."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_rebuilt_requests_share_the_cached_schedule():
    import torch
    import mav_trajectory_generation_amd as m
    ctx = m.Context(0)
    shapes = [(8, 3, 4), (10, 4, 8), (12, 5, 16), (10, 4, 32), (8, 3, 32)]       # config-4 shapes: bodies of the cross-structure kernel
    plans = [m.Plan(ctx, n, 3, k, d, m.ends_full_masks(n, k, 1)) for (n, d, k) in shapes]

    def make_set(seed, sizes):
        items = []
        for (n, d, k), plan, bsz in zip(shapes, plans, sizes):
            t, f = m.random_waypoint_batch(bsz, k, 3, n, plan.fixed_mask, seed=seed + n + k, device="cuda", layout="soa")
            items.append(dict(plan=plan, times=t, d_fixed=f, layout="soa",
                              coeffs=torch.full((bsz, k, 3, n), float("nan"), dtype=torch.float64, device="cuda")))
        return items

    sizes_a, sizes_b = [300, 500, 64, 700, 21], [301, 500, 64, 43, 21]        # two STRUCTURES (tile counts differ)
    created = []
    for rnd, sizes in enumerate([sizes_a, sizes_a, sizes_b, sizes_a, sizes_b, sizes_a]):
        sets = [make_set(100 * rnd + s, sizes) for s in range(3)]
        packed = np.concatenate([m.pack_multi_items(it) for it in sets])       # one request over three sets
        req = m.PackedMultiSolve(ctx, packed, keep=sets)
        assert req.launch_count == 1
        created.append(req.create_us)
        req.solve()
        ctx.sync()
        for items in sets:
            for it in items:
                ref, _, _ = it["plan"].solve(it["times"], it["d_fixed"], layout="soa", dims="dimlane")
                ctx.sync()
                assert torch.equal(it["coeffs"], ref)
        if rnd % 2:
            req.close()          # some requests die early (their item table returns to the free list), some live on
    # the first build of a structure computes and uploads its schedule; later ones only fill a 15-item table
    assert min(created[1], created[3], created[5]) < created[0]
    ctx.close()
