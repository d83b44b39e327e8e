"""Several GPUs from one host process through the C ABI (mtg_device_group_*).  CPU: the shard arithmetic equals the
multi-process module's (dist.shard_range).  GPU (1-GPU box): a group that lists device 0 three times -- three contexts,
plans and streams, concurrent shards, peer-copy / host gather -- against the single-context solve."""
import ctypes

import numpy as np
import pytest


def test_shard_range_matches_the_multi_process_split():
    from mav_trajectory_generation_amd import _lib
    from mav_trajectory_generation_amd.dist import shard_range
    lib = _lib.load()
    lo, hi = ctypes.c_int64(), ctypes.c_int64()
    for batch in (0, 1, 7, 64, 1000, 10_000, 1_000_003):
        for world in (1, 2, 3, 8):
            for r in range(world):
                lib.mtg_shard_range(batch, world, r, ctypes.byref(lo), ctypes.byref(hi))
                assert (lo.value, hi.value) == shard_range(batch, r, world)
    for (batch, world, r) in ((100, 0, 0), (100, -2, 0), (100, 4, 4), (100, 4, -1), (-5, 2, 0)):   # no such shard: empty range
        lo.value, hi.value = 7, 9
        lib.mtg_shard_range(batch, world, r, ctypes.byref(lo), ctypes.byref(hi))
        assert (lo.value, hi.value) == (0, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("soa", [1, 0])
def test_device_group_on_one_gpu(soa):
    import torch
    import mav_trajectory_generation_amd as m
    from mav_trajectory_generation_amd import _lib as L
    lib = L.load()
    n, k, dim, d, batch, G = 10, 8, 3, 4, 10_001, 3
    masks = m.ends_full_masks(n, k)
    arr = (ctypes.c_uint32 * (k + 1))(*masks)
    desc = L.PlanDesc(n, dim, k, d, arr)
    devs = (ctypes.c_int32 * G)(0, 0, 0)
    grp = ctypes.c_void_p()
    assert lib.mtg_device_group_create(G, devs, ctypes.byref(desc), ctypes.byref(grp)) == 0
    assert lib.mtg_device_group_size(grp) == G
    t, f = m.random_waypoint_batch(batch, k, dim, n, masks, seed=21, device="cuda", layout="aos")
    shards, lo, hi = [], ctypes.c_int64(), ctypes.c_int64()
    for s in range(G):
        lib.mtg_shard_range(batch, G, s, ctypes.byref(lo), ctypes.byref(hi))
        ts, fs = t[lo.value:hi.value], f[lo.value:hi.value]
        if soa:
            ts, fs = ts.t().contiguous(), fs.permute(1, 2, 0).contiguous()
        else:
            ts, fs = ts.contiguous(), fs.contiguous()
        shards.append((ts, fs, torch.empty((hi.value - lo.value, k, dim, n), dtype=torch.float64, device="cuda")))
    torch.cuda.synchronize()
    ptrs = lambda j: (ctypes.c_void_p * G)(*[sh[j].data_ptr() for sh in shards])
    tp, fp, cp = ptrs(0), ptrs(1), ptrs(2)
    assert lib.mtg_device_group_solve_linear(grp, batch, soa, tp, fp, cp, 0) == 0
    assert lib.mtg_device_group_sync(grp) == 0
    # reference: one context, whole batch
    ctx = m.Context(0)
    plan = m.Plan(ctx, n, dim, k, d, masks)
    ref, _, _ = plan.solve(t, f)
    ctx.sync()
    got = torch.cat([sh[2] for sh in shards], dim=0)
    den = ref.abs().amax(dim=-1).clamp_min(1e-300)
    assert float(((got - ref).abs().amax(dim=-1) / den).max()) < 1e-11
    # gather to the root device and to host memory
    dst = torch.zeros_like(ref)
    assert lib.mtg_device_group_gather_coeffs(grp, batch, cp, 1, ctypes.c_void_p(dst.data_ptr())) == 0
    assert torch.equal(dst, got)
    host = np.zeros(tuple(ref.shape))
    assert lib.mtg_device_group_gather_coeffs(grp, batch, cp, -1, ctypes.c_void_p(host.ctypes.data)) == 0
    assert np.array_equal(host, got.cpu().numpy())
    # a bad segment time in one shard surfaces at the group sync
    shards[2][0][0 if soa else 5, 5 if soa else 0] = -1.0
    assert lib.mtg_device_group_solve_linear(grp, batch, soa, tp, fp, cp, 0) == 0
    assert lib.mtg_device_group_sync(grp) == -2
    assert lib.mtg_device_group_sync(grp) == 0
    plan.close()
    ctx.close()
    assert lib.mtg_device_group_destroy(grp) == 0
