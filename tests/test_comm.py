"""The C ABI's RCCL communicator (mtg_comm_*, csrc/mtg_comm.hip): the final gather of north_star for a one-process-per-GPU
consumer of libmtg_hip.so, without torch.distributed.  CPU: the entry points exist and reject bad arguments (no RCCL call).
GPU (1-GPU box): a one-rank communicator -- ncclGetUniqueId, ncclCommInitRank on the context's device, ncclAllGather on the
communicator's own stream, the chunked solve + gather with its event choreography -- everything but the cross-device
transport; the two-rank form needs two GPUs (the driver's multi-GPU runs)."""
import ctypes

import numpy as np
import pytest


def test_comm_entry_points_reject_bad_arguments():
    from mav_trajectory_generation_amd import _lib
    lib = _lib.load()
    assert lib.mtg_comm_unique_id(None) == -1
    h = ctypes.c_void_p()
    buf = ctypes.create_string_buffer(128)
    assert lib.mtg_comm_create(None, 0, 1, buf, ctypes.byref(h)) == -1
    assert lib.mtg_comm_all_gather(None, None, 0, None) == -1
    assert lib.mtg_comm_sync(None) == -1
    assert lib.mtg_comm_destroy(None) == 0
    assert lib.mtg_comm_rank(None) == -1 and lib.mtg_comm_world(None) == -1


@pytest.mark.gpu
def test_one_rank_communicator_gathers_through_rccl():
    import torch
    import mav_trajectory_generation_amd as m
    from mav_trajectory_generation_amd.dist import Communicator
    ctx = m.Context(0)
    comm = Communicator(ctx, 0, 1, Communicator.unique_id())
    assert ctx.lib.mtg_comm_rank(comm.handle) == 0 and ctx.lib.mtg_comm_world(comm.handle) == 1
    n, k, dim, d, bsz = 10, 8, 3, 4, 4000
    masks = m.ends_full_masks(n, k)
    plan = m.Plan(ctx, n, dim, k, d, masks)
    t, f = m.random_waypoint_batch(bsz, k, dim, n, masks, seed=4, device="cuda", layout="soa")
    ref, _, _ = plan.solve(t, f, layout="soa")
    # plain all-gather of a solved buffer
    g = comm.all_gather(ref)
    comm.sync()
    assert g.shape == (1, bsz, k, dim, n) and torch.equal(g[0], ref)
    # chunked solve + gather: chunk-major [n_chunks][world][Bc][K][D][N]
    for n_chunks in (1, 4, 5):
        local, gathered = comm.solve_all_gather(plan, t, f, layout="soa", n_chunks=n_chunks)
        comm.sync()
        assert torch.equal(local, ref)
        assert gathered.shape == (n_chunks, 1, bsz // n_chunks, k, dim, n)
        assert torch.equal(gathered.reshape(bsz, k, dim, n), ref)
    # AoS inputs go through the layout's batch strides
    ta, fa = t.t().contiguous(), f.permute(2, 0, 1).contiguous()
    local, gathered = comm.solve_all_gather(plan, ta, fa, layout="aos", n_chunks=4)
    comm.sync()
    assert torch.equal(gathered.reshape(bsz, k, dim, n), ref)
    # a bad segment time in one chunk is reported by the sync
    tb = t.clone()
    tb[3, 2500] = -1.0
    comm.solve_all_gather(plan, tb, f, layout="soa", n_chunks=4)
    with pytest.raises(RuntimeError) as e:
        comm.sync()
    assert "segment" in str(e.value).lower() or "-2" in str(e.value)
    comm.close()
    plan.close()
    ctx.close()
