"""World-size-2 gloo test (CPU) of the multi-GPU path's logic: contiguous sharding + final all_gather.  The
per-shard 'solve' runs the host-emulated lane code (test infrastructure) so no GPU is needed; on the GPU box the
same dist.py functions run over RCCL."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_the_batch():
    from mav_trajectory_generation_amd.dist import shard_range
    for batch in (0, 1, 7, 64, 1000, 10_000, 1_000_000):
        for world in (1, 2, 3, 8):
            spans = [shard_range(batch, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == batch
            for (a, b), (c, d) in zip(spans[:-1], spans[1:]):
                assert b == c
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes
    import torch
    import torch.distributed as dist
    import helpers
    from mav_trajectory_generation_amd.dist import gather_coeffs, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = ctypes.CDLL(os.path.join(ROOT, "tests", "libmtg_host_emu.so"))
    dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
    lib.mtg_emu_run.argtypes = [ctypes.c_int] * 4 + [ip, ctypes.c_longlong, dp, dp, dp, dp, dp, ctypes.c_int, ip]
    batch = 37   # odd on purpose: unequal shards
    masks, times, d_fixed = helpers.reference_batch(batch, 8, 10, 3, 555)
    lo, hi = shard_range(batch, rank, world)
    rc, co, _, _, st = helpers.emu_run(lib, 10, 3, 8, 4, masks, times[lo:hi], d_fixed[lo:hi], mode=1, want_cost=False)
    assert rc == 0 and st == 0
    full = gather_coeffs(torch.from_numpy(co), batch)
    if rank == 0:
        np.save(os.path.join(tmp, "gathered.npy"), full.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_shard_and_gather(host_emu, tmp_path):
    import torch.multiprocessing as mp
    import helpers
    from oracle import oracle_np as onp
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), "gathered.npy"))
    masks, times, d_fixed = helpers.reference_batch(37, 8, 10, 3, 555)
    c_lit, _, _ = onp.solve_batch(10, 4, masks, times, d_fixed)
    assert got.shape == c_lit.shape
    assert helpers.poly_relerr(got, c_lit) < 1e-9
