"""Multi-GPU sharding of a batch of independent trajectories.

The solve needs no communication: rank r of G owns the contiguous slice [r*B/G, (r+1)*B/G) and runs the same
single-GPU path on it.  The only collective is the optional final gather of the coefficient buffer
(torch.distributed all_gather: RCCL over xGMI on GPUs, gloo in the CPU tests).
"""
from __future__ import annotations

from typing import Tuple


def shard_range(batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split: sizes differ by at most one, concatenation over ranks = range(batch)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_coeffs(local, batch: int, group=None):
    """all_gather the per-rank coefficient slices [b_local][K][D][N] into the full [batch][K][D][N] tensor
    (ranks may hold different slice lengths: padded to the longest, then trimmed)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = [shard_range(batch, r, world)[1] - shard_range(batch, r, world)[0] for r in range(world)]
    longest = max(sizes)
    pad = local
    if local.shape[0] < longest:
        pad = torch.zeros((longest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad.contiguous(), group=group)
    return torch.cat([o[:s] for o, s in zip(out, sizes)], dim=0)


def solve_sharded(plan, times, d_fixed, layout: str = "aos", gather: bool = False, group=None):
    """Each rank passes the FULL host-side problem description and solves only its shard on its GPU.
    times / d_fixed: full-batch CUDA tensors in `layout`; returns this rank's coeffs (or all, if gather)."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    batch = times.shape[0] if layout == "aos" else times.shape[1]
    lo, hi = shard_range(batch, rank, world)
    if layout == "aos":
        t, f = times[lo:hi].contiguous(), d_fixed[lo:hi].contiguous()
    else:
        t, f = times[:, lo:hi].contiguous(), d_fixed[:, :, lo:hi].contiguous()
    coeffs, _, _ = plan.solve(t, f, layout=layout)
    return gather_coeffs(coeffs, batch, group) if gather else coeffs


class ChunkedSolveGather:
    """Solve this rank's batch in `n_chunks` pieces and all_gather every piece as soon as it is solved, on a second
    stream: chunk i's gather (RCCL over xGMI) overlaps chunk i+1's solve (at 240 MB per rank the
    gather takes ~10x the solve, so it is worth hiding the solve behind it, not the other way round).

    The gathered buffer is chunk-major: gathered[c][r] = rank r's chunk c, shape [n_chunks][world][Bc][K][D][N]; global
    trajectory index of gathered[c][r][i] = r * B + c * Bc + i (B divisible by n_chunks is required)."""

    def __init__(self, plan, times, d_fixed, layout: str = "soa", n_chunks: int = 4, group=None):
        import torch
        import torch.distributed as dist
        self.plan, self.layout, self.group = plan, layout, group
        self.world = dist.get_world_size(group)
        batch = times.shape[0] if layout == "aos" else times.shape[1]
        while n_chunks > 1 and batch % n_chunks:
            n_chunks -= 1
        self.n_chunks, self.bc = n_chunks, batch // n_chunks
        dev = times.device
        self.chunks = []
        for c in range(n_chunks):
            lo, hi = c * self.bc, (c + 1) * self.bc
            if layout == "aos":
                t, f = times[lo:hi].contiguous(), d_fixed[lo:hi].contiguous()
            else:
                t, f = times[:, lo:hi].contiguous(), d_fixed[:, :, lo:hi].contiguous()
            self.chunks.append((t, f))
        self.local = torch.empty((n_chunks, self.bc, plan.K, plan.D, plan.N), dtype=torch.float64, device=dev)
        self.gathered = torch.empty((n_chunks, self.world, self.bc, plan.K, plan.D, plan.N), dtype=torch.float64, device=dev)
        self.backend = dist.get_backend(group)
        self.comm = torch.cuda.Stream(dev)
        self.host_bounce = None
        if self.backend != "nccl":   # gloo (multi-process tests on a 1-GPU box): the gather goes through host memory
            self.host_bounce = (torch.empty(self.local[0].shape, dtype=torch.float64).pin_memory(),
                                torch.empty(self.gathered[0].shape, dtype=torch.float64).pin_memory())

    def run(self, solve: bool = True, gather: bool = True):
        import torch
        import torch.distributed as dist
        ctx = self.plan.ctx
        for c, (t, f) in enumerate(self.chunks):
            if solve:
                self.plan.solve(t, f, layout=self.layout, coeffs=self.local[c])
            if gather:
                if self.backend == "nccl":
                    self.comm.wait_stream(ctx.stream)
                    with torch.cuda.stream(self.comm):
                        dist.all_gather_into_tensor(self.gathered[c].flatten(0, 1), self.local[c], group=self.group)
                else:
                    ctx.stream.synchronize()
                    h_in, h_out = self.host_bounce
                    h_in.copy_(self.local[c])
                    dist.all_gather_into_tensor(h_out.flatten(0, 1), h_in, group=self.group)
                    self.gathered[c].copy_(h_out)
        if gather and self.backend == "nccl":
            torch.cuda.current_stream().wait_stream(self.comm)
        return self.gathered


class Communicator:
    """The C ABI's own RCCL communicator (mtg_comm_*, csrc/mtg_comm.hip): what a one-process-per-GPU C++ consumer of
    libmtg_hip.so calls -- no torch.distributed involved.  `unique_id`: bytes from Communicator.unique_id() on rank 0, shipped
    to the other ranks by the caller (here typically through torch.distributed's store or a file)."""

    @staticmethod
    def unique_id() -> bytes:
        import ctypes
        from . import _lib as L
        lib = L.load()
        buf = ctypes.create_string_buffer(128)
        rc = lib.mtg_comm_unique_id(buf)
        if rc != 0:
            raise RuntimeError(f"mtg_comm_unique_id: {lib.mtg_status_string(rc).decode()}")
        return buf.raw

    def __init__(self, ctx, rank: int, world: int, unique_id: bytes):
        import ctypes
        self.ctx, self.lib = ctx, ctx.lib
        assert len(unique_id) == 128
        h = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(unique_id, 128)
        rc = self.lib.mtg_comm_create(ctx.handle, rank, world, buf, ctypes.byref(h))
        if rc != 0:
            raise RuntimeError(f"mtg_comm_create: {self.lib.mtg_status_string(rc).decode()}")
        self.handle, self.rank, self.world = h, rank, world

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(f"mtg_comm: {self.lib.mtg_status_string(rc).decode()} ({self.lib.mtg_comm_last_error(self.handle).decode()})")

    def all_gather(self, local, gathered=None):
        """gathered [world][...local.shape] <- every rank's `local` (float64 CUDA tensor, same shape on every rank)."""
        import ctypes
        import torch
        assert local.is_cuda and local.dtype == torch.float64 and local.is_contiguous()
        if gathered is None:
            gathered = torch.empty((self.world,) + tuple(local.shape), dtype=torch.float64, device=local.device)
        cur = self.ctx._enter()
        try:
            rc = self.lib.mtg_comm_all_gather(self.handle, ctypes.c_void_p(local.data_ptr()), local.numel(), ctypes.c_void_p(gathered.data_ptr()))
        finally:
            self.ctx._leave(cur)
        self._check(rc)
        return gathered

    def solve_all_gather(self, plan, times, d_fixed, layout: str = "soa", n_chunks: int = 4, local=None, gathered=None):
        """This rank's batch solved in chunks, every chunk all-gathered under the next chunk's solve (mtg_comm_solve_all_gather).
        Returns (local [B][K][D][N], gathered [n_chunks][world][B / n_chunks][K][D][N])."""
        import ctypes
        import torch
        batch = times.shape[0] if layout == "aos" else times.shape[1]
        while n_chunks > 1 and batch % n_chunks:
            n_chunks -= 1
        if local is None:
            local = torch.empty((batch, plan.K, plan.D, plan.N), dtype=torch.float64, device=times.device)
        if gathered is None:
            gathered = torch.empty((n_chunks, self.world, batch // n_chunks, plan.K, plan.D, plan.N), dtype=torch.float64, device=times.device)
        lay = plan.layout(batch, layout)
        cur = self.ctx._enter()
        try:
            rc = self.lib.mtg_comm_solve_all_gather(self.handle, plan.handle, batch, ctypes.byref(lay), ctypes.c_void_p(times.data_ptr()),
                                                    ctypes.c_void_p(d_fixed.data_ptr()), ctypes.c_void_p(local.data_ptr()),
                                                    ctypes.c_void_p(gathered.data_ptr()), n_chunks, 0)
        finally:
            self.ctx._leave(cur)
        self._check(rc)
        return local, gathered

    def sync(self):
        self._check(self.lib.mtg_comm_sync(self.handle))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.mtg_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
