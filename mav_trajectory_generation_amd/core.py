"""Thin object layer over the C ABI: Context (device, stream), Plan (N, K, D, d, masks), solve.

Mirrors the reference call sequence (polynomial_optimization_linear.h:57-108):
    PolynomialOptimization<N> opt(D); opt.setupFromVertices(vertices, times, d); opt.solveLinear();
as  plan = Plan(ctx, N, D, K, d, fixed_mask); plan.solve(times, d_fixed) -> coeffs [B][K][D][N]
for a whole batch of trajectories sharing the constraint structure.
torch is used only to own device memory and streams.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence

import numpy as np

from . import _lib as L

library_path = L.LIB_PATH


class MtgError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"mtg error {code}: {msg}")
        self.code = code


def _check(lib, code: int, ctx_handle=None):
    if code != 0:
        msg = lib.mtg_status_string(code).decode()
        if ctx_handle:
            extra = lib.mtg_last_error_string(ctx_handle).decode()
            if extra:
                msg += " (" + extra + ")"
        raise MtgError(code, msg)


class Context:
    """(device, stream, scratch).  Owns a torch.cuda.Stream (`.stream`) that the library enqueues on;
    solve calls are ordered after work already queued on torch's current stream and vice versa."""

    def __init__(self, device: int = 0, stream=None):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("mav_trajectory_generation_amd needs a HIP device (a context always owns a GPU; batches have no CPU path)")
        self.lib = L.load()
        self.device = int(device)
        self.stream = stream if stream is not None else torch.cuda.Stream(self.device)
        h = ctypes.c_void_p()
        _check(self.lib, self.lib.mtg_context_create(self.device, ctypes.c_void_p(self.stream.cuda_stream),
                                                     ctypes.byref(h)))
        self.handle = h
        import os
        import weakref
        self._plans = weakref.WeakSet()   # plans must be destroyed before their context
        # measurement knobs (include/mtg_hip_lab.h): the library itself never reads the environment
        for env, (name, conv) in L.ENV_OPTIONS.items():
            if env in os.environ:
                self.set_option(name, conv(os.environ[env]))

    def set_option(self, name: str, value: int):
        """A measurement knob of this context by name (include/mtg_hip_lab.h: A/B runs and form-forcing tests)."""
        _check(self.lib, self.lib.mtg_context_set_option(self.handle, name.encode(), int(value)), self.handle)

    def lab_segment_cost_matrices(self, n_coeffs: int, derivative: int, times, variant: int):
        """EVIDENCE variant (include/mtg_hip_lab.h: mtg_lab_segment_cost_matrices): H_k = A_k^-T Q_k A_k^-1 (LIN:318) of the
        given segment times, [n][N][N]; variant 1 = literally on the FP64 matrix cores, 0 = the scaling identity."""
        import torch
        t = times.contiguous()
        out = torch.empty((t.numel(), n_coeffs, n_coeffs), dtype=torch.float64, device=t.device)
        cur = self._enter()
        try:
            _check(self.lib, self.lib.mtg_lab_segment_cost_matrices(self.handle, n_coeffs, derivative, t.numel(), t.data_ptr(),
                                                                    out.data_ptr(), variant), self.handle)
        finally:
            self._leave(cur)
        return out

    def clock_probe_start(self, duration_us: float):
        """Start a shader-clock probe (include/mtg_hip_lab.h: mtg_lab_clock_probe_start) next to whatever runs on the device for
        the next `duration_us`; returns the handle clock_probe_finish takes."""
        h = ctypes.c_void_p()
        _check(self.lib, self.lib.mtg_lab_clock_probe_start(self.handle, float(duration_us), ctypes.byref(h)), self.handle)
        return h

    def clock_probe_finish(self, probe):
        """(shader clock in MHz over the probed interval, the interval in us)."""
        mhz, us = ctypes.c_double(0), ctypes.c_double(0)
        _check(self.lib, self.lib.mtg_lab_clock_probe_finish(probe, ctypes.byref(mhz), ctypes.byref(us)), self.handle)
        return mhz.value, us.value

    def _enter(self):
        """Order the library's stream after torch's current stream (no-op when they are the same)."""
        import torch
        cur = torch.cuda.current_stream(self.device)
        if cur != self.stream:
            self.stream.wait_stream(cur)
        return cur

    def _leave(self, cur):
        if cur != self.stream:
            cur.wait_stream(self.stream)

    def sync(self):
        _check(self.lib, self.lib.mtg_context_sync(self.handle), self.handle)

    def compare_coefficients(self, a, b):
        """(max over polynomials of ||a - b||_inf / ||b||_inf, max |a - b|) of two coefficient tensors [..., N], computed on
        the device (C ABI: mtg_compare_coefficients).  Synchronous."""
        assert a.shape == b.shape and a.is_cuda and b.is_cuda and a.is_contiguous() and b.is_contiguous()
        n = a.shape[-1]
        rel, ab = ctypes.c_double(0), ctypes.c_double(0)
        cur = self._enter()
        rc = self.lib.mtg_compare_coefficients(self.handle, ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()),
                                               a.numel() // n, n, ctypes.byref(rel), ctypes.byref(ab))
        self._leave(cur)
        _check(self.lib, rc, self.handle)
        return rel.value, ab.value

    def selftest_rcp(self, n: int = 1 << 20) -> float:
        out = ctypes.c_double(0)
        _check(self.lib, self.lib.mtg_selftest_rcp(self.handle, n, ctypes.byref(out)), self.handle)
        return out.value

    def close(self):
        if getattr(self, "handle", None):
            for p in list(getattr(self, "_plans", ())):
                p.close()
            self.lib.mtg_context_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def sample_range(ctx: "Context", coeffs, times, t_start: float, dt: float, n_samples: int, n_derivatives: int = 5,
                 times_layout: str = "aos", want_valid: bool = False):
    """Batched Trajectory::evaluateRange / sampleTrajectoryInRange: coeffs [B][K][D][N] and times ([B][K] 'aos' or
    [K][B] 'soa') CUDA tensors -> out [B][n_samples][n_derivatives][D] (and n_valid [B] int32)."""
    import torch
    bsz, k, dim, n = coeffs.shape
    assert coeffs.is_cuda and coeffs.dtype == torch.float64 and coeffs.is_contiguous() and times.is_contiguous()
    out = torch.empty((bsz, n_samples, n_derivatives, dim), dtype=torch.float64, device=coeffs.device)
    valid = torch.empty((bsz,), dtype=torch.int32, device=coeffs.device) if want_valid else None
    sb, sk = (k, 1) if times_layout == "aos" else (1, bsz)
    cur = ctx._enter()
    rc = ctx.lib.mtg_sample_range(ctx.handle, n, k, dim, bsz, ctypes.c_void_p(coeffs.data_ptr()),
                                  ctypes.c_void_p(times.data_ptr()), sb, sk, float(t_start), float(dt), n_samples,
                                  n_derivatives, ctypes.c_void_p(out.data_ptr()),
                                  ctypes.c_void_p(valid.data_ptr()) if valid is not None else None)
    ctx._leave(cur)
    _check(ctx.lib, rc, ctx.handle)
    return (out, valid) if want_valid else out


def minmax_magnitude(ctx: "Context", coeffs, times, derivative: int, dimensions: Optional[Sequence[int]] = None,
                     times_layout: str = "aos"):
    """Batched Trajectory::computeMinMaxMagnitude: coeffs [B][K][D][N], times ([B][K] 'aos' / [K][B] 'soa') CUDA
    tensors -> (segment_minmax [B][K][4], trajectory_minmax [B][4], trajectory_segment_idx [B][2] int32); the four
    columns are (t_min, v_min, t_max, v_max) with segment-local times."""
    import torch
    bsz, k, dim, n = coeffs.shape
    assert coeffs.is_cuda and coeffs.dtype == torch.float64 and coeffs.is_contiguous() and times.is_contiguous()
    mask = 0
    for d in (dimensions if dimensions is not None else range(dim)):
        if d < 0 or d >= dim:
            raise MtgError(-1, "dimension %d out of bounds [0..%d]" % (d, dim - 1))
        mask |= 1 << d
    seg = torch.empty((bsz, k, 4), dtype=torch.float64, device=coeffs.device)
    traj = torch.empty((bsz, 4), dtype=torch.float64, device=coeffs.device)
    idx = torch.empty((bsz, 2), dtype=torch.int32, device=coeffs.device)
    sb, sk = (k, 1) if times_layout == "aos" else (1, bsz)
    cur = ctx._enter()
    rc = ctx.lib.mtg_minmax_magnitude(ctx.handle, n, k, dim, bsz, ctypes.c_void_p(coeffs.data_ptr()),
                                      ctypes.c_void_p(times.data_ptr()), sb, sk, int(derivative), mask,
                                      ctypes.c_void_p(seg.data_ptr()), ctypes.c_void_p(traj.data_ptr()),
                                      ctypes.c_void_p(idx.data_ptr()))
    ctx._leave(cur)
    _check(ctx.lib, rc, ctx.handle)
    return seg, traj, idx


def scale_segment_times_to_meet_constraints(ctx: "Context", coeffs, times, v_max: float, a_max: float,
                                            max_iterations: int = 2, times_layout: str = "aos", workspace=None):
    """Batched Trajectory::scaleSegmentTimesToMeetConstraints, IN PLACE on coeffs [B][K][D][N] and times.
    Returns (scaling [B], within_range [B] int32, workspace); workspace[2*B*K*4:].view(2, B, 4)[:, :, 3] are the
    maximum |velocity| / |acceleration| seen by the last round's check."""
    import torch
    bsz, k, dim, n = coeffs.shape
    assert coeffs.is_cuda and coeffs.dtype == torch.float64 and coeffs.is_contiguous() and times.is_contiguous()
    need = 8 * bsz * (k + 1)
    if workspace is None:
        workspace = torch.empty((need,), dtype=torch.float64, device=coeffs.device)
    assert workspace.numel() >= need and workspace.dtype == torch.float64
    scaling = torch.empty((bsz,), dtype=torch.float64, device=coeffs.device)
    within = torch.empty((bsz,), dtype=torch.int32, device=coeffs.device)
    sb, sk = (k, 1) if times_layout == "aos" else (1, bsz)
    cur = ctx._enter()
    rc = ctx.lib.mtg_scale_segment_times_to_meet_constraints(
        ctx.handle, n, k, dim, bsz, ctypes.c_void_p(coeffs.data_ptr()), ctypes.c_void_p(times.data_ptr()), sb, sk,
        float(v_max), float(a_max), int(max_iterations), ctypes.c_void_p(workspace.data_ptr()),
        ctypes.c_void_p(scaling.data_ptr()), ctypes.c_void_p(within.data_ptr()))
    ctx._leave(cur)
    _check(ctx.lib, rc, ctx.handle)
    return scaling, within, workspace


class Plan:
    def __init__(self, ctx: Context, n_coeffs: int, dimension: int, n_segments: int,
                 derivative_to_optimize: Optional[int], fixed_mask: Sequence[int]):
        self.ctx = ctx
        self.lib = ctx.lib
        self.N, self.D, self.K = int(n_coeffs), int(dimension), int(n_segments)
        self.deriv = self.N // 2 - 1 if derivative_to_optimize is None else int(derivative_to_optimize)
        if len(fixed_mask) != self.K + 1:
            raise MtgError(-1, "fixed_mask must have n_segments + 1 entries")
        self.fixed_mask = [int(m) for m in fixed_mask]
        arr = (ctypes.c_uint32 * (self.K + 1))(*self.fixed_mask)
        desc = L.PlanDesc(self.N, self.D, self.K, self.deriv, arr)
        h = ctypes.c_void_p()
        _check(self.lib, self.lib.mtg_plan_create(ctx.handle, ctypes.byref(desc), ctypes.byref(h)), ctx.handle)
        self.handle = h
        ctx._plans.add(self)
        info = L.PlanInfo()
        _check(self.lib, self.lib.mtg_plan_get_info(h, ctypes.byref(info)))
        self.n_all, self.n_fixed, self.n_free = info.n_all, info.n_fixed, info.n_free
        self.kernel_variant = info.kernel_variant
        self.bytes_per_trajectory = info.algorithmic_bytes_per_trajectory
        # structural rank deficiency of the free system (0: regular; > 0: every solve flags every trajectory singular and
        # basic_solution=True gives the reference's basic solution, LIN:365-378)
        self.rank_deficiency = self.lib.mtg_plan_rank_deficiency(h)

    def layout(self, batch: int, kind: str) -> L.Layout:
        lay = L.Layout()
        if kind == "aos":
            self.lib.mtg_layout_aos(self.handle, batch, ctypes.byref(lay))
        elif kind == "soa":
            self.lib.mtg_layout_soa(self.handle, batch, ctypes.byref(lay))
        elif kind == "soa16":     # SoA, row stride padded to a multiple of 16 trajectories (mtg_layout_soa_padded)
            self.lib.mtg_layout_soa_padded(self.handle, batch, ctypes.byref(lay))
        else:
            raise ValueError(kind)
        return lay

    @staticmethod
    def _ptr(t):
        return ctypes.c_void_p(0) if t is None else ctypes.c_void_p(t.data_ptr())

    def generate_waypoints(self, batch: int, seed: int = 0, layout: str = "soa", box: float = 10.0, v_max: float = 3.0,
                           a_max: float = 5.0, yaw_dim: bool = False):
        """Random-waypoint inputs of this plan's structure generated by the library's own kernel (C ABI:
        mtg_generate_waypoints; the torch generator of workload.py has the same semantics).  Returns (times, d_fixed) float64
        CUDA tensors in `layout`.  Asynchronous on the library's stream (ordered against torch's current stream)."""
        import torch
        dev = torch.device("cuda", self.ctx.device)
        if layout == "soa":
            t = torch.empty((self.K, batch), dtype=torch.float64, device=dev)
            f = torch.empty((self.D, self.n_fixed, batch), dtype=torch.float64, device=dev)
        else:
            t = torch.empty((batch, self.K), dtype=torch.float64, device=dev)
            f = torch.empty((batch, self.D, self.n_fixed), dtype=torch.float64, device=dev)
        lay = self.layout(batch, layout)
        cur = self.ctx._enter()
        rc = self.lib.mtg_generate_waypoints(self.handle, batch, ctypes.byref(lay), int(seed) & (2**64 - 1), float(box),
                                             float(v_max), float(a_max), 1 if yaw_dim else 0, self._ptr(t), self._ptr(f))
        self.ctx._leave(cur)
        _check(self.lib, rc, self.ctx.handle)
        return t, f

    def solve(self, times, d_fixed, layout: str = "aos", want_free: bool = False, want_cost: bool = False,
              coeffs=None, d_free=None, cost=None, generic: bool = False, dims: str = "auto", ordered: bool = True,
              traj_status=None, basic_solution: bool = False, batch: Optional[int] = None, refine: bool = False):
        """times / d_fixed: float64 CUDA tensors in `layout` ('aos': [B][K], [B][D][n_fixed];
        'soa': [K][B], [D][n_fixed][B]).  Asynchronous; returns (coeffs [B][K][D][N], d_free, cost).
        dims: launch form -- 'auto', 'fused', 'split' (one dimension group per workgroup) or 'dimlane' (all dimensions of
        a trajectory in one wavefront; canonical SoA or AoS inputs, coefficient output only -- falls back to 'auto' where
        not eligible).
        traj_status: optional int32 CUDA tensor [B] that receives the per-trajectory status bits (1 bad time, 2 singular).
        basic_solution: the reference's behaviour on rank-deficient free systems (MTG_FLAG_BASIC_SOLUTION: flagged trajectories
        get the basic solution of a pivoted QR on the host; the call is then synchronous).
        refine: MTG_FLAG_REFINE -- one step of iterative refinement with the residual in double-double (conditioning-limited
        problems: N = 12, d < N/2 - 1); about five plain solves.
        ordered=False skips the automatic ordering against torch's current stream (the caller forks / joins the
        context's stream itself -- MixedBatchSolver runs independent buckets concurrently that way); output tensors
        must then be passed in, allocated by the caller before the fork."""
        import torch
        if layout == "soa16":
            # times [K][Bs], d_fixed [D][n_fixed][Bs] with Bs = batch rounded up to a multiple of 16: the batch size itself
            # cannot be read off the tensors
            assert batch is not None and times.shape[1] == ((batch + 15) & ~15), "layout 'soa16' needs batch= and padded tensors"
        else:
            batch = times.shape[0] if layout == "aos" else times.shape[1]
        assert times.dtype == torch.float64 and times.is_cuda and times.is_contiguous()
        assert d_fixed.dtype == torch.float64 and d_fixed.is_cuda and d_fixed.is_contiguous()
        dev = times.device
        if coeffs is None:
            coeffs = torch.empty((batch, self.K, self.D, self.N), dtype=torch.float64, device=dev)
        if want_free and d_free is None:
            shape = (batch, self.D, self.n_free) if layout == "aos" else (self.D, self.n_free, batch if layout == "soa" else (batch + 15) & ~15)
            d_free = torch.empty(shape, dtype=torch.float64, device=dev)
        if want_cost and cost is None:
            cost = torch.empty((batch,), dtype=torch.float64, device=dev)
        lay = self.layout(batch, layout)
        flags = L.FLAG_GENERIC_KERNEL if generic else 0
        flags |= {"auto": 0, "fused": L.FLAG_FUSED_DIMS, "split": L.FLAG_SPLIT_DIMS, "dimlane": L.FLAG_DIMLANE, "coop": L.FLAG_COOPERATIVE}[dims]
        if basic_solution:
            flags |= L.FLAG_BASIC_SOLUTION
        if refine:
            flags |= L.FLAG_REFINE
        if traj_status is not None:
            assert traj_status.dtype == torch.int32 and traj_status.is_cuda and traj_status.numel() >= batch
        cur = self.ctx._enter() if ordered else None
        rc = self.lib.mtg_solve_linear_status(self.handle, batch, ctypes.byref(lay), self._ptr(times),
                                              self._ptr(d_fixed), self._ptr(coeffs), self._ptr(d_free), self._ptr(cost),
                                              self._ptr(traj_status), flags)
        if ordered:
            self.ctx._leave(cur)
        _check(self.lib, rc, self.ctx.handle)
        return coeffs, d_free, cost

    def solve_sequence(self, sets, layout: str = "soa", dims: str = "auto", one_launch_per_batch: bool = False,
                       start_event=None, stop_event=None, ordered: bool = True, basic_solution: bool = False):
        """A queue of INDEPENDENT batches of equal size (mtg_solve_linear_sequence[_events]): `sets` = sequence of
        (times, d_fixed, coeffs) CUDA tensors in `layout`, coeffs [B][K][D][N] allocated by the caller.  Plans with a
        slab-output kernel run the whole queue as ONE persistent launch (workgroups walk the tiles of all batches);
        one_launch_per_batch=True keeps one launch per batch.  start_event / stop_event: torch.cuda.Event objects that
        have been recorded once (their hipEvent_t exists), recorded by the library around the queue."""
        import torch
        n = len(sets)
        if n == 0:
            return
        batch = sets[0][2].shape[0]
        for (t, f, co) in sets:
            assert t.dtype == torch.float64 and t.is_cuda and t.is_contiguous()
            assert f.dtype == torch.float64 and f.is_cuda and f.is_contiguous()
            assert co.dtype == torch.float64 and co.is_cuda and co.is_contiguous() and co.shape[0] == batch
        arr = [(ctypes.c_void_p * n)(*[x[j].data_ptr() for x in sets]) for j in range(3)]
        lay = self.layout(batch, layout)
        flags = {"auto": 0, "fused": L.FLAG_FUSED_DIMS, "split": L.FLAG_SPLIT_DIMS, "dimlane": L.FLAG_DIMLANE, "coop": L.FLAG_COOPERATIVE}[dims]
        if one_launch_per_batch:
            flags |= L.FLAG_SEQUENCE_ONE_LAUNCH_PER_BATCH
        if basic_solution:   # a structurally rank-deficient plan runs the queue on its shadow (asynchronous; LIN:365-378)
            flags |= L.FLAG_BASIC_SOLUTION
        ev = [ctypes.c_void_p(e.cuda_event) if e is not None else None for e in (start_event, stop_event)]
        cur = self.ctx._enter() if ordered else None
        rc = self.lib.mtg_solve_linear_sequence_events(self.handle, n, batch, ctypes.byref(lay), arr[0], arr[1], arr[2],
                                                       flags, ev[0], ev[1])
        if ordered:
            self.ctx._leave(cur)
        _check(self.lib, rc, self.ctx.handle)

    def solve_cost_only(self, times, d_fixed, layout: str = "aos", cost=None):
        """computeCost() of the optimum for every trajectory without materialising the segments
        (MTG_FLAG_COST_ONLY): what the time optimisers' objective callbacks need.  Returns cost [B]."""
        import torch
        batch = times.shape[0] if layout == "aos" else times.shape[1]
        if cost is None:
            cost = torch.empty((batch,), dtype=torch.float64, device=times.device)
        lay = self.layout(batch, layout)
        cur = self.ctx._enter()
        rc = self.lib.mtg_solve_linear(self.handle, batch, ctypes.byref(lay), self._ptr(times), self._ptr(d_fixed),
                                       None, None, self._ptr(cost), L.FLAG_COST_ONLY)
        self.ctx._leave(cur)
        _check(self.lib, rc, self.ctx.handle)
        return cost

    def update_from_free(self, times, d_fixed, d_free, layout: str = "aos", want_cost: bool = False):
        """setFreeConstraints() path: coefficients from given free constraints, no solve."""
        import torch
        batch = times.shape[0] if layout == "aos" else times.shape[1]
        dev = times.device
        coeffs = torch.empty((batch, self.K, self.D, self.N), dtype=torch.float64, device=dev)
        cost = torch.empty((batch,), dtype=torch.float64, device=dev) if want_cost else None
        lay = self.layout(batch, layout)
        cur = self.ctx._enter()
        rc = self.lib.mtg_update_segments_from_free(self.handle, batch, ctypes.byref(lay), self._ptr(times),
                                                    self._ptr(d_fixed), self._ptr(d_free), self._ptr(coeffs),
                                                    self._ptr(cost), 0)
        self.ctx._leave(cur)
        _check(self.lib, rc, self.ctx.handle)
        return coeffs, cost

    def solve_host(self, times: np.ndarray, d_fixed: np.ndarray, want_free=True, want_cost=True, generic=False,
                   coeffs: Optional[np.ndarray] = None, host_backend: bool = False, basic_solution: bool = False):
        """Host-buffer convenience (AoS numpy in/out, staged through the device by the library).  Pass page-locked
        arrays (e.g. pinned torch tensors viewed as numpy, also for `coeffs`) to have them DMA'd directly."""
        times = np.ascontiguousarray(times, dtype=np.float64)
        d_fixed = np.ascontiguousarray(d_fixed, dtype=np.float64)
        batch = times.shape[0]
        if coeffs is None:
            coeffs = np.empty((batch, self.K, self.D, self.N))
        assert coeffs.dtype == np.float64 and coeffs.flags.c_contiguous and coeffs.size == batch * self.K * self.D * self.N
        d_free = np.empty((batch, self.D, self.n_free)) if want_free else None
        cost = np.empty((batch,)) if want_cost else None
        lay = self.layout(batch, "aos")
        p = lambda a: ctypes.c_void_p(0) if a is None else ctypes.c_void_p(a.ctypes.data)
        flags = L.FLAG_HOST_POINTERS | (L.FLAG_GENERIC_KERNEL if generic else 0)
        if host_backend:   # batch <= 64: the lane code's host build on this thread (latency path), no GPU involved
            flags |= L.FLAG_HOST_BACKEND
        if basic_solution:  # rank-deficient trajectories get the reference's basic solution (LIN:365-378)
            flags |= L.FLAG_BASIC_SOLUTION
        rc = self.lib.mtg_solve_linear(self.handle, batch, ctypes.byref(lay), p(times), p(d_fixed), p(coeffs),
                                       p(d_free), p(cost), flags)
        _check(self.lib, rc, self.ctx.handle)
        return coeffs, d_free, cost

    LAUNCH_FORMS = {0: "generic", 1: "fused", 2: "split", 3: "rolled", 4: "slab", 5: "dimlane", 6: "dimlane_rt", 7: "coop"}

    def launch_form(self, batch: int, layout: str = "soa", dims: str = "auto", extra_outputs: bool = False) -> str:
        """Kernel form a coefficient-only device-pointer solve of `batch` trajectories takes (mtg_plan_launch_form);
        extra_outputs: of a solve that also returns the cost / d_free."""
        lay = self.layout(batch, layout)
        flags = {"auto": 0, "fused": L.FLAG_FUSED_DIMS, "split": L.FLAG_SPLIT_DIMS, "dimlane": L.FLAG_DIMLANE, "coop": L.FLAG_COOPERATIVE}[dims]
        if extra_outputs:
            flags |= L.FLAG_QUERY_EXTRA_OUTPUTS
        rc = self.lib.mtg_plan_launch_form(self.handle, batch, ctypes.byref(lay), flags)
        if rc < 0:
            _check(self.lib, rc, self.ctx.handle)
        return self.LAUNCH_FORMS[rc]

    def time_last_solve(self, iters: int = 100) -> float:
        """Mean device duration (us) of the last solve launch, hipEvents on the context stream."""
        out = ctypes.c_double(0)
        _check(self.lib, self.lib.mtg_time_last_solve(self.handle, iters, ctypes.byref(out)), self.ctx.handle)
        return out.value

    def close(self):
        if getattr(self, "handle", None):
            if getattr(self.ctx, "handle", None):   # a plan outliving its context is already gone on the C side
                self.lib.mtg_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# numpy mirror of mtg_multi_item (include/mtg_hip.h): a request of hundreds of items is put together by concatenating
# pre-packed arrays instead of filling ctypes structures item by item (240 items: ~1 ms of Python against ~20 us)
MULTI_ITEM_DTYPE = np.dtype([("plan", np.uint64), ("batch", np.int64), ("layout", np.int64, (8,)), ("times", np.uint64),
                             ("d_fixed", np.uint64), ("coeffs", np.uint64), ("d_free", np.uint64), ("cost", np.uint64)])
assert MULTI_ITEM_DTYPE.itemsize == ctypes.sizeof(L.MultiItem)


def pack_multi_items(items: Sequence[dict]) -> np.ndarray:
    """items: dicts with plan, times, d_fixed, coeffs (CUDA tensors; layout 'aos' | 'soa') -> packed mtg_multi_item array.
    The tensors must outlive every request created from the array."""
    out = np.zeros(len(items), dtype=MULTI_ITEM_DTYPE)
    for i, it in enumerate(items):
        plan: Plan = it["plan"]
        layout = it.get("layout", "aos")
        t = it["times"]
        batch = t.shape[0] if layout == "aos" else t.shape[1]
        lay = plan.layout(batch, layout)
        out[i]["plan"] = plan.handle.value
        out[i]["batch"] = batch
        out[i]["layout"] = [lay.times_stride_b, lay.times_stride_k, lay.fixed_stride_b, lay.fixed_stride_d, lay.fixed_stride_c,
                            lay.free_stride_b, lay.free_stride_d, lay.free_stride_c]
        out[i]["times"], out[i]["d_fixed"], out[i]["coeffs"] = t.data_ptr(), it["d_fixed"].data_ptr(), it["coeffs"].data_ptr()
    return out


class PackedMultiSolve:
    """A mixed request created from a packed mtg_multi_item array (pack_multi_items / np.concatenate of such arrays):
    coefficient output only.  create_us = host time of the mtg_multi_create call itself."""

    def __init__(self, ctx: Context, packed: np.ndarray, keep=None):
        import time
        assert packed.dtype == MULTI_ITEM_DTYPE and packed.flags.c_contiguous
        self.ctx, self.lib, self.keep = ctx, ctx.lib, keep
        h = ctypes.c_void_p()
        t0 = time.perf_counter()
        rc = self.lib.mtg_multi_create(ctx.handle, len(packed), packed.ctypes.data_as(ctypes.POINTER(L.MultiItem)), 0, ctypes.byref(h))
        self.create_us = (time.perf_counter() - t0) * 1e6
        _check(self.lib, rc, ctx.handle)
        self.handle = h
        self.launch_count = self.lib.mtg_multi_launch_count(h)
        ctx._plans.add(self)

    def solve(self, ordered: bool = True):
        """Enqueue the request on the context's stream (asynchronous).  ordered: the library's stream waits for torch's current
        stream before the launch and torch's current stream for the launch afterwards, as every other wrapper does
        (ordered=False: the caller orders the streams itself -- bench.py records its events on ctx.stream and syncs)."""
        cur = self.ctx._enter() if ordered else None
        try:
            rc = self.lib.mtg_multi_solve(self.handle)
        finally:
            if ordered:
                self.ctx._leave(cur)
        _check(self.lib, rc, self.ctx.handle)

    def close(self):
        if getattr(self, "handle", None):
            if getattr(self.ctx, "handle", None):
                self.lib.mtg_multi_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiSolve:
    """A mixed request (mtg_multi_*): several (plan, batch) items created once, solved as often as wanted with new
    values in the same device tensors.  Items that share N, D, the constraint pattern and the derivative run as ONE
    kernel launch.  items: dicts with plan, times, d_fixed (CUDA tensors), optional layout ('aos' | 'soa'), coeffs,
    d_free, cost tensors (coeffs allocated here when missing; cost / d_free only when want_cost / want_free)."""

    def __init__(self, ctx: Context, items: Sequence[dict], want_cost: bool = False, want_free: bool = False,
                 dims: str = "auto", basic_solution: bool = False):
        import torch
        self.ctx, self.lib = ctx, ctx.lib
        self.items = []
        arr = (L.MultiItem * len(items))()
        for i, it in enumerate(items):
            plan: Plan = it["plan"]
            assert plan.ctx is ctx
            layout = it.get("layout", "aos")
            t, f = it["times"], it["d_fixed"]
            batch = t.shape[0] if layout == "aos" else t.shape[1]
            co = it.get("coeffs")
            if co is None:
                co = torch.empty((batch, plan.K, plan.D, plan.N), dtype=torch.float64, device=t.device)
            cost = it.get("cost")
            if cost is None and want_cost:
                cost = torch.empty((batch,), dtype=torch.float64, device=t.device)
            fr = it.get("d_free")
            if fr is None and want_free:
                shape = (batch, plan.D, plan.n_free) if layout == "aos" else (plan.D, plan.n_free, batch)
                fr = torch.empty(shape, dtype=torch.float64, device=t.device)
            arr[i].plan = plan.handle
            arr[i].batch = batch
            arr[i].layout = plan.layout(batch, layout)
            arr[i].times, arr[i].d_fixed, arr[i].coeffs = t.data_ptr(), f.data_ptr(), co.data_ptr()
            arr[i].d_free = fr.data_ptr() if fr is not None else None
            arr[i].cost = cost.data_ptr() if cost is not None else None
            self.items.append(dict(plan=plan, times=t, d_fixed=f, coeffs=co, d_free=fr, cost=cost, layout=layout))
        h = ctypes.c_void_p()
        flags = {"auto": 0, "fused": L.FLAG_FUSED_DIMS, "split": L.FLAG_SPLIT_DIMS,
                 "concurrent": L.FLAG_CONCURRENT_ITEMS}[dims]   # concurrent: one launch per item on the context's side streams
        if basic_solution:   # items of structurally rank-deficient plans run on the plan's shadow (MTG_FLAG_BASIC_SOLUTION)
            flags |= L.FLAG_BASIC_SOLUTION
        _check(self.lib, self.lib.mtg_multi_create(ctx.handle, len(items), arr, flags, ctypes.byref(h)), ctx.handle)
        self.handle = h
        self.launch_count = self.lib.mtg_multi_launch_count(h)
        ctx._plans.add(self)   # closed with the context

    def solve(self, ordered: bool = True):
        """Enqueue the whole request (asynchronous).  Returns [(coeffs, d_free, cost)] in item order."""
        cur = self.ctx._enter() if ordered else None
        rc = self.lib.mtg_multi_solve(self.handle)
        if ordered:
            self.ctx._leave(cur)
        _check(self.lib, rc, self.ctx.handle)
        return [(it["coeffs"], it["d_free"], it["cost"]) for it in self.items]

    def close(self):
        if getattr(self, "handle", None):
            if getattr(self.ctx, "handle", None):
                self.lib.mtg_multi_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def solve_linear_batch(n_coeffs, derivative, fixed_mask, times: np.ndarray, d_fixed: np.ndarray, device: int = 0,
                       generic: bool = False):
    """One-shot host convenience: numpy AoS in -> (coeffs, d_free, cost) numpy out, via the HIP path."""
    ctx = Context(device)
    dim = d_fixed.shape[1]
    plan = Plan(ctx, n_coeffs, dim, times.shape[1], derivative, fixed_mask)
    out = plan.solve_host(times, d_fixed, generic=generic)
    ctx.sync()
    plan.close()
    ctx.close()
    return out
