"""Mixed batches (BASELINE config 4): trajectories with different N / K / constraint structure are bucketed on the
host by plan key (N, D, K, d, masks).  Mirrors what a caller of the reference would do with a list of independent
PolynomialOptimization<N> problems.

A bucket of a few thousand trajectories fills only a fraction of the 256 CUs, and twelve back-to-back launches each pay
their own latency chain.  The fast path is `MixedBatchSolver.merged(buckets)`: ONE library call per request
(mtg_multi_*), which runs every bucket with canonical SoA or AoS inputs inside one cross-structure kernel launch (config 4, 30k
trajectories: 63-65 us against ~320 us for per-bucket launches from Python).  `solve_device` / `capture` spread per-bucket
launches over `n_streams` HIP streams (one library context per stream, forked from / joined onto the caller's stream,
longest chains first) -- kept for requests the merged path does not cover, but on this runtime kernels of different
streams overlap two at a time at best and every fork-join costs ~23 us (profiles/r02_stream_overlap_microbench.txt)."""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np

from .core import Context, MultiSolve, Plan


class MixedBatchSolver:
    def __init__(self, ctx: Context, n_streams: int = 4):
        self.ctx = ctx
        # lane 0 is the caller's context; further lanes own their own stream (created lazily: n_streams = 1 keeps the
        # plain one-stream behaviour)
        self.lanes: List[Context] = [ctx]
        self.n_streams = max(1, int(n_streams))
        self.plans: Dict[Tuple, Plan] = {}

    def _lane(self, i: int) -> Context:
        while len(self.lanes) < min(self.n_streams, i + 1):
            self.lanes.append(Context(self.ctx.device))
        return self.lanes[i % self.n_streams]

    def plan_for(self, n_coeffs: int, dimension: int, n_segments: int, derivative: int, masks: Sequence[int],
                 lane: int = 0) -> Plan:
        key = (n_coeffs, dimension, n_segments, derivative, tuple(int(m) for m in masks), lane % self.n_streams)
        if key not in self.plans:
            self.plans[key] = Plan(self._lane(lane), n_coeffs, dimension, n_segments, derivative, list(key[4]))
        return self.plans[key]

    def solve_device(self, buckets: List[dict], want_cost: bool = False, _capturing: bool = False):
        """Device-resident mixed batch.  buckets: dicts with n_coeffs, derivative, masks and CUDA tensors times /
        d_fixed in `layout` ('aos' default; all the bucket's trajectories share the plan).  Asynchronous: results are
        ordered on torch's current stream.  Returns [(coeffs [B][K][D][N], cost [B] or None)] in bucket order."""
        import torch
        cur = torch.cuda.current_stream(self.ctx.device)
        order = sorted(range(len(buckets)),
                       key=lambda i: -(buckets[i]["n_coeffs"] ** 2) * len(buckets[i]["masks"]))   # longest first
        jobs = []
        for lane, i in enumerate(order):
            b = buckets[i]
            layout = b.get("layout", "aos")
            t, f = b["times"], b["d_fixed"]
            k = len(b["masks"]) - 1
            dim = f.shape[1] if layout == "aos" else f.shape[0]
            batch = t.shape[0] if layout == "aos" else t.shape[1]
            plan = self.plan_for(int(b["n_coeffs"]), dim, k, int(b["derivative"]), b["masks"], lane)
            co = torch.empty((batch, k, dim, plan.N), dtype=torch.float64, device=t.device)
            cost = torch.empty((batch,), dtype=torch.float64, device=t.device) if want_cost else None
            jobs.append((i, plan, t, f, layout, co, cost))
        used = {id(j[1].ctx): j[1].ctx for j in jobs}
        for c in used.values():          # fork: every lane starts after the work already queued by the caller
            if c.stream != cur:
                c.stream.wait_stream(cur)
        out = [None] * len(buckets)
        for i, plan, t, f, layout, co, cost in jobs:
            plan.solve(t, f, layout=layout, coeffs=co, cost=cost, want_cost=want_cost, ordered=False)
            out[i] = (co, cost)
            if not _capturing:
                for x in (t, f, co, cost):   # tensors allocated on the caller's stream, used on the lane's
                    if x is not None:
                        x.record_stream(plan.ctx.stream)
        for c in used.values():          # join
            if c.stream != cur:
                cur.wait_stream(c.stream)
        return out

    def capture(self, buckets: List[dict], want_cost: bool = False):
        """Capture the whole mixed batch -- every bucket's launch, forked over the lanes' streams and joined -- into
        ONE hipGraph (torch.cuda.CUDAGraph): a request made of many small launches is bound by the host's per-launch
        cost, not by the device; a replay is a single hipGraphLaunch.  The graph reads the buckets' input tensors in
        place (refill them, then `graph.replay()`) and writes the returned output tensors.  This is what a time
        optimiser does with a mixed population: same structure, new segment times, every iteration.
        Returns (graph, [(coeffs, cost)] in bucket order)."""
        import torch
        self.solve_device(buckets, want_cost)          # warm-up: plans, lane contexts, rolled-kernel workspaces
        torch.cuda.synchronize(self.ctx.device)
        self.sync()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = self.solve_device(buckets, want_cost, _capturing=True)
        return graph, out

    def merged(self, buckets: List[dict], want_cost: bool = False, dims: str = "auto") -> "MergedRequest":
        """The same request with buckets that share N, D, the constraint pattern and the derivative merged into ONE
        kernel launch each (mtg_multi_*): see MergedRequest."""
        return MergedRequest(self, buckets, want_cost, dims)

    def concurrent(self, buckets: List[dict], want_cost: bool = False) -> "ConcurrentRequest":
        """The same request as ONE library call (mtg_multi_* with MTG_FLAG_CONCURRENT_ITEMS): every bucket runs as its
        own best launch (dimension-in-lane / static kernels), the launches spread over the context's side streams inside
        the library -- no per-bucket Python or stream bookkeeping on the host: see ConcurrentRequest."""
        return ConcurrentRequest(self, buckets, want_cost)

    def sync(self):
        for c in self.lanes:
            c.sync()

    def solve(self, problems: List[dict], want_cost: bool = False):
        """problems: dicts with n_coeffs, derivative, masks [K+1], times [K], d_fixed [D][n_fixed] (host arrays).
        Returns (list of coeff arrays [K][D][N] in input order, list of costs or None)."""
        import torch
        groups: Dict[Tuple, List[int]] = {}
        for i, p in enumerate(problems):
            t = np.asarray(p["times"], dtype=np.float64)
            f = np.asarray(p["d_fixed"], dtype=np.float64)
            key = (int(p["n_coeffs"]), f.shape[0], t.shape[0], int(p["derivative"]), tuple(int(m) for m in p["masks"]))
            groups.setdefault(key, []).append(i)
        buckets, index = [], []
        for key, idx in groups.items():
            t = torch.from_numpy(np.stack([np.asarray(problems[i]["times"], dtype=np.float64) for i in idx])).cuda()
            f = torch.from_numpy(np.stack([np.asarray(problems[i]["d_fixed"], dtype=np.float64) for i in idx])).cuda()
            buckets.append(dict(n_coeffs=key[0], derivative=key[3], masks=list(key[4]), times=t, d_fixed=f))
            index.append(idx)
        results = self.solve_device(buckets, want_cost=want_cost)
        torch.cuda.current_stream(self.ctx.device).synchronize()
        self.sync()   # raises if any bucket flagged a bad segment time / singular system
        coeffs: List = [None] * len(problems)
        costs: List = [None] * len(problems)
        for idx, (co, cost) in zip(index, results):
            co = co.cpu().numpy()
            cj = cost.cpu().numpy() if cost is not None else None
            for j, i in enumerate(idx):
                coeffs[i] = co[j]
                if cj is not None:
                    costs[i] = float(cj[j])
        return coeffs, (costs if want_cost else None)

    def close(self):
        for p in self.plans.values():
            p.close()
        self.plans.clear()
        for c in self.lanes[1:]:
            c.close()
        self.lanes = self.lanes[:1]


class ConcurrentRequest:
    """A mixed request enqueued by one C call: per bucket the launch the single-plan path would choose, longest chains
    first, on up to four streams owned by the library context (fork from / join onto the context's stream).  A bucket of
    2500 trajectories occupies 60-120 of the 256 CUs, so the buckets overlap; the request takes about as long as its
    longest bucket plus the host's enqueue time.  Created once for a set of device tensors; `solve()` re-solves with
    whatever values those tensors hold."""

    def __init__(self, solver: MixedBatchSolver, buckets: List[dict], want_cost: bool = False):
        self.solver = solver
        items = []
        for b in buckets:
            layout = b.get("layout", "aos")
            dim = b["d_fixed"].shape[1] if layout == "aos" else b["d_fixed"].shape[0]
            plan = solver.plan_for(int(b["n_coeffs"]), dim, len(b["masks"]) - 1, int(b["derivative"]), b["masks"], 0)
            items.append(dict(plan=plan, times=b["times"], d_fixed=b["d_fixed"], layout=layout, coeffs=b.get("coeffs")))   # (caller-owned output tensor, optional)
        self.multi = MultiSolve(solver.ctx, items, want_cost=want_cost, dims="concurrent")
        self.out = [(it["coeffs"], it["cost"]) for it in self.multi.items]
        self.launch_count = self.multi.launch_count

    def solve(self):
        """Asynchronous; results ordered on torch's current stream.  Returns [(coeffs, cost)] in bucket order."""
        self.multi.solve()
        return self.out

    def close(self):
        self.multi.close()


class MergedRequest:
    """A mixed request as merged launches (C ABI: mtg_multi_create / _solve): buckets of equal structure-up-to-K run as
    one launch, and buckets whose structures differ only in N (the library's standard N = 8 / 10 / 12 shapes) join ONE
    cross-structure launch -- BASELINE config 4 is a single launch instead of 12, every tile of every bucket in flight
    at once, so the request takes about as long as its longest chain.  (Streams are no substitute on this runtime:
    kernels of different streams overlap two at a time at best, profiles/r02_stream_overlap_microbench.txt.)  Created
    once for a set of device tensors; `solve()` re-solves with whatever values those tensors hold; `capture()` wraps
    that in one hipGraph."""

    def __init__(self, solver: MixedBatchSolver, buckets: List[dict], want_cost: bool = False, dims: str = "auto"):
        self.solver = solver
        items = []
        for b in buckets:
            layout = b.get("layout", "aos")
            dim = b["d_fixed"].shape[1] if layout == "aos" else b["d_fixed"].shape[0]
            plan = solver.plan_for(int(b["n_coeffs"]), dim, len(b["masks"]) - 1, int(b["derivative"]), b["masks"], 0)
            items.append(dict(plan=plan, times=b["times"], d_fixed=b["d_fixed"], layout=layout, coeffs=b.get("coeffs")))   # (caller-owned output tensor, optional)
        # launch geometry (dims = 'auto'): the library looks at the whole request
        ms = MultiSolve(solver.ctx, items, want_cost=want_cost, dims=dims)
        self.multis: List[MultiSolve] = [ms]
        self.out: List = [(it["coeffs"], it["cost"]) for it in ms.items]
        self.launch_count = ms.launch_count

    def solve(self):
        """Asynchronous; results ordered on torch's current stream.  Returns [(coeffs, cost)] in bucket order."""
        import torch
        cur = torch.cuda.current_stream(self.solver.ctx.device)
        for ms in self.multis:
            if ms.ctx.stream != cur:
                ms.ctx.stream.wait_stream(cur)
        for ms in self.multis:
            ms.solve(ordered=False)
        for ms in self.multis:
            if ms.ctx.stream != cur:
                cur.wait_stream(ms.ctx.stream)
        return self.out

    def capture(self):
        """One hipGraph for the whole request (fork, the merged launches, join)."""
        import torch
        self.solve()
        torch.cuda.synchronize(self.solver.ctx.device)
        self.solver.sync()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self.solve()
        return graph

    def close(self):
        for ms in self.multis:
            ms.close()
        self.multis = []
