"""Mixed batches (BASELINE config 4): trajectories with different N / K / constraint structure are bucketed on the
host by plan key (N, D, K, d, masks); each bucket is one launch of the matching kernel variant.  Mirrors what a
caller of the reference would do with a list of independent PolynomialOptimization<N> problems."""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np

from .core import Context, Plan


class MixedBatchSolver:
    def __init__(self, ctx: Context):
        self.ctx = ctx
        self.plans: Dict[Tuple, Plan] = {}

    def plan_for(self, n_coeffs: int, dimension: int, n_segments: int, derivative: int, masks: Sequence[int]) -> Plan:
        key = (n_coeffs, dimension, n_segments, derivative, tuple(int(m) for m in masks))
        if key not in self.plans:
            self.plans[key] = Plan(self.ctx, n_coeffs, dimension, n_segments, derivative, list(key[4]))
        return self.plans[key]

    def solve(self, problems: List[dict], want_cost: bool = False):
        """problems: dicts with n_coeffs, derivative, masks [K+1], times [K], d_fixed [D][n_fixed] (host arrays).
        Returns (list of coeff arrays [K][D][N] in input order, list of costs or None)."""
        import torch
        buckets: Dict[Tuple, List[int]] = {}
        for i, p in enumerate(problems):
            t = np.asarray(p["times"], dtype=np.float64)
            f = np.asarray(p["d_fixed"], dtype=np.float64)
            key = (int(p["n_coeffs"]), f.shape[0], t.shape[0], int(p["derivative"]), tuple(int(m) for m in p["masks"]))
            buckets.setdefault(key, []).append(i)
        coeffs: List = [None] * len(problems)
        costs: List = [None] * len(problems)
        pending = []
        for key, idx in buckets.items():
            plan = self.plan_for(*key)
            t = torch.from_numpy(np.stack([np.asarray(problems[i]["times"], dtype=np.float64) for i in idx])).cuda()
            f = torch.from_numpy(np.stack([np.asarray(problems[i]["d_fixed"], dtype=np.float64) for i in idx])).cuda()
            co, _, cost = plan.solve(t, f, want_cost=want_cost)
            pending.append((idx, co, cost))
        self.ctx.sync()
        for idx, co, cost in pending:
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
