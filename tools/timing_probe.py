"""Measurement-only: reads the per-wave phase timestamps written by the MTG_TIMING build (via the cost buffer)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import numpy as np, torch
import mav_trajectory_generation_amd as m
B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
dims = sys.argv[2] if len(sys.argv) > 2 else "fused"
masks = m.ends_full_masks(10, 8)
ctx = m.Context(0)
plan = m.Plan(ctx, 10, 3, 8, 4, masks)
with torch.cuda.stream(ctx.stream):
    t, f = m.random_waypoint_batch(B, 8, 3, 10, masks, seed=5, device="cuda", layout="soa")
    co = torch.empty((B, 8, 3, 10), dtype=torch.float64, device="cuda")
    dbg = torch.zeros((1 << 20,), dtype=torch.int64, device="cuda")
    plan.lib.mtg_plan_set_workspace(plan.handle, ctypes.c_void_p(dbg.data_ptr()), dbg.numel() * 8)
    for _ in range(3):
        plan.solve(t, f, layout="soa", coeffs=co, dims=dims)
    torch.cuda.synchronize()
    dbg.zero_()
    plan.solve(t, f, layout="soa", coeffs=co, dims=dims)
    torch.cuda.synchronize()
raw = dbg.cpu().numpy()
ntiles = (B + 63) // 64
ng = 3 if dims == "split" else 1
nw = min(ntiles, 2048 // ng) * ng * 2
T = raw[: nw * 16].reshape(nw, 16)
T = T[T[:, 0] > 0]
t0 = T[:, 0].min()
names = ["start", "fwd_done", "after_bar1", "finish_done", "stores_done", "preload_done", "mid_done", "bwd0", "bwd1", "bwd2", "bwd3"]
print(f"B={B} dims={dims} waves={len(T)}")
for i, n in enumerate(names):
    v = T[:, i] - t0
    print(f"  {n:14s} min {v.min():8d}  median {int(np.median(v)):8d}  max {v.max():8d}")
d = lambda a, b: int(np.median(T[:, a] - T[:, b]))
print("  medians: preload %d | forward %d | barrier %d | mid %d | bwd3 %d | bwd2 %d | bwd1 %d | bwd0 %d | tail %d | store drain %d"
      % (d(5, 0), d(1, 5), d(2, 1), d(6, 2), d(10, 6), d(9, 10), d(8, 9), d(7, 8), d(3, 7), d(4, 3)))
