"""Kernel time of the setFreeConstraints path (mtg_update_segments_from_free) via torch events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mav_trajectory_generation_amd as m
ctx = m.Context(0)
masks = m.ends_full_masks(10, 8)
plan = m.Plan(ctx, 10, 3, 8, 4, masks)
import itertools
for B, lay in itertools.product((125_000, 1_000_000), ("aos", "soa")):
    with torch.cuda.stream(ctx.stream):
        t, f = m.random_waypoint_batch(B, 8, 3, 10, masks, seed=8, device="cuda", layout=lay)
        co, fr, _ = plan.solve(t, f, want_free=True, layout=lay)
        torch.cuda.synchronize()
        for _ in range(3):
            plan.update_from_free(t, f, fr, layout=lay)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(ctx.stream)
        for _ in range(20):
            plan.update_from_free(t, f, fr, layout=lay)
        e1.record(ctx.stream)
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
    bytes_ = B * 8 * (8 + 3 * 17 + 3 * 28 + 240)
    print(f"update_from_free {lay} B={B}: {us:.1f} us/call (incl. output alloc), {B/us:.1f} M traj/s, {bytes_/us*1e-3:.0f} GB/s = {bytes_/us*1e-3/80:.1f}% of 8 TB/s")
