"""Kernel time of the batched sampler: B trajectories x S samples x 5 derivatives x D."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mav_trajectory_generation_amd as m
ctx = m.Context(0)
masks = m.ends_full_masks(10, 8)
plan = m.Plan(ctx, 10, 3, 8, 4, masks)
with torch.cuda.stream(ctx.stream):
    for B, S in ((10_000, 100), (100_000, 100), (100_000, 1000)):
        t, f = m.random_waypoint_batch(B, 8, 3, 10, masks, seed=8, device="cuda")
        co, _, _ = plan.solve(t, f)
        dt = float(t.sum(dim=1).max()) / S
        m.sample_range(ctx, co, t, 0.0, dt, S, 5); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(ctx.stream)
        for _ in range(10): out = m.sample_range(ctx, co, t, 0.0, dt, S, 5)
        e1.record(ctx.stream); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 10
        bytes_ = B * (S * 15 * 8 + 1920 + 64)
        print(f"sample B={B} S={S}: {us:.1f} us, {B*S/us:.0f} M samples/s, {bytes_/us*1e-3:.0f} GB/s = {bytes_/us*1e-3/80:.1f}% of 8 TB/s")

# config 5 shape: K = 16, D = 4 (x, y, z, yaw)
masks = m.ends_full_masks(10, 16, 7)
plan5 = m.Plan(ctx, 10, 4, 16, 4, masks)
with torch.cuda.stream(ctx.stream):
    for B, S in ((12_500, 100), (100_000, 200)):
        t, f = m.random_waypoint_batch(B, 16, 4, 10, masks, seed=8, device="cuda", yaw_dim=True)
        co, _, _ = plan5.solve(t, f)
        dt = float(t.sum(dim=1).max()) / S
        m.sample_range(ctx, co, t, 0.0, dt, S, 5); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(ctx.stream)
        for _ in range(10): out = m.sample_range(ctx, co, t, 0.0, dt, S, 5)
        e1.record(ctx.stream); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 10
        bytes_ = B * (S * 20 * 8 + 16 * 4 * 10 * 8 + 128)
        print(f"sample K=16 D=4 B={B} S={S}: {us:.1f} us, {B*S/us:.0f} M samples/s, {bytes_/us*1e-3:.0f} GB/s = {bytes_/us*1e-3/80:.1f}% of 8 TB/s")
