"""Kernel time of the batched extrema search / feasibility scaling (row N4), with a CPU timing of the reference's own
Jenkins-Traub root finder (oracle/_ref) on a sample for scale."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mav_trajectory_generation_amd as m
ctx = m.Context(0)
with torch.cuda.stream(ctx.stream):
    for (N, K, D, B) in ((10, 8, 3, 10_000), (10, 8, 3, 100_000), (12, 8, 3, 100_000), (10, 16, 4, 100_000)):
        masks = m.ends_full_masks(N, K)
        plan = m.Plan(ctx, N, D, K, N // 2 - 1, masks)
        t, f = m.random_waypoint_batch(B, K, D, N, masks, seed=8, device="cuda")
        co, _, _ = plan.solve(t, f)
        for der in (1, 2):
            m.minmax_magnitude(ctx, co, t, der); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(ctx.stream)
            for _ in range(5): seg, traj, idx = m.minmax_magnitude(ctx, co, t, der)
            e1.record(ctx.stream); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 5
            print(f"minmax N={N} K={K} D={D} B={B} der={der}: {us:.1f} us, {B*K/us:.1f} M segments/s")
        co0, t0 = co.clone(), t.clone()
        ws = torch.empty((8 * B * (K + 1),), dtype=torch.float64, device="cuda")
        best = 1e30
        for _ in range(3):
            co.copy_(co0); t.copy_(t0); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(ctx.stream)
            sc, within, _ = m.scale_segment_times_to_meet_constraints(ctx, co, t, 2.0, 2.0, workspace=ws)
            e1.record(ctx.stream); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3)
        print(f"scale-to-constraints (2 rounds) N={N} K={K} D={D} B={B}: {best:.1f} us = {B/best:.2f} M traj/s; "
              f"scaled {int((sc > 1).sum())}/{B}, all within: {bool(within.all())}")
        plan.close()

# CPU: the reference's root finder + candidate evaluation on a sample (single thread)
try:
    from oracle import oracle_extrema as ox
    if ox.ref_available():
        N, K, D, B = 10, 8, 3, 200
        masks = m.ends_full_masks(N, K)
        plan = m.Plan(ctx, N, D, K, 4, masks)
        t, f = m.random_waypoint_batch(B, K, D, N, masks, seed=8, device="cuda")
        co, _, _ = plan.solve(t, f); ctx.sync()
        ch, th = co.cpu().numpy(), t.cpu().numpy()
        # time only the root finding (the restatement's Python loops around it are not the reference's cost)
        polys = []
        for b in range(B):
            for k in range(K):
                conv = np.zeros(2 * (N - 1) - 2)
                for d in range(D):
                    conv += np.convolve(ox.get_coefficients(ch[b, k, d], 1)[:N - 1], ox.get_coefficients(ch[b, k, d], 2)[:N - 2])
                polys.append(conv)
        t0 = time.perf_counter()
        for p in polys: ox.find_roots(p, "ref")
        dt = time.perf_counter() - t0
        print(f"CPU reference rpoly_ak1 (1 thread, via ctypes): {dt/len(polys)*1e6:.1f} us per segment-derivative "
              f"=> {len(polys)/dt/1e6:.4f} M segments/s (root finding only)")
except Exception as e:  # noqa
    print("cpu timing skipped:", e)
