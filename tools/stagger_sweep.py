"""Phase-stagger experiment (measurement knob "dl_stagger", include/mtg_hip_lab.h): every second workgroup of a dimension-in-lane
launch starts n x 2048 cycles late.  Persistent one-wave-per-SIMD workgroups of equal work start together and stay phase-locked
(everyone in the memory-silent forward phase, then everyone streaming coefficients): launch time ~ issue time + memory time
(profiles/r03d_long_pmc.json).  Does de-phasing them overlap the two?
CAUTION (found in round 4): the settings run one after the other in ONE process, and the same kernel gets ~10 % faster over its
first ~200 launches -- the apparent gain of the middle settings is that warm-up; tools/stagger_check.py (fresh context, plan and
buffers per setting) shows no effect."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mav_trajectory_generation_amd as m

ctx = m.Context(0)
CASES = [(10, 16, 3, 4, 1, 100_000, "auto"), (8, 16, 3, 3, 1, 100_000, "auto"), (10, 8, 3, 4, 1, 125_000, "dimlane"),
         (12, 16, 3, 5, 1, 100_000, "auto"), (10, 32, 3, 4, 1, 100_000, "auto"), (12, 32, 3, 5, 1, 100_000, "auto"),
         (10, 16, 4, 4, 7, 100_000, "auto"), (10, 8, 3, 4, 1, 10_000, "dimlane")]
for (N, K, D, d, mi, B, dims) in CASES:
    masks = m.ends_full_masks(N, K, mi)
    plan = m.Plan(ctx, N, D, K, d, masks)
    with torch.cuda.stream(ctx.stream):
        t, f = m.random_waypoint_batch(B, K, D, N, masks, seed=11, device="cuda", layout="soa", yaw_dim=(D == 4))
        co = torch.empty((B, K, D, N), dtype=torch.float64, device="cuda")
        base = None
        for stag in (0, 1, 2, 3, 4, 6, 8, 12, 16, 24):
            ctx.set_option("dl_stagger", stag)
            plan.solve(t, f, layout="soa", coeffs=co, dims=dims)
            torch.cuda.synchronize()
            ctx.sync()
            us = plan.time_last_solve(20)
            base = base or us
            print(json.dumps(dict(N=N, K=K, D=D, B=B, form=plan.launch_form(B, "soa", dims), stagger_x2048_cycles=stag, kernel_us=round(us, 2),
                                  vs_no_stagger=round(us / base, 3), frac=round(B * plan.bytes_per_trajectory / us * 1e-3 / 8000, 3))), flush=True)
    ctx.set_option("dl_stagger", 0)
    plan.close()
