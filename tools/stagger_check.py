"""Does the DEFAULT phase stagger of the workspace hybrids take effect?  option -1 (default) vs 0 vs 8, fresh plan per setting."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mav_trajectory_generation_amd as m
for (N, K, d) in ((10, 32, 4), (12, 16, 5), (12, 32, 5)):
    masks = m.ends_full_masks(N, K, 1)
    for opt in (-1, 0, 8, -1, 0, 8):
        ctx = m.Context(0)
        ctx.set_option("dl_stagger", opt)
        plan = m.Plan(ctx, N, 3, K, d, masks)
        with torch.cuda.stream(ctx.stream):
            t, f = m.random_waypoint_batch(100_000, K, 3, N, masks, seed=11, device="cuda", layout="soa")
            co = torch.empty((100_000, K, 3, N), dtype=torch.float64, device="cuda")
            plan.solve(t, f, layout="soa", coeffs=co)
            torch.cuda.synchronize(); ctx.sync()
            us = plan.time_last_solve(20)
        print(json.dumps(dict(N=N, K=K, option=opt, form=plan.launch_form(100_000), kernel_us=round(us, 1))), flush=True)
        plan.close(); ctx.close()
        del t, f, co
