"""Cost of MTG_FLAG_REFINE (one refinement step, residual in double-double: csrc/mtg_refine.hip): wall time of the whole call sequence
against the plain solve, N = 12 (and N = 10 / K = 8) at B = 100k and 2500."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mav_trajectory_generation_amd as m
ctx = m.Context(0)
for (N, K, d) in ((12, 8, 5), (12, 16, 5), (12, 32, 5), (10, 8, 4)):
    masks = m.ends_full_masks(N, K, 1)
    plan = m.Plan(ctx, N, 3, K, d, masks)
    for B in (2500, 100_000):
        with torch.cuda.stream(ctx.stream):
            t, f = m.random_waypoint_batch(B, K, 3, N, masks, seed=11, device="cuda", layout="soa")
            co = torch.empty((B, K, 3, N), dtype=torch.float64, device="cuda")
            out = {}
            for refine in (False, True):
                for _ in range(3):
                    plan.solve(t, f, layout="soa", coeffs=co, refine=refine)
                torch.cuda.synchronize(); ctx.sync()
                t0 = time.perf_counter()
                for _ in range(10):
                    plan.solve(t, f, layout="soa", coeffs=co, refine=refine)
                torch.cuda.synchronize()
                out["refined_us" if refine else "plain_us"] = (time.perf_counter() - t0) / 10 * 1e6
        print(json.dumps(dict(N=N, K=K, B=B, **{k: round(v, 1) for k, v in out.items()}, ratio=round(out["refined_us"] / out["plain_us"], 2))))
    plan.close()
