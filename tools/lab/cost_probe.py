"""cost(solve) against cost(update path on the solve's own d_P) for the low-derivative fuzz cases (lab probe)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
import mav_trajectory_generation_amd as m
ctx = m.Context(0)
for (n, k, dim, d, interior, bsz, layout, seed) in ((10, 8, 3, 1, 1, 65, "aos", 50 * 22), (10, 5, 3, 2, 7, 700, "soa", 50 * 38), (10, 7, 3, 1, 7, 700, "soa", 50 * 59),
                                                   (10, 8, 3, 4, 1, 700, "soa", 7), (12, 8, 3, 5, 1, 700, "soa", 7), (8, 8, 3, 3, 1, 700, "soa", 7)):
    masks = m.ends_full_masks(n, k, interior)
    plan = m.Plan(ctx, n, dim, k, d, masks)
    t, f = m.random_waypoint_batch(bsz, k, dim, n, masks, seed=seed, device="cuda", layout=layout)
    rc, rf, rj = plan.solve(t, f, layout=layout, want_free=True, want_cost=True, generic=True)
    cu, ju = plan.update_from_free(t, f, rf, layout=layout, want_cost=True)
    ctx.sync()
    rel = ((ju - rj).abs() / rj.abs()).max().item()
    relc, _ = ctx.compare_coefficients(cu, rc)
    print(f"N={n} K={k} d={d} mi={interior}: cost(update) vs cost(solve) max rel {rel:.3e}; coefficients {relc:.3e}")
    plan.close()
