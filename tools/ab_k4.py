import sys, os, json
sys.path.insert(0, os.getcwd())
import torch
import mav_trajectory_generation_amd as m
ctx = m.Context(0)
for (N,d,K) in ((12,5,4),(10,4,4),(8,3,4)):
  masks = m.ends_full_masks(N, K, 1)
  plan = m.Plan(ctx, N, 3, K, d, masks)
  for B in (2500, 20000, 100000):
    with torch.cuda.stream(ctx.stream):
        t, f = m.random_waypoint_batch(B, K, 3, N, masks, seed=11, device="cuda", layout="soa")
        co = torch.empty((B, K, 3, N), dtype=torch.float64, device="cuda")
        for dims in ("auto","dimlane","fused"):
            os.environ.pop("MTG_NO_SLAB", None)
            if dims=="fused": os.environ["MTG_NO_SLAB"]="1"
            plan.solve(t, f, layout="soa", coeffs=co, dims=dims)
            torch.cuda.synchronize(); ctx.sync()
            us = plan.time_last_solve(30)
            print(N,K,B,dims,plan.kernel_variant, round(us,2), round(B*plan.bytes_per_trajectory/us*1e-3/8000,3))
