"""Throughput of the other BASELINE.json configs on one GPU (kernel time via hipEvents inside the library):
config 4 = mixed-N buckets (N=8 jerk / N=10 snap / N=12, K in {4,8,16,32}, D=3), config 5 = K=16, D=4 with
velocity+acceleration constraints at interior vertices, plus the large-batch points of config 2/3."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mav_trajectory_generation_amd as m

ctx = m.Context(0)
rows = []

def run(name, N, K, D, d, interior, B, layout="soa", dims="auto", yaw=False):
    masks = m.ends_full_masks(N, K, interior)
    plan = m.Plan(ctx, N, D, K, d, masks)
    with torch.cuda.stream(ctx.stream):
        t, f = m.random_waypoint_batch(B, K, D, N, masks, seed=11, device="cuda", layout=layout, yaw_dim=yaw)
        co = torch.empty((B, K, D, N), dtype=torch.float64, device="cuda")
        plan.solve(t, f, layout=layout, coeffs=co, dims=dims)
        torch.cuda.synchronize()
        ctx.sync()
        us = plan.time_last_solve(30)
    bpt = plan.bytes_per_trajectory
    r = dict(config=name, N=N, K=K, D=D, d=d, B=B, variant=plan.kernel_variant, kernel_us=round(us, 2),
             traj_per_s=B / us * 1e6, GBps=B * bpt / us * 1e-3, frac_8TBps=B * bpt / us * 1e-3 / 8000.0, bytes_per_traj=bpt)
    rows.append(r)
    print(json.dumps(r))
    plan.close()

if len(sys.argv) > 1 and sys.argv[1] == "heavy":
    tag = "rolled" if os.environ.get("MTG_PREFER_ROLLED") else "static"
    for B in (12_500, 100_000):
        run("config5-" + tag, 10, 16, 4, 4, 7, B, yaw=True)
        run("N12K8-" + tag, 12, 8, 3, 5, 1, B)
        run("N10K8-" + tag, 10, 8, 3, 4, 1, B)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "long":
    for (N, d) in ((8, 3), (10, 4), (12, 5)):
        for K in (16, 32):
            run("long-dg" + os.environ.get("MTG_FORCE_DG", "auto"), N, K, 3, d, 1, 100_000)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "config5":
    for B in (12_500, 100_000):
        run("config5-dg" + os.environ.get("MTG_FORCE_DG", "auto"), 10, 16, 4, 4, 7, B, yaw=True)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "split-ab":
    for dims in ("fused", "split"):
        run("config5-" + dims, 10, 16, 4, 4, 7, 100_000, yaw=True, dims=dims)
        run("config4-N12K8-" + dims, 12, 8, 3, 5, 1, 100_000, dims=dims)
        run("config4-N8K8-" + dims, 8, 8, 3, 3, 1, 100_000, dims=dims)
        run("config4-N12K8-" + dims, 12, 8, 3, 5, 1, 2500, dims=dims)
    sys.exit(0)
for B in (10_000, 125_000):
    run("config2/3", 10, 8, 3, 4, 1, B)
# config 4: 30k trajectories = 12 buckets x 2500
for (N, d) in ((8, 3), (10, 4), (12, 5)):
    for K in (4, 8, 16, 32):
        run("config4", N, K, 3, d, 1, 2500)
for (N, d) in ((8, 3), (10, 4), (12, 5)):
    for K in (8, 32):
        run("config4-large", N, K, 3, d, 1, 100_000)
run("config5", 10, 16, 4, 4, 7, 12_500, yaw=True)
run("config5", 10, 16, 4, 4, 7, 100_000, yaw=True)
json.dump(rows, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "configs.json"), "w"), indent=1)
