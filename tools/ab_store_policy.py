"""A/B of the coefficient store policy of the long-chain dimension-in-lane kernels (context option dl_policy: 0 nt sc1, 1 sc1,
2 write-back; needs a library built with -DMTG_DL_ALL_POLICIES: MTG_HIP_LIB=.../libmtg_hip_pol.so).  Fresh context, plan and
buffers per setting; rotating over several buffer sets."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mav_trajectory_generation_amd as m

shapes = [(12, 32), (12, 16), (10, 32), (10, 16), (8, 32), (8, 16), (10, 8), (12, 8)]
for bsz in (100_000, 2500):
    for (n, k) in shapes:
        row = {"N": n, "K": k, "B": bsz}
        for pol in (0, 1, 2, 0):
            ctx = m.Context(0)
            ctx.set_option("dl_policy", pol)
            masks = m.ends_full_masks(n, k, 1)
            plan = m.Plan(ctx, n, 3, k, n // 2 - 1, masks)
            nsets = 6 if bsz >= 100_000 else 24
            with torch.cuda.stream(ctx.stream):
                sets = []
                for s in range(nsets):
                    t, f = m.random_waypoint_batch(bsz, k, 3, n, masks, seed=11 + s, device="cuda", layout="soa")
                    sets.append((t, f, torch.empty((bsz, k, 3, n), dtype=torch.float64, device="cuda")))
                for _ in range(2):
                    for (t, f, co) in sets:
                        plan.solve(t, f, layout="soa", coeffs=co, dims="dimlane")
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 4
                e0.record(ctx.stream)
                for _ in range(reps):
                    for (t, f, co) in sets:
                        plan.solve(t, f, layout="soa", coeffs=co, dims="dimlane")
                e1.record(ctx.stream)
                torch.cuda.synchronize()
            ctx.sync()
            us = e0.elapsed_time(e1) * 1e3 / (reps * nsets)
            key = f"policy{pol}" + ("_again" if f"policy{pol}" in row else "")
            row[key] = round(us, 2)
            plan.close(); ctx.close()
        print(json.dumps(row), flush=True)
