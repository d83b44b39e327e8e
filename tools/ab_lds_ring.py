"""A/B of the LDS ring of the long static chains (MtgCfg::kRing, -DMTG_LDS_RING=0/1): one library per process
(MTG_HIP_LIB=...), shapes with global workspace steps + controls, B = 100k and 2500, rotating buffer sets.
Also checks bit identity of the coefficients against a file written by the other build:
  python tools/ab_lds_ring.py write|check DIR"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mav_trajectory_generation_amd as m

mode, d = (sys.argv[1:3] + ["write", "gpurun_out/ring"][len(sys.argv) - 1:])
os.makedirs(d, exist_ok=True)
shapes = [(12, 32), (12, 24), (12, 20), (12, 17), (10, 50), (12, 16), (10, 32)]
for bsz in (100_000, 2500):
    for (n, k) in shapes:
        ctx = m.Context(0)
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
        row = {"N": n, "K": k, "B": bsz, "us": round(e0.elapsed_time(e1) * 1e3 / (reps * nsets), 2)}
        if bsz == 2500:
            fn = os.path.join(d, f"c_{n}_{k}.pt")
            got = torch.stack([s[2] for s in sets[:4]]).cpu()
            if mode == "write":
                torch.save(got, fn)
            else:
                row["bit_identical"] = bool(torch.equal(got, torch.load(fn)))
        print(json.dumps(row), flush=True)
        plan.close(); ctx.close()
