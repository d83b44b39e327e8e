"""Long-chain two-waves-per-SIMD twins (MTG_DLO2, lab option dl_occ2 = 1) against the default one-wave kernels: kernel time at
B = 30k / 100k, results compared.  One fresh context per variant (process warm-up would favour whichever runs second)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mav_trajectory_generation_amd as m
for n in (10, 12, 8):
    K = 32
    masks = m.ends_full_masks(n, K, 1)
    for B in (30_000, 100_000):
        res, outs = {}, {}
        for rep in range(2):
            for occ2 in (0, 1):
                ctx = m.Context(0)
                ctx.set_option("dl_occ2", occ2)
                plan = m.Plan(ctx, n, 3, K, n // 2 - 1, masks)
                with torch.cuda.stream(ctx.stream):
                    t, f = m.random_waypoint_batch(B, K, 3, n, masks, seed=11, device="cuda", layout="soa")
                    co = torch.empty((B, K, 3, n), dtype=torch.float64, device="cuda")
                    torch.cuda.synchronize()
                    plan.solve(t, f, layout="soa", coeffs=co)
                    torch.cuda.synchronize(); ctx.sync()
                    us = min(plan.time_last_solve(20) for _ in range(3))
                res.setdefault(occ2, []).append(round(us, 1))
                outs[occ2] = co
                plan.close(); ctx.close()
        den = outs[0].abs().amax(dim=-1, keepdim=True).clamp_min(1e-300)
        rel = float(((outs[0] - outs[1]).abs() / den).max())
        bpt = 8 * (K + 3 * (n + K - 1) + K * 3 * n)
        print(json.dumps(dict(N=n, K=K, B=B, one_wave_us=res[0], two_waves_us=res[1], ratio=round(min(res[1]) / min(res[0]), 3),
                              frac_two_waves=round(B * bpt / min(res[1]) * 1e-3 / 8000.0, 3), max_rel_diff=rel)), flush=True)
