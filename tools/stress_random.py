"""Randomised parity stress (GPU box): random shapes / masks / batch sizes / layouts, HIP path vs the compiled reference
(oracle/_ref/libmtg_ref.so) for the solve, the merged mixed request, sampling and velocity extrema.  Not part of the
test suite (minutes of run time); prints the worst relative errors seen.  Oracle-side tool."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import helpers
from oracle import ref_linear
import mav_trajectory_generation_amd as m

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = m.Context(0)
worst = dict(solve=0.0, cost=0.0, merged=0.0, sample=0.0, extrema=0.0)
t0, cases = time.time(), 0
while time.time() - t0 < budget:
    n = int(rng.choice([6, 8, 10, 10, 10]))
    h = n // 2
    k = int(rng.choice([1, 2, 3, 5, 8, 8, 13, 16, 24]))
    dim = int(rng.choice([1, 2, 3, 3, 4]))
    bsz = int(rng.choice([1, 7, 64, 65, 200, 700]))
    style = rng.integers(0, 3)
    if style == 0:
        masks = None                                         # ends full, interior position only
    elif style == 1:
        masks = [(1 << h) - 1] + [int(rng.choice([1, 3, 7 & ((1 << h) - 1)]))] * (k - 1) + [(1 << h) - 1]
    else:                                                    # ragged: random per-vertex masks, position always fixed
        masks = [(1 << h) - 1] + [1 | int(rng.integers(0, 1 << h)) for _ in range(k - 1)] + [(1 << h) - 1]
    masks, times, d_fixed = helpers.reference_batch(bsz, k, n, dim, int(rng.integers(1, 1 << 30)), masks)
    d = h - 1
    ref_c, ref_f, ref_j, _ = ref_linear.solve_batch(n, d, masks, times, d_fixed, nthreads=8)
    plan = m.Plan(ctx, n, dim, k, d, masks)
    layout = "soa" if rng.integers(0, 2) else "aos"
    t = torch.from_numpy(times).cuda(); f = torch.from_numpy(d_fixed).cuda()
    tt, ff = (t.t().contiguous(), f.permute(1, 2, 0).contiguous()) if layout == "soa" else (t, f)
    co, _, cost = plan.solve(tt, ff, layout=layout, want_cost=True, dims=str(rng.choice(["auto", "fused", "split"])))
    ctx.sync()
    e_solve = helpers.poly_relerr(co.cpu().numpy(), ref_c)
    if e_solve > worst["solve"] and os.path.isdir("gpurun_out"):
        np.savez("gpurun_out/stress_worst.npz", n=n, d=d, masks=np.array(masks), times=times, d_fixed=d_fixed,
                 hip=co.cpu().numpy(), ref=ref_c)
    worst["solve"] = max(worst["solve"], e_solve)
    worst["cost"] = max(worst["cost"], float(np.abs(cost.cpu().numpy() / ref_j - 1).max()))
    if k >= 2:   # the same plan twice in one merged request
        ms = m.MultiSolve(ctx, [dict(plan=plan, times=tt, d_fixed=ff, layout=layout)] * 2)
        out = ms.solve(); ctx.sync()
        worst["merged"] = max(worst["merged"], helpers.poly_relerr(out[1][0].cpu().numpy(), ref_c))
        ms.close()
    S = int(rng.choice([5, 33, 64, 100]))
    dt_s = float(times.sum(axis=1).min()) / S
    smp = m.sample_range(ctx, torch.from_numpy(ref_c).cuda(), t, 0.0, dt_s, S, 3).cpu().numpy()
    grid = dt_s * np.arange(S)
    for b in range(min(bsz, 3)):
        for der in range(3):
            want = ref_linear.evaluate(ref_c[b], times[b], grid, der)
            sc = max(1e-2 * np.abs(ref_c[b]).max(), np.abs(want).max())
            worst["sample"] = max(worst["sample"], float(np.abs(smp[b, :, der] - want).max() / sc))
    if n >= 6:
        seg, traj, idx = m.minmax_magnitude(ctx, torch.from_numpy(ref_c).cuda(), t, 1)
        ctx.sync()
        for b in range(min(bsz, 3)):
            mn, mx, per = ref_linear.minmax_magnitude(ref_c[b], times[b], 1)
            if mx[1] > 1e-6:
                worst["extrema"] = max(worst["extrema"], abs(float(traj[b, 3]) - mx[1]) / mx[1])
    plan.close()
    cases += 1
print("cases", cases, {k_: float("%.2e" % v) for k_, v in worst.items()})
# the worst solve case, arbitrated: its worst trajectory against the 50-digit solution (oracle/oracle_mp.py)
if os.path.exists("gpurun_out/stress_worst.npz"):
    from oracle import oracle_mp
    w = np.load("gpurun_out/stress_worst.npz")
    hip, ref = w["hip"], w["ref"]
    num = np.abs(hip - ref).max(axis=-1); den = np.maximum(np.abs(ref).max(axis=-1), 1e-300)
    b = int(np.unravel_index(np.argmax(num / den), num.shape)[0])
    exact = np.array(oracle_mp.solve(int(w["n"]), int(w["d"]), [int(x) for x in w["masks"]], w["times"][b], w["d_fixed"][b])[0], dtype=float)
    exact = exact.reshape(hip[b].shape)
    print("worst solve case: N %d K %d D %d masks %s batch %d, trajectory %d: HIP vs reference %.2e, HIP vs 50-digit %.2e, reference vs 50-digit %.2e"
          % (int(w["n"]), hip.shape[1], hip.shape[2], [int(x) for x in w["masks"]], hip.shape[0], b,
             helpers.poly_relerr(hip[b:b + 1], ref[b:b + 1]), helpers.poly_relerr(hip[b:b + 1], exact[None]), helpers.poly_relerr(ref[b:b + 1], exact[None])))
