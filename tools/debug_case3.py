import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import mav_trajectory_generation_amd as m
import helpers
from oracle import oracle_np as onp
np.set_printoptions(precision=4, suppress=True, linewidth=200)
ctx = m.Context(0)
n, d, k, dim, bsz = 2, 0, 3, 2, 5
masks, times, d_fixed = helpers.reference_batch(bsz, k, n, dim, 4242, None)
c_lit, f_lit, j_lit = onp.solve_batch(n, d, masks, times, d_fixed)
plan = m.Plan(ctx, n, dim, k, d, masks)
for (wf, wc) in [(False, False), (False, True), (True, False)]:
    co, fr, cost = plan.solve_host(times, d_fixed, want_free=wf, want_cost=wc, generic=True)
    print("host path free", wf, "cost", wc, "err %.2e" % helpers.poly_relerr(co, c_lit), "cost", None if cost is None else cost[:3], "ref", j_lit[:3])
t = torch.from_numpy(times).cuda(); f = torch.from_numpy(d_fixed).cuda()
torch.cuda.synchronize()
for sz in (5, 4096):
    cost = torch.zeros((sz,), dtype=torch.float64, device="cuda")
    co = torch.full((bsz, k, dim, n), -777.0, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    plan.solve(t, f, coeffs=co, cost=cost, want_cost=True, generic=True)
    ctx.sync()
    print("device path, cost buffer size", sz, "err %.2e" % helpers.poly_relerr(co.cpu().numpy(), c_lit), cost[:5].cpu().numpy(), j_lit)
print(co.cpu().numpy()[0].reshape(k, -1)); print(c_lit[0].reshape(k, -1))
print("--- repeat solves back-to-back (cost on), then with cost off in between")
for i in range(3):
    co = torch.full((bsz, k, dim, n), -777.0, dtype=torch.float64, device="cuda"); cost = torch.zeros((bsz,), dtype=torch.float64, device="cuda")
    plan.solve(t, f, coeffs=co, cost=cost, want_cost=True, generic=True); ctx.sync()
    print(i, "err %.2e" % helpers.poly_relerr(co.cpu().numpy(), c_lit))
# other dims for comparison
for dim2 in (1, 2, 3, 4):
    masks2, times2, d_fixed2 = helpers.reference_batch(bsz, k, n, dim2, 4242, None)
    c2, _, j2 = onp.solve_batch(n, d, masks2, times2, d_fixed2)
    p2 = m.Plan(ctx, n, dim2, k, d, masks2)
    co2, _, cost2 = p2.solve_host(times2, d_fixed2, want_free=False, want_cost=True, generic=True)
    print("D", dim2, "err %.2e" % helpers.poly_relerr(co2, c2), "cost err %.2e" % np.abs(cost2 / j2 - 1).max())
