"""Bit-identity diagnostics between the kernels of one shape (round 4: how the contraction difference of the factor-store
back-substitution was found): coefficient-only launch against the extra-output launch (cost + d_P) and the dimension-split form, per
shape -- number of differing elements, the worst relative difference, and WHERE they sit (coefficient index, segment, dimension).
usage: compare_kernel_forms.py [N K]...      (default: the long-chain shapes)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mav_trajectory_generation_amd as m

shapes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)] or [(10, 32), (12, 32), (12, 16), (10, 16), (8, 32)]
ctx = m.Context(0)
for n, k in shapes:
    masks = m.ends_full_masks(n, k, 1)
    plan = m.Plan(ctx, n, 3, k, n // 2 - 1, masks)
    for bsz, layout in ((300, "soa"), (1000, "aos")):
        t, f = m.random_waypoint_batch(bsz, k, 3, n, masks, seed=17 * n + k, device="cuda", layout=layout)
        torch.cuda.synchronize()
        co, fr, cost = plan.solve(t, f, layout=layout, want_free=True, want_cost=True)
        co0, _, _ = plan.solve(t, f, layout=layout, dims="dimlane")
        cs, _, _ = plan.solve(t, f, layout=layout, dims="split")
        cf, ff, jf = plan.solve(t, f, layout=layout, want_free=True, want_cost=True, dims="fused")
        ctx.sync()
        den = co0.abs().amax(dim=-1, keepdim=True).clamp_min(1e-300)
        bad = (co - co0).abs() > 0
        line = f"N={n} K={k} B={bsz} {layout}: extra-output vs coefficient-only: {int(bad.sum())} of {bad.numel()} elements differ"
        if int(bad.sum()):
            nz = bad.nonzero()
            line += (f", max rel {float(((co - co0).abs() / den).max()):.2e}; by coefficient index "
                     f"{np.bincount(nz[:, 3].cpu().numpy(), minlength=n).tolist()}, segments {sorted(set(nz[:, 1].tolist()))}, "
                     f"dimensions {sorted(set(nz[:, 2].tolist()))}")
        print(line)
        print(f"    vs split form {float(((co0 - cs).abs() / den).max()):.2e} | vs fused: coefficients {float(((co - cf).abs() / den).max()):.2e}, "
              f"cost {float(((cost - jf).abs() / jf.abs()).max()):.2e}, d_P {float((fr - ff).abs().amax() / ff.abs().amax()):.2e}")
    plan.close()
