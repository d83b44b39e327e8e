"""Form-equivalence fuzz: seeded random plans (N, K, D, interior mask), batch sizes, input layouts and output sets; every
launch form the library offers for the plan (auto, dimension-in-lane, fused, split, generic; the host build of the lane code for
small batches) must return the same solution as the generic kernel, must not write past the batch, and must raise no status
flag.  Catches routing mistakes between the forms (layout kinds, extra outputs, ragged tiles) that the per-form tests, which
each pin one route, cannot see."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    import mav_trajectory_generation_amd as m
    c = m.Context(0)
    yield c
    c.close()


def cases(n_cases, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n_cases):
        n = int(rng.choice([8, 10, 12, 8, 10, 12, 8, 10, 12, 4, 6]))
        k = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 12, 16, 17, 24, 32, 33, 40]))
        dim = int(rng.choice([1, 2, 3, 3, 3, 4]))
        interior = int(rng.choice([1, 1, 1, 3, 7]))
        bsz = int(rng.choice([1, 2, 15, 16, 21, 22, 63, 64, 65, 130, 700]))
        layout = str(rng.choice(["soa", "aos"]))
        extras = bool(rng.integers(0, 2))
        interior &= (1 << (n // 2)) - 1
        d = n // 2 - 1 if rng.integers(0, 10) < 7 else int(rng.integers(1, n // 2))      # mostly the standard derivative
        out.append((i, n, k, dim, interior, bsz, layout, extras, d))
    return out


N_FORMS = int(os.environ.get("MTG_FUZZ_CASES", "120"))      # (MTG_FUZZ_CASES=3000 for a long run)


@pytest.mark.parametrize("case", cases(N_FORMS, 20260925), ids=lambda c: "case%d-N%d-K%d-D%d-mi%d-B%d-%s-%s-d%d" % (c[:7] + ("x" if c[7] else "c", c[8])))
def test_every_form_returns_the_generic_kernels_solution(ctx, case):
    import torch
    import mav_trajectory_generation_amd as m
    _, n, k, dim, interior, bsz, layout, extras, d = case
    masks = m.ends_full_masks(n, k, interior)
    plan = m.Plan(ctx, n, dim, k, d, masks)
    t, f = m.random_waypoint_batch(bsz, k, dim, n, masks, seed=1000 + case[0], device="cuda", layout=layout)
    ref_c, ref_f, ref_j = plan.solve(t, f, layout=layout, want_free=True, want_cost=True, generic=True)
    tol = 1e-10 if n <= 10 else 1e-8
    if case[0] % 4 == 0:      # every fourth case: the generic kernel itself against the literal restatement of the reference
        from oracle import oracle_np as onp
        import helpers
        nb = min(bsz, 2)
        ta = (t if layout == "aos" else t.t())[:nb].contiguous().cpu().numpy()
        fa = (f if layout == "aos" else f.permute(2, 0, 1))[:nb].contiguous().cpu().numpy()
        got = ref_c[:nb].cpu().numpy()
        assert helpers.check_path(masks, ta, fa, got) < 1e-6           # fixed values met, derivatives continuous at the vertices
        if d == n // 2 - 1 or n <= 8:
            # (lower derivative orders with N >= 10 are ill-conditioned enough that the LITERAL float64 route is the less
            # accurate side -- DESIGN.md section 1, arbitrated with 50 digits in test_gpu_vs_reference.py; not repeated here)
            c_lit, _, j_lit = onp.solve_batch(n, d, masks, ta, fa)
            assert helpers.poly_relerr(got, c_lit) < (1e-9 if n <= 10 and d == n // 2 - 1 else 5e-7 if d == n // 2 - 1 else 1e-6)
            assert np.allclose(ref_j[:nb].cpu().numpy(), j_lit, rtol=1e-6)
    for dims in ("auto", "dimlane", "fused", "split"):
        co = torch.full((bsz + 1, k, dim, n), 7.0, dtype=torch.float64, device="cuda")
        _, fr, cost = plan.solve(t, f, layout=layout, coeffs=co[:bsz], want_free=extras, want_cost=extras, dims=dims)
        ctx.sync()          # raises on any status flag
        assert float(co[bsz].min()) == 7.0 and float(co[bsz].max()) == 7.0, (dims, "stores past the end of the batch")
        rel, _ = ctx.compare_coefficients(co[:bsz], ref_c)
        assert rel < tol, (dims, plan.launch_form(bsz, layout, dims, extra_outputs=extras), rel)
        if extras:
            assert torch.allclose(cost, ref_j, rtol=1e-9 if n <= 10 else 1e-7, atol=0), dims
            if plan.n_free > 0:
                scale = float(ref_f.abs().max()) + 1e-300
                assert float((fr - ref_f).abs().max()) / scale < tol, dims
    if bsz <= 64:     # the host build of the same lane code (the veneer's single-trajectory route)
        ta = t if layout == "aos" else t.t().contiguous()
        fa = f if layout == "aos" else f.permute(2, 0, 1).contiguous()
        co_h, _, _ = plan.solve_host(ta.cpu().numpy(), fa.cpu().numpy(), want_free=False, want_cost=False, host_backend=True)
        assert np.abs(co_h - ref_c.cpu().numpy()).max() <= tol * max(1.0, float(ref_c.abs().max()))
    plan.close()


@pytest.mark.parametrize("case", cases(max(1, N_FORMS // 2), 777), ids=lambda c: "case%d-N%d-K%d-D%d-mi%d-B%d-%s-d%d" % (c[:7] + (c[8],)))
def test_queue_update_and_mixed_entry_points_agree_with_single_solves(ctx, case):
    """The other entry points on the same random plans: mtg_solve_linear_sequence (queue launch where the plan has one),
    mtg_update_segments_from_free fed with the solver's own d_P, and a two-bucket mtg_multi request."""
    import torch
    import mav_trajectory_generation_amd as m
    _, n, k, dim, interior, bsz, layout, _, d = case
    masks = m.ends_full_masks(n, k, interior)
    plan = m.Plan(ctx, n, dim, k, d, masks)
    sets, singles = [], []
    for s in range(3):
        t, f = m.random_waypoint_batch(bsz, k, dim, n, masks, seed=50 * case[0] + s, device="cuda", layout=layout)
        sets.append((t, f, torch.full((bsz, k, dim, n), float("nan"), dtype=torch.float64, device="cuda")))
        singles.append(plan.solve(t, f, layout=layout, want_free=True, want_cost=True, generic=True))
    plan.solve_sequence(sets, layout=layout)
    ctx.sync()
    tol = 1e-10 if n <= 10 else 1e-8
    for (t, f, co), (rc, rf, rj) in zip(sets, singles):
        rel, _ = ctx.compare_coefficients(co, rc)
        assert rel < tol
    t, f, _ = sets[0]
    rc, rf, rj = singles[0]
    cu, ju = plan.update_from_free(t, f, rf, layout=layout, want_cost=True)
    ctx.sync()
    rel, _ = ctx.compare_coefficients(cu, rc)
    assert rel < tol
    # (the update path starts from the solver's d_P rounded to float64, the solve's own cost from the scaled vertex values of its
    # back-substitution: coefficients one ulp apart.  For d = h - 1 that is 1e-11 in 0.5 c^T Q c; for d < h - 1 the monomial-basis
    # quadratic form cancels and one ulp in c moves the cost by up to 2e-8 -- measured, tools/lab/cost_probe.py)
    assert torch.allclose(ju, rj, rtol=(1e-9 if d == n // 2 - 1 else 1e-7) if n <= 10 else 1e-7, atol=0)
    if dim == 3 and k >= 2:
        solver = m.MixedBatchSolver(ctx, n_streams=1)
        buckets = [dict(n_coeffs=n, derivative=d, masks=masks, times=sets[i][0], d_fixed=sets[i][1], layout=layout) for i in (1, 2)]
        req = solver.merged(buckets)
        out = req.solve()
        solver.sync()
        ctx.sync()
        for (co, _), (rc2, _, _) in zip(out, singles[1:]):
            rel, _ = ctx.compare_coefficients(co.contiguous(), rc2)
            assert rel < tol
        req.close()
        solver.close()
    plan.close()
