"""GPU parity against the REFERENCE ITSELF: the HIP path (through the C ABI, libmtg_hip.so) vs outputs of the
reference's own PolynomialOptimization<N> / Trajectory code compiled from /root/reference (oracle/_ref/libmtg_ref.so,
see oracle/ref_linear.py for what the Eigen/glog stand-ins do and do not pin).

  * fixture tests: tests/golden/reference_solve_linear.npz (generated in the build container by
    tests/golden/make_reference_golden.py) -- no library, no /root/reference needed at run time;
  * live tests: the prebuilt oracle/_ref/libmtg_ref.so travels to the GPU box with the snapshot (it never reads
    /root/reference at run time) and is called on fresh seeded batches.
Tolerance: north_star's 1e-9 relative (norm-wise per polynomial) for N <= 10 with d = N/2-1; where
float64 evaluation of the reference's own formulas is less accurate than that (N = 12, d < N/2-1: the compiled
reference itself is 1e-8..2e-6 from the 50-digit solve) the bound is the reference's own distance to the truth.
"""
import os

import numpy as np
import pytest

import helpers
from oracle import ref_linear

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "solve_linear_golden.npz"))
REF = np.load(os.path.join(HERE, "golden", "reference_solve_linear.npz"))
NAMES = sorted({k.split("/")[0] for k in GOLD.files})
live = pytest.mark.skipif(not ref_linear.available(), reason="oracle/_ref/libmtg_ref.so not shipped")


def tol_for(n, d):
    if n == 12 and d < n // 2 - 1:
        return 1e-5
    if n == 12 or d < n // 2 - 1:
        return 5e-8
    return 1e-9


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    import mav_trajectory_generation_amd as m
    c = m.Context(0)
    yield c
    c.close()


def gpu_solve(ctx, n, d, masks, times, d_fixed, layout="aos", generic=False, dims="auto"):
    import torch
    import mav_trajectory_generation_amd as m
    dim, k = d_fixed.shape[1], times.shape[1]
    plan = m.Plan(ctx, n, dim, k, d, masks)
    t = torch.from_numpy(np.ascontiguousarray(times)).cuda()
    f = torch.from_numpy(np.ascontiguousarray(d_fixed)).cuda()
    if layout == "soa":
        t = t.t().contiguous()
        f = f.permute(1, 2, 0).contiguous()
    co, fr, cost = plan.solve(t, f, layout=layout, want_free=True, want_cost=True, generic=generic, dims=dims)
    ctx.sync()
    if layout == "soa":
        fr = fr.permute(2, 0, 1)
    out = co.cpu().numpy(), fr.cpu().numpy(), cost.cpu().numpy(), plan.kernel_variant
    plan.close()
    return out


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("generic", [False, True])
def test_hip_vs_reference_outputs(ctx, name, generic):
    n, d = int(GOLD[f"{name}/n"]), int(GOLD[f"{name}/d"])
    masks = [int(m) for m in GOLD[f"{name}/masks"]]
    times, d_fixed = GOLD[f"{name}/times"], GOLD[f"{name}/d_fixed"]
    co, fr, cost, _ = gpu_solve(ctx, n, d, masks, times, d_fixed, generic=generic)
    ref_c, ref_f, ref_j = REF[f"{name}/coeffs_ref"], REF[f"{name}/d_free_ref"], REF[f"{name}/cost_ref"]
    assert helpers.poly_relerr(co, ref_c) < tol_for(n, d)
    assert np.allclose(cost, ref_j, rtol=max(1e-8, tol_for(n, d)))
    if ref_f.size and d == n // 2 - 1 and n <= 10:
        assert np.abs(fr - ref_f).max() <= 1e-8 * max(1.0, np.abs(ref_f).max())
    if name == "two_vertices":   # the reference's known-answer vector, TOPT:777-780
        assert np.abs(co[0, 0, 0] - GOLD["two_vertices/matlab_coeffs"]).max() < 1e-12


@live
@pytest.mark.parametrize("n,d,k,dim,masks,bsz,layout,dims", [
    (10, 4, 8, 3, None, 2000, "soa", "fused"),                       # BASELINE config 2/3 shape, bench layout
    (10, 4, 8, 3, None, 2000, "soa", "split"),
    (10, 4, 8, 3, None, 300, "aos", "auto"),
    (10, 4, 16, 4, [31] + [7] * 15 + [31], 500, "soa", "auto"),      # config 5 shape
    (8, 3, 4, 3, None, 300, "soa", "auto"), (8, 3, 32, 3, None, 100, "soa", "auto"),   # config 4 corners
    (10, 4, 16, 3, None, 200, "soa", "auto"), (10, 4, 32, 3, None, 100, "aos", "auto"),
    (10, 4, 6, 3, [31, 1, 3, 1, 5, 9, 31], 200, "aos", "auto"),      # ragged masks -> generic kernel
    (10, 4, 1, 3, None, 100, "aos", "auto"),                         # n_free == 0
    (6, 2, 4, 3, None, 100, "aos", "auto"), (4, 1, 3, 2, None, 100, "aos", "auto"),
])
def test_hip_vs_live_reference_on_fresh_batches(ctx, n, d, k, dim, masks, bsz, layout, dims):
    masks, times, d_fixed = helpers.reference_batch(bsz, k, n, dim, 60221023, masks)
    ref_c, ref_f, ref_j, _ = ref_linear.solve_batch(n, d, masks, times, d_fixed, nthreads=ref_linear.hardware_threads())
    co, fr, cost, _ = gpu_solve(ctx, n, d, masks, times, d_fixed, layout=layout, dims=dims)
    assert helpers.poly_relerr(co, ref_c) < 1e-9
    assert np.allclose(cost, ref_j, rtol=1e-8)
    if ref_f.size:
        assert np.abs(fr - ref_f).max() <= 1e-8 * max(1.0, np.abs(ref_f).max())


@live
def test_update_from_free_vs_reference_set_free_constraints(ctx):
    """mtg_update (setFreeConstraints, LIN:500-508) with an arbitrary d_P vs the reference's own setFreeConstraints."""
    import torch
    import mav_trajectory_generation_amd as m
    masks, times, d_fixed = helpers.reference_batch(200, 8, 10, 3, 1717)
    rng = np.random.default_rng(5)
    d_free = rng.uniform(-2.0, 2.0, (200, 3, 28))
    ref_c, _, ref_j, _ = ref_linear.solve_batch(10, 4, masks, times, d_fixed, d_free_in=d_free)
    plan = m.Plan(ctx, 10, 3, 8, 4, masks)
    co, cost = plan.update_from_free(torch.from_numpy(times).cuda(), torch.from_numpy(d_fixed).cuda(),
                                     torch.from_numpy(d_free).cuda(), want_cost=True)
    ctx.sync()
    assert helpers.poly_relerr(co.cpu().numpy(), ref_c) < 1e-9
    assert np.allclose(cost.cpu().numpy(), ref_j, rtol=1e-8)
    plan.close()


@live
def test_sampling_vs_reference_evaluate(ctx):
    """mtg_sample_range vs the reference's Trajectory::evaluate on the same (reference-solved) coefficients."""
    import torch
    import mav_trajectory_generation_amd as m
    masks, times, d_fixed = helpers.reference_batch(24, 8, 10, 3, 9090)
    ref_c, _, _, _ = ref_linear.solve_batch(10, 4, masks, times, d_fixed)
    S = 40
    dt = float(times.sum(axis=1).min()) / S
    out = m.sample_range(ctx, torch.from_numpy(ref_c).cuda(), torch.from_numpy(times).cuda(), 0.0, dt, S, 5)
    ctx.sync()
    out = out.cpu().numpy()
    grid = dt * np.arange(S)
    for b in range(24):
        for der in range(5):
            want = ref_linear.evaluate(ref_c[b], times[b], grid, der)
            assert np.abs(out[b, :, der] - want).max() <= 1e-11 * max(1.0, np.abs(want).max())


@live
def test_extrema_and_time_scaling_vs_reference(ctx):
    """mtg_minmax_magnitude / mtg_scale_segment_times_to_meet_constraints vs Trajectory::computeMinMaxMagnitude /
    scaleSegmentTimesToMeetConstraints executed by the reference."""
    import torch
    import mav_trajectory_generation_amd as m
    masks, times, d_fixed = helpers.reference_batch(48, 8, 10, 3, 31415)
    ref_c, _, _, _ = ref_linear.solve_batch(10, 4, masks, times, d_fixed)
    co, t = torch.from_numpy(ref_c).cuda(), torch.from_numpy(times).cuda()
    for der in (1, 2):
        seg, traj, idx = m.minmax_magnitude(ctx, co, t, der)
        ctx.sync()
        seg, traj = seg.cpu().numpy(), traj.cpu().numpy()
        for b in range(48):
            mn, mx, per = ref_linear.minmax_magnitude(ref_c[b], times[b], der)
            helpers.assert_extrema_close(per, seg[b], der, (der, b))
            assert abs(traj[b, 3] - mx[1]) <= 1e-9 * mx[1]
    v_max, a_max = 2.0, 2.5
    scaling, within, _ = m.scale_segment_times_to_meet_constraints(ctx, co, t, v_max, a_max)
    ctx.sync()
    got_c, got_t, within = co.cpu().numpy(), t.cpu().numpy(), within.cpu().numpy()
    n_scaled = 0
    for b in range(48):
        c_ref, t_ref, ok = ref_linear.scale_segment_times_to_meet_constraints(ref_c[b], times[b], v_max, a_max)
        n_scaled += int(t_ref[0] != times[b, 0])
        assert bool(within[b]) == ok
        assert np.max(np.abs(got_t[b] - t_ref) / t_ref) <= 1e-9
        cs = np.abs(c_ref).max(axis=-1, keepdims=True)
        assert np.max(np.abs(got_c[b] - c_ref) / cs) <= 1e-9
    assert n_scaled > 0


@pytest.mark.parametrize("mode", ["host_backend", "device"])
def test_integration_binding_on_the_reference_class(mode):
    """tests/cpp/test_reference_binding: the reference's own PolynomialOptimization<N> object (reference sources
    compiled where they lie) solved by its own solveLinear() and by the replacement body of INTEGRATION.md section 1
    forwarding to libmtg_hip.so -- segments, free constraints, computeCost() and Trajectory::evaluate must agree.
    Both ways a single-trajectory call can go: the library's host build of the lane code, and through the GPU."""
    import subprocess
    exe = os.path.join(HERE, "cpp", "test_reference_binding")
    if not os.path.exists(exe):
        pytest.skip("tests/cpp/test_reference_binding not built (needs /root/reference at build time)")
    r = subprocess.run([exe] + (["device"] if mode == "device" else []), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "REFERENCE BINDING OK" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


# ---- the kernels the bench times, directly against the compiled reference -----------------------
# test_hip_vs_live_reference_on_fresh_batches asks for d_free and cost, which excludes the dimension-in-lane, slab-output,
# queue and cross-structure forms (coefficient output only).  Here: coefficient-only calls of exactly those forms on fresh
# 2000-trajectory batches of BASELINE configs 2 / 3 (same shape), 5 and the twelve config-4 buckets.
def coeff_only(ctx, n, d, masks, times, d_fixed, dims):
    import torch
    import mav_trajectory_generation_amd as m
    dim, k = d_fixed.shape[1], times.shape[1]
    plan = m.Plan(ctx, n, dim, k, d, masks)
    t = torch.from_numpy(np.ascontiguousarray(times)).cuda().t().contiguous()
    f = torch.from_numpy(np.ascontiguousarray(d_fixed)).cuda().permute(1, 2, 0).contiguous()
    form = plan.launch_form(times.shape[0], "soa", dims)
    co, _, _ = plan.solve(t, f, layout="soa", dims=dims)
    ctx.sync()
    out = co.cpu().numpy()
    plan.close()
    return out, form


CONFIG4 = [(n, n // 2 - 1, k) for n in (8, 10, 12) for k in (4, 8, 16, 32)]


def log_arbitration(n, d, k, per_traj):
    """What the gate below actually saw, one line per call, where a GPU visit collects it (gpurun_out/parity_arbitration.jsonl ->
    profiles/): how many trajectories of the batch were above 1e-9 against the reference (and therefore arbitrated by the 50-digit
    solve), the maximum and the median."""
    import json
    out = os.path.join(os.path.dirname(HERE), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_arbitration.jsonl"), "a") as fh:
            fh.write(json.dumps({"N": n, "d": d, "K": int(k), "trajectories": int(len(per_traj)), "above_1e-9_vs_reference": int((~(per_traj < 1e-9)).sum()),
                                 "max_vs_reference": float(per_traj.max()), "median_vs_reference": float(np.median(per_traj))}) + "\n")
    except OSError:
        pass


def assert_close_to_reference(n, d, masks, times, d_fixed, co, ref_c):
    """N <= 10: the north-star tolerance, 1e-9 norm-wise per polynomial.  N = 12: a 2000-trajectory random-waypoint batch contains
    trajectories (a very short segment next to long ones) on which the PROBLEM is so ill-conditioned that float64 evaluation
    of the reference's own formulas is 2e-7 .. 1e-6 from the exact solution -- and the HIP result, although it never inverts
    A, up to 6e-7 (observed, K = 16: HIP 5.5e-7, reference 7.9e-7 on the same trajectory; another one: HIP 1.8e-8, reference
    2.1e-7).  Two correct float64 results differ by that much.  The check there: median difference < 1e-8, no trajectory
    beyond 1e-5, and on the three WORST trajectories the 50-digit solve (oracle/oracle_mp.py) must put the HIP result within
    5e-8 of the truth or within twice the reference's own distance to it (i.e. never the side that is clearly off)."""
    per_traj = np.array([helpers.poly_relerr(co[b:b + 1], ref_c[b:b + 1]) for b in range(co.shape[0])])
    log_arbitration(n, d, times.shape[1], per_traj)
    if n <= 10:
        # EVERY trajectory above the north-star tolerance is arbitrated by the 50-digit solve (seen on long chains, K = 100: one
        # of 1000 at 1.1e-9): accepted only when the HIP result is within 1e-9 of the truth, i.e. the reference is the side
        # that is off -- and only a few of them
        over = np.nonzero(~(per_traj < 1e-9))[0]
        assert len(over) <= max(3, co.shape[0] // 200), (len(over), per_traj.max())
        from oracle import oracle_mp
        for b in over:
            truth = np.asarray(oracle_mp.solve(n, d, masks, times[b], d_fixed[b])[0], dtype=np.float64)[None]
            e_hip, e_ref = helpers.poly_relerr(co[b:b + 1], truth), helpers.poly_relerr(ref_c[b:b + 1], truth)
            assert e_hip < 1e-9 and per_traj[b] < 1e-8, (int(b), per_traj[b], e_hip, e_ref)
        return
    assert np.median(per_traj) < 1e-8 and per_traj.max() < 1e-5
    from oracle import oracle_mp
    for b in np.argsort(per_traj)[-3:]:
        truth, _, _ = oracle_mp.solve(n, d, masks, times[b], d_fixed[b])
        truth = np.asarray(truth, dtype=np.float64)[None]
        e_hip, e_ref = helpers.poly_relerr(co[b:b + 1], truth), helpers.poly_relerr(ref_c[b:b + 1], truth)
        assert e_hip < max(5e-8, 2.0 * e_ref), (int(b), e_hip, e_ref)


@live
@pytest.mark.parametrize("n,d,k,dim,interior,dims,want_form", [
    (10, 4, 8, 3, 1, "dimlane", "dimlane"), (10, 4, 8, 3, 1, "fused", "slab"),         # configs 2 / 3: the bench kernels
    (10, 4, 16, 4, 7, "dimlane", "dimlane"), (10, 4, 16, 4, 7, "auto", "dimlane"),     # config 5
] + [(n, d, k, 3, 1, "dimlane", "dimlane") for (n, d, k) in CONFIG4]                     # config 4 buckets
  + [(n, d, k, 3, 1, "fused", "slab") for (n, d, k) in ((8, 3, 4), (8, 3, 8), (10, 4, 4), (12, 5, 4))])
def test_bench_kernels_vs_live_reference(ctx, n, d, k, dim, interior, dims, want_form):
    import mav_trajectory_generation_amd as m
    bsz = 2000 if k <= 16 else 600
    masks = m.ends_full_masks(n, k, interior)
    masks, times, d_fixed = helpers.reference_batch(bsz, k, n, dim, 31415 + 7 * k + n, masks)
    ref_c, _, _, _ = ref_linear.solve_batch(n, d, masks, times, d_fixed, nthreads=ref_linear.hardware_threads())
    co, form = coeff_only(ctx, n, d, masks, times, d_fixed, dims)
    assert form == want_form
    assert_close_to_reference(n, d, masks, times, d_fixed, co, ref_c)


@live
def test_queue_and_cross_structure_launches_vs_live_reference(ctx):
    """mtg_solve_linear_sequence's persistent queue launch (config 2 shape: the bench's `value` kernel; config 5 shape: the
    dimension-in-lane queue) and mtg_multi_solve's cross-structure launch over the twelve config-4 buckets."""
    import torch
    import mav_trajectory_generation_amd as m
    for (n, d, k, dim, interior) in ((10, 4, 8, 3, 1), (10, 4, 16, 4, 7)):
        masks = m.ends_full_masks(n, k, interior)
        plan = m.Plan(ctx, n, dim, k, d, masks)
        sets, refs = [], []
        for s in range(3):
            _, times, d_fixed = helpers.reference_batch(2000, k, n, dim, 2718 + s, masks)
            refs.append(ref_linear.solve_batch(n, d, masks, times, d_fixed, nthreads=ref_linear.hardware_threads())[0])
            t = torch.from_numpy(np.ascontiguousarray(times)).cuda().t().contiguous()
            f = torch.from_numpy(np.ascontiguousarray(d_fixed)).cuda().permute(1, 2, 0).contiguous()
            sets.append((t, f, torch.zeros((2000, k, dim, n), dtype=torch.float64, device="cuda")))
        plan.solve_sequence(sets, layout="soa")
        ctx.sync()
        for (_, _, co), ref_c in zip(sets, refs):
            assert helpers.poly_relerr(co.cpu().numpy(), ref_c) < 1e-9
        plan.close()
    buckets, refs, inputs = [], [], {}
    for (n, d, k) in CONFIG4:
        bsz = 2000 if k <= 16 else 600
        masks = m.ends_full_masks(n, k, 1)
        _, times, d_fixed = helpers.reference_batch(bsz, k, n, 3, 1618 + k + n, masks)
        refs.append(ref_linear.solve_batch(n, d, masks, times, d_fixed, nthreads=ref_linear.hardware_threads())[0])
        inputs[(n, k)] = (masks, times, d_fixed)
        t = torch.from_numpy(np.ascontiguousarray(times)).cuda().t().contiguous()
        f = torch.from_numpy(np.ascontiguousarray(d_fixed)).cuda().permute(1, 2, 0).contiguous()
        buckets.append(dict(n_coeffs=n, derivative=d, masks=masks, times=t, d_fixed=f, layout="soa"))
    solver = m.MixedBatchSolver(ctx, n_streams=1)
    req = solver.merged(buckets)
    assert req.launch_count == 1          # ONE cross-structure dimension-in-lane launch (mtg_solve_dl_any_kernel)
    out = req.solve()
    torch.cuda.synchronize()
    solver.sync()
    for (n, d, k), (co, _), ref_c in zip(CONFIG4, out, refs):
        assert_close_to_reference(n, d, *inputs[(n, k)], co.cpu().numpy(), ref_c)
    req.close()
    solver.close()
