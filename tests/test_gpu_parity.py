"""GPU parity tests: the HIP path, called through the C ABI (libmtg_hip.so), against the oracle, the committed
golden fixtures and the mpmath truth; size-independent properties at BASELINE.json's full batch sizes.
Tolerances: SURVEY.md 8(d) norm-wise metric max_poly ||c - c_ref||_inf / ||c_ref||_inf <= 1e-9 for N <= 10 with
d = h-1; looser where float64 evaluation of the reference's own formulas is itself less accurate (N = 12, d < h-1)."""
import os

import numpy as np
import pytest

import helpers
from oracle import oracle_np as onp

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "solve_linear_golden.npz"))
NAMES = sorted({k.split("/")[0] for k in GOLD.files})


def case(name):
    pre = name + "/"
    return {k[len(pre):]: GOLD[k] for k in GOLD.files if k.startswith(pre)}


def tol_for(n, d):
    if n == 12 and d < n // 2 - 1:
        return 1e-5
    if n == 12 or d < n // 2 - 1:
        return 5e-7
    return 1e-9


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    import mav_trajectory_generation_amd as m
    c = m.Context(0)
    yield c
    c.close()


def gpu_solve(ctx, n, d, masks, times, d_fixed, layout="aos", generic=False, want=True, dims="auto"):
    import torch
    import mav_trajectory_generation_amd as m
    dim, k = d_fixed.shape[1], times.shape[1]
    plan = m.Plan(ctx, n, dim, k, d, masks)
    t = torch.from_numpy(np.ascontiguousarray(times)).cuda()
    f = torch.from_numpy(np.ascontiguousarray(d_fixed)).cuda()
    if layout == "soa":
        t = t.t().contiguous()
        f = f.permute(1, 2, 0).contiguous()
    co, fr, cost = plan.solve(t, f, layout=layout, want_free=want, want_cost=want, generic=generic, dims=dims)
    ctx.sync()
    if fr is not None and layout == "soa":
        fr = fr.permute(2, 0, 1)
    out = (co.cpu().numpy(), None if fr is None else fr.cpu().numpy(), None if cost is None else cost.cpu().numpy(),
           plan.kernel_variant)
    plan.close()
    return out


def test_native_library_is_the_loaded_one(ctx):
    from mav_trajectory_generation_amd import _lib
    maps = open("/proc/self/maps").read()
    assert os.path.realpath(_lib.LIB_PATH) in maps


def test_device_reciprocal_accuracy(ctx):
    """v_rcp_f64 + 2 Newton steps used for LDL^T pivots and 1/T: must be at rounding level."""
    assert ctx.selftest_rcp(1 << 20) < 4.5e-16


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("generic", [False, True])
@pytest.mark.parametrize("layout", ["aos", "soa"])
def test_golden_fixtures(ctx, name, generic, layout):
    c = case(name)
    n, d = int(c["n"]), int(c["d"])
    masks = [int(m) for m in c["masks"]]
    co, fr, cost, variant = gpu_solve(ctx, n, d, masks, c["times"], c["d_fixed"], layout, generic)
    assert helpers.poly_relerr(co, c["coeffs_lit"]) < tol_for(n, d)
    if "coeffs_mp" in c:
        assert helpers.poly_relerr(co, c["coeffs_mp"]) < (1e-7 if n == 12 else 1e-11)
        if c["d_free_mp"].size:
            assert np.abs(fr - c["d_free_mp"]).max() <= 1e-7 * max(1.0, np.abs(c["d_free_mp"]).max())
        assert np.allclose(cost, c["cost_mp"], rtol=1e-7)
    assert helpers.check_path(masks, c["times"], c["d_fixed"], co) < 1e-6
    if name == "two_vertices":
        assert np.abs(co[0, 0, 0] - c["matlab_coeffs"]).max() < 1e-12   # TOPT:777-780


@pytest.mark.parametrize("name", ["config2", "config5", "readme"])
@pytest.mark.parametrize("dims", ["fused", "split"])
@pytest.mark.parametrize("want", [True, False])
def test_dimension_split_and_fused_variants(ctx, name, dims, want):
    """Both launch geometries of the specialised kernels (all dimensions per workgroup / one dimension group
    per workgroup) against the oracle; cost is accumulated atomically across groups."""
    c = case(name)
    n, d = int(c["n"]), int(c["d"])
    masks = [int(m) for m in c["masks"]]
    co, fr, cost, variant = gpu_solve(ctx, n, d, masks, c["times"], c["d_fixed"], dims=dims, want=want)
    assert variant in (1, 2)
    assert helpers.poly_relerr(co, c["coeffs_lit"]) < tol_for(n, d)
    assert helpers.poly_relerr(co, c["coeffs_mp"]) < 1e-11
    if want:
        assert np.abs(fr - c["d_free_mp"]).max() <= 1e-9 * max(1.0, np.abs(c["d_free_mp"]).max())
        assert np.allclose(cost, c["cost_mp"], rtol=1e-9)


def test_specialised_variant_is_selected_for_baseline_config(ctx):
    import mav_trajectory_generation_amd as m
    plan = m.Plan(ctx, 10, 3, 8, 4, m.ends_full_masks(10, 8))
    assert plan.kernel_variant == 1 and plan.n_fixed == 17 and plan.n_free == 28
    assert plan.bytes_per_trajectory == 2392
    plan.close()


@pytest.mark.parametrize("bsz", [1, 63, 64, 65, 1000])
def test_ragged_batch_sizes_vs_oracle(ctx, bsz):
    masks, times, d_fixed = helpers.reference_batch(bsz, 8, 10, 3, 31337)
    c_lit, f_lit, j_lit = onp.solve_batch(10, 4, masks, times, d_fixed)
    for generic in (False, True):
        co, fr, cost, _ = gpu_solve(ctx, 10, 4, masks, times, d_fixed, generic=generic)
        assert helpers.poly_relerr(co, c_lit) < 1e-9
        assert np.abs(fr - f_lit).max() <= 1e-8 * max(1.0, np.abs(f_lit).max())
        assert np.allclose(cost, j_lit, rtol=1e-8)


@pytest.mark.parametrize("n,d,k,dim,masks", [
    (10, 4, 6, 3, [31, 1, 3, 1, 5, 9, 31]),
    (10, 4, 5, 5, None),
    (10, 4, 9, 7, None),
    (10, 4, 3, 2, [3, 1, 1, 7]),
    (10, 4, 1, 3, None),
    (2, 0, 3, 2, None), (4, 1, 3, 2, None), (6, 2, 4, 3, None),
    (8, 3, 32, 3, None), (12, 5, 7, 4, None), (10, 4, 50, 1, None), (10, 4, 100, 3, None),
])
def test_edge_shapes_vs_oracle(ctx, n, d, k, dim, masks):
    masks, times, d_fixed = helpers.reference_batch(5, k, n, dim, 4242, masks)
    c_lit, f_lit, j_lit = onp.solve_batch(n, d, masks, times, d_fixed)
    co, fr, cost, _ = gpu_solve(ctx, n, d, masks, times, d_fixed)
    assert helpers.poly_relerr(co, c_lit) < tol_for(n, d)
    assert helpers.check_path(masks, times, d_fixed, co) < 1e-6
    assert np.allclose(cost, j_lit, rtol=1e-6)


def test_host_pointer_path_and_update_from_free(ctx):
    import torch
    import mav_trajectory_generation_amd as m
    c = case("config2")
    masks = [int(x) for x in c["masks"]]
    plan = m.Plan(ctx, 10, 3, 8, 4, masks)
    co, fr, cost = plan.solve_host(c["times"], c["d_fixed"])
    ctx.sync()
    assert helpers.poly_relerr(co, c["coeffs_lit"]) < 1e-9
    t = torch.from_numpy(c["times"]).cuda()
    f = torch.from_numpy(c["d_fixed"]).cuda()
    p = torch.from_numpy(fr).cuda()
    co2, cost2 = plan.update_from_free(t, f, p, want_cost=True)
    ctx.sync()
    assert helpers.poly_relerr(co2.cpu().numpy(), co) < 1e-13
    assert np.allclose(cost2.cpu().numpy(), cost, rtol=1e-12)
    # setFreeConstraints with perturbed d_P costs more (optimality, SURVEY.md section 4 gap)
    co3, cost3 = plan.update_from_free(t, f, p * 1.01, want_cost=True)
    ctx.sync()
    assert np.all(cost3.cpu().numpy() >= cost * (1 - 1e-12))
    plan.close()


def test_error_codes(ctx):
    import torch
    import mav_trajectory_generation_amd as m
    with pytest.raises(m.MtgError) as e:
        m.Plan(ctx, 11, 3, 8, 4, [1] * 9)
    assert e.value.code == -1
    with pytest.raises(m.MtgError):
        m.Plan(ctx, 10, 3, 8, 5, [1] * 9)          # derivative > N/2-1 (LIN:60-65)
    c = case("config2")
    masks = [int(x) for x in c["masks"]]
    plan = m.Plan(ctx, 10, 3, 8, 4, masks)
    t = c["times"].copy()
    t[2, 5] = -1.0                                  # LIN:297
    plan.solve(torch.from_numpy(t).cuda(), torch.from_numpy(c["d_fixed"]).cuda())
    with pytest.raises(m.MtgError) as e:
        ctx.sync()
    assert e.value.code == -2
    ctx.sync()                                      # status is cleared by the failing sync
    plan.close()


@pytest.mark.parametrize("bsz", [10_000, 125_000])
def test_full_size_properties(ctx, bsz):
    """BASELINE config 2 (10k) and the per-GPU share of config 3 (1M / 8): properties that need no oracle --
    checkPath at 1e-6, specialised vs generic kernel agreement, linearity in d_F, and oracle parity on a
    strided 500-trajectory subset."""
    import torch
    import mav_trajectory_generation_amd as m
    masks = m.ends_full_masks(10, 8)
    plan = m.Plan(ctx, 10, 3, 8, 4, masks)
    t, f = m.random_waypoint_batch(bsz, 8, 3, 10, masks, seed=1, device="cuda")
    co, fr, cost = plan.solve(t, f, want_free=True, want_cost=True)
    co_g, _, _ = plan.solve(t, f, generic=True)
    co_2, _, _ = plan.solve(t, f * 2.0)
    co_s, _, _ = plan.solve(t, f, dims="split")
    co_f, _, _ = plan.solve(t, f, dims="fused")
    ctx.sync()
    assert torch.isfinite(co).all()
    den = co.abs().amax(dim=-1).clamp_min(1e-300)
    assert float(((co - co_g).abs().amax(dim=-1) / den).max()) < 1e-11
    assert float(((co_s - co_f).abs().amax(dim=-1) / den).max()) < 1e-11
    assert float(((co_2 - 2 * co).abs().amax(dim=-1) / den).max()) < 1e-13
    tn, fn, cn = t.cpu().numpy(), f.cpu().numpy(), co.cpu().numpy()
    assert helpers.check_path(masks, tn, fn, cn) < 1e-6
    idx = np.arange(0, bsz, bsz // 500)[:500]
    c_lit, f_lit, j_lit = onp.solve_batch(10, 4, masks, tn[idx], fn[idx])
    assert helpers.poly_relerr(cn[idx], c_lit) < 1e-9
    assert np.allclose(cost.cpu().numpy()[idx], j_lit, rtol=1e-8)
    plan.close()


def _run_bench(extra_args, nproc=1):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable]
    if nproc > 1:
        base += ["-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
                 "--master-port", str(29600 + os.getpid() % 300)]
    cmd = base + [os.path.join(root, "bench.py"), "--steps", "20", "--warmup", "3", "--no-cpu-baseline"] + extra_args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_contract_single_gpu():
    """bench.py prints ONE JSON line with the driver's keys plus roofline (kernel time from hipEvents)."""
    d = _run_bench([])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["dtype"] == "f64" and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["higher_is_better"] is True
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert d["value"] > 1e8 and "workload" in d["config"]   # the BASELINE target is 1e5 trajectories/s


def test_bench_multi_process_path_on_one_gpu():
    """The N > 1 code path of bench.py (one process per rank, barrier + max-reduce of the timing) with two ranks
    sharing the only GPU of this box; gloo carries the two tiny collectives here, RCCL on a real multi-GPU node."""
    d = _run_bench(["--gpus", "2", "--backend", "gloo", "--same-device"], nproc=2)
    assert d["n_gpus"] == 2 and d["value"] > 1e7     # (sanity only: two ranks share the GPU with the other test workers)


def test_mixed_batch_config4_buckets(ctx):
    """BASELINE config 4: N in {8 (jerk), 10 (snap), 12}, K in {4, 8, 16, 32}, D = 3, mixed in one request list; every
    trajectory is checked against the oracle (N <= 10: 1e-9) or the mpmath truth (N = 12: 1e-7)."""
    import mav_trajectory_generation_amd as m
    from oracle import oracle_mp as omp
    rng = np.random.default_rng(4)
    problems, meta = [], []
    for (n, d) in ((8, 3), (10, 4), (12, 5)):
        for k in (4, 8, 16, 32):
            masks, times, d_fixed = helpers.reference_batch(3, k, n, 3, 7000 + n * 100 + k)
            for b in range(3):
                problems.append(dict(n_coeffs=n, derivative=d, masks=masks, times=times[b], d_fixed=d_fixed[b]))
                meta.append((n, d, masks))
    order = rng.permutation(len(problems))
    solver = m.MixedBatchSolver(ctx)
    coeffs, costs = solver.solve([problems[i] for i in order], want_cost=True)
    assert len(solver.plans) == 12
    for pos, i in enumerate(order):
        n, d, masks = meta[i]
        p = problems[i]
        if n == 12:
            ref, _, j = omp.solve(n, d, masks, p["times"], p["d_fixed"])
            tol = 1e-7
        else:
            ref, _, j = onp.solve_batch(n, d, masks, p["times"][None], p["d_fixed"][None])
            ref, j = ref[0], j[0]
            tol = 1e-9
        assert helpers.poly_relerr(coeffs[pos], ref) < tol, (n, len(p["times"]))
        assert abs(costs[pos] - j) <= 1e-6 * abs(j)
    solver.close()


def test_mixed_batch_streams_and_graph_replay(ctx):
    """Config 4 as one device-resident request: buckets spread over 4 HIP streams, then the whole request captured
    into one hipGraph and replayed on refreshed inputs -- every variant must reproduce the one-stream results bit for
    bit, and those match the oracle."""
    import torch
    import mav_trajectory_generation_amd as m
    buckets, hosts = [], []
    for (n, d) in ((8, 3), (10, 4), (12, 5)):
        for k in (4, 8, 16, 32):
            masks, times, d_fixed = helpers.reference_batch(70, k, n, 3, 9000 + n * 100 + k)
            hosts.append((n, d, masks, times, d_fixed))
            buckets.append(dict(n_coeffs=n, derivative=d, masks=masks, times=torch.from_numpy(times).cuda(),
                                d_fixed=torch.from_numpy(d_fixed).cuda()))
    one = m.MixedBatchSolver(ctx, n_streams=1)
    ref = [(c.clone(), j.clone()) for c, j in one.solve_device(buckets, want_cost=True)]
    torch.cuda.synchronize()
    one.sync()
    for (n, d, masks, times, d_fixed), (c, j) in zip(hosts, ref):
        if n <= 10:
            c_lit, _, j_lit = onp.solve_batch(n, d, masks, times[:4], d_fixed[:4])
            assert helpers.poly_relerr(c[:4].cpu().numpy(), c_lit) < 1e-9
            assert np.allclose(j[:4].cpu().numpy(), j_lit, rtol=1e-7)
    four = m.MixedBatchSolver(ctx, n_streams=4)
    got = four.solve_device(buckets, want_cost=True)
    torch.cuda.synchronize()
    four.sync()
    assert len(four.lanes) == 4
    for (c, j), (c0, j0) in zip(got, ref):
        assert torch.equal(c, c0) and torch.allclose(j, j0, rtol=1e-12)
    graph, out = four.capture(buckets, want_cost=True)
    for c, j in out:
        c.zero_()
    graph.replay()
    torch.cuda.synchronize()
    for (c, j), (c0, j0) in zip(out, ref):
        assert torch.equal(c, c0) and torch.allclose(j, j0, rtol=1e-12)
    # new segment times in place (what a time optimiser does between iterations), replay, compare with a fresh solve
    for b in buckets:
        b["times"].mul_(1.25)
    graph.replay()
    fresh = one.solve_device(buckets, want_cost=True)
    torch.cuda.synchronize()
    for (c, j), (c1, j1) in zip(out, fresh):
        assert torch.equal(c, c1)
    del graph
    four.close()
    one.close()


def test_merged_mixed_request(ctx):
    """mtg_multi_*: buckets that share N, D, masks pattern and derivative run as ONE launch, and the N = 8 / 10 / 12 groups
    join one cross-structure launch (config 4: 1 launch instead of 12).  Results must equal the per-bucket launches bit for bit -- eager, replayed from a hipGraph, with cost and d_P
    outputs, with an un-mergeable item (ragged masks -> ordinary path) and a single-item group in the same request."""
    import torch
    import mav_trajectory_generation_amd as m
    buckets = []
    for (n, d) in ((8, 3), (10, 4), (12, 5)):
        for k in (4, 8, 16, 32):
            masks, times, d_fixed = helpers.reference_batch(70, k, n, 3, 9100 + n * 100 + k)
            buckets.append(dict(n_coeffs=n, derivative=d, masks=masks, times=torch.from_numpy(times).cuda(),
                                d_fixed=torch.from_numpy(d_fixed).cuda()))
    # not mergeable: ragged per-vertex masks; and a group of one (N = 6)
    for (n, d, k, masks) in ((10, 4, 6, [31, 1, 3, 1, 5, 9, 31]), (6, 2, 5, None)):
        masks, times, d_fixed = helpers.reference_batch(33, k, n, 3, 777, masks)
        buckets.append(dict(n_coeffs=n, derivative=d, masks=masks, times=torch.from_numpy(times).cuda(),
                            d_fixed=torch.from_numpy(d_fixed).cuda()))
    one = m.MixedBatchSolver(ctx, n_streams=1)
    ref = [(c.clone(), j.clone()) for c, j in one.solve_device(buckets, want_cost=True)]
    torch.cuda.synchronize()
    one.sync()
    solver = m.MixedBatchSolver(ctx, n_streams=3)
    req = solver.merged(buckets, want_cost=True)
    assert req.launch_count == 1 + 2            # one cross-structure launch (N = 8, 10, 12) + the two ordinary launches
    got = req.solve()
    torch.cuda.synchronize()
    solver.sync()
    def same(b, c, j, c0, j0):
        n_, k_ = b["n_coeffs"], b["times"].shape[1]
        if (n_, k_) in ((8, 32), (10, 32), (12, 16), (12, 32)):
            # the per-bucket launches of these shapes are the factor-store dimension-in-lane kernels (MtgCfg::kFS, round 4: the
            # back-substitution works from the LDL^T factor of a step's pivot block), the merged launch WITH cost output runs the
            # G-form bodies: the same solution up to the association of f^2 products per chain step
            den = c0.abs().amax(dim=-1, keepdim=True).clamp_min(1e-300)
            # (N = 12: on this batch either form is up to 1.0e-7 from the 50-digit solution -- K = 32, segment-time ratio 17.8)
            assert float(((c - c0).abs() / den).max()) < (1e-11 if n_ <= 10 else 1e-7)
            assert torch.allclose(j, j0, rtol=1e-10 if n_ <= 10 else 5e-8)
        else:
            assert torch.equal(c, c0) and torch.allclose(j, j0, rtol=1e-12)

    for b, (c, j), (c0, j0) in zip(buckets, got, ref):
        same(b, c, j, c0, j0)
    # (coefficient-only merged launches run the same factor-store bodies as the per-bucket ones: bit for bit, every bucket)
    req_c = solver.merged(buckets)
    got_c = req_c.solve()
    torch.cuda.synchronize()
    solver.sync()
    for (c, _), (c0, _) in zip(got_c, ref):
        assert torch.equal(c, c0)
    graph = req.capture()
    for c, j in req.out:
        c.zero_()
    for b in buckets:
        b["times"].mul_(1.1)
    graph.replay()
    fresh = one.solve_device(buckets, want_cost=True)
    torch.cuda.synchronize()
    for b, (c, j), (c1, j1) in zip(buckets, req.out, fresh):
        same(b, c, j, c1, j1)
    del graph
    req.close()
    req_c.close()
    # d_P output through the C-ABI wrapper directly, SoA layout, two items of one structure
    plan_a = m.Plan(ctx, 10, 3, 16, 4, m.ends_full_masks(10, 16))
    plan_b = m.Plan(ctx, 10, 3, 5, 4, m.ends_full_masks(10, 5))
    items = []
    for plan, k in ((plan_a, 16), (plan_b, 5)):
        t, f = m.random_waypoint_batch(100, k, 3, 10, plan.fixed_mask, seed=3 + k, device="cuda", layout="soa")
        items.append(dict(plan=plan, times=t, d_fixed=f, layout="soa"))
    ms = m.MultiSolve(ctx, items, want_cost=True, want_free=True)
    assert ms.launch_count == 1
    res = ms.solve()
    ctx.sync()
    for it, (co, fr, cost) in zip(items, res):
        co1, fr1, cost1 = it["plan"].solve(it["times"], it["d_fixed"], layout="soa", want_free=True, want_cost=True)
        ctx.sync()
        assert torch.equal(co, co1) and torch.equal(fr, fr1) and torch.allclose(cost, cost1, rtol=1e-12)
    ms.close()
    plan_a.close()
    plan_b.close()
    solver.close()
    one.close()


def test_config5_full_size_properties(ctx):
    """BASELINE config 5 shape at its per-GPU share (100k / 8): K = 16, D = 4 (x, y, z, yaw), interior vertices fix
    position, velocity and acceleration.  checkPath over the whole batch + oracle parity on a subset, for both the
    small-launch (dimension-split static) and large-launch (rolled) kernel choices."""
    import torch
    import mav_trajectory_generation_amd as m
    masks = m.ends_full_masks(10, 16, 7)
    plan = m.Plan(ctx, 10, 4, 16, 4, masks)
    for bsz in (12_500, 70_000):
        t, f = m.random_waypoint_batch(bsz, 16, 4, 10, masks, seed=3, device="cuda", yaw_dim=True)
        co, _, cost = plan.solve(t, f, want_cost=True)
        ctx.sync()
        tn, fn, cn = t.cpu().numpy(), f.cpu().numpy(), co.cpu().numpy()
        assert helpers.check_path(masks, tn, fn, cn) < 1e-6
        idx = np.arange(0, bsz, bsz // 100)[:100]
        c_lit, _, j_lit = onp.solve_batch(10, 4, masks, tn[idx], fn[idx])
        assert helpers.poly_relerr(cn[idx], c_lit) < 1e-9
        assert np.allclose(cost.cpu().numpy()[idx], j_lit, rtol=1e-8)
    plan.close()


def test_update_from_free_large_batch(ctx):
    """setFreeConstraints path at size: coefficients rebuilt from the solver's own d_P equal the solve's."""
    import torch
    import mav_trajectory_generation_amd as m
    masks = m.ends_full_masks(10, 8)
    plan = m.Plan(ctx, 10, 3, 8, 4, masks)
    t, f = m.random_waypoint_batch(50_001, 8, 3, 10, masks, seed=8, device="cuda")
    co, fr, cost = plan.solve(t, f, want_free=True, want_cost=True)
    co2, cost2 = plan.update_from_free(t, f, fr, want_cost=True)
    ctx.sync()
    den = co.abs().amax(dim=-1).clamp_min(1e-300)
    assert float(((co - co2).abs().amax(dim=-1) / den).max()) < 1e-12
    assert torch.allclose(cost, cost2, rtol=1e-10)
    plan.close()


@pytest.mark.parametrize("n,d,k,dim,interior,bsz", [
    (10, 4, 8, 3, 1, 777),      # piece 1920 B: aligned
    (10, 4, 7, 3, 1, 1000),     # piece 1680 B: phases 0 / 16 / 32 / 48
    (10, 4, 1, 3, 1, 200), (10, 4, 2, 3, 1, 65), (10, 4, 3, 3, 1, 64), (10, 4, 33, 3, 1, 130),
    (8, 3, 5, 3, 1, 321), (8, 3, 16, 3, 1, 500), (12, 5, 3, 3, 1, 259), (12, 5, 8, 3, 1, 300),
    (10, 4, 5, 4, 7, 333), (10, 4, 8, 4, 7, 140),
    (8, 3, 7, 1, 1, 500), (10, 4, 9, 1, 1, 129), (12, 5, 4, 1, 1, 400), (10, 4, 3, 1, 7, 97),
])
def test_update_from_free_whole_sector_output(ctx, n, d, k, dim, interior, bsz, monkeypatch):
    """setFreeConstraints path, rolled form with whole-sector output (mtg_update_slab_kernel): coefficients rebuilt from the
    solver's own d_P equal the solve's, and equal -- bit for bit -- the per-segment staging kernel's (MTG_NO_SLAB context);
    ragged batches, aligned and unaligned pieces, one / three / four dimensions."""
    import torch
    import mav_trajectory_generation_amd as m
    masks = m.ends_full_masks(n, k, interior)
    plan = m.Plan(ctx, n, dim, k, d, masks)
    t, f = m.random_waypoint_batch(bsz, k, dim, n, masks, seed=k + n, device="cuda")
    co, fr, cost = plan.solve(t, f, want_free=True, want_cost=True)
    co2, cost2 = plan.update_from_free(t, f, fr, want_cost=True)
    co3 = plan.update_from_free(t, f, fr)
    if isinstance(co3, tuple):
        co3 = co3[0]
    ctx.sync()
    den = co.abs().amax(dim=-1).clamp_min(1e-300)
    assert float(((co - co2).abs().amax(dim=-1) / den).max()) < (1e-11 if n < 12 else 1e-9)
    assert torch.allclose(cost, cost2, rtol=1e-9)
    assert torch.equal(co2, co3)
    plan.close()
    monkeypatch.setenv("MTG_NO_SLAB", "1")
    ctx2 = m.Context(0)
    plan2 = m.Plan(ctx2, n, dim, k, d, masks)
    co4, cost4 = plan2.update_from_free(t, f, fr, want_cost=True)
    ctx2.sync()
    assert torch.equal(co2, co4) and torch.equal(cost2, cost4)
    plan2.close()
    ctx2.close()


MEL = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_mellinger.npz"))
MEL_NAMES = sorted({k.split("/")[0] for k in MEL.files})
# Gradient tolerance against the reference's own member, relative to each trajectory's largest gradient component: the
# gradient is a forward difference (J(T') - J(T)) / 0.1 of two costs that each agree with the reference's to ~1e-10.
MEL_TOL = 1e-9


def mel_err(got, ref):
    scale = np.maximum(np.abs(ref).max(axis=1, keepdims=True), 1e-300)
    return float((np.abs(got - ref) / scale).max())


def mel_assert(got, ref, n, d, masks, times, d_fixed, tol):
    """Gradient against the reference-side value, per trajectory within `tol` of its largest component.  The forward difference
    amplifies the cost's float64 evaluation error by J / (0.1 |g|): a trajectory between tol and 10 tol (seen: 1.14e-9 on the
    16-segment case, against the reference's member and its restatement alike) is arbitrated by the 50-digit solve of the same K + 1
    problems -- accepted only when the HIP gradient is within tol of THAT (as the coefficient parity tests do)."""
    scale = np.maximum(np.abs(ref).max(axis=1), 1e-300)
    err = np.abs(got - ref).max(axis=1) / scale
    assert err.max() <= 10 * tol, err
    for b in np.nonzero(err > tol)[0]:
        from oracle import oracle_mp
        k = times.shape[1]
        j0 = oracle_mp.solve(n, d, masks, times[b], d_fixed[b])[2]
        truth = np.zeros(k)
        for s_ in range(k):
            tb = times[b].copy()
            for i in range(k):
                tb[i] += 0.1 if i == s_ else -0.1 / (k - 1.0)
            truth[s_] = (oracle_mp.solve(n, d, masks, np.maximum(0.1, tb), d_fixed[b])[2] - j0) / 0.1
        e_hip, e_ref = np.abs(got[b] - truth).max() / scale[b], np.abs(ref[b] - truth).max() / scale[b]
        assert e_hip <= tol, (int(b), float(err[b]), float(e_hip), float(e_ref))


@pytest.mark.parametrize("name", MEL_NAMES)
@pytest.mark.parametrize("layout", ["aos", "soa"])
def test_mellinger_entry_vs_the_references_own_member(ctx, name, layout):
    """mtg_mellinger_cost_gradient against the outputs of the reference's OWN
    PolynomialOptimizationNonLinear<N>::getCostAndGradientMellinger (NL:287-364) compiled and run in the build container
    (oracle/ref_nonlinear_wrap.cpp; tests/golden/make_reference_mellinger_golden.py), incl. the lower clamp (NL:338-340), one
    segment (NL:295-302), ragged masks, config 5's shape and N = 12."""
    import torch
    import mav_trajectory_generation_amd as m
    n, d = int(MEL[f"{name}/n"]), int(MEL[f"{name}/d"])
    masks = [int(x) for x in MEL[f"{name}/masks"]]
    times, d_fixed = MEL[f"{name}/times"], MEL[f"{name}/d_fixed"]
    dim, k = d_fixed.shape[1], times.shape[1]
    plan = m.Plan(ctx, n, dim, k, d, masks)
    t, f = torch.from_numpy(times).cuda(), torch.from_numpy(d_fixed).cuda()
    if layout == "soa":
        t, f = t.t().contiguous(), f.permute(1, 2, 0).contiguous()
    j0, grad = m.mellinger_cost_and_gradient(plan, t, f, layout=layout)
    ctx.sync()
    got = grad.cpu().numpy() if layout == "aos" else grad.t().cpu().numpy()
    tol = MEL_TOL if n <= 10 else 5e-8          # N = 12: the reference's own float64 evaluation (tol_for of test_gpu_vs_reference.py)
    assert np.abs(j0.cpu().numpy() / MEL[f"{name}/cost_ref"] - 1).max() < tol
    mel_assert(got, MEL[f"{name}/grad_ref"], n, d, masks, times, d_fixed, 10 * tol if n > 10 else tol)
    if k == 1:
        assert np.all(got == 0.0)
    plan.close()


def test_cost_only_and_mellinger_gradient(ctx):
    """MTG_FLAG_COST_ONLY and the batched Mellinger cost/gradient step (K+1 perturbed-time solves per trajectory in
    one launch) against the REFERENCE's own getCostAndGradientMellinger (polynomial_optimization_nonlinear_impl.h:287-364;
    tests/golden/reference_mellinger.npz) and its restatement on the oracle."""
    import torch
    import mav_trajectory_generation_amd as m
    bsz, k = 12, 8
    masks, times, d_fixed = helpers.reference_batch(bsz, k, 10, 3, 60606)
    plan = m.Plan(ctx, 10, 3, k, 4, masks)
    t, f = torch.from_numpy(times).cuda(), torch.from_numpy(d_fixed).cuda()
    _, _, cost_full = plan.solve(t, f, want_cost=True)
    cost_only = plan.solve_cost_only(t, f)
    j0, grad = m.mellinger_cost_and_gradient(plan, t, f)
    ctx.sync()
    assert torch.allclose(cost_only, cost_full, rtol=1e-12)
    assert torch.allclose(j0, cost_full, rtol=1e-12)
    # the restatement of the reference's loop (pinned on the reference's own member: tests/test_reference_build.py) ...
    _, want = onp.mellinger_cost_gradient(10, 4, masks, times, d_fixed)
    got = grad.cpu().numpy()
    assert np.abs(got - want).max() <= MEL_TOL * np.abs(want).max()
    # ... and the reference's own getCostAndGradientMellinger run in the build container on these inputs (fixture snap_k8_d3)
    assert np.array_equal(MEL["snap_k8_d3/times"], times) and np.array_equal(MEL["snap_k8_d3/d_fixed"], d_fixed)
    assert np.abs(j0.cpu().numpy() / MEL["snap_k8_d3/cost_ref"] - 1).max() < 1e-9
    assert mel_err(got, MEL["snap_k8_d3/grad_ref"]) <= MEL_TOL
    plan.close()


def test_batched_sampling_vs_oracle(ctx):
    """mtg_sample_range (row N3) against the oracle restatement of Trajectory::evaluate / Polynomial::evaluate, incl.
    samples past the end (clamped, counted out by n_valid), SoA times, N = 12 and D = 4."""
    import torch
    import mav_trajectory_generation_amd as m
    # shapes: compile-time kernels (N in {10, 12, 8} x D = 3, N = 10 x D = 1; ND in {1, 3, 5}; K <= 8) with full chunks,
    # a partial last chunk, fewer than 64 samples in total and more workers than chunks; run-time-shape kernel otherwise
    # (even ND * D, K > 8, other N)
    for (n, d, k, dim, bsz, S, nd) in [(10, 4, 8, 3, 37, 101, 5), (12, 5, 3, 4, 5, 64, 5), (8, 3, 5, 1, 9, 33, 3),
                                       (10, 4, 8, 3, 700, 128, 5), (10, 4, 8, 3, 3, 17, 5), (10, 4, 2, 3, 64, 9, 3),
                                       (12, 5, 4, 3, 50, 77, 1), (8, 3, 8, 3, 41, 100, 5), (10, 4, 6, 1, 33, 65, 5),
                                       (10, 4, 16, 3, 21, 90, 5), (10, 4, 8, 3, 20, 50, 2), (6, 2, 4, 3, 11, 70, 3),
                                       (10, 4, 8, 4, 30, 70, 5), (10, 4, 5, 4, 19, 33, 4), (12, 5, 8, 4, 12, 129, 3),
                                       (10, 4, 16, 4, 25, 80, 5), (10, 4, 12, 3, 40, 64, 5), (8, 3, 9, 3, 17, 100, 3)]:
        masks, times, d_fixed = helpers.reference_batch(bsz, k, n, dim, 515 + n)
        plan = m.Plan(ctx, n, dim, k, d, masks)
        t, f = torch.from_numpy(times).cuda(), torch.from_numpy(d_fixed).cuda()
        co, _, _ = plan.solve(t, f)
        dt = float(times.sum(axis=1).max()) / (S - 7)     # the longest trajectory also runs past its end
        out, nv = m.sample_range(ctx, co, t, 0.0, dt, S, nd, want_valid=True)
        out_soa = m.sample_range(ctx, co, t.t().contiguous(), 0.0, dt, S, nd, times_layout="soa")
        ctx.sync()
        want, want_nv = onp.sample_batch(co.cpu().numpy(), times, 0.0, dt, S, nd)
        got = out.cpu().numpy()
        # per (trajectory, derivative) scale, floored by the trajectory's overall magnitude: with few segments and a
        # coarse grid every sample can sit on a rest vertex, where the true derivatives are 0 and both sides hold noise
        scale = np.abs(want).max(axis=(1, 3), keepdims=True)
        scale = np.maximum(scale, 1e-2 * np.abs(want).max(axis=(1, 2, 3), keepdims=True)) + 1e-300
        assert (np.abs(got - want) / scale).max() < 1e-11
        assert np.array_equal(nv.cpu().numpy(), want_nv)
        assert torch.equal(out, out_soa)
        plan.close()


@pytest.mark.parametrize("n,d,k,dim,masks,bsz,layout", [
    (10, 4, 8, 3, None, 70, "soa"),                              # static kernel, batch not a multiple of the tile
    (10, 4, 16, 3, None, 9, "aos"),                              # rolled kernel (run-time K)
    (10, 4, 6, 3, [31, 1, 3, 1, 5, 9, 31], 5, "aos"),            # ragged masks: generic kernel
    (10, 4, 16, 4, [31] + [7] * 15 + [31], 6, "soa"),            # config 5 shape
    (8, 3, 4, 3, None, 130, "aos"),
    (10, 4, 1, 3, None, 4, "aos"),                               # one segment: zero gradient (impl:295-302)
])
def test_mellinger_cost_gradient_entry(ctx, n, d, k, dim, masks, bsz, layout):
    """mtg_mellinger_cost_gradient (C ABI; perturbed times formed inside the kernel) against the restatement of
    getCostAndGradientMellinger (polynomial_optimization_nonlinear_impl.h:287-364; pinned on the reference's own member in
    tests/test_reference_build.py) on the oracle, incl. the lower clamp."""
    import torch
    import mav_trajectory_generation_amd as m
    masks, times, d_fixed = helpers.reference_batch(bsz, k, n, dim, 424242, masks)
    times[0, 0] = 0.12            # T - h/(K-1) falls below the bound 0.1 -> clamped (impl:338-340)
    if k > 1:
        times[1 % bsz, k - 1] = 0.05   # already below the bound: every variant clamps it
    plan = m.Plan(ctx, n, dim, k, d, masks)
    t, f = torch.from_numpy(times).cuda(), torch.from_numpy(d_fixed).cuda()
    if layout == "soa":
        t, f = t.t().contiguous(), f.permute(1, 2, 0).contiguous()
    j0, grad = m.mellinger_cost_and_gradient(plan, t, f, layout=layout)
    ctx.sync()
    got = grad.cpu().numpy() if layout == "aos" else grad.t().cpu().numpy()
    nchk = min(bsz, 6)
    jd, want = onp.mellinger_cost_gradient(n, d, masks, times[:nchk], d_fixed[:nchk])
    assert np.allclose(j0.cpu().numpy()[:nchk], jd, rtol=1e-8)
    mel_assert(got[:nchk], want, n, d, masks, times[:nchk], d_fixed[:nchk], MEL_TOL)
    if k == 1:
        assert np.all(got == 0.0)
    plan.close()


def test_concurrent_mixed_request(ctx):
    """mtg_multi_* with MTG_FLAG_CONCURRENT_ITEMS: one C call enqueues every bucket as its own best launch on the
    context's side streams (fork / join on the context's stream).  Results equal the per-bucket launches bit for bit --
    eager, repeated back to back, with cost outputs, two items of one plan, and after new values in the input buffers."""
    import torch
    import mav_trajectory_generation_amd as m
    buckets = []
    for (n, d) in ((8, 3), (10, 4), (12, 5)):
        for k in (4, 8, 16, 32):
            masks = m.ends_full_masks(n, k)
            t, f = m.random_waypoint_batch(700, k, 3, n, masks, seed=100 * n + k, device="cuda", layout="soa")
            buckets.append(dict(n_coeffs=n, derivative=d, masks=masks, times=t, d_fixed=f, layout="soa"))
    masks = m.ends_full_masks(10, 8)      # a second bucket of an existing structure: shares the plan (and its workspace)
    t, f = m.random_waypoint_batch(300, 8, 3, 10, masks, seed=5, device="cuda", layout="soa")
    buckets.append(dict(n_coeffs=10, derivative=4, masks=masks, times=t, d_fixed=f, layout="soa"))
    one = m.MixedBatchSolver(ctx, n_streams=1)
    for want_cost in (False, True):
        ref = [(c.clone(), None if j is None else j.clone()) for c, j in one.solve_device(buckets, want_cost=want_cost)]
        torch.cuda.synchronize()
        one.sync()
        solver = m.MixedBatchSolver(ctx, n_streams=1)
        req = solver.concurrent(buckets, want_cost=want_cost)
        assert req.launch_count == len(buckets)
        for _ in range(3):
            got = req.solve()
        torch.cuda.synchronize()
        solver.sync()
        for (c, j), (c0, j0) in zip(got, ref):
            assert torch.equal(c, c0)
            if want_cost:
                assert torch.allclose(j, j0, rtol=1e-12)
        for c, _ in req.out:
            c.zero_()
        for b in buckets:
            b["times"].mul_(1.07)
        req.solve()
        fresh = one.solve_device(buckets, want_cost=want_cost)
        torch.cuda.synchronize()
        for (c, j), (c1, j1) in zip(req.out, fresh):
            assert torch.equal(c, c1)
        req.close()
        solver.close()
    one.close()


def test_cross_structure_dimlane_request(ctx):
    """mtg_multi_*: items with canonical SoA inputs whose plans have a static dimension-in-lane configuration join ONE
    cross-structure launch (mtg_solve_dl_any_kernel), whatever their N and K -- BASELINE config 4's twelve buckets in one
    launch, back-substitution data in registers.  Bit-for-bit equal to the per-bucket dimension-in-lane launches; ragged
    bucket sizes (tile tails), two buckets of one plan, an AoS bucket (round 3: part of the same launch; a strided one -> its
    own ordinary launch) in the same request, re-solve with new values, bad segment time reported through the context status."""
    import torch
    import mav_trajectory_generation_amd as m
    buckets, sizes = [], [700, 21, 1, 64, 333, 2500, 43, 700, 700, 22, 640, 100]
    i = 0
    for (n, d) in ((8, 3), (10, 4), (12, 5)):
        for k in (4, 8, 16, 32):
            masks = m.ends_full_masks(n, k)
            t, f = m.random_waypoint_batch(sizes[i], k, 3, n, masks, seed=100 * n + k, device="cuda", layout="soa")
            buckets.append(dict(n_coeffs=n, derivative=d, masks=masks, times=t, d_fixed=f, layout="soa"))
            i += 1
    masks = m.ends_full_masks(10, 8)
    t, f = m.random_waypoint_batch(300, 8, 3, 10, masks, seed=5, device="cuda", layout="soa")
    buckets.append(dict(n_coeffs=10, derivative=4, masks=masks, times=t, d_fixed=f, layout="soa"))
    n_dl = len(buckets)
    masks, times, d_fixed = helpers.reference_batch(50, 16, 10, 3, 4321)
    buckets.append(dict(n_coeffs=10, derivative=4, masks=masks, times=torch.from_numpy(times).cuda(),
                        d_fixed=torch.from_numpy(d_fixed).cuda()))
    solver = m.MixedBatchSolver(ctx, n_streams=1)
    req = solver.merged(buckets)
    assert req.launch_count == 1        # ONE cross-structure launch, the AoS bucket included

    def reference():
        out = []
        for b in buckets:
            plan = solver.plan_for(b["n_coeffs"], 3, len(b["masks"]) - 1, b["derivative"], b["masks"], 0)
            lay = b.get("layout", "aos")
            co, _, _ = plan.solve(b["times"], b["d_fixed"], layout=lay, dims="dimlane" if lay == "soa" else "auto")
            out.append(co.clone())
        return out

    for round_ in range(2):
        for c, _ in req.out:
            c.fill_(7.0)
        got = req.solve()
        torch.cuda.synchronize()
        solver.sync()
        ref = reference()
        torch.cuda.synchronize()
        for j, ((c, _), c0) in enumerate(zip(got, ref)):
            assert torch.equal(c, c0), f"bucket {j}"
        for b in buckets[:n_dl]:
            b["times"].mul_(1.13)
    buckets[3]["times"][5, 17] = -1.0      # a bad segment time inside the cross-structure launch
    req.solve()
    with pytest.raises(m.MtgError) as e:
        solver.sync()
    assert e.value.code == -2
    req.close()
    solver.close()


def test_sequence_with_events(ctx):
    """mtg_solve_linear_sequence_events: n independent batches enqueued by one call, start / stop events recorded inside
    the call on the context's stream; results equal single solves, the events bracket the launches."""
    import ctypes
    import torch
    import mav_trajectory_generation_amd as m
    masks = m.ends_full_masks(10, 8)
    plan = m.Plan(ctx, 10, 3, 8, 4, masks)
    sets = []
    for s in range(3):
        t, f = m.random_waypoint_batch(1000, 8, 3, 10, masks, seed=40 + s, device="cuda", layout="soa")
        sets.append((t, f, torch.zeros((1000, 8, 3, 10), dtype=torch.float64, device="cuda")))
    torch.cuda.synchronize()     # (the raw C call below does not order the context's stream behind torch's: inputs first)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(ctx.stream):
        e0.record(ctx.stream)
        e1.record(ctx.stream)          # materialise the hipEvent_t handles
        arr = [(ctypes.c_void_p * 3)(*[x[j].data_ptr() for x in sets]) for j in range(3)]
        lay = plan.layout(1000, "soa")
        rc = plan.lib.mtg_solve_linear_sequence_events(plan.handle, 3, 1000, ctypes.byref(lay), arr[0], arr[1], arr[2], 0,
                                                       ctypes.c_void_p(e0.cuda_event), ctypes.c_void_p(e1.cuda_event))
        assert rc == 0
        torch.cuda.synchronize()
    ctx.sync()
    assert e0.elapsed_time(e1) > 0.0
    for t, f, co in sets:
        ref, _, _ = plan.solve(t, f, layout="soa")
        ctx.sync()
        assert torch.equal(co, ref)
    plan.close()


@pytest.mark.parametrize("layout", ["soa", "aos"])
def test_device_side_generator_and_compare(ctx, layout):
    """mtg_generate_waypoints / mtg_compare_coefficients: inputs with the reference's
    random-waypoint recipe made by the library's own kernel, results checked on the device."""
    import torch
    import mav_trajectory_generation_amd as m
    for (n, k, dim, d, interior, yaw) in ((10, 8, 3, 4, 1, False), (10, 16, 4, 4, 7, True), (8, 5, 3, 3, 1, False)):
        masks = m.ends_full_masks(n, k, interior)
        plan = m.Plan(ctx, n, dim, k, d, masks)
        bsz = 3000
        t, f = plan.generate_waypoints(bsz, seed=77, layout=layout, yaw_dim=yaw)
        t2, f2 = plan.generate_waypoints(bsz, seed=77, layout=layout, yaw_dim=yaw)
        t3, _ = plan.generate_waypoints(bsz, seed=78, layout=layout, yaw_dim=yaw)
        torch.cuda.synchronize()
        assert torch.equal(t, t2) and torch.equal(f, f2) and not torch.equal(t, t3)      # reproducible per seed
        tb = t.t() if layout == "soa" else t                       # [B][K]
        fb = f.permute(2, 0, 1) if layout == "soa" else f          # [B][D][n_fixed]
        assert torch.isfinite(tb).all() and torch.isfinite(fb).all() and float(tb.min()) > 0.0
        # position columns: the first fixed slot of every vertex
        h = n // 2
        cols, c = [], 0
        for v in range(k + 1):
            cols.append(c)
            c += bin(masks[v]).count("1")
        pos = fb[:, :, cols]                                       # [B][D][K+1]
        lim = torch.full((dim,), 10.0, dtype=torch.float64, device="cuda")
        if yaw:
            lim[3] = 3 * np.pi
        assert bool((pos.abs() <= lim[None, :, None]).all())
        dist = (pos[:, :, 1:] - pos[:, :, :-1]).norm(dim=1)        # [B][K]
        assert float((dist <= 0.2).double().mean()) < 1e-3
        want = dist / 3.0 * 2 * (1.0 + 6.5 * 3.0 / 5.0 * torch.exp(-dist / 3.0 * 2))
        assert float(((tb - want).abs() / want).max()) < 1e-12    # estimateSegmentTimesNfabian
        # end vertices at rest: every fixed derivative above position is zero
        assert float(fb[:, :, 1:h].abs().max()) == 0.0 and float(fb[:, :, c - h + 1:].abs().max()) == 0.0
        # roughly uniform positions: mean ~ 0, std ~ box / sqrt(3)
        p0 = pos[:, 0, :].flatten()
        assert abs(float(p0.mean())) < 0.3 and abs(float(p0.std()) - 10.0 / 3 ** 0.5) < 0.3
        co, _, _ = plan.solve(t.contiguous(), f.contiguous(), layout=layout)
        ctx.sync()
        assert torch.isfinite(co).all()
        # on-device comparison against torch: identical buffers, then a perturbed one
        rel, ab = ctx.compare_coefficients(co, co.clone())
        assert rel == 0.0 and ab == 0.0
        other = co.clone()
        other[5, 1, 0, 2] += 1e-6
        rel, ab = ctx.compare_coefficients(other, co)
        den = co.abs().amax(dim=-1).clamp_min(1e-300)
        want_rel = float(((other - co).abs().amax(dim=-1) / den).max())
        assert abs(rel - want_rel) <= 1e-12 * want_rel and abs(ab - float((other - co).abs().max())) < 1e-18
        # a NaN on either side saturates both outputs (a caller with a loose threshold must not accept NaN output)
        other[7, 0, 1, 3] = float("nan")
        for a_, b_ in ((other, co), (co, other)):
            rel, ab = ctx.compare_coefficients(a_, b_)
            assert rel >= 1e299 and ab >= 1e299
        plan.close()


def test_plain_c_consumer_of_the_abi():
    """tools/c/roundtrip.c: a C program that sees only include/mtg_hip.h generates a batch on the device, solves it with
    two kernels, compares the results on the device and times the launch (exit code 0 = the two kernels agree to 1e-10)."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "c", "roundtrip")
    assert os.path.exists(exe), "built by __graft_entry__.build()"
    for argv in (["20000", "8"], ["3000", "11"]):
        r = subprocess.run([exe] + argv, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "trajectories/s" in r.stdout
        assert "in one call vs one call each: max norm-wise rel diff" in r.stdout     # (exit code 0: the queue agrees to 1e-10)
