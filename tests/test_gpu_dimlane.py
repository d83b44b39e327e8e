"""GPU tests of the dimension-in-lane launch form (csrc/mtg_dimlane.h): all dimensions of a trajectory in one wavefront,
whole-sector coefficient stores.  Called through the C ABI; checked against the oracle (1e-9 norm-wise per polynomial,
the north-star tolerance), against the committed golden fixtures, and bit-for-bit against the dimension-split form,
whose per-lane arithmetic it shares.  Also: the per-trajectory status output and status flags raised by kernels that are
replayed from a captured hipGraph."""
import os

import numpy as np
import pytest

import helpers
from oracle import oracle_np as onp

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "solve_linear_golden.npz"))


def case(name):
    pre = name + "/"
    return {k[len(pre):]: GOLD[k] for k in GOLD.files if k.startswith(pre)}


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    import mav_trajectory_generation_amd as m
    c = m.Context(0)
    yield c
    c.close()


def to_soa(times, d_fixed):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(times)).cuda().t().contiguous()
    f = torch.from_numpy(np.ascontiguousarray(d_fixed)).cuda().permute(1, 2, 0).contiguous()
    return t, f


# (N, K, D, derivative, interior mask): the shapes of csrc/mtg_dimlane_variants.inc
SHAPES = [(10, 8, 3, 4, 1), (10, 8, 4, 4, 1), (10, 8, 1, 4, 1), (10, 4, 3, 4, 1), (10, 2, 3, 4, 1), (8, 4, 3, 3, 1),
          (8, 8, 3, 3, 1), (12, 4, 3, 5, 1), (12, 8, 3, 5, 1), (10, 16, 4, 4, 7),
          (8, 16, 3, 3, 1), (10, 16, 3, 4, 1),     # register-resident K = 16 chains
          (10, 32, 3, 4, 1), (8, 32, 3, 3, 1), (12, 16, 3, 5, 1), (12, 32, 3, 5, 1)]   # long chains: registers + workspace (MtgCfg::WSJ)
# every other chain length up to 15 of the three standard shapes (mtg_dimlane_more_h*.inc), odd ones included
SHAPES += [(n, k, 3, n // 2 - 1, 1) for n in (8, 10, 12) for k in (3, 5, 6, 7, 9, 10, 11, 12, 13, 14, 15)]
SHAPES += [(n, k, 3, n // 2 - 1, 1) for n in (8, 10, 12) for k in range(17, 32)]   # mtg_dimlane_more_h*b.inc
SHAPES += [(10, 50, 3, 4, 1), (8, 50, 3, 3, 1)]


# shapes that also have a one-dimension-per-workgroup static variant (csrc/mtg_variants.inc): same instruction stream per lane
BITWISE_VS_SPLIT = {(10, 8, 3, 4, 1), (12, 8, 3, 5, 1), (12, 4, 3, 5, 1), (8, 8, 3, 3, 1), (10, 16, 4, 4, 7)}


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("bsz", [1, 20, 21, 22, 43, 64, 257, 3000])
def test_dimlane_vs_oracle_and_split_form(ctx, shape, bsz):
    """Ragged sizes around the tile width (64 / D trajectories per wave) for every instantiated shape."""
    import torch
    import mav_trajectory_generation_amd as m
    n, k, dim, d, mi = shape
    masks = helpers.masks_ends_full(n, k, mi)
    _, times, d_fixed = helpers.reference_batch(min(bsz, 64), k, n, dim, 4242 + bsz, masks)
    if bsz > 64:   # beyond 64 the bit-exact generator is slow: tile the 64 with scaled times (still distinct problems)
        reps = (bsz + 63) // 64
        scale = 1.0 + 0.01 * np.arange(reps).repeat(64)[:bsz]
        times = np.tile(times, (reps, 1))[:bsz] * scale[:, None]
        d_fixed = np.tile(d_fixed, (reps, 1, 1))[:bsz]
    plan = m.Plan(ctx, n, dim, k, d, masks)
    t, f = to_soa(times, d_fixed)
    co = torch.full((bsz + 1, k, dim, n), 7.0, dtype=torch.float64, device="cuda")   # one guard row behind the batch
    plan.solve(t, f, layout="soa", coeffs=co[:bsz], dims="dimlane")
    ref, _, _ = plan.solve(t, f, layout="soa", dims="split")
    ctx.sync()
    assert float(co[bsz].min()) == 7.0 and float(co[bsz].max()) == 7.0, "stores past the end of the batch"
    got = co[:bsz].cpu().numpy()
    ref = ref.cpu().numpy()
    # Long chains keep the LDL^T FACTOR of a step's pivot block where the split form keeps G = Dtilde^-1 U (MtgCfg::kFS, round 4):
    # x_l = g - (L D L^T)^-1 (U x_r) against g - G x_r -- the same numbers up to the association of f^2 products per step
    # (and the forward sweep eliminates the 2f x 2f step block partially instead of solving for G; measured: <= 6e-12 for N <= 10; both
    # forms are equally far from the 50-digit solution, tests/test_lane_emu.py; N = 12: <= 1e-9, a tenth of what either form is from it)
    assert helpers.poly_relerr(got, ref) < (1e-12 if k <= 10 else (3e-11 if n <= 10 else 3e-9))
    if shape in BITWISE_VS_SPLIT:
        assert np.array_equal(got, ref), "dimension-in-lane and dimension-split forms share the lane arithmetic"
    nchk = min(bsz, 64)
    c_lit, _, _ = onp.solve_batch(n, d, masks, times[:nchk], d_fixed[:nchk])
    tol = 1e-9 if n <= 10 else 5e-7     # N = 12: the literal float64 route itself is ~1e-8 from the 50-digit solution
    assert helpers.poly_relerr(got[:nchk], c_lit) < tol
    assert helpers.check_path(masks, times[:nchk], d_fixed[:nchk], got[:nchk]) < 1e-6
    plan.close()


@pytest.mark.parametrize("name", ["config2", "config5", "readme"])
def test_dimlane_golden_fixtures(ctx, name):
    import mav_trajectory_generation_amd as m
    c = case(name)
    n, d = int(c["n"]), int(c["d"])
    masks = [int(x) for x in c["masks"]]
    dim, k = c["d_fixed"].shape[1], c["times"].shape[1]
    plan = m.Plan(ctx, n, dim, k, d, masks)
    t, f = to_soa(c["times"], c["d_fixed"])
    co, _, _ = plan.solve(t, f, layout="soa", dims="dimlane")
    ctx.sync()
    co = co.cpu().numpy()
    assert helpers.poly_relerr(co, c["coeffs_lit"]) < 1e-9
    assert helpers.poly_relerr(co, c["coeffs_mp"]) < 1e-11
    plan.close()


def test_dimlane_is_the_default_for_the_bench_call(ctx):
    """bench.py's call (SoA inputs, coefficient output only, B = 10k) must take the dimension-in-lane form: the result is
    bit-identical to the forced form and the library's own timing hook replays a dimension-in-lane launch."""
    import torch
    import mav_trajectory_generation_amd as m
    masks = m.ends_full_masks(10, 8)
    plan = m.Plan(ctx, 10, 3, 8, 4, masks)
    t, f = m.random_waypoint_batch(10_000, 8, 3, 10, masks, seed=3, device="cuda", layout="soa")
    a, _, _ = plan.solve(t, f, layout="soa")
    us_auto = min(plan.time_last_solve(20) for _ in range(4))      # (best of four replays: other test workers share the GPU)
    b, _, _ = plan.solve(t, f, layout="soa", dims="dimlane")
    c, _, _ = plan.solve(t, f, layout="soa", dims="fused")
    us_fused = min(plan.time_last_solve(20) for _ in range(4))
    ctx.sync()
    assert torch.equal(a, b)
    den = c.abs().amax(dim=-1).clamp_min(1e-300)
    assert float(((a - c).abs().amax(dim=-1) / den).max()) < 1e-11
    assert plan.launch_form(10_000, "soa") == "dimlane"
    # (timing as a sanity check only: under `pytest -n 4` the workers share the GPU and both figures triple -- seen 29.9 vs 28.1 us;
    # alone: 8.3 vs 9.9 us)
    assert us_auto < 1.25 * us_fused
    plan.close()


def test_full_size_dimlane_properties(ctx):
    """BASELINE config 2 at full size through the default path: checkPath at 1e-6, linearity in d_F, oracle parity on
    a strided subset."""
    import torch
    import mav_trajectory_generation_amd as m
    bsz = 10_000
    masks = m.ends_full_masks(10, 8)
    plan = m.Plan(ctx, 10, 3, 8, 4, masks)
    t, f = m.random_waypoint_batch(bsz, 8, 3, 10, masks, seed=11, device="cuda", layout="soa")
    co, _, _ = plan.solve(t, f, layout="soa")
    co2, _, _ = plan.solve(t, (f * 2.0).contiguous(), layout="soa")
    ctx.sync()
    assert torch.isfinite(co).all()
    den = co.abs().amax(dim=-1).clamp_min(1e-300)
    assert float(((co2 - 2.0 * co).abs().amax(dim=-1) / den).max()) < 1e-11
    idx = np.arange(0, bsz, 20)
    th = t.t().contiguous().cpu().numpy()[idx]
    fh = f.permute(2, 0, 1).contiguous().cpu().numpy()[idx]
    c_lit, _, _ = onp.solve_batch(10, 4, masks, th, fh)
    got = co.cpu().numpy()[idx]
    assert helpers.poly_relerr(got, c_lit) < 1e-9
    assert helpers.check_path(masks, th, fh, got) < 1e-6
    plan.close()


@pytest.mark.parametrize("dims", ["dimlane", "split", "fused", "generic"])
def test_per_trajectory_status(ctx, dims):
    """A batch with some non-positive segment times (LIN:297): the batch code says THAT a trajectory failed, the
    per-trajectory status says which -- every launch form."""
    import torch
    import mav_trajectory_generation_amd as m
    bsz = 500
    masks = m.ends_full_masks(10, 8)
    plan = m.Plan(ctx, 10, 3, 8, 4, masks)
    t, f = m.random_waypoint_batch(bsz, 8, 3, 10, masks, seed=5, device="cuda", layout="soa")
    bad = [3, 20, 21, 63, 64, 255, 499]
    for i, b in enumerate(bad):
        t[i % 8, b] = -1.0 if i % 2 else 0.0
    st = torch.full((bsz,), 99, dtype=torch.int32, device="cuda")
    kw = dict(generic=True) if dims == "generic" else dict(dims=dims)
    co, _, _ = plan.solve(t, f, layout="soa", traj_status=st, **kw)
    with pytest.raises(m.MtgError) as e:
        ctx.sync()
    assert e.value.code == -2
    got = st.cpu().numpy()
    want = np.zeros(bsz, dtype=np.int32)
    want[bad] = 1
    assert np.array_equal(got & 1, want)
    # T < 0 makes the first pivot of that segment's vertex negative (T^(1-2d) < 0): the singular bit names it too
    neg = [b for i, b in enumerate(bad) if i % 2]
    assert np.all(got[neg] == 3)
    ok = np.setdiff1d(np.arange(bsz), bad)
    assert np.all(got[ok] == 0)
    assert torch.isfinite(co[torch.from_numpy(ok).cuda()]).all()
    ctx.sync()
    plan.close()


def test_status_of_graph_replays_is_reported(ctx):
    """Kernels replayed from a captured hipGraph never pass through the library: mtg_context_sync must still see the
    flags they raise (the host-side dirty flag hid them)."""
    import torch
    import mav_trajectory_generation_amd as m
    masks = m.ends_full_masks(10, 8)
    plan = m.Plan(ctx, 10, 3, 8, 4, masks)
    t, f = m.random_waypoint_batch(2000, 8, 3, 10, masks, seed=8, device="cuda", layout="soa")
    co = torch.empty((2000, 8, 3, 10), dtype=torch.float64, device="cuda")
    plan.solve(t, f, layout="soa", coeffs=co)     # warm-up outside the capture (LDS attribute, allocations)
    ctx.sync()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(ctx.stream):
        with torch.cuda.graph(g, stream=ctx.stream):
            plan.solve(t, f, layout="soa", coeffs=co, ordered=False)
    g.replay()
    ctx.sync()                                      # clean replay: no error
    t[2, 77] = -3.0                                 # new values in the captured buffers
    g.replay()
    with pytest.raises(m.MtgError) as e:
        ctx.sync()
    assert e.value.code == -2
    t[2, 77] = 1.0
    g.replay()
    ctx.sync()
    plan.close()


@pytest.mark.parametrize("bsz_host", [1, 40, 9000])      # bounce-buffer calls and staged pageable copies
def test_host_pointer_calls_have_their_own_status_word(ctx, bsz_host):
    """flags raised by an asynchronous device-pointer solve belong to the next mtg_context_sync; a
    host-pointer call in between reports (and clears) only the flags of its own launch."""
    import torch
    import mav_trajectory_generation_amd as m
    masks = m.ends_full_masks(10, 8)
    plan = m.Plan(ctx, 10, 3, 8, 4, masks)
    t, f = m.random_waypoint_batch(2000, 8, 3, 10, masks, seed=8, device="cuda", layout="soa")
    t[4, 1234] = -2.0
    plan.solve(t, f, layout="soa")                     # asynchronous, raises the bad-time flag on the device
    th, fh = m.random_waypoint_batch(bsz_host, 8, 3, 10, masks, seed=9, device="cpu")
    co, _, cost = plan.solve_host(th.numpy(), fh.numpy())         # a correct solve: must not be blamed
    assert np.isfinite(co).all() and np.isfinite(cost).all()
    with pytest.raises(m.MtgError) as e:               # ... and the failed batch is still reported
        ctx.sync()
    assert e.value.code == -2
    ctx.sync()
    # the other way round: a failing host-pointer call reports itself and leaves nothing behind
    th2 = th.clone()
    th2[0, 3] = 0.0
    with pytest.raises(m.MtgError) as e:
        plan.solve_host(th2.numpy(), fh.numpy())
    assert e.value.code == -2
    ctx.sync()
    plan.close()


def test_default_form_cross_over_is_pinned(ctx):
    """mtg_dimlane_variants.inc's HI counts HALF workgroups per CU (HI = 3: the dimension-in-lane form is the
    default up to 1.5 workgroups of 42 trajectories per CU, B ~ 16k on 256 CUs; profiles/r02_sweep_forms.txt: the slab-output
    fused kernel wins from 20k on).  mtg_plan_launch_form reports the choice without launching."""
    import torch
    import mav_trajectory_generation_amd as m
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    plan = m.Plan(ctx, 10, 3, 8, 4, m.ends_full_masks(10, 8))
    edge = (3 * cus // 2) * 42                  # largest batch with 2 * workgroups <= 3 * CUs
    assert plan.launch_form(10_000) == "dimlane" and plan.launch_form(edge) == "dimlane"
    assert plan.launch_form(edge + 1) == "slab" and plan.launch_form(125_000) == "slab"
    assert plan.launch_form(125_000, dims="dimlane") == "dimlane" and plan.launch_form(10_000, dims="split") == "split"
    # the dimension-in-lane form reads canonical SoA and canonical AoS inputs (same cross-over); coefficient-only calls of a
    # shape with a slab-output kernel never take the split form by default (partial sectors: 15.4 vs 10.4 us at 10k with
    # rotating buffers)
    assert plan.launch_form(10_000, layout="aos") == "dimlane" and plan.launch_form(edge + 1, layout="aos") == "slab"
    assert plan.launch_form(20_000) == "slab"
    plan.close()
    p5 = m.Plan(ctx, 10, 4, 16, 4, m.ends_full_masks(10, 16, 7))      # config 5: no upper limit
    assert p5.launch_form(12_500) == "dimlane" and p5.launch_form(1_000_000) == "dimlane"
    p5.close()
    p12 = m.Plan(ctx, 12, 3, 4, 5, m.ends_full_masks(12, 4))          # HI = 3 here as well
    assert p12.launch_form(edge) == "dimlane" and p12.launch_form(edge + 1) == "slab"
    p12.close()
    odd = m.Plan(ctx, 10, 3, 8, 4, [31, 3, 1, 1, 7, 1, 1, 1, 31])     # ragged masks: the generic kernel
    assert odd.launch_form(5000) == "generic"
    odd.close()


# ---- the run-time-K body (csrc/mtg_dimlane_rt.h): one kernel per polynomial order for every chain length ------------------
@pytest.fixture(scope="module")
def ctx_rt():
    """A context that takes the run-time-K dimension-in-lane body even where a static variant exists (MTG_DL_RT=1 is read at
    context creation)."""
    import mav_trajectory_generation_amd as m
    old = os.environ.get("MTG_DL_RT")
    os.environ["MTG_DL_RT"] = "1"
    try:
        c = m.Context(0)
    finally:
        if old is None:
            del os.environ["MTG_DL_RT"]
        else:
            os.environ["MTG_DL_RT"] = old
    yield c
    c.close()


@pytest.mark.parametrize("n", [8, 10, 12])
@pytest.mark.parametrize("k", [2, 3, 4, 5, 7, 8, 12, 13, 16, 21, 27, 32, 33, 40, 50, 51, 64, 100])
@pytest.mark.parametrize("bsz", [1, 21, 22, 64, 300])
def test_runtime_k_body(ctx, ctx_rt, n, k, bsz):
    """Every chain length through ONE body per N: half-chains shorter than the register tail (skipped tail positions), chains
    that fill the registers + the LDS step area exactly, chains that spill into the global workspace (K = 100: 50 steps per
    direction), odd K (the two directions differ by one step), ragged batch sizes around the tile width; against the oracle
    and -- where a static variant exists -- against that variant (same per-step arithmetic)."""
    import torch
    import mav_trajectory_generation_amd as m
    d = n // 2 - 1
    masks = m.ends_full_masks(n, k, 1)
    t, f = m.random_waypoint_batch(bsz, k, 3, n, masks, seed=100 * n + k, device="cuda", layout="soa")
    plan_rt = m.Plan(ctx_rt, n, 3, k, d, masks)
    assert plan_rt.launch_form(bsz) == "dimlane_rt"
    co = torch.full((bsz, k, 3, n), float("nan"), dtype=torch.float64, device="cuda")
    plan_rt.solve(t, f, layout="soa", coeffs=co)
    ctx_rt.sync()
    assert torch.isfinite(co).all()
    plan = m.Plan(ctx, n, 3, k, d, masks)
    ref, _, _ = plan.solve(t, f, layout="soa")
    ctx.sync()
    rel, _ = ctx.compare_coefficients(co, ref)
    assert rel < (1e-11 if n <= 10 else 3e-9), (plan.launch_form(bsz), rel)     # (factor-store body against the G form)
    nb = min(bsz, 6)
    th, fh = t.t()[:nb].contiguous().cpu().numpy(), f.permute(2, 0, 1)[:nb].contiguous().cpu().numpy()
    c_lit, _, _ = onp.solve_batch(n, d, masks, th, fh)
    assert helpers.poly_relerr(co[:nb].cpu().numpy(), c_lit) < (1e-9 if n <= 10 else 5e-7)
    plan_rt.close()
    plan.close()


@pytest.mark.parametrize("n,k", [(8, 6), (10, 9), (10, 40), (12, 24), (12, 50)])
def test_runtime_k_body_four_dimensions(ctx, n, k):
    """x, y, z, yaw with position-only interior vertices: no static variant beyond K = 8 -- the run-time-K body by default."""
    import torch
    import mav_trajectory_generation_amd as m
    d = n // 2 - 1
    masks = m.ends_full_masks(n, k, 1)
    t, f = m.random_waypoint_batch(700, k, 4, n, masks, seed=9 * n + k, device="cuda", layout="soa", yaw_dim=True)
    plan = m.Plan(ctx, n, 4, k, d, masks)
    assert plan.launch_form(700) == "dimlane_rt"
    co, _, _ = plan.solve(t, f, layout="soa")
    ref, _, _ = plan.solve(t, f, layout="soa", dims="fused")
    ctx.sync()
    rel, _ = ctx.compare_coefficients(co, ref)
    assert rel < (1e-11 if n <= 10 else 3e-9)
    th, fh = t.t()[:5].contiguous().cpu().numpy(), f.permute(2, 0, 1)[:5].contiguous().cpu().numpy()
    c_lit, _, _ = onp.solve_batch(n, d, masks, th, fh)
    assert helpers.poly_relerr(co[:5].cpu().numpy(), c_lit) < (1e-9 if n <= 10 else 5e-7)
    plan.close()


@pytest.mark.parametrize("k,dim,interior", [(5, 4, 7), (24, 4, 7), (57, 4, 7), (12, 3, 7), (40, 3, 7), (10, 3, 3), (33, 3, 3)])
def test_runtime_k_body_other_interior_masks(ctx, k, dim, interior):
    """N = 10 with velocity (and acceleration) fixed at the interior vertices as well -- BASELINE config 5's constraint pattern
    at chain lengths other than 16: the run-time-K body by default."""
    import torch
    import mav_trajectory_generation_amd as m
    masks = m.ends_full_masks(10, k, interior)
    t, f = m.random_waypoint_batch(500, k, dim, 10, masks, seed=31 * k + dim, device="cuda", layout="soa", yaw_dim=(dim == 4))
    plan = m.Plan(ctx, 10, dim, k, 4, masks)
    assert plan.launch_form(500) == "dimlane_rt"
    co, _, _ = plan.solve(t, f, layout="soa")
    ref, _, _ = plan.solve(t, f, layout="soa", dims="fused")
    ctx.sync()
    rel, _ = ctx.compare_coefficients(co, ref)
    assert rel < 1e-11
    th, fh = t.t()[:5].contiguous().cpu().numpy(), f.permute(2, 0, 1)[:5].contiguous().cpu().numpy()
    c_lit, _, _ = onp.solve_batch(10, 4, masks, th, fh)
    assert helpers.poly_relerr(co[:5].cpu().numpy(), c_lit) < 1e-9
    plan.close()


RT_MANY_TILES = [  # (N, K, D, interior mask, layout, B): B > 2 * CUs * tile width and not a multiple of it
    (10, 6, 3, 1, "soa", 12001), (10, 9, 3, 1, "aos", 12001), (10, 35, 3, 1, "soa", 25000), (10, 40, 3, 1, "soa", 12001),
    (12, 7, 3, 1, "soa", 12001), (12, 19, 3, 1, "aos", 25000), (12, 37, 3, 1, "soa", 12001),
    (8, 5, 3, 1, "soa", 25000), (8, 70, 3, 1, "aos", 12001),
    (10, 9, 4, 1, "soa", 12001), (10, 24, 4, 7, "aos", 9001), (10, 57, 4, 7, "soa", 9001),
]


@pytest.mark.parametrize("case", RT_MANY_TILES)
def test_runtime_k_body_many_tiles_per_workgroup(ctx, ctx_rt, case):
    """the persistent workgroups of the run-time-K body (grid = 2 x CUs) must walk SEVERAL tiles each -- reuse of
    the LDS slab / exchange area, the step area and the workspace column across tiles, the per-tile phase reset of pieces that
    are not a multiple of 64 bytes (N = 10: K mod 4 != 0, N = 12: odd K) for tiles with b0 > 0, and a ragged last tile after a
    full round.  Against another launch form of the same plan (static variant / fused), NaN-prefilled outputs, a sentinel row
    behind the last trajectory, and the oracle on rows of the first, a middle and the last tile."""
    import torch
    import mav_trajectory_generation_amd as m
    n, k, dim, mi, layout, bsz = case
    d = n // 2 - 1
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    tpw = 64 // dim
    assert bsz > 2 * cus * tpw and bsz % tpw != 0
    masks = m.ends_full_masks(n, k, mi)
    t, f = m.random_waypoint_batch(bsz, k, dim, n, masks, seed=7 * n + k + dim, device="cuda", layout=layout, yaw_dim=(dim == 4))
    plan_rt = m.Plan(ctx_rt, n, dim, k, d, masks)
    assert plan_rt.launch_form(bsz, layout) == "dimlane_rt"
    co = torch.full((bsz + 1, k, dim, n), float("nan"), dtype=torch.float64, device="cuda")
    co[bsz] = 7.0
    plan_rt.solve(t, f, layout=layout, coeffs=co[:bsz])
    ctx_rt.sync()
    assert torch.isfinite(co[:bsz]).all()
    assert float(co[bsz].min()) == 7.0 and float(co[bsz].max()) == 7.0      # nothing written past the end
    plan = m.Plan(ctx, n, dim, k, d, masks)
    other = "fused" if plan.launch_form(bsz, layout) == "dimlane_rt" else "auto"
    ref, _, _ = plan.solve(t, f, layout=layout, dims=other)
    ctx.sync()
    assert plan.launch_form(bsz, layout, other) != "dimlane_rt"
    rel, _ = ctx.compare_coefficients(co[:bsz].contiguous(), ref)
    if n <= 10:
        assert rel < 1e-10, (plan.launch_form(bsz, layout, other), rel)
    else:
        # N = 12: the run-time-K body eliminates each step's block partially and keeps factors (MtgCfg::kFS), the other form solves
        # for G -- different roundings of an R_PP whose condition number reaches 1e9 on a few of these 12k - 25k trajectories.
        # Measured (K = 37): median 1.4e-11, 99.9 % below 4e-9, max 1.8e-7 on a trajectory where BOTH forms are 1.1e-6 / 1.3e-6
        # from the 50-digit solution (this body the closer one).
        den = ref.abs().amax(dim=-1, keepdim=True).clamp_min(1e-300)
        per = ((co[:bsz] - ref).abs() / den).amax(dim=(1, 2, 3))
        assert float(per.median()) < 1e-9 and float(torch.quantile(per, 0.999)) < 1e-7 and float(per.max()) < 5e-6, \
            (plan.launch_form(bsz, layout, other), float(per.median()), float(per.max()))
    rows = torch.tensor([0, 1, tpw, bsz // 2, bsz // 2 + 1, 2 * cus * tpw + 3, bsz - 2, bsz - 1], device="cuda")
    th = (t[:, rows].t() if layout == "soa" else t[rows]).contiguous().cpu().numpy()
    fh = (f[:, :, rows].permute(2, 0, 1) if layout == "soa" else f[rows]).contiguous().cpu().numpy()
    c_lit, _, _ = onp.solve_batch(n, d, masks, th, fh)
    assert helpers.poly_relerr(co[rows].cpu().numpy(), c_lit) < (1e-9 if n <= 10 else 5e-7)
    plan_rt.close()
    plan.close()


def test_runtime_k_body_is_the_default_beyond_the_static_variants(ctx):
    import mav_trajectory_generation_amd as m
    for (n, k, want) in ((12, 50, "dimlane_rt"), (10, 100, "dimlane_rt"), (8, 33, "dimlane_rt"), (10, 32, "dimlane"), (10, 50, "dimlane")):
        plan = m.Plan(ctx, n, 3, k, n // 2 - 1, m.ends_full_masks(n, k, 1))
        assert plan.launch_form(2500) == want, (n, k)
        plan.close()


def test_runtime_k_body_status(ctx_rt):
    """Bad segment times are flagged per trajectory by the run-time-K body as well (head, LDS and tail steps)."""
    import torch
    import mav_trajectory_generation_amd as m
    masks = m.ends_full_masks(10, 40)
    plan = m.Plan(ctx_rt, 10, 3, 40, 4, masks)
    t, f = m.random_waypoint_batch(100, 40, 3, 10, masks, seed=3, device="cuda", layout="soa")
    bad = {5: 0, 17: 3, 44: 19, 63: 20, 99: 39}          # trajectory -> segment with a non-positive time
    for b, seg in bad.items():
        t[seg, b] = 0.0
    st = torch.full((100,), 77, dtype=torch.int32, device="cuda")
    plan.solve(t, f, layout="soa", traj_status=st)
    with pytest.raises(m.MtgError) as e:
        ctx_rt.sync()
    assert e.value.code == -2
    got = st.cpu().numpy()
    assert sorted(np.nonzero(got & 1)[0].tolist()) == sorted(bad)
    plan.close()


# ---- canonical AoS inputs (times[B][K], d_fixed[B][D][n_fixed]: the reference's natural order) through the same kernels ----
AOS_SHAPES = [(10, 8, 3, 4, 1), (10, 8, 4, 4, 1), (10, 8, 1, 4, 1), (10, 2, 3, 4, 1), (8, 7, 3, 3, 1), (12, 5, 3, 5, 1),
              (10, 16, 4, 4, 7), (10, 16, 3, 4, 1), (10, 32, 3, 4, 1), (8, 32, 3, 3, 1), (12, 16, 3, 5, 1), (12, 32, 3, 5, 1),
              (10, 27, 3, 4, 1), (12, 21, 3, 5, 1),
              (10, 40, 3, 4, 1), (12, 50, 3, 5, 1), (8, 100, 3, 3, 1), (10, 9, 4, 4, 1), (10, 24, 4, 4, 7), (10, 33, 3, 4, 3)]   # run-time-K body


@pytest.mark.parametrize("shape", AOS_SHAPES)
@pytest.mark.parametrize("bsz", [1, 22, 300])
def test_dimlane_reads_aos_inputs(ctx, shape, bsz):
    """The dimension-in-lane forms (static variants, long-chain hybrids with on-demand input loads, the run-time-K body) take
    canonical AoS inputs as well: same lanes, same arithmetic, only the load addresses differ -> bit-identical coefficients."""
    import torch
    import mav_trajectory_generation_amd as m
    n, k, dim, d, mi = shape
    masks = m.ends_full_masks(n, k, mi)
    plan = m.Plan(ctx, n, dim, k, d, masks)
    ta, fa = m.random_waypoint_batch(bsz, k, dim, n, masks, seed=31 * n + k, device="cuda", layout="aos")
    ts, fs = ta.t().contiguous(), fa.permute(1, 2, 0).contiguous()
    form = plan.launch_form(bsz, "aos", "dimlane")
    assert form in ("dimlane", "dimlane_rt") and form == plan.launch_form(bsz, "soa", "dimlane")
    co_a = torch.full((bsz + 1, k, dim, n), 7.0, dtype=torch.float64, device="cuda")
    plan.solve(ta, fa, layout="aos", coeffs=co_a[:bsz], dims="dimlane")
    co_s, _, _ = plan.solve(ts, fs, layout="soa", dims="dimlane")
    ctx.sync()
    assert float(co_a[bsz].min()) == 7.0 and float(co_a[bsz].max()) == 7.0
    assert torch.equal(co_a[:bsz], co_s)
    nb = min(bsz, 4)
    c_lit, _, _ = onp.solve_batch(n, d, masks, ta[:nb].cpu().numpy(), fa[:nb].cpu().numpy())
    assert helpers.poly_relerr(co_a[:nb].cpu().numpy(), c_lit) < (1e-9 if n <= 10 else 5e-7)
    plan.close()


def test_aos_inputs_through_queue_mixed_request_and_host_pointers(ctx):
    """AoS inputs on the other entry points that run dimension-in-lane kernels: mtg_solve_linear_sequence's queue launch,
    mtg_multi_solve's cross-structure launch, and a host-pointer call (the C++ veneer's layout)."""
    import torch
    import mav_trajectory_generation_amd as m
    # queue (config 5 shape: dimension-in-lane only)
    masks = m.ends_full_masks(10, 16, 7)
    plan = m.Plan(ctx, 10, 4, 16, 4, masks)
    sets_a, sets_s = [], []
    for s in range(3):
        ta, fa = m.random_waypoint_batch(700, 16, 4, 10, masks, seed=900 + s, device="cuda", layout="aos")
        sets_a.append((ta, fa, torch.zeros((700, 16, 4, 10), dtype=torch.float64, device="cuda")))
        sets_s.append((ta.t().contiguous(), fa.permute(1, 2, 0).contiguous(), torch.zeros((700, 16, 4, 10), dtype=torch.float64, device="cuda")))
    plan.solve_sequence(sets_a, layout="aos")
    plan.solve_sequence(sets_s, layout="soa")
    ctx.sync()
    for a, s_ in zip(sets_a, sets_s):
        assert float(a[2].abs().max()) > 0 and torch.equal(a[2], s_[2])
    # host pointers, AoS numpy (the veneer's batch call): dimension-in-lane is the default form at this size
    assert plan.launch_form(700, "aos") == "dimlane"
    co_h, _, _ = plan.solve_host(sets_a[0][0].cpu().numpy(), sets_a[0][1].cpu().numpy(), want_free=False, want_cost=False)
    ctx.sync()
    assert np.array_equal(co_h, sets_a[0][2].cpu().numpy())
    plan.close()
    # cross-structure launch: buckets in mixed layouts
    buckets_a, buckets_s = [], []
    for i, (n, k) in enumerate(((8, 4), (10, 8), (12, 16), (10, 32))):
        mk = m.ends_full_masks(n, k, 1)
        ta, fa = m.random_waypoint_batch(333, k, 3, n, mk, seed=77 + i, device="cuda", layout="aos")
        ts, fs = ta.t().contiguous(), fa.permute(1, 2, 0).contiguous()
        mixed = (ta, fa, "aos") if i % 2 == 0 else (ts, fs, "soa")
        buckets_a.append(dict(n_coeffs=n, derivative=n // 2 - 1, masks=mk, times=mixed[0], d_fixed=mixed[1], layout=mixed[2]))
        buckets_s.append(dict(n_coeffs=n, derivative=n // 2 - 1, masks=mk, times=ts, d_fixed=fs, layout="soa"))
    solver = m.MixedBatchSolver(ctx, n_streams=1)
    ra, rs = solver.merged(buckets_a), solver.merged(buckets_s)
    assert ra.launch_count == 1 and rs.launch_count == 1
    out_a, out_s = ra.solve(), rs.solve()
    solver.sync()
    ctx.sync()
    for (ca, _), (cs, _) in zip(out_a, out_s):
        assert torch.equal(ca, cs)


# ---- solves that also return the cost / d_P through the dimension-in-lane kernels (mtg_solve_dl_extra_kernel) ------------
EXTRA_SHAPES = [(10, 8, 3, 4, 1), (10, 8, 4, 4, 1), (10, 8, 1, 4, 1), (10, 4, 3, 4, 1), (10, 2, 3, 4, 1), (8, 4, 3, 3, 1),
                (8, 8, 3, 3, 1), (12, 4, 3, 5, 1), (12, 8, 3, 5, 1), (10, 16, 4, 4, 7), (8, 16, 3, 3, 1), (10, 16, 3, 4, 1),
                (12, 16, 3, 5, 1), (10, 32, 3, 4, 1), (8, 32, 3, 3, 1), (12, 32, 3, 5, 1)]


@pytest.mark.parametrize("shape", EXTRA_SHAPES)
@pytest.mark.parametrize("bsz,layout", [(1, "soa"), (22, "aos"), (300, "soa"), (1000, "aos")])
def test_dimlane_extra_outputs(ctx, shape, bsz, layout):
    """cost and d_P from the dimension-in-lane kernels: against the fused kernels' (same lane arithmetic; the cost is summed
    in another order) and the oracle; bit-reproducible from run to run (the dimension lanes are summed in a fixed order, one
    atomic per chain direction); coefficients bit-identical to the coefficient-only launch."""
    import torch
    import mav_trajectory_generation_amd as m
    n, k, dim, d, mi = shape
    masks = m.ends_full_masks(n, k, mi)
    plan = m.Plan(ctx, n, dim, k, d, masks)
    assert plan.launch_form(bsz, layout, extra_outputs=True) == "dimlane"
    t, f = m.random_waypoint_batch(bsz, k, dim, n, masks, seed=17 * n + k, device="cuda", layout=layout)
    co, fr, cost = plan.solve(t, f, layout=layout, want_free=True, want_cost=True)
    co2, fr2, cost2 = plan.solve(t, f, layout=layout, want_free=True, want_cost=True)
    co0, _, _ = plan.solve(t, f, layout=layout, dims="dimlane")
    _, _, cost_only = plan.solve(t, f, layout=layout, want_cost=True)
    cf, ff, jf = plan.solve(t, f, layout=layout, want_free=True, want_cost=True, dims="fused")
    ctx.sync()
    assert torch.equal(co, co2) and torch.equal(fr, fr2) and torch.equal(cost, cost2) and torch.equal(cost, cost_only)
    assert torch.equal(co, co0)
    tol = (1e-12 if k < 32 else 1e-11) if n <= 10 else (1e-10 if k < 16 else 3e-9)     # (N = 12 long chains: factor-store form against the fused kernel's G form)
    rel, _ = ctx.compare_coefficients(co, cf)
    assert rel < tol
    # (N = 12 long chains: factor-store back-substitution here, G in the fused kernel; the cost is second order in the
    # coefficients but N = 12's are themselves only good to ~1e-8 -- measured 1.2e-9 at N = 12 / K = 32, 1.3e-11 at N = 10 / K = 32)
    crel = float(((cost - jf).abs() / jf.abs()).max())
    assert crel <= ((1e-11 if k < 32 else 5e-11) if n <= 10 else (1e-9 if k < 16 else 3e-8)), crel
    scale = ff.abs().amax().clamp_min(1e-300)
    assert float((fr - ff).abs().amax() / scale) < tol
    nb = min(bsz, 4)
    ta = t[:nb] if layout == "aos" else t.t()[:nb]
    fa = f[:nb] if layout == "aos" else f.permute(2, 0, 1)[:nb]
    fra = fr[:nb] if layout == "aos" else fr.permute(2, 0, 1)[:nb]
    c_lit, f_lit, j_lit = onp.solve_batch(n, d, masks, ta.contiguous().cpu().numpy(), fa.contiguous().cpu().numpy())
    assert helpers.poly_relerr(co[:nb].cpu().numpy(), c_lit) < (1e-9 if n <= 10 else 5e-7)
    assert np.allclose(cost[:nb].cpu().numpy(), j_lit, rtol=1e-6)
    assert np.allclose(fra.contiguous().cpu().numpy(), f_lit, rtol=1e-6, atol=1e-6 * np.abs(f_lit).max())
    plan.close()


# ---- SoA with the row stride padded to a multiple of 16 trajectories (mtg_layout_soa_padded; round 4) ----------------------------
@pytest.mark.parametrize("shape", [(10, 16, 4, 4, 7, 12_500), (10, 8, 3, 4, 1, 10_007), (12, 8, 3, 5, 1, 333), (8, 5, 3, 3, 1, 1)])
def test_dimlane_reads_padded_soa_inputs(ctx, shape):
    """The dimension-in-lane kernels (single launch and queue) read the padded SoA layout like the canonical ones: same lanes,
    same arithmetic, only the row stride differs -> bit-identical coefficients; the launch form stays dimension-in-lane (with the
    plain stride a 12 500-trajectory batch -- BASELINE config 5 per GPU -- reads 1.75x its input bytes: every 128-byte row
    piece straddles two lines)."""
    import torch
    import mav_trajectory_generation_amd as m
    n, k, dim, d, mi, bsz = shape
    masks = m.ends_full_masks(n, k, mi)
    plan = m.Plan(ctx, n, dim, k, d, masks)
    ts, fs = m.random_waypoint_batch(bsz, k, dim, n, masks, seed=5 * n + k, device="cuda", layout="soa", yaw_dim=(dim == 4))
    bs = (bsz + 15) & ~15
    tp = torch.full((k, bs), float("nan"), dtype=torch.float64, device="cuda")
    fp = torch.full((dim, fs.shape[1], bs), float("nan"), dtype=torch.float64, device="cuda")
    tp[:, :bsz], fp[:, :, :bsz] = ts, fs                  # (the padding columns are never read: NaN would show)
    assert plan.launch_form(bsz, "soa16", "dimlane") == "dimlane" and plan.launch_form(bsz, "soa16") == plan.launch_form(bsz, "soa")
    co_p = torch.full((bsz + 1, k, dim, n), 7.0, dtype=torch.float64, device="cuda")
    plan.solve(tp, fp, layout="soa16", coeffs=co_p[:bsz], dims="dimlane", batch=bsz)
    co_s, _, _ = plan.solve(ts, fs, layout="soa", dims="dimlane")
    ctx.sync()
    assert float(co_p[bsz].min()) == 7.0 and float(co_p[bsz].max()) == 7.0
    assert torch.equal(co_p[:bsz], co_s)
    # the queue form (what the bench times)
    outs = [torch.full((bsz, k, dim, n), float("nan"), dtype=torch.float64, device="cuda") for _ in range(3)]
    plan.solve_sequence([(tp, fp, o) for o in outs], layout="soa16", dims="dimlane")
    ctx.sync()
    for o in outs:
        assert torch.equal(o, co_s)
    plan.close()
