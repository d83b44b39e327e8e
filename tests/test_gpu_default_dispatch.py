"""The reference-anchored suites under the PRODUCTION default dispatch.

tests/conftest.py takes the row-cooperative form (launch form 7, csrc/mtg_coop.hip) out of the default choice for the suites
whose invariants are bit-identity between the lane-per-half forms.  A caller of libmtg_hip.so does not do that: with the
context's option "coop" at its default (-1) a coefficient-only solve of a long chain in a small launch (N = 12 / K >= 16,
N = 10 / K >= 64, N = 8 / K >= 80, at most one or two 4-trajectory workgroups per CU) IS the cooperative kernel.  This module
runs on a context with the library's own defaults and

  * asserts per case which form `mtg_plan_launch_form` reports (the cooperative one wherever production would pick it),
  * compares the WHOLE batch with the reference's own code (oracle/_ref/libmtg_ref.so, impl/polynomial_optimization_linear_impl.h:
    339-379) -- 1e-9 norm-wise per polynomial for N <= 10, the arbitration rule of test_gpu_vs_reference.py for N = 12,
  * runs the committed golden fixtures, queue / merged launches, status flags and host-pointer calls through the same defaults.
Inside the cooperative form's default range single launches and queue / merged launches of the same plan are NOT bit-identical
(another elimination order): there the comparison is a tolerance (include/mtg_hip.h says so).
"""
import os

import numpy as np
import pytest

import helpers
from oracle import ref_linear
from test_gpu_vs_reference import assert_close_to_reference, tol_for

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "solve_linear_golden.npz"))
REF = np.load(os.path.join(HERE, "golden", "reference_solve_linear.npz"))
NAMES = sorted({k.split("/")[0] for k in GOLD.files})
live = pytest.mark.skipif(not ref_linear.available(), reason="oracle/_ref/libmtg_ref.so not shipped")


@pytest.fixture(scope="module")
def ctx():
    """A context with the library's OWN defaults, whatever the environment of the test process says."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    import mav_trajectory_generation_amd as m
    c = m.Context(0)
    c.set_option("coop", -1)
    yield c
    c.close()


def to_dev(times, d_fixed, layout):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(times)).cuda()
    f = torch.from_numpy(np.ascontiguousarray(d_fixed)).cuda()
    if layout == "soa":
        t, f = t.t().contiguous(), f.permute(1, 2, 0).contiguous()
    return t, f


def cus():
    import torch
    return torch.cuda.get_device_properties(0).multi_processor_count


# (n, k, batch, layout, form production picks).  Batches are sized relative to a 256-CU device; `want` is re-derived from the
# device's CU count below so the test also holds on a cut-down part.
CASES = [
    (12, 16, 1000, "soa"), (12, 16, 1024, "aos"), (12, 17, 333, "soa"), (12, 24, 500, "aos"), (12, 32, 600, "soa"),
    (12, 32, 2048, "soa"), (12, 50, 256, "aos"),
    (10, 64, 300, "soa"), (10, 100, 1000, "aos"), (8, 80, 200, "soa"), (8, 100, 1024, "soa"),
    # the same shapes just OUTSIDE the default range (batch or chain length): the lane-per-half forms
    (12, 16, 2000, "soa"), (12, 15, 600, "soa"), (10, 32, 600, "soa"), (10, 63, 300, "soa"), (8, 64, 200, "soa"),
]


def expect_coop(n, k, bsz):
    kmin = {12: 16, 10: 64, 8: 80}[n]
    per_cu = 2 if (n == 12 and k >= 32) else 1
    return k >= kmin and (bsz + 3) // 4 <= per_cu * cus()


@live
@pytest.mark.parametrize("n,k,bsz,layout", CASES)
def test_default_dispatch_vs_live_reference(ctx, n, k, bsz, layout):
    """Coefficient-only device-pointer solves exactly as a caller issues them (no form flag), whole batch against the reference."""
    import torch
    import mav_trajectory_generation_amd as m
    d = n // 2 - 1
    masks = m.ends_full_masks(n, k, 1)
    masks, times, d_fixed = helpers.reference_batch(bsz, k, n, 3, 8128 + 31 * k + n, masks)
    ref_c, _, _, _ = ref_linear.solve_batch(n, d, masks, times, d_fixed, nthreads=ref_linear.hardware_threads())
    plan = m.Plan(ctx, n, 3, k, d, masks)
    want = "coop" if expect_coop(n, k, bsz) else None
    form = plan.launch_form(bsz, layout)
    assert (form == "coop") == (want == "coop"), (form, want)
    t, f = to_dev(times, d_fixed, layout)
    st = torch.zeros((bsz,), dtype=torch.int32, device="cuda")
    co = torch.full((bsz + 1, k, 3, n), float("nan"), dtype=torch.float64, device="cuda")
    co[bsz] = 7.0
    plan.solve(t, f, layout=layout, coeffs=co[:bsz], traj_status=st)
    ctx.sync()
    assert int(st.abs().max()) == 0
    assert float(co[bsz].min()) == 7.0 and float(co[bsz].max()) == 7.0        # nothing written behind the batch
    assert_close_to_reference(n, d, masks, times, d_fixed, co[:bsz].cpu().numpy(), ref_c)
    plan.close()


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("extra", [False, True])
def test_golden_fixtures_default_dispatch(ctx, name, extra):
    """The committed reference outputs (tests/golden/reference_solve_linear.npz) through a default-option context: coefficient-only
    calls (the dimension-in-lane / cooperative / slab candidates) and calls that also ask for d_P and the cost."""
    import mav_trajectory_generation_amd as m
    n, d = int(GOLD[f"{name}/n"]), int(GOLD[f"{name}/d"])
    masks = [int(x) for x in GOLD[f"{name}/masks"]]
    times, d_fixed = GOLD[f"{name}/times"], GOLD[f"{name}/d_fixed"]
    dim, k = d_fixed.shape[1], times.shape[1]
    plan = m.Plan(ctx, n, dim, k, d, masks)
    t, f = to_dev(times, d_fixed, "aos")
    co, fr, cost = plan.solve(t, f, want_free=extra, want_cost=extra)
    ctx.sync()
    assert helpers.poly_relerr(co.cpu().numpy(), REF[f"{name}/coeffs_ref"]) < tol_for(n, d)
    if extra:
        assert np.allclose(cost.cpu().numpy(), REF[f"{name}/cost_ref"], rtol=max(1e-8, tol_for(n, d)))
    plan.close()


@live
@pytest.mark.parametrize("n,k,bsz", [(12, 16, 800), (12, 32, 512), (10, 64, 256)])
def test_queue_and_merged_launches_in_the_cooperative_range(ctx, n, k, bsz):
    """A single launch of these batches is the cooperative kernel, a queue (mtg_solve_linear_sequence) or a merged request
    (mtg_multi_*) of the same plan runs the lane-per-half bodies: every route against the reference, and against each other to
    round-off x conditioning (NOT bit for bit -- the documented exception to the form-equivalence invariants)."""
    import torch
    import mav_trajectory_generation_amd as m
    d = n // 2 - 1
    masks = m.ends_full_masks(n, k, 1)
    plan = m.Plan(ctx, n, 3, k, d, masks)
    assert plan.launch_form(bsz, "soa") == "coop"
    sets, refs, inputs = [], [], []
    for s in range(3):
        _, times, d_fixed = helpers.reference_batch(bsz, k, n, 3, 4004 + s + k, masks)
        refs.append(ref_linear.solve_batch(n, d, masks, times, d_fixed, nthreads=ref_linear.hardware_threads())[0])
        inputs.append((times, d_fixed))
        t, f = to_dev(times, d_fixed, "soa")
        sets.append((t, f, torch.zeros((bsz, k, 3, n), dtype=torch.float64, device="cuda")))
    single = [plan.solve(t, f, layout="soa")[0] for (t, f, _) in sets]
    plan.solve_sequence(sets, layout="soa")
    ctx.sync()
    req = m.MultiSolve(ctx, [dict(plan=plan, times=t, d_fixed=f, layout="soa") for (t, f, _) in sets])
    merged = [o[0] for o in req.solve()]
    ctx.sync()
    for (times, d_fixed), ref_c, one, (_, _, queued), mg in zip(inputs, refs, single, sets, merged):
        for co in (one, queued, mg):
            assert_close_to_reference(n, d, masks, times, d_fixed, co.cpu().numpy(), ref_c)
        for other in (queued, mg):
            per = np.array([helpers.poly_relerr(one[b:b + 1].cpu().numpy(), other[b:b + 1].cpu().numpy()) for b in range(bsz)])
            assert np.median(per) < (1e-12 if n <= 10 else 1e-10) and per.max() < (5e-9 if n <= 10 else 5e-6)
    req.close()
    plan.close()


def test_status_and_host_pointer_calls_in_the_cooperative_range(ctx):
    """Per-trajectory status of the default (cooperative) launch, the context-wide word, and a host-pointer call of the same
    plan and size -- the paths tests/test_gpu_parity.py covers for the lane-per-half forms."""
    import torch
    import mav_trajectory_generation_amd as m
    n, k, bsz = 12, 20, 96
    masks = m.ends_full_masks(n, k, 1)
    masks, times, d_fixed = helpers.reference_batch(bsz, k, n, 3, 515, masks)
    plan = m.Plan(ctx, n, 3, k, 5, masks)
    assert plan.launch_form(bsz, "aos") == "coop"
    bad = {3: 0, 40: 7, 41: 19, 95: 10}
    tb = times.copy()
    for b, seg in bad.items():
        tb[b, seg] = -0.5
    t, f = to_dev(tb, d_fixed, "aos")
    st = torch.full((bsz,), 77, dtype=torch.int32, device="cuda")
    co, _, _ = plan.solve(t, f, traj_status=st)
    with pytest.raises(m.MtgError) as e:
        ctx.sync()
    assert e.value.code == -2
    ctx.sync()                                   # reported once, then cleared
    assert sorted(np.nonzero(st.cpu().numpy() & 1)[0].tolist()) == sorted(bad)
    good = [b for b in range(bsz) if b not in bad]
    co_h, _, _ = plan.solve_host(times, d_fixed, want_free=False, want_cost=False)
    ctx.sync()
    assert helpers.poly_relerr(co.cpu().numpy()[good], co_h[good]) < 5e-6
    if ref_linear.available():
        ref_c = ref_linear.solve_batch(n, 5, masks, times, d_fixed)[0]
        assert_close_to_reference(n, 5, masks, times, d_fixed, co_h, ref_c)
    plan.close()
