"""The row-cooperative form (csrc/mtg_coop.h / mtg_coop.hip): 16 lanes per trajectory-half, DPP row broadcasts inside the FMAs.
CPU: the SAME header run on a 16-lane lock-step host emulation (tests/coop_emu.cpp) against the oracles and against the host
build of the lane-per-half code (the product's other forms).  GPU: the kernel through the C ABI against the other launch forms,
the oracle, ragged batch sizes, both input layouts, status flags."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import helpers
from oracle import cpu_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mav_trajectory_generation_amd", "csrc")


@pytest.fixture(scope="module")
def emu():
    so, src = os.path.join(ROOT, "tests", "libmtg_coop_emu.so"), os.path.join(ROOT, "tests", "coop_emu.cpp")
    deps = [src] + [os.path.join(CSRC, f) for f in ("mtg_coop.h", "mtg_lane.h", "mtg_tables.inc")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-o", so, src])
    lib = ctypes.CDLL(so)
    dp = ctypes.POINTER(ctypes.c_double)
    lib.coop_emu_solve.argtypes = [ctypes.c_int] * 4 + [ctypes.c_longlong, dp, dp, dp]
    return lib


def emu_solve(lib, n, k, deriv, times, d_fixed):
    dp = ctypes.POINTER(ctypes.c_double)
    t, f = np.ascontiguousarray(times), np.ascontiguousarray(d_fixed)
    co = np.full((t.shape[0], k, 3, n), np.nan)
    rc = lib.coop_emu_solve(n, k, 3, deriv, t.shape[0], t.ctypes.data_as(dp), f.ctypes.data_as(dp), co.ctypes.data_as(dp))
    return rc, co


@pytest.mark.parametrize("n", [8, 10, 12])
@pytest.mark.parametrize("k", [2, 3, 4, 5, 8, 9, 16, 17, 33, 64])
def test_emulated_rows_vs_oracle_and_lane_code(emu, n, k):
    """Every parity combination of the two half-chains (K even / odd, half-chain lengths even / odd), the first-step special case
    (K = 2: nothing but end steps), long chains: against the C++ restatement of the reference (float64 evaluation error of the
    reference's formulas: 1e-10 for N = 10, 1e-8 for N = 12) and, tightly, against the host build of the product's lane code."""
    d = n // 2 - 1
    masks, times, d_fixed = helpers.reference_batch(5, k, n, 3, 1000 + 10 * n + k)
    rc, co = emu_solve(emu, n, k, d, times, d_fixed)
    assert rc == 0 and np.isfinite(co).all()
    ref = cpu_ref.solve_batch(n, d, masks, times, d_fixed)[0]
    assert helpers.poly_relerr(co, ref) < {8: 1e-10, 10: 2e-9, 12: 5e-6}[n]   # (N = 12: the float64 restatement is the side that is off)
    lane = ctypes.CDLL(os.path.join(ROOT, "tests", "libmtg_host_emu.so"))
    _, co_lane, _, _, st = helpers.emu_run(lane, n, 3, k, d, masks, times, d_fixed, want_cost=False)
    assert st == 0 and helpers.poly_relerr(co, co_lane) < (1e-11 if n <= 10 else 1e-8)   # (another elimination order: round-off x cond; N = 12 / K = 33: each side is 1.4e-9 ... 1.8e-9 from the 50-digit solution on the trajectory with the segment-time ratio 16.5)


def test_emulated_rows_other_derivatives_and_flags(emu):
    """derivative_to_optimize below h - 1 (run-time exponent path), a non-positive segment time (flag 1)."""
    masks, times, d_fixed = helpers.reference_batch(4, 6, 10, 3, 77)
    for d in (2, 3):
        rc, co = emu_solve(emu, 10, 6, d, times, d_fixed)
        ref = cpu_ref.solve_batch(10, d, masks, times, d_fixed)[0]
        assert rc == 0 and helpers.poly_relerr(co, ref) < 1e-6
    bad = times.copy()
    bad[2, 4] = -1.0
    rc, _ = emu_solve(emu, 10, 6, 4, bad, d_fixed)
    assert rc & 1


def test_no_dpp_hazard_in_the_compiled_kernel(tmp_path):
    """The kernel's v_fmac_f64_dpp are inline asm, which the compiler's hazard recogniser does not see: the gfx9 rule "VALU writes
    a VGPR -> DPP reads it: 2 wait states" is checked on the emitted code (tools/check_dpp_hazards.py)."""
    out = str(tmp_path / "coop.s")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-Wno-unused-result",
                           "--cuda-device-only", "-S", os.path.join(CSRC, "mtg_coop.hip"), "-o", out])
    r = subprocess.run(["python", os.path.join(ROOT, "tools", "check_dpp_hazards.py"), out, "coop"], capture_output=True, text=True)
    assert r.returncode == 0 and "DPP hazard check: ok" in r.stdout, r.stdout[-2000:]
    assert r.stdout.count("DPP") >= 3          # three instantiations with DPP instructions were inspected


@pytest.fixture(scope="module")
def ctx():
    import mav_trajectory_generation_amd as m
    c = m.Context(0)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [8, 10, 12])
@pytest.mark.parametrize("k", [2, 3, 5, 8, 16, 17, 32, 50])
@pytest.mark.parametrize("bsz,layout", [(1, "soa"), (3, "aos"), (4, "soa"), (301, "aos"), (2500, "soa")])
def test_gpu_cooperative_form_vs_other_forms(ctx, n, k, bsz, layout):
    import torch
    import mav_trajectory_generation_amd as m
    d = n // 2 - 1
    masks = m.ends_full_masks(n, k, 1)
    plan = m.Plan(ctx, n, 3, k, d, masks)
    assert plan.launch_form(bsz, layout, "coop") == "coop"
    t, f = m.random_waypoint_batch(bsz, k, 3, n, masks, seed=17 * n + k, device="cuda", layout=layout)
    co = torch.full((bsz + 1, k, 3, n), float("nan"), dtype=torch.float64, device="cuda")
    co[bsz] = 7.0
    plan.solve(t, f, layout=layout, coeffs=co[:bsz], dims="coop")
    ref, _, _ = plan.solve(t, f, layout=layout)
    ctx.sync()
    assert plan.launch_form(bsz, layout) != "coop"
    assert torch.isfinite(co[:bsz]).all() and float(co[bsz].min()) == 7.0 and float(co[bsz].max()) == 7.0
    # two float64 eliminations in different orders differ by round-off x conditioning: 1e-12 typically, 1e-9 (N = 10) ... 1e-6
    # (N = 12) on the rare ill-conditioned trajectory of a large batch (a very short segment between long ones) -- where the
    # 50-digit solution says which side is off (on the CPU emulation of both codes: the cooperative form is the closer one)
    num = (co[:bsz] - ref).abs().amax(dim=-1)
    den = ref.abs().amax(dim=-1).clamp_min(1e-300)
    per_traj = (num / den).reshape(bsz, -1).amax(dim=1)
    rel = float(per_traj.max())
    assert rel < (5e-9 if n <= 10 else 5e-6), rel
    assert float(per_traj.median()) < (3e-12 if n <= 10 else 1e-10)     # (a one-trajectory batch: the median IS that trajectory; 1.1e-12 seen for N = 10 / K = 17)
    if rel > (1e-11 if n <= 10 else 1e-9):
        from oracle import oracle_mp
        w = int(per_traj.argmax())
        tw = (t[:, w:w + 1].t() if layout == "soa" else t[w:w + 1]).contiguous().cpu().numpy()
        fw = (f[:, :, w:w + 1].permute(2, 0, 1) if layout == "soa" else f[w:w + 1]).contiguous().cpu().numpy()
        truth = np.asarray(oracle_mp.solve_batch(n, d, masks, tw, fw)[0], dtype=np.float64)
        e_coop = helpers.poly_relerr(co[w:w + 1].cpu().numpy(), truth)
        e_other = helpers.poly_relerr(ref[w:w + 1].cpu().numpy(), truth)
        assert e_coop <= max(2.0 * e_other, 1e-11 if n <= 10 else 1e-9), (w, e_coop, e_other)
    # ... and the WHOLE batch against the reference's own code (oracle/_ref, LIN:339-379): 1e-9 for N <= 10, the arbitration rule
    # of tests/test_gpu_vs_reference.py for N = 12 (round 4 compared four trajectories with the numpy restatement)
    th = (t.t() if layout == "soa" else t).contiguous().cpu().numpy()
    fh = (f.permute(2, 0, 1) if layout == "soa" else f).contiguous().cpu().numpy()
    from oracle import ref_linear
    if ref_linear.available():
        from test_gpu_vs_reference import assert_close_to_reference
        ref_c = ref_linear.solve_batch(n, d, masks, th, fh, nthreads=ref_linear.hardware_threads())[0]
        assert_close_to_reference(n, d, masks, th, fh, co[:bsz].cpu().numpy(), ref_c)
    else:
        from oracle import oracle_np as onp
        nb = min(bsz, 4)
        c_lit, _, _ = onp.solve_batch(n, d, masks, th[:nb], fh[:nb])
        assert helpers.poly_relerr(co[:nb].cpu().numpy(), c_lit) < (1e-9 if n <= 10 else 5e-7)
    plan.close()


@pytest.mark.gpu
def test_gpu_cooperative_form_default_range(ctx):
    """Where the form is the DEFAULT (measured: profiles/r04d_coop_vs_default.jsonl): long chains in launches of at most one
    workgroup (four trajectories) per CU; results as accurate as any other form's."""
    import torch
    import mav_trajectory_generation_amd as m
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    c = m.Context(0)
    c.set_option("coop", -1)          # (tests/conftest.py takes the form out of the default choice for the other suites)
    try:
        for (n, k, bsz, want) in ((12, 16, 4 * cus, True), (12, 16, 4 * cus + 1, False), (12, 32, 8 * cus, True), (12, 15, 64, False),
                                  (10, 100, 4 * cus, True), (10, 50, 64, False), (10, 64, 100, True), (8, 100, 200, True),
                                  (8, 64, 200, False), (10, 100, 100_000, False)):
            plan = m.Plan(c, n, 3, k, n // 2 - 1, m.ends_full_masks(n, k, 1))
            assert (plan.launch_form(bsz) == "coop") == want, (n, k, bsz, plan.launch_form(bsz))
            plan.close()
        masks = m.ends_full_masks(12, 24, 1)
        plan = m.Plan(c, 12, 3, 24, 5, masks)
        t, f = m.random_waypoint_batch(500, 24, 3, 12, masks, seed=9, device="cuda", layout="aos")
        co, _, _ = plan.solve(t, f)                       # default choice: the cooperative form
        assert plan.launch_form(500, "aos") == "coop"
        c.sync()
        from oracle import ref_linear
        from test_gpu_vs_reference import assert_close_to_reference
        th, fh = t.cpu().numpy(), f.cpu().numpy()
        if ref_linear.available():
            ref_c = ref_linear.solve_batch(12, 5, masks, th, fh, nthreads=ref_linear.hardware_threads())[0]
            assert_close_to_reference(12, 5, masks, th, fh, co.cpu().numpy(), ref_c)
        else:
            from oracle import oracle_np as onp
            c_lit, _, _ = onp.solve_batch(12, 5, masks, th[:6], fh[:6])
            assert helpers.poly_relerr(co[:6].cpu().numpy(), c_lit) < 5e-7
        plan.close()
    finally:
        c.close()


@pytest.mark.gpu
def test_gpu_cooperative_form_status_and_eligibility(ctx):
    import torch
    import mav_trajectory_generation_amd as m
    masks = m.ends_full_masks(10, 20)
    plan = m.Plan(ctx, 10, 3, 20, 4, masks)
    t, f = m.random_waypoint_batch(50, 20, 3, 10, masks, seed=3, device="cuda", layout="soa")
    bad = {4: 0, 17: 9, 33: 10, 49: 19}
    for b, seg in bad.items():
        t[seg, b] = 0.0
    st = torch.full((50,), 77, dtype=torch.int32, device="cuda")
    plan.solve(t, f, layout="soa", traj_status=st, dims="coop")
    with pytest.raises(m.MtgError) as e:
        ctx.sync()
    assert e.value.code == -2
    assert sorted(np.nonzero(st.cpu().numpy() & 1)[0].tolist()) == sorted(bad)
    plan.close()
    # not eligible: other constraint patterns, four dimensions, extra outputs -> the flag falls back to the ordinary choice
    p5 = m.Plan(ctx, 10, 4, 16, 4, m.ends_full_masks(10, 16, 7))
    assert p5.launch_form(100, "soa", "coop") != "coop"
    p5.close()
    p = m.Plan(ctx, 10, 3, 16, 4, m.ends_full_masks(10, 16, 1))
    assert p.launch_form(100, "soa", "coop", extra_outputs=True) != "coop"
    p.close()
