"""MTG_FLAG_REFINE: one step of iterative refinement on the free derivatives with the residual in double-double
(csrc/mtg_refine.hip), for problems whose float64 solution is conditioning-limited (N = 12 with uneven segment times: the float64
sweep, the numpy restatement and the compiled reference alike sit 1e-8 ... 2e-7 from the 50-digit solution).

CPU layer: the lane code's hook for an explicit right-hand side (MtgParams::rhs, generic mode -- the correction solve), through
the host emulation, with the residual from the 50-digit oracle.  GPU layer: the flag itself, every piece on the device."""
import ctypes

import mpmath as mp
import numpy as np
import pytest

import helpers
from oracle import oracle_mp
from test_pivot_threshold import build_emu

dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)


def worst_ratio_problems(n, k, dim, seed, pool, count):
    masks = helpers.masks_ends_full(n, k, 1)
    masks, times, dfx = helpers.reference_batch(pool, k, n, dim, seed, masks)
    worst = np.argsort(times.max(axis=1) / times.min(axis=1))[-count:]
    return masks, np.ascontiguousarray(times[worst]), np.ascontiguousarray(dfx[worst])


def exact_residual(n, d, masks, times, dfx, x):
    """r = -(R_PP x + R_PF d_F) of one trajectory at 50 digits, rounded once (R = sum over segments of T^(1-2d) S H(1) S)."""
    mp.mp.dps = 50
    h, k = n // 2, len(times)
    a1, q1 = oracle_mp.mapping_matrix(n, 1.0), oracle_mp.cost_matrix(n, d, 1.0)
    ai = a1 ** -1
    h1 = ai.T * q1 * ai
    fcol = {key: i for i, key in enumerate((v, p) for v in range(k + 1) for p in range(h) if (masks[v] >> p) & 1)}
    pcol = {key: i for i, key in enumerate((v, p) for v in range(k + 1) for p in range(h) if not (masks[v] >> p) & 1)}
    out = np.zeros((dfx.shape[0], len(pcol)))
    for dm in range(dfx.shape[0]):
        acc = [mp.mpf(0)] * len(pcol)
        for i in range(k):
            t = mp.mpf(float(times[i]))
            dv = [mp.mpf(float(dfx[dm][fcol[(vv, p)]])) if (vv, p) in fcol else mp.mpf(float(x[dm][pcol[(vv, p)]]))
                  for vv in (i, i + 1) for p in range(h)]
            for a in range(n):
                key = (i if a < h else i + 1, a % h)
                if key in pcol:
                    acc[pcol[key]] += t ** (1 - 2 * d) * t ** (a % h) * sum(h1[a, c] * t ** (c % h) * dv[c] for c in range(n))
        out[dm] = [-float(v) for v in acc]
    return out


@pytest.mark.parametrize("n,d,k,pool", [(12, 5, 16, 400), (12, 5, 5, 60), (10, 4, 8, 60)])
def test_explicit_rhs_correction_solve_through_the_lane_code(n, d, k, pool):
    """x1 = x0 + (R_PP)^-1 r with r exact: the worst-conditioned trajectory of the N = 12 / K = 16 pool goes from 2e-7 to 1e-15 of the
    50-digit solution; well-conditioned ones are not made worse."""
    lib = build_emu(False)
    lib.mtg_emu_run_rhs.argtypes = [ctypes.c_int] * 4 + [ip, ctypes.c_longlong, dp, dp, dp, dp, dp, ip]
    lib.mtg_emu_run_rhs.restype = ctypes.c_int
    dim = 3
    masks, t, f = worst_ratio_problems(n, k, dim, 31415 + 7 * k + n, pool, 1)
    rc, _, fr, _, st = helpers.emu_run(lib, n, dim, k, d, masks, t, f, want_cost=False)
    assert rc == 0 and st == 0
    truth = oracle_mp.solve(n, d, masks, t[0], f[0])[1]
    e0 = np.abs(fr[0] - truth).max() / np.abs(truth).max()
    r = np.ascontiguousarray(exact_residual(n, d, masks, t[0], f[0], fr[0])[None])
    zf, delta, cc, stt = np.zeros_like(f), np.zeros_like(fr), np.zeros((1, k, dim, n)), ctypes.c_int(0)
    m = np.array(masks, dtype=np.int32)
    rc = lib.mtg_emu_run_rhs(n, dim, k, d, m.ctypes.data_as(ip), 1, t.ctypes.data_as(dp), zf.ctypes.data_as(dp), cc.ctypes.data_as(dp),
                             delta.ctypes.data_as(dp), r.ctypes.data_as(dp), ctypes.byref(stt))
    assert rc == 0 and stt.value == 0
    e1 = np.abs(fr[0] + delta[0] - truth).max() / np.abs(truth).max()
    assert e1 <= 1e-13 and e1 <= max(e0, 1e-15), (e0, e1)
    if (n, k) == (12, 16):
        assert e0 > 1e-8          # the case the flag exists for (ratio of segment times 16.8)


def build_refine_emu():
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so, src = os.path.join(root, "tests", "librefine_emu.so"), os.path.join(root, "tests", "refine_emu.cpp")
    deps = [src] + [os.path.join(root, "mav_trajectory_generation_amd", "csrc", f) for f in ("mtg_refine_dd.h", "mtg_tables.inc", "mtg_tables_dd.inc")]
    if not os.path.exists(so) or any(os.path.getmtime(x) > os.path.getmtime(so) for x in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", "-o", so, src])
    lib = ctypes.CDLL(so)
    lib.mtg_refine_emu_residual.argtypes = [ctypes.c_int] * 4 + [ip, ctypes.c_longlong, dp, dp, dp, dp]
    lib.mtg_refine_emu_residual.restype = ctypes.c_int
    return lib


@pytest.mark.parametrize("n,d,k,pool", [(12, 5, 16, 400), (12, 2, 8, 60), (10, 4, 8, 60), (8, 3, 5, 40), (6, 0, 3, 20)])
def test_double_double_residual_host_build_against_the_50_digit_residual(n, d, k, pool):
    """csrc/mtg_refine_dd.h (the code the device kernel runs) built for the host: the residual of the lane code's own d_P within
    2e-15 of the residual formed at 50 digits, relative to its largest entry -- which is itself ~1e-8 ... 1e-15 of the terms it is the
    difference of."""
    lib, emu = build_refine_emu(), build_emu(False)
    dim = 3
    masks, t, f = worst_ratio_problems(n, k, dim, 31415 + 7 * k + n, pool, 2)
    rc, _, fr, _, st = helpers.emu_run(emu, n, dim, k, d, masks, t, f, want_cost=False)
    assert rc == 0 and st == 0
    fr = np.ascontiguousarray(fr)
    got = np.zeros_like(fr)
    m = np.array(masks, dtype=np.int32)
    assert lib.mtg_refine_emu_residual(n, dim, k, d, m.ctypes.data_as(ip), 2, t.ctypes.data_as(dp), f.ctypes.data_as(dp), fr.ctypes.data_as(dp),
                                       got.ctypes.data_as(dp)) == 0
    for b in range(2):
        want = exact_residual(n, d, masks, t[b], f[b], fr[b])
        assert np.abs(got[b] - want).max() <= 2e-15 * np.abs(want).max()


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    import mav_trajectory_generation_amd as m
    c = m.Context(0)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n,d,k,pool,layout", [(12, 5, 16, 2000, "soa"), (12, 5, 32, 1000, "aos"), (12, 5, 8, 2000, "soa"), (12, 5, 4, 2000, "aos"),
                                               (12, 3, 8, 100, "aos"), (10, 2, 8, 100, "soa"), (10, 4, 8, 60, "aos"), (8, 3, 50, 40, "soa")])
def test_refined_solve_against_the_50_digit_solution(ctx, n, d, k, pool, layout):
    """The flag on the device (double-double residual kernel, generic correction solve, update path): the worst-conditioned
    trajectories of a pool (N = 12: config 4's buckets K = 4 / 8 / 16 / 32, the three largest segment-time ratios of 1000-2000
    random-waypoint trajectories) against the 50-digit solve -- d_P and coefficients within 1e-11 (measured ~1e-14), where the plain solve
    is up to 2e-7 off; results with and without a caller-side d_free buffer agree bit for bit; cost as the plain solve's."""
    import torch
    import mav_trajectory_generation_amd as m
    dim, count = 3, 3
    masks, t_h, f_h = worst_ratio_problems(n, k, dim, 31415 + 7 * k + n, pool, count)
    # the sampled trajectories inside a batch of ordinary ones (the flag applies to the whole batch)
    _, t_all, f_all = helpers.reference_batch(130, k, n, dim, 99, masks)
    t_all[:count], f_all[:count] = t_h, f_h
    plan = m.Plan(ctx, n, dim, k, d, masks)
    t, f = torch.from_numpy(t_all).cuda(), torch.from_numpy(f_all).cuda()
    if layout == "soa":
        t, f = t.t().contiguous(), f.permute(1, 2, 0).contiguous()
    co0, fr0, j0 = plan.solve(t, f, layout=layout, want_free=True, want_cost=True)
    st = torch.full((130,), 7, dtype=torch.int32, device="cuda")
    co1, fr1, j1 = plan.solve(t, f, layout=layout, want_free=True, want_cost=True, refine=True, traj_status=st)
    co2, _, _ = plan.solve(t, f, layout=layout, refine=True)
    ctx.sync()
    assert torch.equal(co1, co2) and int(st.abs().max()) == 0
    fr0n, fr1n = (x.cpu().numpy() if layout == "aos" else x.permute(2, 0, 1).cpu().numpy() for x in (fr0, fr1))
    co0n, co1n = co0.cpu().numpy(), co1.cpu().numpy()
    e_plain, e_ref = [], []
    for b in range(count):
        c_mp, f_mp, j_mp = oracle_mp.solve(n, d, masks, t_h[b], f_h[b])
        e_plain.append((np.abs(fr0n[b] - f_mp).max() / np.abs(f_mp).max(), helpers.poly_relerr(co0n[b], c_mp)))
        e_ref.append((np.abs(fr1n[b] - f_mp).max() / np.abs(f_mp).max(), helpers.poly_relerr(co1n[b], c_mp)))
        # (the cost is 0.5 c^T Q c evaluated in float64 FROM the coefficients: its own cancellation, 3e-11 ... 6e-9 measured)
        assert abs(float(j1[b]) - j_mp) <= (1e-7 if n == 12 or d < n // 2 - 1 else 1e-10) * abs(j_mp)
    # (d_P error, coefficient error) per sampled trajectory
    assert max(max(e) for e in e_ref) <= 1e-11, (e_plain, e_ref)
    assert max(max(e) for e in e_ref) <= max(max(max(e) for e in e_plain), 1e-13), (e_plain, e_ref)
    if (n, d, k) == (12, 5, 16) and pool >= 400:
        assert max(max(e) for e in e_plain) > 1e-8
    # the rest of the batch: the refined coefficients stay within the plain solve's own error of it
    assert helpers.poly_relerr(co1n[count:], co0n[count:]) < (1e-6 if n == 12 or d < n // 2 - 1 else 1e-9)
    assert helpers.check_path(masks, t_all, f_all, co1n) < 1e-6
    plan.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n,d,k,pool,layout", [(12, 5, 16, 400, "soa"), (12, 5, 8, 200, "aos"), (10, 4, 8, 60, "soa"), (8, 2, 5, 40, "aos")])
def test_device_residual_against_the_50_digit_residual(ctx, n, d, k, pool, layout):
    """The double-double residual kernel alone (include/mtg_hip_lab.h: mtg_lab_refine_residual) on the plain solve's d_P: within
    1e-13 of the residual formed at 50 digits, relative to its largest entry (a float64 evaluation of the same expression is wrong in
    the FIRST digit there: the residual is ~1e-8 of the terms it is the difference of)."""
    import torch
    import mav_trajectory_generation_amd as m
    dim, count = 3, 2
    masks, t_h, f_h = worst_ratio_problems(n, k, dim, 31415 + 7 * k + n, pool, count)
    plan = m.Plan(ctx, n, dim, k, d, masks)
    t, f = torch.from_numpy(t_h).cuda(), torch.from_numpy(f_h).cuda()
    if layout == "soa":
        t, f = t.t().contiguous(), f.permute(1, 2, 0).contiguous()
    _, fr, _ = plan.solve(t, f, layout=layout, want_free=True)
    rhs = torch.zeros((count, dim, plan.n_free), dtype=torch.float64, device="cuda")
    lay = plan.layout(count, layout)
    cur = ctx._enter()
    rc = plan.lib.mtg_lab_refine_residual(plan.handle, count, ctypes.byref(lay), t.data_ptr(), f.data_ptr(), fr.data_ptr(), rhs.data_ptr())
    ctx._leave(cur)
    assert rc == 0
    ctx.sync()
    frn = fr.cpu().numpy() if layout == "aos" else fr.permute(2, 0, 1).cpu().numpy()
    got = rhs.cpu().numpy()
    for b in range(count):
        want = exact_residual(n, d, masks, t_h[b], f_h[b], frn[b])
        assert np.abs(got[b] - want).max() <= 1e-13 * np.abs(want).max(), (b, np.abs(got[b] - want).max() / np.abs(want).max())
    plan.close()


@pytest.mark.gpu
def test_refine_flag_argument_checks(ctx):
    import mav_trajectory_generation_amd as m
    masks = m.ends_full_masks(12, 4)
    plan = m.Plan(ctx, 12, 3, 4, 5, masks)
    _, times, d_fixed = helpers.reference_batch(3, 4, 12, 3, 1, masks)
    from mav_trajectory_generation_amd import _lib as L
    lay = plan.layout(3, "aos")
    co = np.zeros((3, 4, 3, 12))
    rc = plan.lib.mtg_solve_linear(plan.handle, 3, ctypes.byref(lay), times.ctypes.data_as(ctypes.c_void_p), d_fixed.ctypes.data_as(ctypes.c_void_p),
                                   co.ctypes.data_as(ctypes.c_void_p), None, None, L.FLAG_HOST_POINTERS | L.FLAG_REFINE)
    assert rc == -1          # MTG_ERR_INVALID_ARGUMENT: device pointers only
    plan.close()
