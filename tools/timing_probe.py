"""Measurement-only: reads the per-wave phase timestamps written by the MTG_TIMING build (via the cost buffer)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import numpy as np, torch
import mav_trajectory_generation_amd as m
B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
dims = sys.argv[2] if len(sys.argv) > 2 else "fused"
masks = m.ends_full_masks(10, 8)
ctx = m.Context(0)
plan = m.Plan(ctx, 10, 3, 8, 4, masks)
with torch.cuda.stream(ctx.stream):
    t, f = m.random_waypoint_batch(B, 8, 3, 10, masks, seed=5, device="cuda", layout="soa")
    co = torch.empty((B, 8, 3, 10), dtype=torch.float64, device="cuda")
    dbg = torch.zeros((1 << 20,), dtype=torch.int64, device="cuda")
    plan.lib.mtg_plan_set_workspace(plan.handle, ctypes.c_void_p(dbg.data_ptr()), dbg.numel() * 8)
    for _ in range(3):
        plan.solve(t, f, layout="soa", coeffs=co, dims=dims)
    torch.cuda.synchronize()
    dbg.zero_()
    plan.solve(t, f, layout="soa", coeffs=co, dims=dims)
    torch.cuda.synchronize()
raw = dbg.cpu().numpy()
ntiles = (B + 63) // 64
ng = 3 if dims == "split" else 1
nw = min(ntiles, 2048 // ng) * ng * 2
T = raw[: nw * 16].reshape(nw, 16)
T = T[T[:, 0] > 0]
t0 = T[:, 0].min()
names = ["start", "fwd_done", "after_bar1", "finish_done", "stores_done", "preload_done", "mid_done", "bwd0", "bwd1", "bwd2", "bwd3"]
print(f"B={B} dims={dims} waves={len(T)}")
for i, n in enumerate(names):
    v = T[:, i] - t0
    print(f"  {n:14s} min {v.min():8d}  median {int(np.median(v)):8d}  max {v.max():8d}")
d = lambda a, b: int(np.median(T[:, a] - T[:, b]))
print("  medians: preload %d | forward %d | barrier %d | mid %d | bwd3 %d | bwd2 %d | bwd1 %d | bwd0 %d | tail %d | store drain %d"
      % (d(5, 0), d(1, 5), d(2, 1), d(6, 2), d(10, 6), d(9, 10), d(8, 9), d(7, 8), d(3, 7), d(4, 3)))

# device-wide 100 MHz clock (comparable across XCDs, 10 ns resolution): dispatch skew and in-kernel span
w0, w1 = T[:, 14], T[:, 15]
ok = (w0 > 0) & (w1 > 0)
if ok.any():
    base = w0[ok].min()
    s0 = (w0[ok] - base) * 0.01
    s1 = (w1[ok] - base) * 0.01
    print("  wave start  [us after the first wave]: median %.2f  p90 %.2f  max %.2f" % (np.median(s0), np.percentile(s0, 90), s0.max()))
    print("  wave end    [us after the first wave]: median %.2f  p90 %.2f  max %.2f" % (np.median(s1), np.percentile(s1, 90), s1.max()))
    print("  wave lifetime [us]: median %.2f  max %.2f" % (np.median(s1 - s0), (s1 - s0).max()))

    # placement: HW_ID bits  simd [5:4], cu [11:8], sh [12], se [15:13]; XCC_ID bits [3:0]
    hw = T[ok, 13] & 0xffffffff
    xcc = (T[ok, 13] >> 32) & 0xf
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
    key_cu = xcc * 10000 + se * 1000 + sh * 100 + cu
    key_simd = key_cu * 10 + simd
    import collections
    per_simd = collections.Counter(key_simd.tolist())
    per_cu = collections.Counter(key_cu.tolist())
    print("  CUs used %d, SIMDs used %d; waves per SIMD histogram %s; waves per CU histogram %s" % (
        len(per_cu), len(per_simd), dict(collections.Counter(per_simd.values())), dict(sorted(collections.Counter(per_cu.values()).items()))))
    life = s1 - s0
    share = np.array([per_simd[k] for k in key_simd.tolist()])
    for c in sorted(set(share.tolist())):
        print("  lifetime of waves with %d wave(s) on their SIMD: median %.2f max %.2f (n=%d)" % (c, np.median(life[share == c]), life[share == c].max(), (share == c).sum()))
    late = s0 > np.percentile(s0, 75)
    print("  lifetime of the last-started quarter: median %.2f; of the first-started quarter: median %.2f" % (np.median(life[late]), np.median(life[s0 <= np.percentile(s0, 25)])))

    # absolute timeline on the device-wide clock: when does each phase begin / end across the launch?
    # (shader-cycle stamps converted per wave: phase offset in cycles / that wave's cycles-per-us)
    cyc = (T[ok, 4] - T[ok, 0]).astype(float)              # start .. stores_done in shader cycles
    per_us = cyc / np.maximum(s1 - s0, 1e-9)
    print("  shader clock during the wave lifetimes: median %.0f MHz (p10 %.0f, p90 %.0f)" % (
        np.median(per_us), np.percentile(per_us, 10), np.percentile(per_us, 90)))
    def abs_us(col):
        return s0 + (T[ok, col] - T[ok, 0]) / per_us
    pct = lambda v: "min %.2f p25 %.2f med %.2f p75 %.2f max %.2f" % (v.min(), np.percentile(v, 25), np.median(v), np.percentile(v, 75), v.max())
    print("  [us after the first wave start]")
    print("  wave start      : " + pct(s0))
    print("  preload done    : " + pct(abs_us(5)))
    print("  forward done    : " + pct(abs_us(1)))
    print("  middle solved   : " + pct(abs_us(6)))
    print("  last seg recov. : " + pct(abs_us(3)))
    print("  stores acked    : " + pct(s1))
