"""Pick the workspace / LDS step counts (MtgCfg::WSJ, ::LSJ) of a dimension-in-lane variant: the smallest number of
workspace steps whose kernel compiles without scratch spills (hipcc -Rpass-analysis=kernel-resource-usage), the last
LS of them in LDS (as many as fit with two 2-wave workgroups per CU).  Prints one line per shape in the form used by
csrc/mtg_dimlane_more_h*.inc.  CPU only (cross-compiles for gfx950).

usage: python tools/search_ws.py K_FIRST K_LAST [H ...]      (H = N / 2 in {4, 5, 6}; default all three)"""
import concurrent.futures as cf
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mav_trajectory_generation_amd", "csrc")
SHAPES = {4: (15, 1, 15, 3), 5: (31, 1, 31, 4), 6: (63, 1, 63, 5)}      # (start mask, interior mask, end mask, derivative)
LDS_STEPS = {4: 8, 5: 6, 6: 4}                                          # steps that fit the LDS next to the output slabs
REG_STEPS = {4: 12, 5: 9, 6: 5}                                          # where the search starts (steps the registers hold)


def resources(H, K, WS, LS, RS=0, full=False):
    ms, mi, me, dv = SHAPES[H]
    cfg = f"MtgCfg<{H},1,{K},{ms},{mi},{me},{dv},0,{WS},{3 if (WS > 0 or RS) else 0},{LS},{RS}>"
    src = f"/tmp/search_ws_{os.environ.get('SEARCH_KERNEL', 'plain')}_{H}_{K}_{WS}_{LS}_{RS}.hip"
    with open(src, "w") as f:
        if os.environ.get("SEARCH_KERNEL") == "extra":      # the extra-output kernel (cost / d_P) of the same configuration
            f.write('#include "mtg_dimlane.h"\ntemplate __global__ void mtg_solve_dl_extra_kernel<' + cfg +
                    ', 3, 1, 18>(const double*, const double*, double*, int*, int*, int, int, int, int, double*, MtgDlExtra);\n')
        else:
            f.write('#include "mtg_dimlane.h"\ntemplate __global__ void mtg_solve_dl_kernel<' + cfg +
                    ', 3, 1, 0, 18>(const double*, const double*, double*, int*, int*, int, int, int, int, double*);\n')
    p = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + CSRC, "-I" + os.path.join(ROOT, "include"),
                        "-mllvm", "-disable-machine-licm", "-mllvm", "-amdgpu-kernarg-preload-count=14", "-mllvm",
                        "-pragma-unroll-threshold=1000000", "--cuda-device-only", "-c", src, "-o", src + ".o",
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    if "error" in p.stderr:
        return None
    scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", p.stderr).group(1))
    spills = int(re.search(r"VGPRs Spill: (\d+)", p.stderr).group(1))
    if full:
        return dict(scratch=scratch, spills=spills, vgpr=int(re.search(r" VGPRs: (\d+)", p.stderr).group(1)),
                    agpr=int(re.search(r"AGPRs: (\d+)", p.stderr).group(1)), sgpr_spill=int(re.search(r"SGPRs Spill: (\d+)", p.stderr).group(1)),
                    lds=int(re.search(r"LDS Size \[bytes/block\]: (\d+)", p.stderr).group(1)))
    return scratch, spills


RS_REG_STEPS = {4: 16, 5: 12, 6: 8}                                    # register steps with the shared storage (search start)


def best(H, K, RS=1):
    """RS = 1 (round 3): register steps with G shared between the dimension lanes (MtgCfg::kRegShared, MTG_DLR lines)."""
    kc = (K + 1) // 2
    start = kc - (RS_REG_STEPS if RS else REG_STEPS)[H]
    for WS in range(max(0, start), kc + 1):
        LS = min(WS, LDS_STEPS[H])
        r = resources(H, K, WS, LS, RS)
        if r is None:
            return f"// H={H} K={K}: compile error"
        if r[0] == 0 and r[1] <= 16:      # no scratch; a few registers parked in AGPRs are fine
            ms, mi, me, dv = SHAPES[H]
            macro = "MTG_DLR" if RS else "MTG_DLW"
            if WS == 0 and kc <= REG_STEPS[H]:
                macro, tail = "MTG_DL", ""      # fits without sharing: keep the plain variant
                return f"MTG_DL({H}, {K}, {ms}, {mi}, {me}, {dv}, 3, {2 if K <= 8 else 1}, 0, 0)"
            return f"{macro}({H}, {K}, {ms}, {mi}, {me}, {dv}, 3, {2 if K <= 8 else 1}, 0, 0, {WS}, {LS})   // spilled VGPRs: {r[1]}"
    return f"// H={H} K={K}: no setting without scratch"


if __name__ == "__main__" and sys.argv[1] == "probe":
    # python tools/search_ws.py probe H K WS LS RS [WS LS RS ...]: resource usage of explicit settings
    H, K = int(sys.argv[2]), int(sys.argv[3])
    trip = [tuple(int(x) for x in sys.argv[i:i + 3]) for i in range(4, len(sys.argv), 3)]
    with cf.ThreadPoolExecutor(max(1, (os.cpu_count() or 2) - 1)) as ex:
        for t, r in zip(trip, ex.map(lambda a: resources(H, K, a[0], a[1], a[2], True), trip)):
            print(f"H={H} K={K} WS={t[0]} LS={t[1]} RS={t[2]}: {r}", flush=True)
    sys.exit(0)

if __name__ == "__main__":
    k0, k1 = int(sys.argv[1]), int(sys.argv[2])
    hs = [int(x) for x in sys.argv[3:]] or [4, 5, 6]
    jobs = [(H, K) for H in hs for K in range(k0, k1 + 1)]
    with cf.ThreadPoolExecutor(max(1, (os.cpu_count() or 2) - 1)) as ex:
        rs = int(os.environ.get('SEARCH_RS', '1'))      # SEARCH_RS=0: register steps unshared (MTG_DLW lines)
        for line in ex.map(lambda a: best(*a, RS=rs), jobs):
            print(line, flush=True)
