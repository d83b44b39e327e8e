"""Generates tests/golden/stress_outliers.json: seeded random shapes / ragged masks (the generator of
tools/stress_random.py) on which the compiled reference (oracle/_ref/libmtg_ref.so -- the reference's own
solveLinear()) and the product's lane code (host emulation of csrc/mtg_lane.h) differ by more than the north-star
tolerance 1e-9.  The 50-digit solve (oracle/oracle_mp.py) says which side is off: tests/test_gpu_stress.py pins these
cases on the GPU.  Needs /root/reference (to build oracle/_ref); run from the repo root:
    python tests/golden/make_stress_outliers.py
"""
import ctypes
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import helpers  # noqa: E402
from oracle import ref_linear  # noqa: E402

SEED, NCASES, KEEP = 20260925, 6000, 6


def case_stream(seed, ncases):
    """The shape / mask / seed stream shared with tests/test_gpu_stress.py."""
    rng = np.random.default_rng(seed)
    for case in range(ncases):
        n = int(rng.choice([6, 8, 10, 10, 10]))
        h = n // 2
        k = int(rng.choice([1, 2, 3, 5, 8, 8, 13, 16, 24]))
        dim = int(rng.choice([1, 2, 3, 3, 4]))
        bsz = int(rng.choice([1, 7, 33, 64]))
        style = int(rng.integers(0, 3))
        if style == 0:
            masks = None
        elif style == 1:
            masks = [(1 << h) - 1] + [int(rng.choice([1, 3, 7 & ((1 << h) - 1)]))] * (k - 1) + [(1 << h) - 1]
        else:
            masks = [(1 << h) - 1] + [1 | int(rng.integers(0, 1 << h)) for _ in range(k - 1)] + [(1 << h) - 1]
        yield dict(case=case, n=n, k=k, dim=dim, bsz=bsz, masks=masks, seed=int(rng.integers(1, 1 << 30)))


def main():
    lib = ctypes.CDLL(os.path.join(ROOT, "tests", "libmtg_host_emu.so"))
    dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
    lib.mtg_emu_run.argtypes = [ctypes.c_int] * 4 + [ip, ctypes.c_longlong, dp, dp, dp, dp, dp, ctypes.c_int, ip]
    lib.mtg_emu_run.restype = ctypes.c_int
    found = []
    for c in case_stream(SEED, NCASES):
        masks, times, d_fixed = helpers.reference_batch(c["bsz"], c["k"], c["n"], c["dim"], c["seed"], c["masks"])
        ref_c, _, _, _ = ref_linear.solve_batch(c["n"], c["n"] // 2 - 1, masks, times, d_fixed, nthreads=8)
        rc, co, _, _, st = helpers.emu_run(lib, c["n"], c["dim"], c["k"], c["n"] // 2 - 1, masks, times, d_fixed, 0)
        assert rc == 0 and st == 0
        e = helpers.poly_relerr(co, ref_c)
        if e > 1e-9:
            found.append(dict(c, masks=[int(x) for x in masks], lane_code_vs_reference=e))
    found.sort(key=lambda r: -r["lane_code_vs_reference"])
    out = dict(stream_seed=SEED, stream_cases=NCASES, above_1e9=len(found), cases=found[:KEEP])
    json.dump(out, open(os.path.join(HERE, "stress_outliers.json"), "w"), indent=1)
    print(len(found), "of", NCASES, "cases above 1e-9; worst", found[0]["lane_code_vs_reference"])


if __name__ == "__main__":
    main()
