#!/usr/bin/env python3
"""Generates tests/golden/reference_solve_linear.npz: outputs of the REFERENCE ITSELF run in this container.

The reference's own PolynomialOptimization<N> (setupFromVertices + solveLinear + computeCost, and the Trajectory
analysis calls) is compiled from /root/reference where it lies into oracle/_ref/libmtg_ref.so (oracle/Makefile target
`ref`; Eigen/glog replaced by the container stand-ins in oracle/ref_shim/, see oracle/ref_linear.py for what that
does and does not pin).  Inputs are those of tests/golden/solve_linear_golden.npz (bit-exact reference generators,
parameter sets of the reference's tests), so every case has three answers side by side: the compiled reference
(`*_ref`, here), the numpy restatement (`*_lit`) and the 50-digit solve (`*_mp`).

Needs /root/reference (build container only).  Run from the repo root:
    make -C oracle ref && python tests/golden/make_reference_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_linear  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
gold = np.load(os.path.join(HERE, "solve_linear_golden.npz"))
names = sorted({k.split("/")[0] for k in gold.files})
out = {}
for name in names:
    n, d = int(gold[f"{name}/n"]), int(gold[f"{name}/d"])
    masks = [int(m) for m in gold[f"{name}/masks"]]
    times, d_fixed = gold[f"{name}/times"], gold[f"{name}/d_fixed"]
    co, fr, cost, _ = ref_linear.solve_batch(n, d, masks, times, d_fixed)
    out[f"{name}/coeffs_ref"], out[f"{name}/d_free_ref"], out[f"{name}/cost_ref"] = co, fr, cost
    # the post-solve calls on the first trajectory of the case: velocity / acceleration extrema, sampled positions
    seg, t = co[0], times[0]
    if n >= 6:
        for der in (1, 2):
            mn, mx, per = ref_linear.minmax_magnitude(seg, t, der)
            out[f"{name}/minmax_d{der}"] = np.array([*mn, *mx])
            out[f"{name}/minmax_per_segment_d{der}"] = per
    grid = np.linspace(0.0, float(np.sum(t)) * (1 - 1e-12), 33)
    out[f"{name}/sample_t"] = grid
    out[f"{name}/sample_pos"] = ref_linear.evaluate(seg, t, grid, 0)
    out[f"{name}/sample_acc"] = ref_linear.evaluate(seg, t, grid, 2) if n >= 4 else np.zeros((33, seg.shape[1]))
    print(name, "done")
np.savez_compressed(os.path.join(HERE, "reference_solve_linear.npz"), **out)
print("wrote", len(out), "arrays")
