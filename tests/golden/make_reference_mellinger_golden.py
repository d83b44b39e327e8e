#!/usr/bin/env python3
"""Generates tests/golden/reference_mellinger.npz: cost and finite-difference time gradient of the REFERENCE's own
PolynomialOptimizationNonLinear<N>::getCostAndGradientMellinger (impl/polynomial_optimization_nonlinear_impl.h:287-364) run in
this container -- oracle/_ref/libmtg_ref_nl.so, compiled from /root/reference where it lies (oracle/Makefile target `ref`;
Eigen / glog container stand-ins of oracle/ref_shim/, types-only nlopt stand-in oracle/ref_shim_nlopt/nlopt.hpp: the member never
calls nlopt).  The cases are those of tests/test_gpu_parity.py (seeded bit-exact reference generators, incl. the lower clamp
of impl:338-340); inputs are stored next to the outputs.

Needs /root/reference (build container only).  Run from the repo root:
    make -C oracle ref && python tests/golden/make_reference_mellinger_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
from oracle import ref_linear  # noqa: E402

CASES = [  # name, n, d, k, dim, masks, bsz, seed, clamp tweaks
    ("snap_k8_d3", 10, 4, 8, 3, None, 12, 60606, False),
    ("snap_k8_b70", 10, 4, 8, 3, None, 6, 424242, True),
    ("snap_k16", 10, 4, 16, 3, None, 6, 424242, True),
    ("ragged_k6", 10, 4, 6, 3, [31, 1, 3, 1, 5, 9, 31], 5, 424242, True),
    ("config5_k16_d4", 10, 4, 16, 4, [31] + [7] * 15 + [31], 6, 424242, True),
    ("jerk_k4", 8, 3, 4, 3, None, 6, 424242, True),
    ("one_segment", 10, 4, 1, 3, None, 4, 424242, True),
    ("n12_k8", 12, 5, 8, 3, None, 6, 424242, False),   # (no clamp tweak: an N = 12 chain with a 0.05 s segment is beyond float64 on every side)
]


def inputs(n, k, dim, masks, bsz, seed, tweak):
    masks, times, d_fixed = helpers.reference_batch(bsz, k, n, dim, seed, masks)
    if tweak:
        times[0, 0] = 0.12            # T - h/(K-1) falls below the bound 0.1 -> clamped (impl:338-340)
        if k > 1:
            times[1 % bsz, k - 1] = 0.05   # already below the bound: every variant clamps it
    return masks, times, d_fixed


if __name__ == "__main__":
    assert ref_linear.nonlinear_available(), "make -C oracle ref first"
    out = {}
    for name, n, d, k, dim, masks, bsz, seed, tweak in CASES:
        masks, times, d_fixed = inputs(n, k, dim, masks, bsz, seed, tweak)
        cost, grad = ref_linear.mellinger_cost_gradient(n, d, masks, times, d_fixed)
        out[f"{name}/n"], out[f"{name}/d"] = np.int64(n), np.int64(d)
        out[f"{name}/masks"], out[f"{name}/times"], out[f"{name}/d_fixed"] = np.array(masks, dtype=np.int64), times, d_fixed
        out[f"{name}/cost_ref"], out[f"{name}/grad_ref"] = cost, grad
        print(name, "cost", cost[:2], "grad scale", np.abs(grad).max())
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_mellinger.npz"), **out)
    print("wrote", len(out), "arrays")
