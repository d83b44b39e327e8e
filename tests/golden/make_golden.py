#!/usr/bin/env python3
"""Generates tests/golden/solve_linear_golden.npz.

The reference (C++/Eigen) cannot be built or imported in this image, so the fixtures are
  * the reference's own known-answer vector (test_polynomial_optimization.cpp:777-780, MATLAB),
  * outputs of the literal float64 restatement (oracle/oracle_np.py, pinned on that vector), and
  * outputs of the 50-digit mpmath solve (oracle/oracle_mp.py) = ground truth,
on inputs produced by bit-exact re-implementations of the reference generators
(createRandomVertices vertex.cpp:27-82 with std::mt19937, estimateSegmentTimesNfabian :255-272)
with the parameter sets of test_polynomial_optimization.cpp:790-867 and
mav_trajectory_generation_ros/test/test_feasibility.cpp:54-116.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle_mp as omp  # noqa: E402
from oracle import oracle_np as onp  # noqa: E402
import helpers  # noqa: E402

cases = {}


def add(name, n, d, masks, times, d_fixed, with_mp=True):
    c_lit, f_lit, j_lit = onp.solve_batch(n, d, masks, times, d_fixed)
    entry = dict(n=n, d=d, masks=np.array(masks), times=times, d_fixed=d_fixed, coeffs_lit=c_lit, d_free_lit=f_lit,
                 cost_lit=j_lit)
    if with_mp:
        c_mp, f_mp, j_mp = omp.solve_batch(n, d, masks, times, d_fixed)
        entry.update(coeffs_mp=c_mp, d_free_mp=f_mp, cost_mp=j_mp)
    for k, v in entry.items():
        cases[f"{name}/{k}"] = np.asarray(v)
    print(name, "done")


# 1. TwoVerticesSetup (test_polynomial_optimization.cpp:743-787): 1-D, N=10, rest(0) -> rest(5), T=5, snap.
masks = [31, 31]
times = np.array([[5.0]])
d_fixed = np.zeros((1, 1, 10))
d_fixed[0, 0, 5] = 5.0
add("two_vertices", 10, 4, masks, times, d_fixed)
cases["two_vertices/matlab_coeffs"] = np.array([-0.000000000000004, 0.000000000000004, -0.000000000000006,
                                                0.000000000000003, -0.000000000000001, 0.201600000000015,
                                                -0.134400000000012, 0.034560000000004, -0.004032000000000,
                                                0.000179200000000])

# 2. README example (README.md:104-140): (0,0,1) -> (1,2,3) -> (2,1,5), v_max = a_max = 2, N=10, snap.
vs = [onp.Vertex(3) for _ in range(3)]
vs[0].make_start_or_end([0, 0, 1], 4)
vs[1].add_constraint(0, [1, 2, 3])
vs[2].make_start_or_end([2, 1, 5], 4)
t = onp.estimate_segment_times(vs, 2.0, 2.0)
masks = [31, 1, 31]
d_fixed = np.zeros((1, 3, 11))
col = 0
for v in vs:
    for p in range(5):
        c = v.get_constraint(p)
        if c is not None:
            d_fixed[0, :, col] = c
            col += 1
add("readme", 10, 4, masks, np.array([t]), d_fixed)

# 3. BASELINE config 2 shape: 8 segments, N=10, 3-D, snap; seeds 0..15.
masks, times, d_fixed = helpers.reference_batch(16, 8, 10, 3, 0)
add("config2", 10, 4, masks, times, d_fixed)

# 4. Parameter sets of test_polynomial_optimization.cpp:790-867 (D, d, K, seed, pos_max, v, a).
params = [(1, 4, 1, 100, 3.0, 5.0), (1, 4, 10, 102, 3.0, 5.0), (1, 4, 50, 103, 3.0, 5.0), (3, 4, 1, 104, 3.0, 5.0),
          (3, 4, 10, 105, 3.0, 5.0), (3, 4, 50, 106, 3.0, 5.0), (1, 2, 5, 107, 1.0, 2.0), (3, 2, 1, 108, 1.0, 2.0),
          (3, 2, 5, 109, 1.0, 2.0), (3, 3, 5, 110, 1.0, 2.0)]
for (dim, d, k, seed, v, a) in params:
    # the test fixture builds vertices with maximum_derivative = derivative_to_optimize (TOPT:60-62)
    vs = onp.create_random_vertices(d, k, [-10.0] * dim, [10.0] * dim, seed)
    t = onp.estimate_segment_times(vs, v, a)
    masks = [(1 << (d + 1)) - 1] + [1] * (k - 1) + [(1 << (d + 1)) - 1]
    nf = sum(bin(m).count("1") for m in masks)
    d_fixed = np.zeros((1, dim, nf))
    col = 0
    for vv in vs:
        for p in range(5):
            c = vv.get_constraint(p)
            if c is not None:
                d_fixed[0, :, col] = c
                col += 1
    add(f"topt_D{dim}_d{d}_K{k}_s{seed}", 10, d, masks, np.array([t]), d_fixed, with_mp=(k <= 10))

# 5. N=12 instantiations of test_feasibility.cpp:54-116 (free end-vertex derivatives, K=1).
rng = np.random.default_rng(1234567)
vs_list, times = [], []
bsz = 8
d_fixed = np.zeros((bsz, 3, 10))
for b in range(bsz):
    vs = onp.create_random_vertices(4, 1, [-5.0] * 3, [5.0] * 3, b)
    for e in (0, 1):
        dirv = rng.uniform(0.01, 1.0, 3)
        vs[e].add_constraint(1, dirv / np.linalg.norm(dirv) * rng.uniform(0.0, 2.0))
    dist = np.linalg.norm(vs[1].get_constraint(0) - vs[0].get_constraint(0))
    times.append([dist / rng.uniform(0.5, 2.0)])
    col = 0
    for vv in vs:
        for p in range(5):
            d_fixed[b, :, col] = vv.get_constraint(p)
            col += 1
add("feas_pos_N12", 12, 4, [31, 31], np.array(times), d_fixed)
d_fixed = np.zeros((bsz, 1, 4))
for b in range(bsz):
    vs = onp.create_random_vertices(1, 1, [-3 * np.pi], [3 * np.pi], b)
    col = 0
    for vv in vs:
        for p in range(2):
            d_fixed[b, :, col] = vv.get_constraint(p)
            col += 1
add("feas_yaw_N12", 12, 2, [3, 3], np.array(times), d_fixed)

# 6. config 5 shape (K=16, D=4, interior pos+vel+acc fixed) and a mixed-N sample (config 4 buckets).
masks = helpers.masks_ends_full(10, 16, 7)
masks, times, d_fixed = helpers.reference_batch(4, 16, 10, 4, 500, masks)
add("config5", 10, 4, masks, times, d_fixed)
for (n, d, k) in [(8, 3, 4), (12, 5, 8), (8, 3, 32)]:
    masks, times, d_fixed = helpers.reference_batch(4, k, n, 3, 900 + n + k)
    add(f"config4_N{n}_K{k}", n, d, masks, times, d_fixed, with_mp=(k <= 8))

out = os.path.join(ROOT, "tests", "golden", "solve_linear_golden.npz")
np.savez_compressed(out, **cases)
print("wrote", out, os.path.getsize(out), "bytes")
