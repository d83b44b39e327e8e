"""CPU latency protocol of the reference's own benchmark (src/polynomial_timing_evaluation.cpp:93-128): wall clock of
setupFromVertices + solveLinear per trajectory, N = 10, D = 3, K in {2, 8, 10, 50, 100}, 1000 runs, ONE host thread --
for (a) the reference's own code compiled from /root/reference against the Eigen/glog container stand-ins
(oracle/_ref/libmtg_ref.so: slower than real Eigen) and (b) the C++ restatement (oracle/libcpu_ref.so: no sparse
bookkeeping / allocation, i.e. faster than real Eigen).  Real Eigen lies between the two.  Oracle-side tool."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from oracle import cpu_ref, ref_linear

runs = 1000
for k in (2, 8, 10, 50, 100):
    masks = [31] + [1] * (k - 1) + [31]
    pos, times = cpu_ref.generate(runs, k, 3, 1)
    nf = 10 + (k - 1)
    d_fixed = np.zeros((runs, 3, nf))
    d_fixed[:, :, 0] = pos[:, 0]
    for v in range(1, k):
        d_fixed[:, :, 4 + v] = pos[:, v]
    d_fixed[:, :, 5 + (k - 1)] = pos[:, k]
    row = dict(K=k, runs=runs, threads=1)
    _, _, _, s = cpu_ref.solve_batch(10, 4, masks, times, d_fixed, nthreads=1, want_free=False, want_cost=False)
    row["port_us_per_trajectory"] = round(s / runs * 1e6, 2)
    if ref_linear.available():
        _, _, _, s = ref_linear.solve_batch(10, 4, masks, times, d_fixed, want_free=False, want_cost=False)
        row["reference_build_us_per_trajectory"] = round(s / runs * 1e6, 2)
    print(json.dumps(row))
