"""The MFMA evidence variant of the per-segment contraction H = A^-T Q A^-1 (impl/polynomial_optimization_linear_impl.h:318;
SURVEY.md section 7 K1-alt, the MFMA clause of BASELINE.json's north_star), reachable through include/mtg_hip_lab.h:
literal form on v_mfma_f64_16x16x4_f64 and the scaling identity the solve kernels use, both against the 50-digit oracle
(oracle/oracle_mp.py: mapping_matrix LIN:112-121, cost_matrix LIN:568-583) and against the float64 restatement of what the
reference computes (oracle/oracle_np.py)."""
import numpy as np
import pytest

from oracle import oracle_mp, oracle_np

pytestmark = pytest.mark.gpu


def exact_h(n, d, t):
    mp = oracle_mp.mp
    a = oracle_mp.mapping_matrix(n, mp.mpf(t))
    q = oracle_mp.cost_matrix(n, d, mp.mpf(t))
    ai = a ** -1
    h = ai.T * q * ai
    return np.array([[float(h[r, c]) for c in range(n)] for r in range(n)])


@pytest.mark.parametrize("n,d", [(8, 3), (10, 4), (12, 5), (10, 2), (10, 3), (12, 2), (6, 2), (4, 1), (2, 0)])
def test_literal_and_scaled_match_the_exact_contraction(n, d):
    import torch
    import mav_trajectory_generation_amd as m

    ctx = m.Context(0)
    rng = np.random.default_rng(n * 100 + d)
    times = np.concatenate([rng.uniform(0.3, 20.0, size=61), [1.0, 0.5, 7.25]])
    t = torch.tensor(times, dtype=torch.float64, device="cuda")
    lit = ctx.lab_segment_cost_matrices(n, d, t, 1).cpu().numpy()
    sc = ctx.lab_segment_cost_matrices(n, d, t, 0).cpu().numpy()
    ctx.sync()
    worst_lit = worst_sc = worst_ref = 0.0
    for k in range(len(times)):
        ex = exact_h(n, d, times[k])
        scale = np.abs(ex).max()
        worst_lit = max(worst_lit, np.abs(lit[k] - ex).max() / scale)
        worst_sc = max(worst_sc, np.abs(sc[k] - ex).max() / scale)
        # what the reference computes in float64 (LIN:318 with its Schur-form inverse, LIN:143-179)
        ai = oracle_np.invert_mapping_matrix(oracle_np.setup_mapping_matrix(n, times[k]))
        ref = ai.T @ oracle_np.compute_quadratic_cost_jacobian(n, d, times[k]) @ ai
        worst_ref = max(worst_ref, np.abs(ref - ex).max() / scale)
    # the identity multiplies exact table constants: round-off only.  The literal contraction inherits the conditioning of
    # the products (entries of A^-1 span 1e0 .. 1e8 for N = 12) exactly as the reference's own float64 evaluation does.
    assert worst_sc < 5e-15, worst_sc
    assert worst_lit < max(1e-9, 50 * worst_ref), (worst_lit, worst_ref)
    asym = (np.abs(lit - np.swapaxes(lit, 1, 2)).max(axis=(1, 2)) / np.abs(lit).max(axis=(1, 2))).max()   # H is symmetric
    assert asym < max(1e-9, 50 * worst_ref), (asym, worst_ref)


def test_argument_checks():
    import torch
    import mav_trajectory_generation_amd as m

    ctx = m.Context(0)
    t = torch.ones(4, dtype=torch.float64, device="cuda")
    for bad in ((7, 2), (14, 3), (10, 5), (10, -1)):
        with pytest.raises(Exception):
            ctx.lab_segment_cost_matrices(bad[0], bad[1], t, 1)
    with pytest.raises(Exception):
        ctx.lab_segment_cost_matrices(10, 4, t, 2)
    ctx.sync()
