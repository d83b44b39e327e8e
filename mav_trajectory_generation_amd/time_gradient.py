"""Batched cost-and-time-gradient step of the Mellinger outer loop.

The reference evaluates, per nlopt iteration and per trajectory, K+1 linear solves that differ only in the
segment times (polynomial_optimization_nonlinear_impl.h:287-364, getCostAndGradientMellinger): the base times and,
for each segment n, times with T_n += h and every other T_i -= h/(K-1) (h = 0.1), clamped to
kOptimizationTimeLowerBound = 0.1 (polynomial_optimization_nonlinear.h:31); gradient_n = (J_n - J_0) / h.
Here all (K+1)*B problems of a batch go through ONE cost-only launch of the solve kernel.
"""
from __future__ import annotations

K_OPTIMIZATION_TIME_LOWER_BOUND = 0.1   # polynomial_optimization_nonlinear.h:31
INCREMENT_TIME = 0.1                    # polynomial_optimization_nonlinear_impl.h:312


def mellinger_cost_and_gradient(plan, times, d_fixed):
    """times [B][K], d_fixed [B][D][n_fixed] (AoS CUDA tensors) -> (J [B], dJ/dT [B][K]).
    K == 1: zero gradient, as in the reference (:295-302)."""
    import torch
    bsz, k = times.shape
    if k == 1:
        return plan.solve_cost_only(times, d_fixed), torch.zeros_like(times)
    h = INCREMENT_TIME
    corr = h / (k - 1.0)
    eye = torch.eye(k, dtype=times.dtype, device=times.device)
    # variant 0 = base times; variant n+1 = T_n + h, others - h/(K-1), clamped from below
    pert = times[:, None, :] + eye[None] * h - (1.0 - eye[None]) * corr
    pert = torch.clamp(pert, min=K_OPTIMIZATION_TIME_LOWER_BOUND)
    allt = torch.cat([times[:, None, :], pert], dim=1).reshape(bsz * (k + 1), k).contiguous()
    allf = d_fixed[:, None].expand(bsz, k + 1, *d_fixed.shape[1:]).reshape(bsz * (k + 1), *d_fixed.shape[1:]).contiguous()
    cost = plan.solve_cost_only(allt, allf).reshape(bsz, k + 1)
    j0 = cost[:, 0]
    grad = (cost[:, 1:] - j0[:, None]) / h
    return j0, grad
