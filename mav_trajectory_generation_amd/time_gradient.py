"""Batched cost-and-time-gradient step of the Mellinger outer loop.

The reference evaluates, per nlopt iteration and per trajectory, K+1 linear solves that differ only in the
segment times (polynomial_optimization_nonlinear_impl.h:287-364, getCostAndGradientMellinger): the base times and,
for each segment n, times with T_n += h and every other T_i -= h/(K-1) (h = 0.1), clamped to
kOptimizationTimeLowerBound = 0.1 (polynomial_optimization_nonlinear.h:31); gradient_n = (J_n - J_0) / h.
The C ABI entry mtg_mellinger_cost_gradient runs all (K+1)*B problems as ONE cost-only launch and forms the perturbed
times inside the kernel; this module is its thin Python binding.
"""
from __future__ import annotations

import ctypes

K_OPTIMIZATION_TIME_LOWER_BOUND = 0.1   # polynomial_optimization_nonlinear.h:31
INCREMENT_TIME = 0.1                    # polynomial_optimization_nonlinear_impl.h:312


def mellinger_cost_and_gradient(plan, times, d_fixed, layout: str = "aos", increment_time: float = INCREMENT_TIME,
                                lower_bound: float = K_OPTIMIZATION_TIME_LOWER_BOUND):
    """times / d_fixed: float64 CUDA tensors in `layout` ('aos': [B][K], [B][D][n_fixed]; 'soa': [K][B], [D][n_fixed][B])
    -> (J [B], dJ/dT shaped like `times`).  K == 1: zero gradient, as in the reference (:295-302).  Asynchronous."""
    import torch
    assert times.is_cuda and times.dtype == torch.float64 and times.is_contiguous() and d_fixed.is_contiguous()
    batch = times.shape[0] if layout == "aos" else times.shape[1]
    cost = torch.empty((batch,), dtype=torch.float64, device=times.device)
    grad = torch.empty_like(times)
    lay = plan.layout(batch, layout)
    cur = plan.ctx._enter()
    rc = plan.lib.mtg_mellinger_cost_gradient(plan.handle, batch, ctypes.byref(lay), ctypes.c_void_p(times.data_ptr()),
                                              ctypes.c_void_p(d_fixed.data_ptr()), float(increment_time), float(lower_bound),
                                              ctypes.c_void_p(cost.data_ptr()), ctypes.c_void_p(grad.data_ptr()))
    plan.ctx._leave(cur)
    from .core import _check
    _check(plan.lib, rc, plan.ctx.handle)
    return cost, grad
