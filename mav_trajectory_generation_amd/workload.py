"""Synthetic "random-waypoint" batches with the semantics of the reference's test generators
(mav_trajectory_generation/src/vertex.cpp:27-82 createRandomVertices, :255-272
estimateSegmentTimesNfabian), vectorised and generated directly on the device so that large
batches never cross PCIe.  Distribution-equivalent, not bit-equal, to the mt19937 original
(the bit-exact generator lives in oracle/ for the parity tests).
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np


def ends_full_masks(n_coeffs: int, n_segments: int, interior_mask: int = 1) -> List[int]:
    """Start/end vertices fix all h derivatives (makeStartOrEnd, vertex.cpp:147-153); interior
    vertices fix `interior_mask` (bit p = derivative p; 1 = position only)."""
    h = n_coeffs // 2
    return [(1 << h) - 1] + [interior_mask] * (n_segments - 1) + [(1 << h) - 1]


def random_waypoint_batch(batch: int, n_segments: int, dimension: int, n_coeffs: int, fixed_mask: Sequence[int],
                          seed: int = 0, device="cuda", layout: str = "aos", box: float = 10.0, v_max: float = 3.0,
                          a_max: float = 5.0, magic: float = 6.5, yaw_dim: bool = False):
    """Returns (times, d_fixed) float64 tensors on `device` in `layout` ('aos', 'soa' or the padded 'soa16').

    Positions uniform in [-box, box]^D with consecutive spacing > 0.2 (re-drawn otherwise); end vertices:
    position random, higher fixed derivatives zero; interior vertices: fixed velocity uniform direction with
    speed <= v_max, fixed acceleration magnitude <= a_max, higher fixed derivatives uniform in [-1, 1]
    (SURVEY.md section 8d config 5).  times = nfabian(v_max, a_max, magic)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    k, d, h = n_segments, dimension, n_coeffs // 2
    pos = (torch.rand((batch, k + 1, d), generator=g, device=device, dtype=torch.float64) * 2 - 1) * box
    if yaw_dim and d == 4:
        pos[..., 3] = (torch.rand((batch, k + 1), generator=g, device=device, dtype=torch.float64) * 2 - 1) * (3 * np.pi)
    for v in range(1, k + 1):
        for _ in range(8):
            close = (pos[:, v] - pos[:, v - 1]).norm(dim=-1) <= 0.2
            n_close = int(close.sum())
            if n_close == 0:
                break
            pos[close, v] = (torch.rand((n_close, d), generator=g, device=device, dtype=torch.float64) * 2 - 1) * box
    dist = (pos[:, 1:] - pos[:, :-1]).norm(dim=-1)
    times = dist / v_max * 2 * (1.0 + magic * v_max / a_max * torch.exp(-dist / v_max * 2))
    cols = []
    for v in range(k + 1):
        for p in range(h):
            if not (fixed_mask[v] >> p) & 1:
                continue
            if p == 0:
                cols.append(pos[:, v])
            elif v in (0, k):
                cols.append(torch.zeros((batch, d), device=device, dtype=torch.float64))
            else:
                scale = {1: v_max, 2: a_max}.get(p, 1.0)
                u = torch.randn((batch, d), generator=g, device=device, dtype=torch.float64)
                u = u / u.norm(dim=-1, keepdim=True)
                r = torch.rand((batch, 1), generator=g, device=device, dtype=torch.float64)
                cols.append(u * r * scale)
    d_fixed = torch.stack(cols, dim=-1)  # [B][D][n_fixed]
    if layout == "soa":
        return times.t().contiguous(), d_fixed.permute(1, 2, 0).contiguous()
    if layout == "soa16":
        # SoA with the row stride padded to a multiple of 16 trajectories (mtg_layout_soa_padded): [K][Bs], [D][n_fixed][Bs];
        # the padding columns hold 1.0 / 0.0 and are never read (pass batch= to Plan.solve)
        bs = (batch + 15) & ~15
        t = torch.ones((k, bs), device=device, dtype=torch.float64)
        f = torch.zeros((d, d_fixed.shape[-1], bs), device=device, dtype=torch.float64)
        t[:, :batch] = times.t()
        f[:, :, :batch] = d_fixed.permute(1, 2, 0)
        return t, f
    return times.contiguous(), d_fixed.contiguous()
