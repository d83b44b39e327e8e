"""CPU ORACLE (test infrastructure, NOT product code) -- restatement of the reference's magnitude-extrema search
and feasibility time scaling.

Follows, step by step:
  SEG   = mav_trajectory_generation/src/segment.cpp        (:83-184 candidates / selection)
  TRAJ  = mav_trajectory_generation/src/trajectory.cpp     (:190-227 computeMinMaxMagnitude, :343-361
          computeMaxVelocityAndAcceleration, :385-429 scaleSegmentTimesToMeetConstraints)
  POLYC = mav_trajectory_generation/src/polynomial.cpp     (:28-83 roots -> candidates, :199-205 scalePolynomialInTime)
  POLYH = .../include/mav_trajectory_generation/polynomial.h (:137-149 evaluate, :230-250 convolve)
  RPOLY = mav_trajectory_generation/src/rpoly/rpoly_ak1.cpp (:57-121 findRootsJenkinsTraub)

Root finder: the reference uses Jenkins-Traub (RPOLY).  Two back ends here:
  * "ref"   -- the reference's own rpoly_ak1.cpp compiled where it lies into oracle/_ref/librpoly_ref.so
               (oracle/Makefile target `ref`; PARITY PINNED against the reference's arithmetic for this step);
  * "numpy" -- numpy.roots (companion-matrix eigenvalues); a port, used when the _ref library is absent and
               cross-checked against "ref" in tests/test_extrema.py.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes
import math
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_REF_LIB = os.path.join(_HERE, "_ref", "librpoly_ref.so")
_ref = None


def ref_available() -> bool:
    return os.path.exists(_REF_LIB)


def _load_ref():
    global _ref
    if _ref is None:
        lib = ctypes.CDLL(_REF_LIB)
        lib.rpoly_ref_find_roots.restype = ctypes.c_int
        lib.rpoly_ref_find_roots.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        _ref = lib
    return _ref


def find_last_non_zero_coeff(c: np.ndarray) -> int:
    """RPOLY:57-68."""
    for i in range(len(c) - 1, -1, -1):
        if abs(c[i]) >= np.finfo(np.float64).tiny:
            return i
    return -1


def find_roots(coefficients_increasing: np.ndarray, backend: str = "auto") -> Tuple[bool, np.ndarray]:
    """findRootsJenkinsTraub (RPOLY:70-121): complex roots of sum c_i t^i after stripping trailing zeros."""
    c = np.ascontiguousarray(coefficients_increasing, dtype=np.float64)
    if backend == "auto":
        backend = "ref" if ref_available() else "numpy"
    if backend == "ref":
        lib = _load_ref()
        re = np.zeros(max(len(c), 1))
        im = np.zeros(max(len(c), 1))
        m = lib.rpoly_ref_find_roots(c.ctypes.data, len(c), re.ctypes.data, im.ctypes.data)
        ok = m >= 0
        if not ok:
            m = -1 - m
        return ok, re[:m] + 1j * im[:m]
    last = find_last_non_zero_coeff(c)
    if last == -1:
        return True, np.zeros(0, dtype=complex)           # RPOLY:75-79
    dec = c[:last + 1][::-1]
    if len(dec) < 2:
        return True, np.zeros(0, dtype=complex)           # RPOLY:86-90
    return True, np.roots(dec).astype(complex)


def base_coefficient(derivative: int, i: int) -> float:
    """POLYC:145-160: i (i-1) ... (i-derivative+1)."""
    r = 1.0
    for k in range(derivative):
        r *= (i - k)
    return r


def get_coefficients(c: np.ndarray, derivative: int) -> np.ndarray:
    """Polynomial::getCoefficients(derivative) (POLYH:100-116): length N, low powers first, tail zero."""
    n = len(c)
    out = np.zeros(n)
    if derivative == 0:
        return np.array(c, dtype=np.float64)
    for i in range(derivative, n):
        out[i - derivative] = base_coefficient(derivative, i) * c[i]
    return out


def evaluate(c: np.ndarray, t: float, derivative: int) -> float:
    """Polynomial::evaluate(t, derivative) (POLYH:137-149): Horner from the highest power."""
    n = len(c)
    if derivative >= n:
        return 0.0
    r = base_coefficient(derivative, n - 1) * c[n - 1]
    for j in range(n - 2, derivative - 1, -1):
        r = r * t + base_coefficient(derivative, j) * c[j]
    return r


def select_min_max_candidates_from_roots(t_start: float, t_end: float, roots: np.ndarray) -> List[float]:
    """POLYC:32-63."""
    cands = [t_start, t_end]
    for r in roots:
        if abs(r.imag) > np.finfo(np.float64).eps:
            continue
        if r.real < t_start or r.real > t_end:
            continue
        cands.append(float(r.real))
    return cands


def compute_min_max_candidates(c: np.ndarray, t_start: float, t_end: float, derivative: int, backend="auto") -> List[float]:
    """Polynomial::computeMinMaxCandidates (POLYC:65-83): roots of derivative+1."""
    _, roots = find_roots(get_coefficients(c, derivative + 1), backend)
    return select_min_max_candidates_from_roots(t_start, t_end, roots)


def segment_candidate_times(seg: np.ndarray, derivative: int, t_start: float, t_end: float,
                            dimensions: Sequence[int], backend="auto") -> List[float]:
    """Segment::computeMinMaxMagnitudeCandidateTimes (SEG:83-131).  seg = [D][N]."""
    n = seg.shape[1]
    if len(dimensions) > 1:
        n_d = n - derivative
        n_dd = n_d - 1
        conv = np.zeros(n_d + n_dd - 1)
        for dim in dimensions:
            d = get_coefficients(seg[dim], derivative)[:n_d]
            dd = get_coefficients(seg[dim], derivative + 1)[:n_dd]
            conv += np.convolve(d, dd)                       # Polynomial::convolve POLYH:234-250
        # computeMinMaxCandidates(t_start, t_end, -1): roots of getCoefficients(0) = conv itself
        _, roots = find_roots(conv, backend)
        return select_min_max_candidates_from_roots(t_start, t_end, roots)
    return compute_min_max_candidates(seg[dimensions[0]], t_start, t_end, derivative, backend)


def segment_min_max_magnitude(seg: np.ndarray, derivative: int, t_start: float, t_end: float,
                              dimensions: Sequence[int], backend="auto"):
    """computeMinMaxMagnitudeCandidates + selectMinMaxMagnitudeFromCandidates (SEG:133-184).
    Returns ((t_min, v_min), (t_max, v_max))."""
    times = segment_candidate_times(seg, derivative, t_start, t_end, dimensions, backend)
    vmin, vmax = (0.0, np.finfo(np.float64).max), (0.0, -np.finfo(np.float64).max)
    for t in times:
        mag = math.sqrt(sum(evaluate(seg[dim], t, derivative) ** 2 for dim in dimensions))
        if t < t_start or t > t_end:
            continue
        if vmax[1] < mag:            # std::max(*maximum, candidate): replaced only when strictly greater
            vmax = (t, mag)
        if mag < vmin[1]:
            vmin = (t, mag)
    return vmin, vmax


def trajectory_min_max_magnitude(segments: np.ndarray, times: Sequence[float], derivative: int,
                                 dimensions: Optional[Sequence[int]] = None, backend="auto"):
    """Trajectory::computeMinMaxMagnitude (TRAJ:190-227).  segments = [K][D][N].
    Returns (t_min, v_min, seg_min), (t_max, v_max, seg_max) and the per-segment [K][4] table
    (t_min, v_min, t_max, v_max)."""
    K, D, _ = segments.shape
    if dimensions is None:
        dimensions = list(range(D))
    mn = (0.0, np.finfo(np.float64).max, 0)
    mx = (0.0, -np.finfo(np.float64).max, 0)
    per_seg = np.zeros((K, 4))
    for k in range(K):
        smin, smax = segment_min_max_magnitude(segments[k], derivative, 0.0, float(times[k]), dimensions, backend)
        per_seg[k] = [smin[0], smin[1], smax[0], smax[1]]
        if smin[1] < mn[1]:
            mn = (smin[0], smin[1], k)
        if smax[1] > mx[1]:
            mx = (smax[0], smax[1], k)
    return mn, mx, per_seg


def compute_max_velocity_and_acceleration(segments, times, backend="auto"):
    """TRAJ:343-361."""
    _, v, _ = trajectory_min_max_magnitude(segments, times, 1, None, backend)
    _, a, _ = trajectory_min_max_magnitude(segments, times, 2, None, backend)
    return v[1], a[1]


def scale_polynomial_in_time(c: np.ndarray, scaling_factor: float) -> np.ndarray:
    """POLYC:199-205."""
    out = np.array(c, dtype=np.float64)
    scale = 1.0
    for n in range(len(out)):
        out[n] *= scale
        scale *= scaling_factor
    return out


def scale_segment_times_to_meet_constraints(segments: np.ndarray, times: Sequence[float], v_max: float,
                                            a_max: float, backend="auto", max_counter: int = 20):
    """TRAJ:385-429.  Returns (within_range, new_segments, new_times, n_iterations_that_scaled)."""
    k_tolerance = 1e-3
    segs = np.array(segments, dtype=np.float64)
    times = np.array(times, dtype=np.float64)
    within_range = False
    n_scaled = 0
    for _ in range(max_counter):
        v_act, a_act = compute_max_velocity_and_acceleration(segs, times, backend)
        velocity_violation = v_act / v_max
        acceleration_violation = a_act / a_max
        within_range = velocity_violation <= 1.0 + k_tolerance and acceleration_violation <= 1.0 + k_tolerance
        if within_range:
            break
        violation_scaling = max(1.0, max(velocity_violation, math.sqrt(acceleration_violation)))
        inv = 1.0 / violation_scaling
        for i in range(segs.shape[0]):
            for d in range(segs.shape[1]):
                segs[i, d] = scale_polynomial_in_time(segs[i, d], inv)
            times[i] = times[i] * violation_scaling
        n_scaled += 1
    return within_range, segs, times, n_scaled
