"""The reference's on-disk trajectory format (mav_trajectory_generation/src/io.cpp:27-31, :126-218) for the
coefficient buffers this package produces.

    segments:
      - N: 10
        D: 3
        time: 3970847833  # [ns]
        coefficients:
          - [c0, ..., c9]        one flow sequence per dimension, increasing powers
          - [...]

Segment times travel as integer nanoseconds, truncated like Segment::getTimeNSec (segment.h:58-60):
`uint64(1e9 * t)`; reading gives `ns * 1e-9` (setTimeNSec, :63).  Coefficients are written with 17 significant
digits, so a write/read cycle is exact for them.  Host-side code (numpy + PyYAML for reading)."""
from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np

K_SEGMENTS, K_N, K_D, K_TIME, K_COEFFS = "segments", "N", "D", "time", "coefficients"   # io.cpp:27-31


def time_to_nsec(t: float) -> int:
    """Segment::getTimeNSec: static_cast<uint64_t>(1e9 * time)."""
    return int(np.uint64(1.0e9 * float(t)))


def segments_to_yaml(coeffs, times: Sequence[float]) -> str:
    """coeffs [K][D][N] (increasing powers), times [K] seconds -> the document segmentsToFile writes (io.cpp:126-167)."""
    c = np.asarray(coeffs, dtype=np.float64)
    if c.ndim != 3 or len(times) != c.shape[0]:
        raise ValueError("coeffs must be [K][D][N] with one time per segment")
    lines = [K_SEGMENTS + ":"]
    for k in range(c.shape[0]):
        lines.append("  - %s: %d" % (K_N, c.shape[2]))
        lines.append("    %s: %d" % (K_D, c.shape[1]))
        lines.append("    %s: %d  # [ns]" % (K_TIME, time_to_nsec(times[k])))
        lines.append("    %s:" % K_COEFFS)
        for d in range(c.shape[1]):
            lines.append("      - [" + ", ".join(_fmt(v) for v in c[k, d]) + "]")
    return "\n".join(lines) + "\n"


def _fmt(v: float) -> str:
    if np.isnan(v):
        return ".nan"
    if np.isinf(v):
        return ".inf" if v > 0 else "-.inf"
    return "%.17g" % v


def segments_to_file(filename: str, coeffs, times: Sequence[float]) -> bool:
    """segmentsToFile / trajectoryToFile (io.cpp:126-167, io.h:45-51)."""
    try:
        with open(filename, "w") as f:
            f.write(segments_to_yaml(coeffs, times))
    except OSError:
        return False
    return True


def segments_from_yaml(text: str) -> Tuple[np.ndarray, np.ndarray]:
    """Inverse of segments_to_yaml; follows segmentsFromFile's checks (io.cpp:169-218).  Raises ValueError where the
    reference returns false.  All segments must share (N, D) to come back as one array (Trajectory::setSegments
    CHECKs the same, trajectory.cpp:158-170)."""
    import yaml
    node = yaml.safe_load(text)
    if not isinstance(node, dict) or K_SEGMENTS not in node:
        raise ValueError("no segments element")
    segs = node[K_SEGMENTS] or []
    coeffs, times = [], []
    for s in segs:
        if not isinstance(s, dict) or any(k not in s for k in (K_N, K_D, K_TIME, K_COEFFS)):
            raise ValueError("wrong format, missing elements")
        n, d = int(s[K_N]), int(s[K_D])
        rows = s[K_COEFFS]
        if len(rows) != d:
            raise ValueError("coefficients and dimensions do not coincide")
        if any(len(r) != n for r in rows):
            raise ValueError("number of coefficients does not coincide")
        coeffs.append([[float(v) for v in r] for r in rows])
        times.append(int(s[K_TIME]) * 1.0e-9)
    if not coeffs:
        return np.zeros((0, 0, 0)), np.zeros(0)
    if any(len(c) != len(coeffs[0]) or len(c[0]) != len(coeffs[0][0]) for c in coeffs):
        raise ValueError("segments of one trajectory must share N and D")
    return np.array(coeffs, dtype=np.float64), np.array(times, dtype=np.float64)


def segments_from_file(filename: str) -> Tuple[np.ndarray, np.ndarray]:
    with open(filename) as f:
        return segments_from_yaml(f.read())


def batch_to_files(pattern: str, coeffs, times) -> int:
    """coeffs [B][K][D][N], times [B][K] (host arrays or CUDA tensors): one file per trajectory, pattern % b."""
    c = coeffs.cpu().numpy() if hasattr(coeffs, "cpu") else np.asarray(coeffs)
    t = times.cpu().numpy() if hasattr(times, "cpu") else np.asarray(times)
    n = 0
    for b in range(c.shape[0]):
        n += bool(segments_to_file(pattern % b, c[b], t[b]))
    return n
