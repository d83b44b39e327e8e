"""ctypes wrapper of oracle/libcpu_ref.so (C++ restatement of the reference algorithm; oracle / CPU baseline only)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcpu_ref.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "cpu_ref.cpp")
        if not os.path.exists(_SO) or os.path.getmtime(src) > os.path.getmtime(_SO):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        lib = ctypes.CDLL(_SO)
        dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
        lib.cpu_ref_solve_batch.restype = ctypes.c_double
        lib.cpu_ref_solve_batch.argtypes = [ctypes.c_int] * 4 + [ip, ctypes.c_longlong, dp, dp, dp, dp, dp, ctypes.c_int]
        lib.cpu_ref_solve_batch_repeat.restype = ctypes.c_double
        lib.cpu_ref_solve_batch_repeat.argtypes = [ctypes.c_int] * 4 + [ip, ctypes.c_longlong, dp, dp, dp, dp, dp,
                                                   ctypes.c_int, ctypes.c_int]
        lib.cpu_ref_generate.restype = None
        lib.cpu_ref_generate.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_ulonglong] + \
            [ctypes.c_double] * 4 + [dp, dp]
        lib.cpu_ref_hardware_threads.restype = ctypes.c_int
        _lib = lib
    return _lib


def solve_batch(n, deriv, masks, times, d_fixed, nthreads=1, want_free=True, want_cost=True):
    """AoS in (times [B][K], d_fixed [B][D][n_fixed]) -> coeffs [B][K][D][N], d_free, cost, seconds."""
    lib = load()
    dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
    times = np.ascontiguousarray(times, dtype=np.float64)
    d_fixed = np.ascontiguousarray(d_fixed, dtype=np.float64)
    bsz, k = times.shape
    dim = d_fixed.shape[1]
    h = n // 2
    nfree = (k + 1) * h - sum(bin(m).count("1") for m in masks)
    m = np.array(masks, dtype=np.int32)
    co = np.empty((bsz, k, dim, n))
    fr = np.empty((bsz, dim, nfree)) if want_free else None
    cost = np.empty(bsz) if want_cost else None
    p = lambda a: None if a is None else a.ctypes.data_as(dp)
    secs = lib.cpu_ref_solve_batch(n, deriv, k, dim, m.ctypes.data_as(ip), bsz, p(times), p(d_fixed), p(co), p(fr),
                                   p(cost), nthreads)
    return co, fr, cost, secs


def generate(bsz, k, dim, seed0, box=10.0, v_max=3.0, a_max=5.0, magic=6.5):
    """Bit-exact createRandomVertices / nfabian through real libstdc++ <random>: positions [B][K+1][D], times [B][K]."""
    lib = load()
    dp = ctypes.POINTER(ctypes.c_double)
    pos = np.empty((bsz, k + 1, dim))
    times = np.empty((bsz, k))
    lib.cpu_ref_generate(k, dim, bsz, seed0, box, v_max, a_max, magic, pos.ctypes.data_as(dp), times.ctypes.data_as(dp))
    return pos, times


def timed_baseline(n, deriv, masks, times, d_fixed, target_seconds=12.0):
    """bench.py cpu_baseline: reference-algorithm restatement on all host cores over a bounded sample.
    One calibration pass over the sample, then ONE timed call in which every thread re-solves its slice `repeat`
    times (sized from the calibration for ~target_seconds; a single thread spawn, so start-up cost is amortised)."""
    lib = load()
    dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
    from oracle import effective_cpus
    cores = min(max(1, lib.cpu_ref_hardware_threads()), effective_cpus())
    times = np.ascontiguousarray(times, dtype=np.float64)
    d_fixed = np.ascontiguousarray(d_fixed, dtype=np.float64)
    bsz, k = times.shape
    dim = d_fixed.shape[1]
    m = np.array(masks, dtype=np.int32)
    co = np.empty((bsz, k, dim, n))

    def run(nthreads, count, repeat):
        return lib.cpu_ref_solve_batch_repeat(n, deriv, k, dim, m.ctypes.data_as(ip), count, times.ctypes.data_as(dp),
                                              d_fixed.ctypes.data_as(dp), co.ctypes.data_as(dp), None, None, nthreads,
                                              repeat)

    s_cal = run(cores, bsz, 1)
    repeat = int(max(1, min(10_000, target_seconds / max(s_cal, 1e-6))))
    s_all = run(cores, bsz, repeat)
    probe = min(bsz, 2000)
    s_one = run(1, probe, 1)
    return {"value": bsz * repeat / s_all, "unit": "trajectories/s", "cores": cores, "kind": "port",
            "single_thread_value": probe / s_one,
            "sample": f"{repeat} x {bsz} trajectories of the bench workload (setupFromVertices + solveLinear per "
                      f"trajectory), C++17 -O3 -march=x86-64-v3 restatement of the reference algorithm (real Eigen unavailable "
                      f"offline), std::thread over {cores} host threads, {s_all:.1f} s"}
