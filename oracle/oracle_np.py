"""CPU ORACLE (test infrastructure, NOT product code) — numpy float64 restatement of
mav_trajectory_generation's PolynomialOptimization<N>::solveLinear() hot path.

PARITY STATUS: PINNED against the reference itself run in the build container.  The
reference's own PolynomialOptimization<N> code is compiled from /root/reference where it
lies into oracle/_ref/libmtg_ref.so (oracle/Makefile `ref`; Eigen and glog -- un-vendored,
un-pinned dependencies fetched by install/mav_trajectory_generation_https.rosinstall:1-6,
absent offline -- are replaced by the container stand-ins in oracle/ref_shim/, whose
header says exactly what is not Eigen's: eager dense containers; inverse() and SparseQR
follow Eigen's algorithm CLASSES -- cofactor formulas up to 4x4, partial-pivoting LU above,
left-looking column Householder QR in a minimum-degree column order with Eigen's pivot
threshold -- not its exact operation order; `make -C oracle ref_eigen EIGEN_DIR=...` builds
the same sources against a real Eigen on a box that has one).  tests/test_reference_build.py checks this restatement against
that library step by step (A, Q bit/ulp-equal; M identical; A^-1, R, d_P, coefficients,
cost to round-off x cond: <= 7e-11 norm-wise for N = 10 snap) and against its committed
outputs (tests/golden/reference_solve_linear.npz).  It is additionally pinned on the
reference's golden vector (test_polynomial_optimization.cpp:777-780, n_free == 0) and its
property tests (AMatrixInversion :731-741, ConstraintPacking :505-564, checkPath
:113-174); oracle/oracle_mp.py (mpmath, 50 digits) arbitrates below the float64
evaluation error of the reference's own formulas (1e-11 for N = 10 .. 1e-8 for N = 12).
What stays unpinned in THIS image: real Eigen's last-bits behaviour (its LU / SparseQR
operation order), which no reference test constrains either; tests/test_reference_build.py::
test_real_eigen_build_vs_stand_in pins it wherever _ref/libmtg_ref_eigen.so exists.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Every function follows the reference step by step and cites file:line, with
  LIN   = mav_trajectory_generation/include/mav_trajectory_generation/impl/polynomial_optimization_linear_impl.h
  LINH  = .../include/mav_trajectory_generation/polynomial_optimization_linear.h
  POLYH = .../include/mav_trajectory_generation/polynomial.h
  POLYC = mav_trajectory_generation/src/polynomial.cpp
  VERT  = mav_trajectory_generation/src/vertex.cpp
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence

import numpy as np

K_MAX_N = 12                        # POLYH:44
K_MAX_CONVOLUTION_SIZE = 2 * K_MAX_N - 2   # POLYH:47
DBL_EPS = float(np.finfo(np.float64).eps)


# --------------------------------------------------------------------------- basis
def compute_base_coefficients(n: int) -> np.ndarray:
    """POLYC:145-160.  base(n,i) = i*(i-1)*...*(i-n+1)."""
    bc = np.zeros((n, n))
    bc[0, :] = 1.0
    deg = n - 1
    order = deg
    for d in range(1, n):
        for i in range(deg - order, n):
            bc[d, i] = (order - deg + i) * bc[d - 1, i]
        order -= 1
    return bc


BASE_COEFFICIENTS = compute_base_coefficients(K_MAX_CONVOLUTION_SIZE)  # POLYC:213-214


def base_coeffs_with_time(n: int, derivative: int, t: float) -> np.ndarray:
    """POLYH:201-219 (including the |t| < eps early-out and repeated-multiply powers)."""
    assert 0 <= derivative < n
    c = np.zeros(n)
    c[derivative] = BASE_COEFFICIENTS[derivative, derivative]
    if abs(t) < DBL_EPS:
        return c
    t_power = t
    for j in range(derivative + 1, n):
        c[j] = BASE_COEFFICIENTS[derivative, j] * t_power
        t_power = t_power * t
    return c


# --------------------------------------------------------------------------- per-segment matrices
def setup_mapping_matrix(n: int, segment_time: float) -> np.ndarray:
    """LIN:112-121.  A = [A(t=0); A(t=T)]."""
    a = np.zeros((n, n))
    h = n // 2
    for i in range(h):
        a[i, :] = base_coeffs_with_time(n, i, 0.0)
        a[i + h, :] = base_coeffs_with_time(n, i, segment_time)
    return a


def invert_mapping_matrix(a: np.ndarray) -> np.ndarray:
    """LIN:143-179.  Schur-complement inverse; the h x h block uses a dense LU inverse
    (Eigen fixed-size .inverse(), LIN:170-171) -> numpy/LAPACK LU inverse here."""
    n = a.shape[0]
    h = n // 2
    a_diag = np.diag(a[:h, :h]).copy()
    a_inv = np.diag(1.0 / a_diag)
    c = a[h:, :h]
    d_inv = np.linalg.inv(a[h:, h:])
    out = np.zeros((n, n))
    out[:h, :h] = a_inv
    out[h:, :h] = -d_inv @ c @ a_inv
    out[h:, h:] = d_inv
    return out


def compute_quadratic_cost_jacobian(n: int, derivative: int, t: float) -> np.ndarray:
    """LIN:568-583 (pow-based)."""
    assert derivative < n
    q = np.zeros((n, n))
    for col in range(n - derivative):
        for row in range(n - derivative):
            exponent = (n - 1 - derivative) * 2 + 1 - row - col
            q[n - 1 - row, n - 1 - col] = (
                BASE_COEFFICIENTS[derivative, n - 1 - row]
                * BASE_COEFFICIENTS[derivative, n - 1 - col]
                * math.pow(t, exponent) * 2.0 / exponent)
    return q


# --------------------------------------------------------------------------- Vertex (VERT / vertex.h)
class Vertex:
    """vertex.h:42-112.  constraints: derivative order -> np.ndarray(D)."""

    def __init__(self, dimension: int):
        self.D = int(dimension)
        self.constraints: Dict[int, np.ndarray] = {}

    def add_constraint(self, derivative_order: int, value) -> None:      # VERT:130-134
        v = np.asarray(value, dtype=np.float64)
        if v.ndim == 0:
            v = np.full(self.D, float(v))
        assert v.shape == (self.D,)
        self.constraints[int(derivative_order)] = v.copy()

    def make_start_or_end(self, value, up_to_derivative: int) -> None:   # VERT:147-153
        self.add_constraint(0, value)
        for i in range(1, up_to_derivative + 1):
            self.constraints[i] = np.zeros(self.D)

    def get_constraint(self, derivative_order: int):                    # VERT:155-163
        return self.constraints.get(int(derivative_order))

    def copy(self) -> "Vertex":
        v = Vertex(self.D)
        v.constraints = {k: x.copy() for k, x in self.constraints.items()}
        return v


# --------------------------------------------------------------------------- the optimiser
class PolynomialOptimization:
    """Literal restatement of PolynomialOptimization<N> (LINH:45-284, LIN)."""

    def __init__(self, n_coeffs: int, dimension: int):
        assert n_coeffs % 2 == 0                       # LINH:47
        self.N = int(n_coeffs)
        self.dimension = int(dimension)
        self.k_highest_derivative_to_optimize = self.N // 2 - 1   # LINH:51
        self.derivative_to_optimize = -1
        self.n_vertices = self.n_segments = 0
        self.n_all = self.n_fixed = self.n_free = 0
        self.fixed_constraints_compact = [np.zeros(0) for _ in range(self.dimension)]
        self.free_constraints_compact = [np.zeros(0) for _ in range(self.dimension)]
        self.segments = None    # [K][D][N]

    # LIN:57-109
    def setup_from_vertices(self, vertices: Sequence[Vertex], times: Sequence[float],
                            derivative_to_optimize: int | None = None) -> bool:
        if derivative_to_optimize is None:
            derivative_to_optimize = self.k_highest_derivative_to_optimize
        assert 0 <= derivative_to_optimize <= self.k_highest_derivative_to_optimize  # LIN:60
        self.derivative_to_optimize = derivative_to_optimize
        self.vertices = [v.copy() for v in vertices]
        self.segment_times = [float(t) for t in times]
        self.n_vertices = len(vertices)
        self.n_segments = self.n_vertices - 1
        assert self.n_vertices == len(times) + 1       # LIN:76
        # LIN:84-105: silently drop over-order constraints.
        for v in self.vertices:
            for k in list(v.constraints.keys()):
                if k > self.k_highest_derivative_to_optimize:
                    del v.constraints[k]
        self.update_segment_times(times)
        self._setup_constraint_reordering_matrix()
        return True

    # LIN:286-305
    def update_segment_times(self, segment_times: Sequence[float]) -> None:
        assert len(segment_times) == self.n_segments   # LIN:289
        self.segment_times = [float(t) for t in segment_times]
        self.cost_matrices = []
        self.inverse_mapping_matrices = []
        for t in self.segment_times:
            assert t > 0                                # LIN:297
            self.cost_matrices.append(
                compute_quadratic_cost_jacobian(self.N, self.derivative_to_optimize, t))
            a = setup_mapping_matrix(self.N, t)
            self.inverse_mapping_matrices.append(invert_mapping_matrix(a))

    # LIN:182-260
    def _setup_constraint_reordering_matrix(self) -> None:
        h = self.N // 2
        all_constraints = []        # (vertex_idx, constraint_idx)
        fixed = {}                  # std::set ordered by (vertex_idx, constraint_idx), LINH:288-295
        free = set()
        for vertex_idx, vertex in enumerate(self.vertices):
            occ = 1 if vertex_idx in (0, self.n_segments) else 2      # LIN:202-204
            for _ in range(occ):
                for constraint_idx in range(h):
                    key = (vertex_idx, constraint_idx)
                    val = vertex.get_constraint(constraint_idx)
                    all_constraints.append(key)
                    if val is not None:
                        fixed[key] = val
                    else:
                        free.add(key)
        fixed_keys = sorted(fixed.keys())
        free_keys = sorted(free)
        self.n_all = len(all_constraints)
        self.n_fixed = len(fixed_keys)
        self.n_free = len(free_keys)
        m = np.zeros((self.n_all, self.n_fixed + self.n_free))
        self.fixed_constraints_compact = [np.zeros(self.n_fixed) for _ in range(self.dimension)]
        for row, ca in enumerate(all_constraints):
            for col, cf in enumerate(fixed_keys):
                if ca == cf:
                    m[row, col] = 1.0
                    for d in range(self.dimension):
                        self.fixed_constraints_compact[d][col] = fixed[cf][d]
            for col, cp in enumerate(free_keys):
                if ca == cp:
                    m[row, self.n_fixed + col] = 1.0
        self.constraint_reordering = m
        self.fixed_keys = fixed_keys
        self.free_keys = free_keys

    # LIN:308-336
    def construct_r(self) -> np.ndarray:
        n = self.N
        big = np.zeros((n * self.n_segments, n * self.n_segments))
        for i in range(self.n_segments):
            ai = self.inverse_mapping_matrices[i]
            q = self.cost_matrices[i]
            hmat = ai.T @ q @ ai                       # LIN:318
            big[i * n:(i + 1) * n, i * n:(i + 1) * n] = hmat
        m = self.constraint_reordering
        return m.T @ big @ m                           # LIN:334-335

    # LIN:339-379
    def solve_linear(self) -> bool:
        assert 0 <= self.derivative_to_optimize <= self.k_highest_derivative_to_optimize
        if self.n_free == 0:                           # LIN:343-349
            self.free_constraints_compact = [np.zeros(0) for _ in range(self.dimension)]
            self._update_segments_from_compact_constraints()
            return True
        r = self.construct_r()
        nf = self.n_fixed
        rpf = r[nf:, :nf]                              # LIN:360-361
        rpp = r[nf:, nf:]                              # LIN:362-364
        # LIN:365-367: Eigen::SparseQR<COLAMD>; here dense Householder QR (LAPACK).
        qmat, rmat = np.linalg.qr(rpp)
        self.free_constraints_compact = []
        for d in range(self.dimension):
            df = -rpf @ self.fixed_constraints_compact[d]          # LIN:371-372
            self.free_constraints_compact.append(
                np.linalg.solve(rmat, qmat.T @ df))                # LIN:373-374
        self._update_segments_from_compact_constraints()
        return True

    # LIN:263-283
    def _update_segments_from_compact_constraints(self) -> None:
        n = self.N
        seg = np.zeros((self.n_segments, self.dimension, n))
        for d in range(self.dimension):
            d_all = np.concatenate([self.fixed_constraints_compact[d],
                                    self.free_constraints_compact[d]])
            for i in range(self.n_segments):
                new_d = self.constraint_reordering[i * n:(i + 1) * n, :] @ d_all
                seg[i, d, :] = self.inverse_mapping_matrices[i] @ new_d
        self.segments = seg

    # LIN:500-508
    def set_free_constraints(self, free_constraints: Sequence[np.ndarray]) -> None:
        assert len(free_constraints) == self.dimension
        for v in free_constraints:
            assert len(v) == self.n_free
        self.free_constraints_compact = [np.asarray(v, dtype=np.float64).copy()
                                         for v in free_constraints]
        self._update_segments_from_compact_constraints()

    # LIN:124-140
    def compute_cost(self) -> float:
        cost = 0.0
        for i in range(self.n_segments):
            q = self.cost_matrices[i]
            for d in range(self.dimension):
                c = self.segments[i, d, :]
                cost += float(c @ q @ c)
        return 0.5 * cost

    # accessors LIN:511-565
    def get_a_inverse(self) -> np.ndarray:
        n = self.N
        out = np.zeros((n * self.n_segments, n * self.n_segments))
        for i in range(self.n_segments):
            out[i * n:(i + 1) * n, i * n:(i + 1) * n] = self.inverse_mapping_matrices[i]
        return out

    def get_a(self) -> np.ndarray:
        n = self.N
        out = np.zeros((n * self.n_segments, n * self.n_segments))
        for i in range(self.n_segments):
            out[i * n:(i + 1) * n, i * n:(i + 1) * n] = setup_mapping_matrix(n, self.segment_times[i])
        return out

    def get_m(self) -> np.ndarray:
        return self.constraint_reordering.copy()

    def get_m_pinv(self) -> np.ndarray:
        mp = self.constraint_reordering.T.copy()
        for r in range(mp.shape[0]):
            mp[r, :] = mp[r, :] / mp[r, :].sum()
        return mp

    def fixed_mask(self) -> List[int]:
        """bit k of entry v set <=> derivative k is fixed at vertex v (C-ABI plan descriptor)."""
        h = self.N // 2
        out = []
        for v in self.vertices:
            mk = 0
            for k in range(h):
                if v.get_constraint(k) is not None:
                    mk |= 1 << k
            out.append(mk)
        return out


# --------------------------------------------------------------------------- input generators (VERT)
class Mt19937:
    """std::mt19937 (libstdc++ <random>), seeded with a single 32-bit value."""

    def __init__(self, seed: int):
        self.mt = [0] * 624
        self.mt[0] = seed & 0xFFFFFFFF
        for i in range(1, 624):
            self.mt[i] = (1812433253 * (self.mt[i - 1] ^ (self.mt[i - 1] >> 30)) + i) & 0xFFFFFFFF
        self.idx = 624

    def _twist(self) -> None:
        mt = self.mt
        for i in range(624):
            y = (mt[i] & 0x80000000) | (mt[(i + 1) % 624] & 0x7FFFFFFF)
            mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ (0x9908B0DF if (y & 1) else 0)
        self.idx = 0

    def next_u32(self) -> int:
        if self.idx >= 624:
            self._twist()
        y = self.mt[self.idx]
        self.idx += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF

    def canonical(self) -> float:
        """std::generate_canonical<double,53>(mt19937): two 32-bit draws (libstdc++)."""
        lo = self.next_u32()
        hi = self.next_u32()
        s = (float(lo) + float(hi) * 4294967296.0) / 18446744073709551616.0
        if s >= 1.0:
            s = math.nextafter(1.0, 0.0)
        return s

    def uniform(self, a: float, b: float) -> float:
        """std::uniform_real_distribution<double>(a,b)(gen) (libstdc++)."""
        return self.canonical() * (b - a) + a


def create_random_vertices(maximum_derivative: int, n_segments: int, pos_min, pos_max,
                           seed: int = 0) -> List[Vertex]:
    """VERT:27-82.  One mt19937(seed); per-dimension uniform draws in order; re-draw the
    whole point until it is > 0.2 away from the previous one; ends makeStartOrEnd."""
    pos_min = np.asarray(pos_min, dtype=np.float64)
    pos_max = np.asarray(pos_max, dtype=np.float64)
    assert n_segments >= 1 and pos_min.shape == pos_max.shape
    assert np.linalg.norm(pos_max - pos_min) >= 0.2 and maximum_derivative > 0
    gen = Mt19937(seed)
    dim = pos_min.size
    min_distance = 0.2
    last = np.array([gen.uniform(pos_min[i], pos_max[i]) for i in range(dim)])
    vertices = [Vertex(dim)]
    vertices[0].make_start_or_end(last, maximum_derivative)
    for _ in range(1, n_segments + 1):
        while True:
            pos = np.array([gen.uniform(pos_min[d], pos_max[d]) for d in range(dim)])
            if np.linalg.norm(pos - last) > min_distance:
                break
        v = Vertex(dim)
        v.add_constraint(0, pos)
        vertices.append(v)
        last = pos
    vertices[-1].make_start_or_end(last, maximum_derivative)
    return vertices


def estimate_segment_times_nfabian(vertices: Sequence[Vertex], v_max: float, a_max: float,
                                   magic_fabian_constant: float = 6.5) -> List[float]:
    """VERT:255-272."""
    assert len(vertices) >= 2
    out = []
    for i in range(len(vertices) - 1):
        start = vertices[i].get_constraint(0)
        end = vertices[i + 1].get_constraint(0)
        distance = float(np.linalg.norm(end - start))
        t = distance / v_max * 2 * (1.0 + magic_fabian_constant * v_max / a_max
                                    * math.exp(-distance / v_max * 2))
        out.append(t)
    return out


def estimate_segment_times(vertices, v_max, a_max):
    """VERT:228-231 (default = nfabian)."""
    return estimate_segment_times_nfabian(vertices, v_max, a_max)


# --------------------------------------------------------------------------- batch helpers for tests/bench
def solve_batch(n_coeffs: int, derivative: int, fixed_mask: Sequence[int],
                times: np.ndarray, d_fixed: np.ndarray):
    """Run the literal path on a batch given in the C-ABI layout.

    times   [B][K]; d_fixed [B][D][n_fixed] ordered by (vertex, derivative) over fixed slots.
    returns coeffs [B][K][D][N], d_free [B][D][n_free], cost [B]
    """
    times = np.asarray(times, dtype=np.float64)
    d_fixed = np.asarray(d_fixed, dtype=np.float64)
    bsz, k = times.shape
    dim = d_fixed.shape[1]
    h = n_coeffs // 2
    keys = [(v, p) for v in range(k + 1) for p in range(h) if (fixed_mask[v] >> p) & 1]
    assert d_fixed.shape[2] == len(keys)
    coeffs = np.zeros((bsz, k, dim, n_coeffs))
    n_free = (k + 1) * h - len(keys)
    d_free = np.zeros((bsz, dim, n_free))
    cost = np.zeros(bsz)
    for b in range(bsz):
        verts = [Vertex(dim) for _ in range(k + 1)]
        for col, (v, p) in enumerate(keys):
            verts[v].add_constraint(p, d_fixed[b, :, col])
        opt = PolynomialOptimization(n_coeffs, dim)
        opt.setup_from_vertices(verts, times[b], derivative)
        opt.solve_linear()
        coeffs[b] = opt.segments
        for d in range(dim):
            d_free[b, d] = opt.free_constraints_compact[d]
        cost[b] = opt.compute_cost()
    return coeffs, d_free, cost


# --------------------------------------------------------------------------- Mellinger time gradient (next-step row N2)
K_OPTIMIZATION_TIME_LOWER_BOUND = 0.1    # polynomial_optimization_nonlinear.h: kOptimizationTimeLowerBound


def mellinger_cost_gradient(n_coeffs: int, derivative: int, fixed_mask: Sequence[int], times: np.ndarray, d_fixed: np.ndarray):
    """PolynomialOptimizationNonLinear<N>::getCostAndGradientMellinger (impl/polynomial_optimization_nonlinear_impl.h:287-364),
    literally: J_d of the given times, and per segment n the forward difference (J_d(T + h e_n - h/(m-1) sum_{i != n} e_i) - J_d) / h
    with h = 0.1 (:311), every perturbed time clamped from below by kOptimizationTimeLowerBound (:338-340); one segment: zero
    gradient (:295-302).  Pinned on the reference's own member run in this container (tests/golden/reference_mellinger.npz,
    tests/test_reference_build.py).  times [B][K], d_fixed [B][D][n_fixed] -> (cost [B], gradient [B][K])."""
    times = np.asarray(times, dtype=np.float64)
    bsz, k = times.shape
    cost, grad = np.zeros(bsz), np.zeros((bsz, k))
    for b in range(bsz):
        def cost_of(tt):
            return solve_batch(n_coeffs, derivative, fixed_mask, tt[None], d_fixed[b][None])[2][0]
        cost[b] = cost_of(times[b])
        if k == 1:
            continue
        increment_time = 0.1
        corr = increment_time / (k - 1.0)
        for s in range(k):
            bigger = times[b].copy()
            for i in range(k):
                bigger[i] += increment_time if i == s else -corr
            bigger = np.maximum(K_OPTIMIZATION_TIME_LOWER_BOUND, bigger)
            grad[b, s] = (cost_of(bigger) - cost[b]) / increment_time
    return cost, grad


# --------------------------------------------------------------------------- sampling (next-step row N3)
def polynomial_evaluate(coeffs: np.ndarray, t: float, derivative: int) -> float:
    """polynomial.h:137-149: Horner on base_coefficients_(derivative, j) * c_j, highest power first."""
    n = len(coeffs)
    if derivative >= n:
        return 0.0
    tmp = n - 1
    acc = BASE_COEFFICIENTS[derivative, tmp] * coeffs[tmp]
    for j in range(tmp - 1, derivative - 1, -1):
        acc *= t
        acc += BASE_COEFFICIENTS[derivative, j] * coeffs[j]
    return acc


def trajectory_evaluate(segments: np.ndarray, times: Sequence[float], t: float, derivative: int) -> np.ndarray:
    """src/trajectory.cpp:48-79 Trajectory::evaluate for segments [K][D][N] (t beyond the end: last segment, as the
    reference does for t == max time; the batched kernel clamps such t to the end time)."""
    accumulated = 0.0
    i = 0
    for i in range(len(times)):
        accumulated += times[i]
        if accumulated > t:
            break
    else:
        i = len(times) - 1
    accumulated -= times[i]
    local = min(t - accumulated, times[i])
    return np.array([polynomial_evaluate(segments[i, d], local, derivative) for d in range(segments.shape[1])])


def sample_batch(coeffs: np.ndarray, times: np.ndarray, t_start: float, dt: float, n_samples: int,
                 n_derivatives: int = 5):
    """Batched restatement of sampleTrajectoryInRange (src/trajectory_sampling.cpp:45-110) with closed-form sample
    times t_i = t_start + i*dt: out [B][S][ND][D], n_valid [B]."""
    bsz, k, dim, _ = coeffs.shape
    out = np.zeros((bsz, n_samples, n_derivatives, dim))
    n_valid = np.zeros(bsz, dtype=np.int32)
    for b in range(bsz):
        total = 0.0
        for i in range(k):
            total += times[b, i]
        for s in range(n_samples):
            t = t_start + dt * s
            if t <= total:
                n_valid[b] += 1
            for der in range(n_derivatives):
                out[b, s, der] = trajectory_evaluate(coeffs[b], times[b], t, der)
    return out, n_valid
