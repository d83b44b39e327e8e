// C entries around the REFERENCE's own solveLinear() path, compiled from the reference sources where they lie
// (oracle/Makefile target _ref/libmtg_ref.so):
//   include/mav_trajectory_generation/polynomial_optimization_linear.h + impl/polynomial_optimization_linear_impl.h,
//   src/{polynomial,vertex,segment,trajectory,motion_defines}.cpp, src/rpoly/rpoly_ak1.cpp
// against the container stand-ins in oracle/ref_shim/ (Eigen and glog are un-vendored dependencies of the reference,
// absent from this image -- see ref_shim/mini_eigen.h for exactly what is and is not Eigen's).  Every matrix
// construction, the constraint ordering and the call sequence executed here are the reference's own code.
// TEST INFRASTRUCTURE ONLY: parity anchor for tests/ and the "reference" CPU baseline of bench.py; never linked
// into the product library.
#include <mav_trajectory_generation/polynomial_optimization_linear.h>
#include <mav_trajectory_generation/trajectory.h>
#include <mav_trajectory_generation/vertex.h>

#include <chrono>
#include <cstring>
#include <thread>
#include <vector>

namespace mtg = mav_trajectory_generation;

namespace {

int popcount(int m) { return __builtin_popcount((unsigned)m); }

// Vertex::Vector from the C-ABI layout: masks[v] bit p = derivative p fixed at vertex v; d_fixed [D][n_fixed] ordered
// by (vertex, derivative) -- the reference's own fixed_constraints_compact_ order (polynomial_optimization_linear.h:288-295).
mtg::Vertex::Vector make_vertices(int h, int k, int dim, const int* masks, const double* d_fixed, int n_fixed) {
  mtg::Vertex::Vector vertices(k + 1, mtg::Vertex(dim));
  int col = 0;
  for (int v = 0; v <= k; ++v)
    for (int p = 0; p < h; ++p)
      if ((masks[v] >> p) & 1) {
        Eigen::VectorXd value(dim);
        for (int d = 0; d < dim; ++d) value[d] = d_fixed[(size_t)d * n_fixed + col];
        vertices[v].addConstraint(p, value);
        ++col;
      }
  return vertices;
}

template <int N>
void solve_one(int deriv, int k, int dim, const int* masks, const double* times, const double* d_fixed, int n_fixed,
               int n_free, double* coeffs, double* d_free, double* cost, const double* d_free_in) {
  const mtg::Vertex::Vector vertices = make_vertices(N / 2, k, dim, masks, d_fixed, n_fixed);
  const std::vector<double> segment_times(times, times + k);
  // the sequence timed by the reference's own benchmark (src/polynomial_timing_evaluation.cpp:104-110)
  mtg::PolynomialOptimization<N> opt(dim);
  opt.setupFromVertices(vertices, segment_times, deriv);
  if (d_free_in == nullptr) {
    opt.solveLinear();
  } else {  // setFreeConstraints path (impl/polynomial_optimization_linear_impl.h:500-508)
    std::vector<Eigen::VectorXd> free_constraints(dim, Eigen::VectorXd(n_free));
    for (int d = 0; d < dim; ++d)
      for (int j = 0; j < n_free; ++j) free_constraints[d][j] = d_free_in[(size_t)d * n_free + j];
    opt.setFreeConstraints(free_constraints);
  }
  if (coeffs != nullptr) {
    mtg::Segment::Vector segments;
    opt.getSegments(&segments);
    for (int s = 0; s < k; ++s)
      for (int d = 0; d < dim; ++d) {
        const Eigen::VectorXd c = segments[s][d].getCoefficients(0);
        for (int j = 0; j < N; ++j) coeffs[((size_t)s * dim + d) * N + j] = c[j];
      }
  }
  if (d_free != nullptr) {
    std::vector<Eigen::VectorXd> free_constraints;
    opt.getFreeConstraints(&free_constraints);
    for (int d = 0; d < dim; ++d)
      for (int j = 0; j < n_free; ++j) d_free[(size_t)d * n_free + j] = n_free ? free_constraints[d][j] : 0.0;
  }
  if (cost != nullptr) *cost = opt.computeCost();
}

typedef void (*solve_fn)(int, int, int, const int*, const double*, const double*, int, int, double*, double*, double*,
                         const double*);
solve_fn pick(int n) {
  switch (n) {
    case 2: return &solve_one<2>;
    case 4: return &solve_one<4>;
    case 6: return &solve_one<6>;
    case 8: return &solve_one<8>;
    case 10: return &solve_one<10>;
    case 12: return &solve_one<12>;
    default: return nullptr;
  }
}

mtg::Trajectory make_trajectory(int n, int k, int dim, const double* coeffs, const double* times) {
  mtg::Segment::Vector segments;
  for (int s = 0; s < k; ++s) {
    mtg::Segment seg(n, dim);
    for (int d = 0; d < dim; ++d) {
      Eigen::VectorXd c(n);
      for (int j = 0; j < n; ++j) c[j] = coeffs[((size_t)s * dim + d) * n + j];
      seg[d] = mtg::Polynomial(n, c);
    }
    seg.setTime(times[s]);
    segments.push_back(seg);
  }
  mtg::Trajectory trajectory;
  trajectory.setSegments(segments);
  return trajectory;
}

void read_trajectory(const mtg::Trajectory& trajectory, double* coeffs, double* times) {
  const mtg::Segment::Vector& segments = trajectory.segments();
  const int n = trajectory.N(), dim = trajectory.D();
  for (size_t s = 0; s < segments.size(); ++s) {
    times[s] = segments[s].getTime();
    for (int d = 0; d < dim; ++d) {
      const Eigen::VectorXd c = segments[s][d].getCoefficients(0);
      for (int j = 0; j < n; ++j) coeffs[((size_t)s * dim + d) * n + j] = c[j];
    }
  }
}

}  // namespace

extern "C" {

int mtg_ref_hardware_threads() { return (int)std::thread::hardware_concurrency(); }

// times [B][K], d_fixed [B][D][n_fixed] -> coeffs [B][K][D][N], d_free [B][D][n_free] (opt), cost [B] (opt);
// d_free_in (opt) [B][D][n_free] switches to the setFreeConstraints path.  Every thread re-solves its slice `repeat`
// times.  Returns wall seconds, < 0 on bad arguments.
double mtg_ref_solve_batch(int n, int deriv, int k, int dim, const int* masks, long long bsz, const double* times,
                           const double* d_fixed, double* coeffs, double* d_free, double* cost,
                           const double* d_free_in, int nthreads, int repeat) {
  solve_fn fn = pick(n);
  if (fn == nullptr || k < 1 || dim < 1) return -1.0;
  int n_fixed = 0;
  for (int v = 0; v <= k; ++v) n_fixed += popcount(masks[v] & ((1 << (n / 2)) - 1));
  const int n_free = (k + 1) * (n / 2) - n_fixed;
  const size_t cs = (size_t)k * dim * n, fs = (size_t)dim * n_fixed, ps = (size_t)dim * n_free;
  auto work = [&](long long b0, long long b1) {
    for (int r = 0; r < repeat; ++r)
      for (long long b = b0; b < b1; ++b)
        fn(deriv, k, dim, masks, times + b * k, d_fixed + b * fs, n_fixed, n_free, coeffs ? coeffs + b * cs : nullptr,
           d_free ? d_free + b * ps : nullptr, cost ? cost + b : nullptr, d_free_in ? d_free_in + b * ps : nullptr);
  };
  const auto t0 = std::chrono::steady_clock::now();
  if (nthreads <= 1) {
    work(0, bsz);
  } else {
    std::vector<std::thread> pool;
    for (int t = 0; t < nthreads; ++t) pool.emplace_back(work, bsz * t / nthreads, bsz * (t + 1) / nthreads);
    for (auto& th : pool) th.join();
  }
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// The three static builders (impl/polynomial_optimization_linear_impl.h:112-121, :143-179, :568-583), column-major N x N.
int mtg_ref_segment_matrices(int n, int deriv, double t, double* a, double* a_inv, double* q) {
#define MTG_REF_CASE(NN)                                                                     \
  case NN: {                                                                                 \
    typedef mtg::PolynomialOptimization<NN> Opt;                                             \
    Opt::SquareMatrix A, Ai, Q;                                                              \
    Opt::setupMappingMatrix(t, &A);                                                          \
    Opt::invertMappingMatrix(A, &Ai);                                                        \
    Opt::computeQuadraticCostJacobian(deriv, t, &Q);                                         \
    std::memcpy(a, A.data(), sizeof(double) * NN * NN);                                      \
    std::memcpy(a_inv, Ai.data(), sizeof(double) * NN * NN);                                 \
    std::memcpy(q, Q.data(), sizeof(double) * NN * NN);                                      \
    return 0;                                                                                \
  }
  switch (n) {
    MTG_REF_CASE(2)
    MTG_REF_CASE(4)
    MTG_REF_CASE(6)
    MTG_REF_CASE(8)
    MTG_REF_CASE(10)
    MTG_REF_CASE(12)
    default: return -1;
  }
#undef MTG_REF_CASE
}

// Dense M (n_all x (n_fixed + n_free)) and R of one problem (getM / getR), row-major out; returns n_all, sizes via out.
int mtg_ref_m_and_r(int n, int deriv, int k, int dim, const int* masks, const double* times, const double* d_fixed,
                    double* m_out, double* r_out, int* n_fixed_out, int* n_free_out) {
  if (n != 10 && n != 8 && n != 12) return -1;
  int n_fixed = 0;
  for (int v = 0; v <= k; ++v) n_fixed += popcount(masks[v] & ((1 << (n / 2)) - 1));
  const mtg::Vertex::Vector vertices = make_vertices(n / 2, k, dim, masks, d_fixed, n_fixed);
  const std::vector<double> segment_times(times, times + k);
  Eigen::MatrixXd M, R;
  size_t nf, np;
#define MTG_REF_CASE(NN)                                   \
  if (n == NN) {                                           \
    mtg::PolynomialOptimization<NN> opt(dim);              \
    opt.setupFromVertices(vertices, segment_times, deriv); \
    opt.getM(&M);                                          \
    opt.getR(&R);                                          \
    nf = opt.getNumberFixedConstraints();                  \
    np = opt.getNumberFreeConstraints();                   \
  }
  MTG_REF_CASE(8) MTG_REF_CASE(10) MTG_REF_CASE(12)
#undef MTG_REF_CASE
  for (Eigen::Index r = 0; r < M.rows(); ++r)
    for (Eigen::Index c = 0; c < M.cols(); ++c) m_out[r * M.cols() + c] = M(r, c);
  for (Eigen::Index r = 0; r < R.rows(); ++r)
    for (Eigen::Index c = 0; c < R.cols(); ++c) r_out[r * R.cols() + c] = R(r, c);
  *n_fixed_out = (int)nf;
  *n_free_out = (int)np;
  return (int)M.rows();
}

// createRandomVertices (src/vertex.cpp:27-82) + estimateSegmentTimesNfabian (:255-272): positions [K+1][D], times [K].
void mtg_ref_random_vertices(int max_derivative, int k, int dim, double lo, double hi, unsigned long long seed,
                             double v_max, double a_max, double magic, double* positions, double* times) {
  const mtg::Vertex::Vector vertices = mtg::createRandomVertices(
      max_derivative, k, Eigen::VectorXd::Constant(dim, lo), Eigen::VectorXd::Constant(dim, hi), seed);
  for (int v = 0; v <= k; ++v) {
    Eigen::VectorXd p;
    vertices[v].getConstraint(mtg::derivative_order::POSITION, &p);
    for (int d = 0; d < dim; ++d) positions[v * dim + d] = p[d];
  }
  const std::vector<double> t = mtg::estimateSegmentTimesNfabian(vertices, v_max, a_max, magic);
  for (int s = 0; s < k; ++s) times[s] = t[s];
}

// Trajectory::evaluateRange (src/trajectory.cpp:81-141) for one trajectory: out [count][D], sample_times [count].
int mtg_ref_evaluate_range(int n, int k, int dim, const double* coeffs, const double* times, double t_start,
                           double t_end, double dt, int derivative, double* out, double* sample_times, int capacity) {
  const mtg::Trajectory trajectory = make_trajectory(n, k, dim, coeffs, times);
  std::vector<Eigen::VectorXd> result;
  std::vector<double> sampling_times;
  trajectory.evaluateRange(t_start, t_end, dt, derivative, &result, &sampling_times);
  const int count = (int)std::min<size_t>(result.size(), (size_t)capacity);
  for (int i = 0; i < count; ++i) {
    for (int d = 0; d < dim; ++d) out[i * dim + d] = result[i][d];
    if (sample_times) sample_times[i] = sampling_times[i];
  }
  return (int)result.size();
}

// Trajectory::evaluate (src/trajectory.cpp:48-79) at arbitrary times: out [count][D].
void mtg_ref_evaluate(int n, int k, int dim, const double* coeffs, const double* times, const double* t, int count,
                      int derivative, double* out) {
  const mtg::Trajectory trajectory = make_trajectory(n, k, dim, coeffs, times);
  for (int i = 0; i < count; ++i) {
    const Eigen::VectorXd v = trajectory.evaluate(t[i], derivative);
    for (int d = 0; d < dim; ++d) out[i * dim + d] = v[d];
  }
}

// Trajectory::computeMinMaxMagnitude over all dimensions (src/trajectory.cpp:190-227):
// out = {min.time, min.value, min.segment_idx, max.time, max.value, max.segment_idx}; per_segment (opt) [K][4] =
// {min.time, min.value, max.time, max.value} from Segment::computeMinMaxMagnitudeCandidates + select (segment.cpp:83-184).
int mtg_ref_minmax_magnitude(int n, int k, int dim, const double* coeffs, const double* times, int derivative,
                             double* out, double* per_segment) {
  const mtg::Trajectory trajectory = make_trajectory(n, k, dim, coeffs, times);
  std::vector<int> dimensions(dim);
  for (int d = 0; d < dim; ++d) dimensions[d] = d;
  mtg::Extremum mn, mx;
  const bool ok = trajectory.computeMinMaxMagnitude(derivative, dimensions, &mn, &mx);
  out[0] = mn.time; out[1] = mn.value; out[2] = mn.segment_idx;
  out[3] = mx.time; out[4] = mx.value; out[5] = mx.segment_idx;
  if (per_segment != nullptr) {
    const mtg::Segment::Vector& segments = trajectory.segments();
    for (int s = 0; s < k; ++s) {
      std::vector<mtg::Extremum> candidates;
      segments[s].computeMinMaxMagnitudeCandidates(derivative, 0.0, segments[s].getTime(), dimensions, &candidates);
      mtg::Extremum smn, smx;
      segments[s].selectMinMaxMagnitudeFromCandidates(derivative, 0.0, segments[s].getTime(), dimensions, candidates,
                                                      &smn, &smx);
      per_segment[s * 4 + 0] = smn.time; per_segment[s * 4 + 1] = smn.value;
      per_segment[s * 4 + 2] = smx.time; per_segment[s * 4 + 3] = smx.value;
    }
  }
  return ok ? 0 : -1;
}

// Trajectory::scaleSegmentTimesToMeetConstraints (src/trajectory.cpp:385-429), in place; returns within_range.
int mtg_ref_scale_segment_times(int n, int k, int dim, double* coeffs, double* times, double v_max, double a_max) {
  mtg::Trajectory trajectory = make_trajectory(n, k, dim, coeffs, times);
  const bool ok = trajectory.scaleSegmentTimesToMeetConstraints(v_max, a_max);
  read_trajectory(trajectory, coeffs, times);
  return ok ? 1 : 0;
}

}  // extern "C"
