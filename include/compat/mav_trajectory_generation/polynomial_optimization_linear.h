// Compat veneer: PolynomialOptimization<N> with the reference's public hot-path API
// (reference: polynomial_optimization_linear.h:45-284) whose solveLinear() / setFreeConstraints() forward to the
// MI355X library through the C ABI (include/mtg_hip.h).  Host code only: no HIP headers, Eigen optional.
//
// Kept: every public member of the reference class -- constructor, setupFromVertices, updateSegmentTimes,
// solveLinear, getTrajectory, getSegments, getVertices, getSegmentTimes, get/setFreeConstraints,
// getFixedConstraints, computeCost, the static matrix helpers, the counters, the dense accessors getA /
// getAInverse / getM / getR / getMpinv and the magnitude-extremum helpers (host code; real roots by derivative
// recursion + bisection instead of the reference's Jenkins-Traub translation).  Only setupFromPositons, which the
// reference declares but never defines (LINH:79), is absent.  New: PolynomialOptimizationBatch<N>, the batched entry.
//
// Value semantics are preserved (the optimiser is copy-assigned in the wild, time_evaluation_node.cpp:357): the
// object owns only host data plus a shared, immutable plan handle; device scratch lives in a per-thread context.
#ifndef MAV_TRAJECTORY_GENERATION_POLYNOMIAL_OPTIMIZATION_LINEAR_H_
#define MAV_TRAJECTORY_GENERATION_POLYNOMIAL_OPTIMIZATION_LINEAR_H_
#include <atomic>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <memory>
#include <string>
#include <numeric>
#include <ostream>
#include <vector>

#include <map>

#include "../../mtg_hip.h"
#include "motion_defines.h"
#include "polynomial.h"
#include "extremum.h"
#include "segment.h"
#include "trajectory.h"
#include "vertex.h"

namespace mav_trajectory_generation {

namespace mtg_compat_detail {
// One library context per host thread, plus that thread's plans keyed by constraint structure: the reference's callers
// set the same structure up over and over (every nlopt objective call, every run of the timing benchmark), and a plan
// costs a device allocation + a synchronous table upload to create and a stream synchronisation to destroy.
// Lifetime: the context is reference-counted and every plan's deleter holds a reference, so an optimiser that was
// created (or copied) on one thread and outlives that thread keeps its context alive; solve calls with host pointers are
// synchronous and return their own status, so no per-thread status ever has to be collected from another thread's context.
struct ThreadContext {
  std::shared_ptr<mtg_context> ctx;
  std::map<std::vector<uint32_t>, std::shared_ptr<mtg_plan>> plans;
};
inline ThreadContext& thread_state() {
  static thread_local ThreadContext tc;
  return tc;
}
inline std::shared_ptr<mtg_context> context_ref() {
  ThreadContext& tc = thread_state();
  if (!tc.ctx) {
    mtg_context* raw = nullptr;
    const int rc = mtg_context_create(0, nullptr, &raw);
    CHECK(rc == MTG_OK) << "mtg_context_create: " << mtg_status_string(rc)
                        << " (a library context always owns a gfx950 device: batches have no CPU path; only single-trajectory "
                           "calls may run on the library's host build of the kernel code, see setSingleCallsOnDevice)";
    tc.ctx = std::shared_ptr<mtg_context>(raw, [](mtg_context* c) { mtg_context_destroy(c); });
  }
  return tc.ctx;
}
inline mtg_context* context() { return context_ref().get(); }
inline std::shared_ptr<mtg_plan> make_plan(int N, int D, int K, int derivative, const std::vector<uint32_t>& mask) {
  std::vector<uint32_t> key{(uint32_t)N, (uint32_t)D, (uint32_t)K, (uint32_t)derivative};
  key.insert(key.end(), mask.begin(), mask.end());
  std::shared_ptr<mtg_context> ctx = context_ref();
  ThreadContext& tc = thread_state();
  auto it = tc.plans.find(key);
  if (it != tc.plans.end()) return it->second;
  mtg_plan_desc desc{N, D, K, derivative, mask.data()};
  mtg_plan* p = nullptr;
  const int rc = mtg_plan_create(ctx.get(), &desc, &p);
  CHECK(rc == MTG_OK) << "mtg_plan_create: " << mtg_status_string(rc) << " " << mtg_last_error_string(ctx.get());
  std::shared_ptr<mtg_plan> sp(p, [ctx](mtg_plan* q) { mtg_plan_destroy(q); });   // the deleter keeps the context alive
  if (tc.plans.size() >= 256) tc.plans.clear();   // bounded; live optimisers keep their own references
  tc.plans.emplace(std::move(key), sp);
  return sp;
}
// Synchronises the context a PLAN lives on (not the calling thread's): a batch object may be used from, or outlive,
// another thread than the one that created it, and its launches, status flags and error text belong to the plan's context.
inline void check_sync(const mtg_plan* plan) {
  mtg_context* ctx = plan ? mtg_plan_context(plan) : context();
  const int rc = mtg_context_sync(ctx);
  // LIN:297 CHECK_GT(segment_time, 0) and friends surface here
  CHECK(rc == MTG_OK) << mtg_status_string(rc) << " " << mtg_last_error_string(ctx);
}
inline void check_sync() { check_sync(nullptr); }

// Where the single-trajectory calls of the reference API (solveLinear(), setFreeConstraints() on ONE optimiser object)
// run.  One trajectory per call is latency-bound (a launch plus a PCIe round trip: ~25 us), so by default they run on
// the library's HOST BUILD of the kernels' lane code on the calling thread (MTG_FLAG_HOST_BACKEND: the same functions the
// kernels are made of, ~2-8 us per call; product code, not the test oracle) -- what SURVEY 8(b) asks for the reference's
// nlopt-style callers.  A run-time switch, process-wide:
//   * environment MTG_COMPAT_SINGLE_CALLS = device | host (read once), or
//   * mav_trajectory_generation::setSingleCallsOnDevice(true / false),
//   * compile-time default `device` with -DMTG_COMPAT_SINGLE_CALLS_ON_DEVICE.
// Batched entries (PolynomialOptimizationBatch, solveLinearMixed) always run on the GPU.
inline std::atomic<int>& single_calls_on_device_state() {   // (atomic: optimiser threads read it while another may toggle it)
  static std::atomic<int> state{[] {
#if defined(MTG_COMPAT_SINGLE_CALLS_ON_DEVICE)
    int v = 1;
#else
    int v = 0;
#endif
    if (const char* e = std::getenv("MTG_COMPAT_SINGLE_CALLS")) {
      if (std::string(e) == "device") v = 1;
      else if (std::string(e) == "host") v = 0;
    }
    return v;
  }()};
  return state;
}
}  // namespace mtg_compat_detail

inline void setSingleCallsOnDevice(bool on_device) { mtg_compat_detail::single_calls_on_device_state().store(on_device ? 1 : 0, std::memory_order_relaxed); }
inline bool singleCallsOnDevice() { return mtg_compat_detail::single_calls_on_device_state().load(std::memory_order_relaxed) != 0; }

template <int _N = 10>
class PolynomialOptimization {
  static_assert(_N % 2 == 0, "The number of coefficients has to be even.");

 public:
  enum { N = _N };
  static constexpr int kHighestDerivativeToOptimize = N / 2 - 1;
  typedef Eigen::Matrix<double, N, N> SquareMatrix;
  typedef std::vector<SquareMatrix, Eigen::aligned_allocator<SquareMatrix>> SquareMatrixVector;

  explicit PolynomialOptimization(size_t dimension)
      : dimension_(dimension), derivative_to_optimize_(derivative_order::INVALID), n_vertices_(0), n_segments_(0),
        n_all_constraints_(0), n_fixed_constraints_(0), n_free_constraints_(0) {
    fixed_constraints_compact_.resize(dimension_);
    free_constraints_compact_.resize(dimension_);
  }

  bool setupFromVertices(const Vertex::Vector& vertices, const std::vector<double>& segment_times,
                         int derivative_to_optimize = kHighestDerivativeToOptimize) {
    CHECK(derivative_to_optimize >= 0 && derivative_to_optimize <= kHighestDerivativeToOptimize)
        << "You tried to optimize the " << derivative_to_optimize << "th derivative on a " << N << " coefficient polynomial.";
    CHECK(vertices.size() == segment_times.size() + 1) << "Size of times must be one less than positions.";
    derivative_to_optimize_ = derivative_to_optimize;
    vertices_ = vertices;
    n_vertices_ = vertices.size();
    n_segments_ = n_vertices_ - 1;
    segments_.assign(n_segments_, Segment(N, (int)dimension_));
    // constraints of order > N/2-1 are dropped (the reference warns and ignores them)
    for (Vertex& v : vertices_) {
      for (int k = N / 2; k <= 2 * Polynomial::kMaxN; ++k) v.removeConstraint(k);
    }
    // constraint structure: per-vertex masks, fixed values in (vertex, derivative) order
    mask_.assign(n_vertices_, 0u);
    n_fixed_constraints_ = 0;
    for (size_t v = 0; v < n_vertices_; ++v) {
      for (int p = 0; p < N / 2; ++p) {
        if (vertices_[v].hasConstraint(p)) { mask_[v] |= 1u << p; ++n_fixed_constraints_; }
      }
    }
    n_all_constraints_ = (size_t)N * n_segments_;
    n_free_constraints_ = n_vertices_ * (N / 2) - n_fixed_constraints_;
    for (size_t d = 0; d < dimension_; ++d) {
      fixed_constraints_compact_[d] = Eigen::VectorXd::Zero(n_fixed_constraints_);
      free_constraints_compact_[d] = Eigen::VectorXd::Zero(n_free_constraints_);
    }
    size_t col = 0;
    for (size_t v = 0; v < n_vertices_; ++v) {
      for (int p = 0; p < N / 2; ++p) {
        Eigen::VectorXd value;
        if (!vertices_[v].getConstraint(p, &value)) continue;
        for (size_t d = 0; d < dimension_; ++d) fixed_constraints_compact_[d][col] = value[d];
        ++col;
      }
    }
    plan_ = mtg_compat_detail::make_plan(N, (int)dimension_, (int)n_segments_, derivative_to_optimize_, mask_);
    updateSegmentTimes(segment_times);
    return true;
  }

  void updateSegmentTimes(const std::vector<double>& segment_times) {
    CHECK(segment_times.size() == n_segments_) << "Number of segment times does not match number of segments";
    for (double t : segment_times) CHECK_GT(t, 0) << "Segment times need to be greater than zero";
    segment_times_ = segment_times;
  }

  // Always returns true, like the reference (LIN:339-379).  A rank-deficient free-constraint system (under-constrained
  // problems: e.g. one segment with only the end positions fixed -- any cubic through them has zero snap) makes the
  // library's LDL^T sweep report MTG_ERR_SINGULAR; the reference's rank-revealing SparseQR (LIN:365-367) returns a BASIC
  // solution there, and so does solveLinearBasic() below (a column-pivoted Householder QR of the dense R_PP on the host:
  // such problems are tiny and rare).  The minimum cost is unique; the basic solution is not (free variables beyond the
  // numerical rank are zero, and WHICH ones depends on the column order: Eigen processes COLAMD order, this code pivots by
  // column norm), so coefficients may differ from the reference's while cost and all constraints agree.
  bool solveLinear() {
    CHECK(derivative_to_optimize_ >= 0 && derivative_to_optimize_ <= kHighestDerivativeToOptimize);
    const int rc = run(/*solve=*/true);
    if (rc == MTG_OK) return true;
    return solveLinearBasic();
  }

  // d_P = basic solution of R_PP d_P = -R_PF d_F (LIN:360-375 with a rank-revealing factorisation; the library's host code,
  // mtg_basic_solution_host: column-pivoted Householder QR of the dense R_PP, Eigen's SparseQR rank threshold), then the
  // setFreeConstraints path (LIN:263-283) for the coefficients.
  bool solveLinearBasic() {
    const size_t nf = n_fixed_constraints_, np = n_free_constraints_;
    if (np == 0) return run(/*solve=*/false) == MTG_OK;
    std::vector<double> d_fixed(dimension_ * nf), d_free(dimension_ * np);
    for (size_t d = 0; d < dimension_; ++d) for (size_t c = 0; c < nf; ++c) d_fixed[d * nf + c] = fixed_constraints_compact_[d][c];
    int32_t rank = 0;
    const int rc = mtg_basic_solution_host(plan_.get(), segment_times_.data(), d_fixed.data(), d_free.data(), &rank);
    CHECK(rc == MTG_OK) << mtg_status_string(rc);
    for (size_t d = 0; d < dimension_; ++d) for (size_t c = 0; c < np; ++c) free_constraints_compact_[d][c] = d_free[d * np + c];
    last_solve_rank_ = (size_t)rank;
    static std::atomic<bool> logged{false};
    if (!logged.exchange(true))
      LOG(WARNING) << "solveLinear(): rank-deficient free-constraint system (rank " << rank << " of " << np
                   << "): basic solution from a pivoted QR, like the reference's SparseQR but not in its column order "
                      "(getLastSolveRank() < getNumberFreeConstraints() tells; logged once)";
    return run(/*solve=*/false) == MTG_OK;
  }
  // numerical rank found by the last solveLinearBasic() (n_free when the last solveLinear() went through the device path)
  size_t getLastSolveRank() const { return last_solve_rank_; }

  void setFreeConstraints(const std::vector<Eigen::VectorXd>& free_constraints) {
    CHECK(free_constraints.size() == dimension_);
    for (const Eigen::VectorXd& v : free_constraints) CHECK(static_cast<size_t>(v.size()) == n_free_constraints_);
    free_constraints_compact_ = free_constraints;
    run(/*solve=*/false);
  }

  double computeCost() const {   // 0.5 * sum c^T Q c over segments and dimensions
    double cost = 0.0;
    for (size_t i = 0; i < n_segments_; ++i) {
      SquareMatrix Q;
      computeQuadraticCostJacobian(derivative_to_optimize_, segment_times_[i], &Q);
      for (size_t d = 0; d < dimension_; ++d) {
        const Eigen::VectorXd c = segments_[i][d].getCoefficients(0);
        for (int r = 0; r < N; ++r) for (int q = 0; q < N; ++q) cost += c[r] * Q(r, q) * c[q];
      }
    }
    return 0.5 * cost;
  }

  void getTrajectory(Trajectory* trajectory) const { CHECK_NOTNULL(trajectory); trajectory->setSegments(segments_); }

  // ---- extrema of |p^(derivative)| (post-solve helpers of the reference API; host code) ---------------------
  template <int Derivative>
  static bool computeSegmentMaximumMagnitudeCandidates(const Segment& segment, double t_start, double t_stop,
                                                       std::vector<double>* candidates) {
    return computeSegmentMaximumMagnitudeCandidates(Derivative, segment, t_start, t_stop, candidates);
  }
  static bool computeSegmentMaximumMagnitudeCandidates(int derivative, const Segment& segment, double t_start,
                                                       double t_stop, std::vector<double>* candidates) {
    CHECK(candidates != nullptr);
    CHECK(N - derivative - 1 > 0) << "N-Derivative-1 has to be greater 0";
    std::vector<int> dimensions(segment.D());
    std::iota(dimensions.begin(), dimensions.end(), 0);
    return segment.computeMinMaxMagnitudeCandidateTimes(derivative, t_start, t_stop, dimensions, candidates);
  }
  // Sampling variant (debugging aid in the reference): sign changes of d|.|/dt on a dt grid, plus both ends.
  template <int Derivative>
  static void computeSegmentMaximumMagnitudeCandidatesBySampling(const Segment& segment, double t_start, double t_stop,
                                                                 double sampling_interval, std::vector<double>* candidates) {
    CHECK_NOTNULL(candidates);
    candidates->push_back(t_start);
    double t_prev = t_start + sampling_interval;
    double n_prev = segment.evaluate(t_prev, Derivative).norm();
    double direction = n_prev - segment.evaluate(t_start, Derivative).norm();
    for (double t = t_start + 2 * sampling_interval; t <= t_stop; t += sampling_interval) {
      const double n_new = segment.evaluate(t, Derivative).norm();
      const double direction_new = n_new - n_prev;
      if (std::signbit(direction) != std::signbit(direction_new) && segment.evaluate(t_prev, Derivative + 1).norm() < 1e-2)
        candidates->push_back(t_prev);
      direction = direction_new;
      n_prev = n_new;
      t_prev = t;
    }
    if (candidates->back() != t_stop) candidates->push_back(t_stop);
  }
  template <int Derivative>
  Extremum computeMaximumOfMagnitude(std::vector<Extremum>* candidates) const {
    return computeMaximumOfMagnitude(Derivative, candidates);
  }
  Extremum computeMaximumOfMagnitude(int derivative, std::vector<Extremum>* candidates) const {
    if (candidates != nullptr) candidates->clear();
    Extremum best;
    int idx = 0;
    for (const Segment& s : segments_) {
      std::vector<double> times;
      computeSegmentMaximumMagnitudeCandidates(derivative, s, 0.0, s.getTime(), &times);
      times.push_back(0.0);
      for (double t : times) {
        const Extremum c(t, s.evaluate(t, derivative).norm(), idx);
        if (best < c) best = c;
        if (candidates != nullptr) candidates->push_back(c);
      }
      ++idx;
    }
    return best;
  }

  void getVertices(Vertex::Vector* vertices) const { CHECK_NOTNULL(vertices); *vertices = vertices_; }
  void getSegments(Segment::Vector* segments) const { CHECK_NOTNULL(segments); *segments = segments_; }
  void getSegmentTimes(std::vector<double>* t) const { CHECK(t != nullptr); *t = segment_times_; }
  void getFreeConstraints(std::vector<Eigen::VectorXd>* f) const { CHECK(f != nullptr); *f = free_constraints_compact_; }
  void getFixedConstraints(std::vector<Eigen::VectorXd>* f) const { CHECK(f != nullptr); *f = fixed_constraints_compact_; }

  size_t getDimension() const { return dimension_; }
  size_t getNumberSegments() const { return n_segments_; }
  size_t getNumberAllConstraints() const { return n_all_constraints_; }
  size_t getNumberFixedConstraints() const { return n_fixed_constraints_; }
  size_t getNumberFreeConstraints() const { return n_free_constraints_; }
  int getDerivativeToOptimize() const { return derivative_to_optimize_; }

  // ---- static matrix helpers (host; not on the GPU path, kept for API compatibility) ----------------------
  static void setupMappingMatrix(double segment_time, SquareMatrix* A) {   // A = [A(0); A(T)]
    for (int i = 0; i < N / 2; ++i) {
      const Eigen::VectorXd r0 = Polynomial::baseCoeffsWithTime(N, i, 0.0);
      const Eigen::VectorXd r1 = Polynomial::baseCoeffsWithTime(N, i, segment_time);
      for (int j = 0; j < N; ++j) { (*A)(i, j) = r0[j]; (*A)(i + N / 2, j) = r1[j]; }
    }
  }
  // Inverse of the block matrix [[diag, 0], [C, D]]: [[diag^-1, 0], [-D^-1 C diag^-1, D^-1]].
  // A(T)^-1 through the time-scaling identity the kernels use (csrc/mtg_lane.h): with tau = t / T the mapping factorises as
  // A(T) = S^-1 A(1) diag(T^i), S = diag(T^0 .. T^(h-1), T^0 .. T^(h-1)), hence A(T)^-1[i][j] = T^-i A(1)^-1[i][j] T^(j mod h).
  // A(1)^-1 is computed once per N in extended precision and rounded (its entries are integers and simple fractions), so the
  // result is good to an ulp or two for every T -- the reference's Schur-complement route with a numeric h x h inverse
  // (LIN:143-179) loses cond(A) ~ 1e7 .. 1e17 of it.  T is read off the matrix (row h is [1, T, T^2, ...]); anything that is not a
  // mapping matrix of a positive time goes through the generic elimination below.  Found wanting by the reference's own
  // AMatrixInversion test (TOPT:731-741, 1e-10 absolute) run against this header: the Gauss-Jordan version of rounds 1-4 was
  // 2.9e-10 off on the entry 420 of A(1)^-1.
  static void invertMappingMatrix(const SquareMatrix& A, SquareMatrix* Ai) {
    constexpr int h = N / 2;
    const double T = A(h, 1);
    if (T > 0.0 && A(h, 0) == 1.0) {
      static const UnitInverse unit;
      double tp[h], tn[N];
      tp[0] = 1.0;
      for (int k = 1; k < h; ++k) tp[k] = tp[k - 1] * T;
      tn[0] = 1.0;
      for (int i = 1; i < N; ++i) tn[i] = tn[i - 1] / T;
      for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) (*Ai)(i, j) = tn[i] * unit.v[i][j] * tp[j % h];
      return;
    }
    invertGeneric(A, Ai);
  }

 private:
  struct UnitInverse {          // A(1)^-1, Gauss-Jordan with partial pivoting in long double, rounded once
    double v[N][N];
    UnitInverse() {
      constexpr int h = N / 2;
      long double w[N][2 * N];
      for (int i = 0; i < N; ++i)
        for (int j = 0; j < 2 * N; ++j) w[i][j] = 0.0L;
      for (int k = 0; k < h; ++k) {
        w[k][k] = (long double)Polynomial::baseCoefficient(k, k);
        for (int j = k; j < N; ++j) w[h + k][j] = (long double)Polynomial::baseCoefficient(k, j);
      }
      for (int i = 0; i < N; ++i) w[i][N + i] = 1.0L;
      for (int c = 0; c < N; ++c) {
        int p = c;
        for (int r = c + 1; r < N; ++r)
          if (std::abs(w[r][c]) > std::abs(w[p][c])) p = r;
        if (p != c)
          for (int j = 0; j < 2 * N; ++j) std::swap(w[c][j], w[p][j]);
        const long double s = 1.0L / w[c][c];
        for (int j = 0; j < 2 * N; ++j) w[c][j] *= s;
        for (int r = 0; r < N; ++r) {
          if (r == c) continue;
          const long double f = w[r][c];
          if (f != 0.0L)
            for (int j = 0; j < 2 * N; ++j) w[r][j] -= f * w[c][j];
        }
      }
      for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) v[i][j] = (double)w[i][N + j];
    }
  };
  static void invertGeneric(const SquareMatrix& A, SquareMatrix* Ai) {
    constexpr int h = N / 2;
    double Dinv[h][h], W[h][2 * h];
    for (int i = 0; i < h; ++i) for (int j = 0; j < h; ++j) { W[i][j] = A(h + i, h + j); W[i][h + j] = (i == j); }
    for (int c = 0; c < h; ++c) {   // Gauss-Jordan with partial pivoting on the h x h block
      int p = c;
      for (int r = c + 1; r < h; ++r) if (std::abs(W[r][c]) > std::abs(W[p][c])) p = r;
      if (p != c) for (int j = 0; j < 2 * h; ++j) std::swap(W[c][j], W[p][j]);
      const double s = 1.0 / W[c][c];
      for (int j = 0; j < 2 * h; ++j) W[c][j] *= s;
      for (int r = 0; r < h; ++r) {
        if (r == c) continue;
        const double f = W[r][c];
        for (int j = 0; j < 2 * h; ++j) W[r][j] -= f * W[c][j];
      }
    }
    for (int i = 0; i < h; ++i) for (int j = 0; j < h; ++j) Dinv[i][j] = W[i][h + j];
    Ai->setZero();
    for (int i = 0; i < h; ++i) (*Ai)(i, i) = 1.0 / A(i, i);
    for (int i = 0; i < h; ++i) {
      for (int j = 0; j < h; ++j) {
        double acc = 0.0;
        for (int k = 0; k < h; ++k) acc += Dinv[i][k] * A(h + k, j);
        (*Ai)(h + i, j) = -acc / A(j, j);
        (*Ai)(h + i, h + j) = Dinv[i][j];
      }
    }
  }

 public:
  // Q(r, c) = base(d, r) base(d, c) t^(r+c-2d+1) * 2 / (r+c-2d+1) so that 0.5 c^T Q c = int_0^t (p^(d))^2
  static void computeQuadraticCostJacobian(int derivative, double t, SquareMatrix* cost_jacobian) {
    CHECK_LT(derivative, N);
    cost_jacobian->setZero();
    for (int r = derivative; r < N; ++r) {
      for (int c = derivative; c < N; ++c) {
        const double e = r + c - 2 * derivative + 1;
        (*cost_jacobian)(r, c) = Polynomial::baseCoefficient(derivative, r) * Polynomial::baseCoefficient(derivative, c) *
                                 std::pow(t, e) * 2.0 / e;
      }
    }
  }

  // ---- dense accessors ----------------------------------------------------------------------------------------
  void getA(Eigen::MatrixXd* A) const {
    CHECK_NOTNULL(A);
    A->resize(N * n_segments_, N * n_segments_);
    A->setZero();
    for (size_t s = 0; s < n_segments_; ++s) {
      SquareMatrix As;
      setupMappingMatrix(segment_times_[s], &As);
      for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) (*A)(N * s + i, N * s + j) = As(i, j);
    }
  }
  void getAInverse(Eigen::MatrixXd* A_inv) const {
    CHECK_NOTNULL(A_inv);
    A_inv->resize(N * n_segments_, N * n_segments_);
    A_inv->setZero();
    for (size_t s = 0; s < n_segments_; ++s) {
      SquareMatrix As, Ai;
      setupMappingMatrix(segment_times_[s], &As);
      invertMappingMatrix(As, &Ai);
      for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) (*A_inv)(N * s + i, N * s + j) = Ai(i, j);
    }
  }
  // M: row s*N + p <-> (vertex s, derivative p), row s*N + N/2 + p <-> (vertex s+1, p); columns = fixed slots in
  // (vertex, derivative) order followed by free slots in the same order.
  void getM(Eigen::MatrixXd* M) const {
    CHECK_NOTNULL(M);
    M->resize(n_all_constraints_, n_fixed_constraints_ + n_free_constraints_);
    M->setZero();
    const std::vector<int> col = columnOfSlot();
    for (size_t s = 0; s < n_segments_; ++s) {
      for (int p = 0; p < N / 2; ++p) {
        (*M)(s * N + p, col[s * (N / 2) + p]) = 1.0;
        (*M)(s * N + N / 2 + p, col[(s + 1) * (N / 2) + p]) = 1.0;
      }
    }
  }
  // LIN:382-385: the constraint reordering ("mapping") matrix M, rows separated by newlines, entries by blanks
  void printReorderingMatrix(std::ostream& stream) const {
    Eigen::MatrixXd M;
    getM(&M);
    stream << "Mapping matrix:\n";
    for (std::ptrdiff_t r = 0; r < M.rows(); ++r) {
      for (std::ptrdiff_t c = 0; c < M.cols(); ++c) stream << (c ? " " : "") << M(r, c);
      if (r + 1 < M.rows()) stream << "\n";
    }
    stream << std::endl;
  }
  void getMpinv(Eigen::MatrixXd* M_pinv) const {   // M^T with every row normalised by its sum
    CHECK_NOTNULL(M_pinv);
    Eigen::MatrixXd M;
    getM(&M);
    M_pinv->resize(M.cols(), M.rows());
    for (std::ptrdiff_t r = 0; r < M.cols(); ++r) {
      double sum = 0.0;
      for (std::ptrdiff_t c = 0; c < M.rows(); ++c) sum += M(c, r);
      for (std::ptrdiff_t c = 0; c < M.rows(); ++c) (*M_pinv)(r, c) = M(c, r) / sum;
    }
  }
  void getR(Eigen::MatrixXd* R) const {   // R = M^T blkdiag(A^-T Q A^-1) M
    CHECK_NOTNULL(R);
    const size_t na = n_fixed_constraints_ + n_free_constraints_;
    R->resize(na, na);
    R->setZero();
    const std::vector<int> col = columnOfSlot();
    for (size_t s = 0; s < n_segments_; ++s) {
      SquareMatrix As, Ai, Q;
      setupMappingMatrix(segment_times_[s], &As);
      invertMappingMatrix(As, &Ai);
      computeQuadraticCostJacobian(derivative_to_optimize_, segment_times_[s], &Q);
      double QA[N][N];
      for (int a = 0; a < N; ++a) for (int b = 0; b < N; ++b) { double acc = 0; for (int c = 0; c < N; ++c) acc += Q(a, c) * Ai(c, b); QA[a][b] = acc; }
      for (int a = 0; a < N; ++a) {
        const int ca = col[(a < N / 2 ? s : s + 1) * (N / 2) + a % (N / 2)];
        for (int b = 0; b < N; ++b) {
          const int cb = col[(b < N / 2 ? s : s + 1) * (N / 2) + b % (N / 2)];
          double acc = 0;
          for (int c = 0; c < N; ++c) acc += Ai(c, a) * QA[c][b];
          (*R)(ca, cb) += acc;
        }
      }
    }
  }

 private:
  std::vector<int> columnOfSlot() const {
    std::vector<int> col(n_vertices_ * (N / 2));
    int nf = 0, np = 0;
    for (size_t v = 0; v < n_vertices_; ++v) for (int p = 0; p < N / 2; ++p) if ((mask_[v] >> p) & 1u) col[v * (N / 2) + p] = nf++;
    for (size_t v = 0; v < n_vertices_; ++v) for (int p = 0; p < N / 2; ++p) if (!((mask_[v] >> p) & 1u)) col[v * (N / 2) + p] = nf + np++;
    return col;
  }

  // One trajectory per call is latency-bound: see mtg_compat_detail::single_calls_on_device_state (run-time switch;
  // default: the library's host build of the kernels' lane code on this thread instead of a launch + PCIe round trip).
  static uint32_t single_call_backend() { return singleCallsOnDevice() ? 0u : (uint32_t)MTG_FLAG_HOST_BACKEND; }
  int run(bool solve) {
    CHECK(plan_ != nullptr) << "setupFromVertices() has to be called first";
    const size_t D = dimension_, K = n_segments_, nf = n_fixed_constraints_, np = n_free_constraints_;
    std::vector<double> d_fixed(D * nf), d_free(D * np + 1), coeffs(K * D * N);
    for (size_t d = 0; d < D; ++d) {
      for (size_t c = 0; c < nf; ++c) d_fixed[d * nf + c] = fixed_constraints_compact_[d][c];
      if (!solve) for (size_t c = 0; c < np; ++c) d_free[d * np + c] = free_constraints_compact_[d][c];
    }
    mtg_layout lay;
    mtg_layout_aos(plan_.get(), 1, &lay);
    int rc;
    if (solve) {
      rc = mtg_solve_linear(plan_.get(), 1, &lay, segment_times_.data(), d_fixed.data(), coeffs.data(),
                            np ? d_free.data() : nullptr, nullptr, MTG_FLAG_HOST_POINTERS | single_call_backend());
    } else {
      rc = mtg_update_segments_from_free(plan_.get(), 1, &lay, segment_times_.data(), d_fixed.data(), d_free.data(),
                                         coeffs.data(), nullptr, MTG_FLAG_HOST_POINTERS | single_call_backend());
    }
    // host-pointer calls are synchronous and return the batch status themselves (on the plan's own context, whichever
    // thread created it)
    if (rc == MTG_ERR_SINGULAR) return rc;
    if (solve) last_solve_rank_ = np;
    CHECK(rc == MTG_OK) << mtg_status_string(rc);   // LIN:297 CHECK_GT(segment_time, 0) and friends surface here
    for (size_t d = 0; d < D; ++d) {
      if (solve) for (size_t c = 0; c < np; ++c) free_constraints_compact_[d][c] = d_free[d * np + c];
      for (size_t k = 0; k < K; ++k) {
        Eigen::VectorXd c(N);
        for (int j = 0; j < N; ++j) c[j] = coeffs[(k * D + d) * N + j];
        segments_[k].setTime(segment_times_[k]);
        segments_[k][d] = Polynomial(N, c);
      }
    }
    return MTG_OK;
  }

  Vertex::Vector vertices_;
  Segment::Vector segments_;
  std::vector<Eigen::VectorXd> fixed_constraints_compact_, free_constraints_compact_;
  std::vector<double> segment_times_;
  std::vector<uint32_t> mask_;
  std::shared_ptr<mtg_plan> plan_;
  size_t dimension_;
  int derivative_to_optimize_;
  size_t n_vertices_, n_segments_, n_all_constraints_, n_fixed_constraints_, n_free_constraints_;
  size_t last_solve_rank_ = 0;
};

// Batched entry: B trajectories sharing one constraint structure (masks).  Buffers are flat, AoS:
// times [B][K], d_fixed [B][D][n_fixed], coeffs [B][K][D][N], d_free [B][D][n_free], cost [B]; host or device.
template <int _N = 10>
class PolynomialOptimizationBatch {
 public:
  enum { N = _N };
  PolynomialOptimizationBatch(size_t dimension, const std::vector<uint32_t>& fixed_mask, int derivative_to_optimize = N / 2 - 1)
      : dimension_(dimension), n_segments_(fixed_mask.size() - 1), mask_(fixed_mask), derivative_(derivative_to_optimize) {
    plan_ = mtg_compat_detail::make_plan(N, (int)dimension, (int)n_segments_, derivative_to_optimize, fixed_mask);
    mtg_plan_get_info(plan_.get(), &info_);
  }
  // structure taken from a prototype vertex list (values ignored)
  static std::vector<uint32_t> masksFromVertices(const Vertex::Vector& vertices) {
    std::vector<uint32_t> m(vertices.size(), 0u);
    for (size_t v = 0; v < vertices.size(); ++v) for (int p = 0; p < N / 2; ++p) m[v] |= uint32_t(vertices[v].hasConstraint(p)) << p;
    return m;
  }
  size_t getNumberFixedConstraints() const { return info_.n_fixed; }
  size_t getNumberFreeConstraints() const { return info_.n_free; }

  // Host pointers: synchronous; returns true like the reference -- trajectories whose free-constraint system is rank
  // deficient get the reference's behaviour, a BASIC solution from a rank-revealing factorisation (LIN:365-378), inside the
  // library (MTG_FLAG_BASIC_SOLUTION; under-constrained problems are rare and tiny); every other trajectory of the batch
  // keeps its device result.
  // Device pointers: asynchronous, no fallback (sync() reports MTG_ERR_SINGULAR; mtg_solve_linear_status names the
  // trajectories; callers who want the reference's behaviour pass MTG_FLAG_BASIC_SOLUTION to mtg_solve_linear themselves --
  // that call is synchronous) -- returns true when the launch was enqueued.
  bool solveLinear(size_t batch, const double* times, const double* d_fixed, double* coeffs, double* d_free = nullptr,
                   double* cost = nullptr, bool device_pointers = false) {
    mtg_layout lay;
    mtg_layout_aos(plan_.get(), (int64_t)batch, &lay);
    if (device_pointers) {
      const int rc = mtg_solve_linear(plan_.get(), (int64_t)batch, &lay, times, d_fixed, coeffs, d_free, cost, 0u);
      CHECK(rc == MTG_OK) << mtg_status_string(rc) << " " << mtg_last_error_string(mtg_plan_context(plan_.get()));
      return true;
    }
    // (MTG_FLAG_BASIC_SOLUTION: the library replaces the outputs of rank-deficient trajectories by the basic solution)
    const int rc = mtg_solve_linear(plan_.get(), (int64_t)batch, &lay, times, d_fixed, coeffs, d_free, cost,
                                    (uint32_t)MTG_FLAG_HOST_POINTERS | (uint32_t)MTG_FLAG_BASIC_SOLUTION);
    CHECK(rc == MTG_OK) << mtg_status_string(rc) << " " << mtg_last_error_string(mtg_plan_context(plan_.get()));
    return true;
  }
  // waits for the device-pointer solves of THIS object (the context its plan was created on, whichever thread calls)
  void sync() { mtg_compat_detail::check_sync(plan_.get()); }

  // Segment::Vector view of trajectory b of a host coefficient buffer
  void getSegments(const double* coeffs, const double* times, size_t b, Segment::Vector* segments) const {
    segments->assign(n_segments_, Segment(N, (int)dimension_));
    for (size_t k = 0; k < n_segments_; ++k) {
      (*segments)[k].setTime(times[b * n_segments_ + k]);
      for (size_t d = 0; d < dimension_; ++d) {
        Eigen::VectorXd c(N);
        for (int j = 0; j < N; ++j) c[j] = coeffs[((b * n_segments_ + k) * dimension_ + d) * N + j];
        (*segments)[k][d] = Polynomial(N, c);
      }
    }
  }

 private:
  size_t dimension_, n_segments_;
  std::vector<uint32_t> mask_;
  int derivative_;
  std::shared_ptr<mtg_plan> plan_;
  mtg_plan_info info_;
};

// New API: a heterogeneous list of problems (same N and dimension; any number of segments / constraint structure per
// problem) solved as one mixed request -- what a caller of the reference does with a loop over independent
// PolynomialOptimization<N> objects.  Problems are bucketed by structure, uploaded once, and solved through
// mtg_multi_create / mtg_multi_solve: buckets that differ only in their number of segments share ONE kernel launch.
// segments->at(i) receives problem i's solution (Segment::Vector as from PolynomialOptimization<N>::getSegments).
template <int _N>
bool solveLinearMixed(size_t dimension, const std::vector<Vertex::Vector>& problems,
                      const std::vector<std::vector<double>>& segment_times, int derivative_to_optimize,
                      std::vector<Segment::Vector>* segments, std::vector<double>* costs = nullptr) {
  CHECK_NOTNULL(segments);
  CHECK(problems.size() == segment_times.size());
  constexpr int N = _N;
  mtg_context* ctx = mtg_compat_detail::context();
  struct Bucket {
    std::vector<uint32_t> mask;
    std::vector<size_t> members;
    std::shared_ptr<mtg_plan> plan;
    mtg_plan_info info;
    std::vector<double> t, f, c, j;
    double *dt = nullptr, *df = nullptr, *dc = nullptr, *dj = nullptr;
  };
  std::vector<Bucket> buckets;
  std::map<std::vector<uint32_t>, size_t> index;
  for (size_t i = 0; i < problems.size(); ++i) {
    CHECK(problems[i].size() == segment_times[i].size() + 1) << "Size of times must be one less than positions.";
    std::vector<uint32_t> mask = PolynomialOptimizationBatch<N>::masksFromVertices(problems[i]);
    auto it = index.find(mask);
    if (it == index.end()) {
      it = index.emplace(mask, buckets.size()).first;
      buckets.emplace_back();
      buckets.back().mask = mask;
    }
    buckets[it->second].members.push_back(i);
  }
  auto check = [&](int rc) { CHECK(rc == MTG_OK) << mtg_status_string(rc) << " " << mtg_last_error_string(ctx); };
  std::vector<mtg_multi_item> items(buckets.size());
  for (size_t bi = 0; bi < buckets.size(); ++bi) {
    Bucket& bk = buckets[bi];
    const size_t K = bk.mask.size() - 1, B = bk.members.size();
    bk.plan = mtg_compat_detail::make_plan(N, (int)dimension, (int)K, derivative_to_optimize, bk.mask);
    mtg_plan_get_info(bk.plan.get(), &bk.info);
    const size_t nf = bk.info.n_fixed;
    bk.t.resize(B * K);
    bk.f.assign(B * dimension * nf, 0.0);
    for (size_t m = 0; m < B; ++m) {
      const size_t i = bk.members[m];
      for (size_t k = 0; k < K; ++k) bk.t[m * K + k] = segment_times[i][k];
      size_t col = 0;
      for (size_t v = 0; v <= K; ++v)
        for (int p = 0; p < N / 2; ++p) {
          Eigen::VectorXd value;
          if (!problems[i][v].getConstraint(p, &value)) continue;
          for (size_t d = 0; d < dimension; ++d) bk.f[(m * dimension + d) * nf + col] = value[d];
          ++col;
        }
    }
    bk.c.resize(B * K * dimension * N);
    bk.j.resize(B);
    check(mtg_device_malloc(ctx, bk.t.size() * sizeof(double), (void**)&bk.dt));
    check(mtg_device_malloc(ctx, std::max<size_t>(1, bk.f.size()) * sizeof(double), (void**)&bk.df));
    check(mtg_device_malloc(ctx, bk.c.size() * sizeof(double), (void**)&bk.dc));
    check(mtg_device_malloc(ctx, bk.j.size() * sizeof(double), (void**)&bk.dj));
    check(mtg_copy_to_device(ctx, bk.dt, bk.t.data(), bk.t.size() * sizeof(double)));
    if (!bk.f.empty()) check(mtg_copy_to_device(ctx, bk.df, bk.f.data(), bk.f.size() * sizeof(double)));
    mtg_multi_item& it = items[bi];
    it.plan = bk.plan.get();
    it.batch = (int64_t)B;
    mtg_layout_aos(bk.plan.get(), (int64_t)B, &it.layout);
    it.times = bk.dt;
    it.d_fixed = bk.df;
    it.coeffs = bk.dc;
    it.d_free = nullptr;
    it.cost = costs ? bk.dj : nullptr;
  }
  mtg_multi* multi = nullptr;
  check(mtg_multi_create(ctx, (int32_t)items.size(), items.data(), 0, &multi));
  check(mtg_multi_solve(multi));
  mtg_compat_detail::check_sync();
  segments->assign(problems.size(), Segment::Vector());
  if (costs) costs->assign(problems.size(), 0.0);
  for (Bucket& bk : buckets) {
    const size_t K = bk.mask.size() - 1, B = bk.members.size();
    check(mtg_copy_to_host(ctx, bk.c.data(), bk.dc, bk.c.size() * sizeof(double)));
    if (costs) check(mtg_copy_to_host(ctx, bk.j.data(), bk.dj, bk.j.size() * sizeof(double)));
    for (size_t m = 0; m < B; ++m) {
      Segment::Vector& out = (*segments)[bk.members[m]];
      out.assign(K, Segment(N, (int)dimension));
      for (size_t k = 0; k < K; ++k) {
        out[k].setTime(bk.t[m * K + k]);
        for (size_t d = 0; d < dimension; ++d) {
          Eigen::VectorXd c(N);
          for (int n = 0; n < N; ++n) c[n] = bk.c[((m * K + k) * dimension + d) * N + n];
          out[k][d] = Polynomial(N, c);
        }
      }
      if (costs) (*costs)[bk.members[m]] = bk.j[m];
    }
  }
  mtg_multi_destroy(multi);
  for (Bucket& bk : buckets) {
    mtg_device_free(ctx, bk.dt);
    mtg_device_free(ctx, bk.df);
    mtg_device_free(ctx, bk.dc);
    mtg_device_free(ctx, bk.dj);
  }
  return true;
}

}  // namespace mav_trajectory_generation
#endif
