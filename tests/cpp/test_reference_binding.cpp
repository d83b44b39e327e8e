// Drop-in check against the REFERENCE'S OWN CLASS: the reference's PolynomialOptimization<N> (compiled from
// /root/reference where it lies, Eigen/glog container stand-ins of oracle/ref_shim) is set up through its public
// API, solved once by its own solveLinear() (impl/polynomial_optimization_linear_impl.h:339-379) and once by the
// replacement body that INTEGRATION.md section 1 proposes for that member function -- the same code, written here as
// a free function over the object's members -- which forwards to libmtg_hip.so through the C ABI.  Afterwards the
// object's segments_, free_constraints_compact_ and computeCost() must agree with the reference's own results.
// Built by __graft_entry__.build() only where /root/reference exists; the binary travels to the GPU box.
#include <cstdio>
#include <vector>

// the replacement body is a member function in INTEGRATION.md; a test cannot edit the read-only reference header, so it
// reaches the same private members from outside
// (every system / stand-in header the reference pulls in is included first, so the access hack touches only the
// reference's own class definitions)
#include <algorithm>
#include <chrono>
#include <cmath>
#include <iostream>
#include <limits>
#include <map>
#include <numeric>
#include <random>
#include <set>
#include <sstream>
#include <string>
#include <tuple>
#include <utility>
#include <Eigen/Sparse>
#include <glog/logging.h>
#define private public
#include <mav_trajectory_generation/polynomial_optimization_linear.h>
#undef private
#include <mtg_hip.h>

using namespace mav_trajectory_generation;

// MTG_FLAG_HOST_BACKEND (the library's host build of the lane code; what INTEGRATION.md section 1 uses for a single
// trajectory) or 0 ("device" argument: the same call through the GPU)
static uint32_t kSingleCallBackend = MTG_FLAG_HOST_BACKEND;

// ---- INTEGRATION.md section 1, verbatim except `opt.` in front of the members ------------------------------------
template <int N>
bool solveLinearViaMtgHip(PolynomialOptimization<N>& opt) {
  CHECK(opt.derivative_to_optimize_ >= 0 && opt.derivative_to_optimize_ <= PolynomialOptimization<N>::kHighestDerivativeToOptimize);
  static thread_local mtg_context* ctx = nullptr;               // one context per host thread
  if (!ctx) CHECK_EQ(mtg_context_create(/*device*/0, /*stream*/nullptr, &ctx), MTG_OK);

  std::vector<uint32_t> mask(opt.n_vertices_);
  for (size_t v = 0; v < opt.n_vertices_; ++v)
    for (int p = 0; p < N / 2; ++p) mask[v] |= uint32_t(opt.vertices_[v].hasConstraint(p)) << p;
  mtg_plan_desc desc{N, int(opt.dimension_), int(opt.n_segments_), opt.derivative_to_optimize_, mask.data()};
  mtg_plan* plan = nullptr;
  CHECK_EQ(mtg_plan_create(ctx, &desc, &plan), MTG_OK);         // cache per constraint structure in real use

  std::vector<double> d_fixed(opt.dimension_ * opt.n_fixed_constraints_), d_free(opt.dimension_ * opt.n_free_constraints_),
      coeffs(opt.n_segments_ * opt.dimension_ * N);
  for (size_t d = 0; d < opt.dimension_; ++d)
    std::copy_n(opt.fixed_constraints_compact_[d].data(), opt.n_fixed_constraints_, &d_fixed[d * opt.n_fixed_constraints_]);
  mtg_layout lay;
  mtg_layout_aos(plan, 1, &lay);
  const int rc = mtg_solve_linear(plan, 1, &lay, opt.segment_times_.data(), d_fixed.data(), coeffs.data(),
                                  d_free.data(), nullptr, MTG_FLAG_HOST_POINTERS | kSingleCallBackend);
  CHECK_EQ(rc, MTG_OK) << mtg_status_string(rc);   // host-pointer calls are synchronous: LIN:297 etc. surface here

  for (size_t d = 0; d < opt.dimension_; ++d) {
    opt.free_constraints_compact_[d] =
        Eigen::Map<Eigen::VectorXd>(&d_free[d * opt.n_free_constraints_], opt.n_free_constraints_);
    for (size_t k = 0; k < opt.n_segments_; ++k) {
      opt.segments_[k].setTime(opt.segment_times_[k]);
      opt.segments_[k][d] = Polynomial(N, Eigen::Map<Eigen::VectorXd>(&coeffs[(k * opt.dimension_ + d) * N], N));
    }
  }
  mtg_plan_destroy(plan);
  return true;
}
// ------------------------------------------------------------------------------------------------------------------

static int g_fail = 0;

template <int N>
void run_case(const char* name, const Vertex::Vector& vertices, const std::vector<double>& times, int derivative, int dim,
              double tol) {
  PolynomialOptimization<N> ref(dim), ours(dim);
  ref.setupFromVertices(vertices, times, derivative);
  ours.setupFromVertices(vertices, times, derivative);
  ref.solveLinear();                    // the reference's own solve
  solveLinearViaMtgHip<N>(ours);        // the proposed replacement body
  Segment::Vector sr, so;
  ref.getSegments(&sr);
  ours.getSegments(&so);
  double worst = 0.0;
  for (size_t k = 0; k < sr.size(); ++k)
    for (int d = 0; d < dim; ++d) {
      const Eigen::VectorXd a = sr[k][d].getCoefficients(0), b = so[k][d].getCoefficients(0);
      double num = 0, den = 0;
      for (int j = 0; j < N; ++j) {
        num = std::max(num, std::abs(a[j] - b[j]));
        den = std::max(den, std::abs(a[j]));
      }
      worst = std::max(worst, num / (den > 0 ? den : 1.0));
    }
  std::vector<Eigen::VectorXd> fr, fo;
  ref.getFreeConstraints(&fr);
  ours.getFreeConstraints(&fo);
  double worst_free = 0.0, scale_free = 1.0;
  for (int d = 0; d < dim; ++d)
    for (Eigen::Index j = 0; j < fr[d].size(); ++j) {
      worst_free = std::max(worst_free, std::abs(fr[d][j] - fo[d][j]));
      scale_free = std::max(scale_free, std::abs(fr[d][j]));
    }
  const double jr = ref.computeCost(), jo = ours.computeCost();   // the reference's computeCost() on both objects
  // downstream use through the reference's API: a Trajectory from our segments evaluates like the reference's
  Trajectory tr, to;
  ref.getTrajectory(&tr);
  ours.getTrajectory(&to);
  double worst_eval = 0.0;
  for (int i = 0; i <= 20; ++i) {
    const double t = tr.getMaxTime() * i / 20.0 * (1.0 - 1e-12);
    const Eigen::VectorXd pa = tr.evaluate(t, derivative_order::POSITION), pb = to.evaluate(t, derivative_order::POSITION);
    for (int d = 0; d < dim; ++d) worst_eval = std::max(worst_eval, std::abs(pa[d] - pb[d]));
  }
  const bool ok = worst < tol && worst_free <= 10 * tol * scale_free && std::abs(jo - jr) <= 1e-7 * std::abs(jr) + 1e-300 &&
                  worst_eval < 1e-7;
  std::printf("%-28s N=%d K=%zu D=%d d=%d  coeff rel err %.2e  d_free err %.2e  cost %.9g vs %.9g  eval err %.1e  %s\n", name,
              N, times.size(), dim, derivative, worst, worst_free, jo, jr, worst_eval, ok ? "ok" : "FAIL");
  if (!ok) ++g_fail;
}

int main(int argc, char** argv) {
  if (argc > 1 && std::string(argv[1]) == "device") kSingleCallBackend = 0;
  // README example (README.md:104-140)
  {
    Vertex::Vector vertices;
    Vertex start(3), middle(3), end(3);
    start.makeStartOrEnd(Eigen::Vector3d(0, 0, 1), derivative_order::SNAP);
    middle.addConstraint(derivative_order::POSITION, Eigen::Vector3d(1, 2, 3));
    end.makeStartOrEnd(Eigen::Vector3d(2, 1, 5), derivative_order::SNAP);
    vertices = {start, middle, end};
    run_case<10>("readme", vertices, estimateSegmentTimes(vertices, 2.0, 2.0), derivative_order::SNAP, 3, 1e-9);
  }
  // TwoVerticesSetup (test_polynomial_optimization.cpp:743-787): n_free == 0 branch
  {
    Vertex v0(1), v1(1);
    v0.makeStartOrEnd(0.0, derivative_order::SNAP);
    v1.makeStartOrEnd(5.0, derivative_order::SNAP);
    run_case<10>("two_vertices", {v0, v1}, {5.0}, derivative_order::SNAP, 1, 1e-9);
  }
  // BASELINE config 2 shape and the reference's own test parameter sets (:790-867)
  for (int seed = 0; seed < 20; ++seed) {
    const Vertex::Vector vertices = createRandomVertices(derivative_order::SNAP, 8, Eigen::VectorXd::Constant(3, -10.0),
                                                         Eigen::VectorXd::Constant(3, 10.0), seed);
    run_case<10>("config2_shape", vertices, estimateSegmentTimes(vertices, 3.0, 5.0), derivative_order::SNAP, 3, 1e-9);
  }
  {
    const Vertex::Vector v1 = createRandomVertices1D(derivative_order::SNAP, 10, -10.0, 10.0, 102);
    run_case<10>("topt_1D_K10", v1, estimateSegmentTimes(v1, 3.0, 5.0), derivative_order::SNAP, 1, 1e-9);
    const Vertex::Vector v3 = createRandomVertices(derivative_order::SNAP, 50, Eigen::VectorXd::Constant(3, -10.0),
                                                   Eigen::VectorXd::Constant(3, 10.0), 106);
    run_case<10>("topt_3D_K50", v3, estimateSegmentTimes(v3, 3.0, 5.0), derivative_order::SNAP, 3, 1e-9);
    const Vertex::Vector v8 = createRandomVertices(derivative_order::JERK, 6, Eigen::VectorXd::Constant(3, -10.0),
                                                   Eigen::VectorXd::Constant(3, 10.0), 7);
    run_case<8>("N8_jerk", v8, estimateSegmentTimes(v8, 3.0, 5.0), derivative_order::JERK, 3, 1e-9);
    // N = 12 with ends fixed only up to snap (test_feasibility.cpp:65-66,97-99): float64 evaluation of the
    // reference's own formulas is ~1e-8 accurate here
    const Vertex::Vector v12 = createRandomVertices(derivative_order::SNAP, 5, Eigen::VectorXd::Constant(3, -10.0),
                                                    Eigen::VectorXd::Constant(3, 10.0), 11);
    run_case<12>("N12_free_ends", v12, estimateSegmentTimes(v12, 3.0, 5.0), derivative_order::SNAP, 3, 5e-7);
  }
  std::printf(g_fail ? "REFERENCE BINDING: %d FAILED\n" : "REFERENCE BINDING OK\n", g_fail);
  return g_fail ? 1 : 0;
}
