// COMPILE-ONLY stand-in for the reference's polynomial_optimization_nonlinear.h (which needs nlopt: not in this image, no
// network).  /root/reference's test/test_polynomial_optimization.cpp includes that header and two of its tests
// (UnconstrainedNonlinear :402-503, TimeScaling :608-686) drive the nlopt-based time optimiser; those two are FILTERED OUT when
// the binary runs (tests/ref_tests/README.md).  This header exists only so that the file compiles unmodified: the names the
// two tests touch, forwarding the linear part to PolynomialOptimization<N> and refusing to optimise.  Out of scope by SURVEY
// section 2 (nonlinear optimiser); nothing in the product includes it.
#ifndef MTG_REF_TESTS_NONLINEAR_STANDIN_H_
#define MTG_REF_TESTS_NONLINEAR_STANDIN_H_
#include <string>
#include <vector>

#include "mav_trajectory_generation/polynomial_optimization_linear.h"

namespace nlopt {
enum algorithm { LN_BOBYQA, LN_SBPLX, LN_COBYLA, LD_LBFGS, GN_ISRES, GN_ORIG_DIRECT, GN_ORIG_DIRECT_L };
enum result { FAILURE = -1, INVALID_ARGS = -2, OUT_OF_MEMORY = -3, ROUNDOFF_LIMITED = -4, FORCED_STOP = -5, SUCCESS = 1 };
inline std::string returnValueToString(int r) { return r == FAILURE ? "FAILURE (nlopt is not available in this image)" : std::to_string(r); }
}  // namespace nlopt

namespace mav_trajectory_generation {
struct NonlinearOptimizationParameters {
  double f_abs = -1, f_rel = 0.05, x_rel = -1, x_abs = -1, initial_stepsize_rel = 0.1, equality_constraint_tolerance = 1.0e-3,
         inequality_constraint_tolerance = 0.1, time_penalty = 500.0, soft_constraint_weight = 100.0;
  int max_iterations = 3000, random_seed = 0;
  nlopt::algorithm algorithm = nlopt::LN_BOBYQA;
  bool use_soft_constraints = true, print_debug_info = false, print_debug_info_time_allocation = false;
  enum TimeAllocMethod { kSquaredTime, kRichterTime, kMellingerOuterLoop, kSquaredTimeAndConstraints, kRichterTimeAndConstraints, kUnknown }
      time_alloc_method = kSquaredTimeAndConstraints;
};

template <int _N>
class PolynomialOptimizationNonLinear {
 public:
  PolynomialOptimizationNonLinear(size_t dimension, const NonlinearOptimizationParameters&) : linear_(dimension) {}
  bool setupFromVertices(const Vertex::Vector& vertices, const std::vector<double>& segment_times, int derivative_to_optimize) {
    return linear_.setupFromVertices(vertices, segment_times, derivative_to_optimize);
  }
  bool addMaximumMagnitudeConstraint(int, double) { return true; }
  bool solveLinear() { return linear_.solveLinear(); }
  int optimize() { return nlopt::FAILURE; }
  void scaleSegmentTimesWithViolation() {}
  double getCost() const { return linear_.computeCost(); }
  double getTotalCostWithSoftConstraints() const { return linear_.computeCost(); }
  void getTrajectory(Trajectory* trajectory) const { linear_.getTrajectory(trajectory); }
  PolynomialOptimization<_N>& getPolynomialOptimizationRef() { return linear_; }
  const PolynomialOptimization<_N>& getPolynomialOptimizationRef() const { return linear_; }

 private:
  PolynomialOptimization<_N> linear_;
};
}  // namespace mav_trajectory_generation
#endif
