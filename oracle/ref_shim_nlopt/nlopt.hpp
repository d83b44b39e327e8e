// nlopt.hpp -- TYPES ONLY.  nlopt is an un-vendored dependency of the reference's polynomial_optimization_nonlinear.h
// (package.xml: nlopt via catkin; absent from this image, no network).  This header declares exactly the names that header and
// its impl mention so that the reference's class template COMPILES; there is no optimiser behind it: opt::optimize() reports
// nlopt::FAILURE.  The one member the oracle build calls -- getCostAndGradientMellinger,
// impl/polynomial_optimization_nonlinear_impl.h:287-364 -- does not touch nlopt at all: it calls updateSegmentTimes(),
// solveLinear() and computeCost() of the linear optimiser, all of it the reference's own code (oracle/ref_nonlinear_wrap.cpp).
// TEST INFRASTRUCTURE ONLY.
#ifndef MTG_REF_SHIM_NLOPT_HPP_
#define MTG_REF_SHIM_NLOPT_HPP_
#include <stdexcept>
#include <vector>

namespace nlopt {
enum algorithm {
  GN_DIRECT = 0, GN_DIRECT_L, GN_DIRECT_L_RAND, GN_DIRECT_NOSCAL, GN_DIRECT_L_NOSCAL, GN_DIRECT_L_RAND_NOSCAL, GN_ORIG_DIRECT,
  GN_ORIG_DIRECT_L, GD_STOGO, GD_STOGO_RAND, LD_LBFGS_NOCEDAL, LD_LBFGS, LN_PRAXIS, LD_VAR1, LD_VAR2, LD_TNEWTON,
  LD_TNEWTON_RESTART, LD_TNEWTON_PRECOND, LD_TNEWTON_PRECOND_RESTART, GN_CRS2_LM, GN_MLSL, GD_MLSL, GN_MLSL_LDS, GD_MLSL_LDS,
  LD_MMA, LN_COBYLA, LN_NEWUOA, LN_NEWUOA_BOUND, LN_NELDERMEAD, LN_SBPLX, LN_AUGLAG, LD_AUGLAG, LN_AUGLAG_EQ, LD_AUGLAG_EQ,
  LN_BOBYQA, GN_ISRES, AUGLAG, AUGLAG_EQ, G_MLSL, G_MLSL_LDS, LD_SLSQP, LD_CCSAQ, GN_ESCH, NUM_ALGORITHMS
};
enum result {
  FAILURE = -1, INVALID_ARGS = -2, OUT_OF_MEMORY = -3, ROUNDOFF_LIMITED = -4, FORCED_STOP = -5,
  SUCCESS = 1, STOPVAL_REACHED = 2, FTOL_REACHED = 3, XTOL_REACHED = 4, MAXEVAL_REACHED = 5, MAXTIME_REACHED = 6
};
typedef double (*vfunc)(const std::vector<double>& x, std::vector<double>& grad, void* data);
class opt {
 public:
  opt(algorithm, unsigned n) : n_(n) {}
  void set_ftol_rel(double) {}
  void set_ftol_abs(double) {}
  void set_xtol_rel(double) {}
  void set_xtol_abs(double) {}
  void set_maxeval(int) {}
  void set_initial_step(const std::vector<double>&) {}
  void set_upper_bounds(double) {}
  void set_lower_bounds(double) {}
  void set_upper_bounds(const std::vector<double>&) {}
  void set_lower_bounds(const std::vector<double>&) {}
  void set_min_objective(vfunc, void*) {}
  void add_inequality_constraint(vfunc, void*, double) {}
  result optimize(std::vector<double>&, double&) { return FAILURE; }   // no optimiser in this image
 private:
  unsigned n_;
};
}  // namespace nlopt
inline void nlopt_srand(unsigned long) {}
inline void nlopt_srand_time() {}
#endif
