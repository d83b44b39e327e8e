// C entry around the REFERENCE's own getCostAndGradientMellinger (impl/polynomial_optimization_nonlinear_impl.h:287-364), the
// finite-difference time gradient of the Mellinger outer loop: PolynomialOptimizationNonLinear<N> compiled
// from the reference headers where they lie (oracle/Makefile target _ref/libmtg_ref_nl.so) against the Eigen / glog container
// stand-ins of ref_shim/ and the TYPES-ONLY nlopt stand-in ref_shim_nlopt/nlopt.hpp -- the member called here never touches
// nlopt: it perturbs the segment times, calls the linear optimiser's updateSegmentTimes() / solveLinear() / computeCost() and
// restores the times.  It is a private member; this translation unit reaches it by compiling the reference's header with
// `private` spelt `public` (after every system header has been included).
// TEST INFRASTRUCTURE ONLY: parity anchor of tests/ for mtg_mellinger_cost_gradient; never linked into the product library.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <iostream>
#include <limits>
#include <map>
#include <memory>
#include <set>
#include <sstream>
#include <string>
#include <tuple>
#include <vector>
#include <nlopt.hpp>
#include <mav_trajectory_generation/polynomial_optimization_linear.h>
#include <mav_trajectory_generation/trajectory.h>
#include <mav_trajectory_generation/vertex.h>
#define private public
#include <mav_trajectory_generation/polynomial_optimization_nonlinear.h>
#undef private

namespace mtg = mav_trajectory_generation;

namespace {
template <int N>
int mellinger_one(int deriv, int k, int dim, const int* masks, const double* times, const double* d_fixed, int n_fixed,
                  double* cost, double* grad) {
  mtg::Vertex::Vector vertices(k + 1, mtg::Vertex(dim));
  int col = 0;
  for (int v = 0; v <= k; ++v)
    for (int p = 0; p < N / 2; ++p)
      if ((masks[v] >> p) & 1) {
        Eigen::VectorXd value(dim);
        for (int d = 0; d < dim; ++d) value[d] = d_fixed[(size_t)d * n_fixed + col];
        vertices[v].addConstraint(p, value);
        ++col;
      }
  mtg::NonlinearOptimizationParameters params;
  params.time_alloc_method = mtg::NonlinearOptimizationParameters::kMellingerOuterLoop;
  mtg::PolynomialOptimizationNonLinear<N> opt(dim, params);
  opt.setupFromVertices(vertices, std::vector<double>(times, times + k), deriv);
  opt.solveLinear();                                   // the state the objective holds when it asks for the gradient (impl:556-571)
  std::vector<double> g;
  *cost = opt.getCostAndGradientMellinger(&g);
  if ((int)g.size() != k) return -1;
  for (int i = 0; i < k; ++i) grad[i] = g[i];
  return 0;
}
}  // namespace

extern "C" int mtg_ref_mellinger_cost_gradient(int n, int deriv, int k, int dim, const int* masks, const double* times,
                                               const double* d_fixed, int n_fixed, double* cost, double* grad) {
  switch (n) {
    case 4: return mellinger_one<4>(deriv, k, dim, masks, times, d_fixed, n_fixed, cost, grad);
    case 6: return mellinger_one<6>(deriv, k, dim, masks, times, d_fixed, n_fixed, cost, grad);
    case 8: return mellinger_one<8>(deriv, k, dim, masks, times, d_fixed, n_fixed, cost, grad);
    case 10: return mellinger_one<10>(deriv, k, dim, masks, times, d_fixed, n_fixed, cost, grad);
    case 12: return mellinger_one<12>(deriv, k, dim, masks, times, d_fixed, n_fixed, cost, grad);
    default: return -2;
  }
}
