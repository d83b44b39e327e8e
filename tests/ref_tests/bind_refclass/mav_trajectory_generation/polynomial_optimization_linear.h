// The REFERENCE'S OWN PolynomialOptimization<N> with the replacement solveLinear() of INTEGRATION.md section 1.
// The reference tree is read-only, so its header is taken as it is (#include_next) and the member function is given its new
// body as an explicit specialisation for the N the reference's tests instantiate (a member of the class like the original:
// it sees the same private members).  Everything else -- setupFromVertices, the constraint reordering, updateSegmentTimes, the
// matrix accessors, computeCost, the extrema helpers -- stays the reference's code.
// Backend of the single-trajectory call: the library's host build of the lane code (MTG_FLAG_HOST_BACKEND) unless the
// environment says MTG_REF_TESTS_BACKEND=device.
#ifndef MTG_REF_TESTS_BIND_REFCLASS_H_
#define MTG_REF_TESTS_BIND_REFCLASS_H_
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include_next <mav_trajectory_generation/polynomial_optimization_linear.h>
#include <mtg_hip.h>

namespace mav_trajectory_generation {
inline uint32_t mtgRefTestsBackendFlag() {
  const char* e = std::getenv("MTG_REF_TESTS_BACKEND");
  return (e && std::strcmp(e, "device") == 0) ? 0u : (uint32_t)MTG_FLAG_HOST_BACKEND;
}

#define MTG_BIND_SOLVE_LINEAR(NN)                                                                                           \
  template <>                                                                                                               \
  inline bool PolynomialOptimization<NN>::solveLinear() {                                                                   \
    CHECK(derivative_to_optimize_ >= 0 && derivative_to_optimize_ <= kHighestDerivativeToOptimize);                         \
    static thread_local mtg_context* ctx = nullptr;                                                                         \
    if (!ctx) CHECK_EQ(mtg_context_create(0, nullptr, &ctx), MTG_OK);                                                       \
    std::vector<uint32_t> mask(n_vertices_);                                                                                \
    for (size_t v = 0; v < n_vertices_; ++v)                                                                                \
      for (int p = 0; p < N / 2; ++p) mask[v] |= uint32_t(vertices_[v].hasConstraint(p)) << p;                              \
    mtg_plan_desc desc{N, int(dimension_), int(n_segments_), derivative_to_optimize_, mask.data()};                         \
    mtg_plan* plan = nullptr;                                                                                               \
    CHECK_EQ(mtg_plan_create(ctx, &desc, &plan), MTG_OK);                                                                   \
    std::vector<double> d_fixed(dimension_ * n_fixed_constraints_), d_free(dimension_ * n_free_constraints_),               \
        coeffs(n_segments_ * dimension_ * N);                                                                               \
    for (size_t d = 0; d < dimension_; ++d)                                                                                 \
      std::copy_n(fixed_constraints_compact_[d].data(), n_fixed_constraints_, &d_fixed[d * n_fixed_constraints_]);          \
    mtg_layout lay;                                                                                                         \
    mtg_layout_aos(plan, 1, &lay);                                                                                          \
    const int rc = mtg_solve_linear(plan, 1, &lay, segment_times_.data(), d_fixed.data(), coeffs.data(), d_free.data(),     \
                                    nullptr, MTG_FLAG_HOST_POINTERS | mtgRefTestsBackendFlag() | MTG_FLAG_BASIC_SOLUTION);  \
    CHECK_EQ(rc, MTG_OK) << mtg_status_string(rc);                                                                          \
    for (size_t d = 0; d < dimension_; ++d) {                                                                               \
      free_constraints_compact_[d] = Eigen::Map<Eigen::VectorXd>(&d_free[d * n_free_constraints_], n_free_constraints_);    \
      for (size_t k = 0; k < n_segments_; ++k) {                                                                            \
        segments_[k].setTime(segment_times_[k]);                                                                            \
        segments_[k][d] = Polynomial(N, Eigen::Map<Eigen::VectorXd>(&coeffs[(k * dimension_ + d) * N], N));                 \
      }                                                                                                                     \
    }                                                                                                                       \
    mtg_plan_destroy(plan);                                                                                                 \
    return true;                                                                                                            \
  }
MTG_BIND_SOLVE_LINEAR(10)
#undef MTG_BIND_SOLVE_LINEAR
}  // namespace mav_trajectory_generation
#endif
