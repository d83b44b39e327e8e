// Stand-in for ethz-asl/eigen_checks' gtest front end (not in this image): EIGEN_MATRIX_NEAR as the reference's tests use it
// (test/test_polynomial_optimization.cpp:135, :555, :738) -- element-wise |a - b| <= tolerance, same dimensions, with the
// offending element in the message.
#ifndef MTG_EIGEN_CHECKS_GTEST_H_
#define MTG_EIGEN_CHECKS_GTEST_H_
#include <cmath>
#include <gtest/gtest.h>

namespace eigen_checks_standin {
template <class A, class B>
::testing::AssertionResult MatricesNear(const A& a, const char* na, const B& b, const char* nb, double tol) {
  if (a.rows() != b.rows() || a.cols() != b.cols())
    return ::testing::AssertionFailure() << "Matrix size mismatch: " << na << " is " << a.rows() << "x" << a.cols() << ", " << nb << " is "
                                         << b.rows() << "x" << b.cols();
  for (long r = 0; r < (long)a.rows(); ++r)
    for (long c = 0; c < (long)a.cols(); ++c) {
      const double x = a(r, c), y = b(r, c);
      if (!(std::fabs(x - y) <= tol))
        return ::testing::AssertionFailure() << na << "(" << r << "," << c << ") = " << x << " and " << nb << "(" << r << "," << c << ") = " << y
                                             << " differ by " << std::fabs(x - y) << " > " << tol;
    }
  return ::testing::AssertionSuccess();
}
}  // namespace eigen_checks_standin
#define EIGEN_MATRIX_NEAR(A, B, tol) ::eigen_checks_standin::MatricesNear(A, #A, B, #B, tol)
#define EIGEN_MATRIX_EQUAL_DOUBLE(A, B) ::eigen_checks_standin::MatricesNear(A, #A, B, #B, 1e-12)
#endif
