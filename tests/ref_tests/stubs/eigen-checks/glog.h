// Stand-in for eigen_checks' glog front end: CHECK_EIGEN_MATRIX_EQUAL_DOUBLE(A, B) aborts unless the matrices agree element-wise
// to Eigen's dummy_precision of double (1e-12) -- test/test_polynomial_optimization.cpp:786, the MATLAB vector.
#ifndef MTG_EIGEN_CHECKS_GLOG_H_
#define MTG_EIGEN_CHECKS_GLOG_H_
#include <glog/logging.h>
#include <eigen-checks/gtest.h>
#define CHECK_EIGEN_MATRIX_EQUAL_DOUBLE(A, B)                                        \
  do {                                                                                \
    const auto mtg_ec_r = ::eigen_checks_standin::MatricesNear(A, #A, B, #B, 1e-12); \
    CHECK(static_cast<bool>(mtg_ec_r)) << mtg_ec_r.message();                        \
  } while (0)
#define CHECK_EIGEN_MATRIX_EQUAL(A, B)                                               \
  do {                                                                                \
    const auto mtg_ec_r = ::eigen_checks_standin::MatricesNear(A, #A, B, #B, 0.0);   \
    CHECK(static_cast<bool>(mtg_ec_r)) << mtg_ec_r.message();                        \
  } while (0)
#define CHECK_EIGEN_MATRIX_NEAR(A, B, tol)                                          \
  do {                                                                                \
    const auto mtg_ec_r = ::eigen_checks_standin::MatricesNear(A, #A, B, #B, tol);   \
    CHECK(static_cast<bool>(mtg_ec_r)) << mtg_ec_r.message();                        \
  } while (0)
#endif
