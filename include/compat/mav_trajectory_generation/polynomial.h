// Compat veneer: the subset of the reference's Polynomial that sits on the solveLinear() path
// (reference: polynomial.h:37-251, src/polynomial.cpp:145-160,213-214): coefficient container with increasing
// powers, derivative evaluation, the static derivative-coefficient table and baseCoeffsWithTime.
// Root finding / convolution / min-max are post-solve analysis and out of scope.
#ifndef MAV_TRAJECTORY_GENERATION_POLYNOMIAL_H_
#define MAV_TRAJECTORY_GENERATION_POLYNOMIAL_H_
#include <limits>
#include <vector>

#include "mtg_compat_base.h"

namespace mav_trajectory_generation {

class Polynomial {
 public:
  typedef std::vector<Polynomial> Vector;
  static constexpr int kMaxN = 12;
  static constexpr int kMaxConvolutionSize = 2 * kMaxN - 2;

  explicit Polynomial(int N) : N_(N), coefficients_(N) { coefficients_.setZero(); }
  Polynomial(int N, const Eigen::VectorXd& coeffs) : N_(N), coefficients_(coeffs) {
    CHECK_EQ(N_, (int)coeffs.size()) << "Number of coefficients has to match.";
  }
  explicit Polynomial(const Eigen::VectorXd& coeffs) : N_((int)coeffs.size()), coefficients_(coeffs) {}

  int N() const { return N_; }
  bool operator==(const Polynomial& rhs) const { return coefficients_ == rhs.coefficients_; }
  bool operator!=(const Polynomial& rhs) const { return !(*this == rhs); }

  // base(n, i) = i * (i-1) * ... * (i-n+1): coefficient of t^(i-n) in the n-th derivative of t^i.
  static double baseCoefficient(int derivative, int i) {
    if (i < derivative) return 0.0;
    double b = 1.0;
    for (int k = 0; k < derivative; ++k) b *= (i - k);
    return b;
  }

  void setCoefficients(const Eigen::VectorXd& coeffs) {
    CHECK_EQ(N_, (int)coeffs.size());
    coefficients_ = coeffs;
  }

  // Coefficients of the derivative-th derivative, increasing powers, zero padded to N.
  Eigen::VectorXd getCoefficients(int derivative = 0) const {
    CHECK(derivative >= 0 && derivative <= N_);
    Eigen::VectorXd out = Eigen::VectorXd::Zero(N_);
    for (int i = derivative; i < N_; ++i) out[i - derivative] = baseCoefficient(derivative, i) * coefficients_[i];
    return out;
  }

  // Value of the derivative-th derivative at t (Horner on the derivative's coefficients).
  double evaluate(double t, int derivative = 0) const {
    if (derivative >= N_) return 0.0;
    double acc = 0.0;
    for (int i = N_ - 1; i >= derivative; --i) acc = acc * t + baseCoefficient(derivative, i) * coefficients_[i];
    return acc;
  }

  // Row of the mapping matrix: r[j] = base(derivative, j) * t^(j - derivative); for |t| < eps only the leading
  // entry is set (the reference treats such t as exactly zero, polynomial.h:211).
  static Eigen::VectorXd baseCoeffsWithTime(int N, int derivative, double t) {
    CHECK_LT(derivative, N);
    CHECK_GE(derivative, 0);
    Eigen::VectorXd c = Eigen::VectorXd::Zero(N);
    c[derivative] = baseCoefficient(derivative, derivative);
    if (std::abs(t) < std::numeric_limits<double>::epsilon()) return c;
    double tp = t;
    for (int j = derivative + 1; j < N; ++j) {
      c[j] = baseCoefficient(derivative, j) * tp;
      tp *= t;
    }
    return c;
  }

 private:
  int N_;
  Eigen::VectorXd coefficients_;
};

}  // namespace mav_trajectory_generation
#endif
