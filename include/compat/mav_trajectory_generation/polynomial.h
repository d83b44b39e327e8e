// Compat veneer: the subset of the reference's Polynomial that sits on the solveLinear() path
// (reference: polynomial.h:37-251, src/polynomial.cpp:145-160,213-214): coefficient container with increasing
// powers, derivative evaluation, the static derivative-coefficient table and baseCoeffsWithTime.
// Convolution and the min/max candidate helpers the linear optimiser's extrema functions rely on are provided too,
// with a derivative-recursion + bisection real-root finder in place of the reference's Jenkins-Traub translation
// (rpoly_ak1.cpp): same candidate semantics (interval ends + real critical points inside the interval).
#ifndef MAV_TRAJECTORY_GENERATION_POLYNOMIAL_H_
#define MAV_TRAJECTORY_GENERATION_POLYNOMIAL_H_
#include <algorithm>
#include <cmath>
#include <limits>
#include <utility>
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

  // Discrete convolution: out[m] = sum_n data[m - n] * kernel[n] (product of the two polynomials).
  static Eigen::VectorXd convolve(const Eigen::VectorXd& data, const Eigen::VectorXd& kernel) {
    Eigen::VectorXd out = Eigen::VectorXd::Zero(getConvolutionLength((int)data.size(), (int)kernel.size()));
    for (int i = 0; i < data.size(); ++i) for (int j = 0; j < kernel.size(); ++j) out[i + j] += data[i] * kernel[j];
    return out;
  }
  static inline int getConvolutionLength(int data_size, int kernel_size) { return data_size + kernel_size - 1; }

  // All real roots of sum c[i] t^i inside [a, b]: the roots of the derivative (found recursively) split the
  // interval into monotone pieces; each piece with a sign change is bisected.  Degenerate (identically zero)
  // polynomials have no isolated roots.
  static void realRootsInInterval(std::vector<double> c, double a, double b, std::vector<double>* roots) {
    roots->clear();
    double scale = 0.0;
    for (double x : c) scale = std::max(scale, std::abs(x));
    while (!c.empty() && std::abs(c.back()) <= 1e-14 * scale) c.pop_back();   // true degree
    if (c.size() < 2 || !(a <= b)) return;
    auto eval = [&](double t) { double v = 0.0; for (size_t i = c.size(); i-- > 0;) v = v * t + c[i]; return v; };
    if (c.size() == 2) {
      const double r = -c[0] / c[1];
      if (r >= a && r <= b) roots->push_back(r);
      return;
    }
    std::vector<double> dc(c.size() - 1), crit;
    for (size_t i = 1; i < c.size(); ++i) dc[i - 1] = c[i] * (double)i;
    realRootsInInterval(dc, a, b, &crit);
    std::vector<double> knots;
    knots.push_back(a);
    for (double t : crit) if (t > knots.back()) knots.push_back(t);
    if (b > knots.back()) knots.push_back(b);
    const double ftol = 1e-13 * scale;
    for (size_t i = 0; i + 1 < knots.size(); ++i) {
      double lo = knots[i], hi = knots[i + 1];
      double flo = eval(lo), fhi = eval(hi);
      if (std::abs(flo) <= ftol * std::max(1.0, std::pow(std::max(std::abs(lo), 1.0), (double)c.size() - 1))) {
        if (roots->empty() || lo > roots->back()) roots->push_back(lo);   // root at a knot (incl. double roots)
        continue;
      }
      if ((flo < 0) == (fhi < 0) || fhi == 0.0) {
        if (fhi == 0.0 && i + 2 == knots.size()) roots->push_back(hi);
        continue;
      }
      for (int it = 0; it < 200 && hi - lo > 1e-15 * std::max(1.0, std::abs(lo)); ++it) {
        const double mid = 0.5 * (lo + hi), fm = eval(mid);
        if ((fm < 0) == (flo < 0)) { lo = mid; flo = fm; } else { hi = mid; fhi = fm; }
      }
      roots->push_back(0.5 * (lo + hi));
    }
    if (!knots.empty()) {
      const double fb = eval(knots.back());
      if (std::abs(fb) <= ftol && (roots->empty() || knots.back() > roots->back())) roots->push_back(knots.back());
    }
  }

  // Candidates for the extrema of the derivative-th derivative on [t_start, t_end]: both interval ends plus the real
  // roots of the (derivative+1)-th derivative inside the interval.  derivative = -1: roots of the polynomial itself.
  bool computeMinMaxCandidates(double t_start, double t_end, int derivative, std::vector<double>* candidates) const {
    CHECK_NOTNULL(candidates);
    candidates->clear();
    if (N_ - derivative - 1 < 0 || t_start > t_end) return false;
    const Eigen::VectorXd dc = getCoefficients(derivative + 1);
    std::vector<double> c(dc.size()), roots;
    for (int i = 0; i < dc.size(); ++i) c[i] = dc[i];
    realRootsInInterval(c, t_start, t_end, &roots);
    candidates->push_back(t_start);
    candidates->push_back(t_end);
    for (double r : roots) candidates->push_back(r);
    return true;
  }
  bool selectMinMaxFromCandidates(const std::vector<double>& candidates, int derivative, std::pair<double, double>* minimum,
                                  std::pair<double, double>* maximum) const {
    CHECK_NOTNULL(minimum);
    CHECK_NOTNULL(maximum);
    if (candidates.empty()) return false;
    *minimum = std::make_pair(candidates[0], std::numeric_limits<double>::max());
    *maximum = std::make_pair(candidates[0], std::numeric_limits<double>::lowest());
    for (double t : candidates) {
      const double v = evaluate(t, derivative);
      if (v < minimum->second) *minimum = std::make_pair(t, v);
      if (v > maximum->second) *maximum = std::make_pair(t, v);
    }
    return true;
  }
  bool computeMinMax(double t_start, double t_end, int derivative, std::pair<double, double>* minimum,
                     std::pair<double, double>* maximum) const {
    std::vector<double> candidates;
    if (!computeMinMaxCandidates(t_start, t_end, derivative, &candidates)) return false;
    return selectMinMaxFromCandidates(candidates, derivative, minimum, maximum);
  }

 private:
  int N_;
  Eigen::VectorXd coefficients_;
};

}  // namespace mav_trajectory_generation
#endif
