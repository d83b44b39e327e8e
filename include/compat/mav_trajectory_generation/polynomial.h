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

  // The derivative-coefficient table as the reference publishes it (polynomial.h:50: a static kMaxConvolutionSize-square
  // matrix, row n = coefficients of the n-th derivative of sum t^i); the veneer's own code calls baseCoefficient() below.
  struct BaseCoefficientTable : Eigen::MatrixXd {
    BaseCoefficientTable() : Eigen::MatrixXd(kMaxConvolutionSize, kMaxConvolutionSize) {
      for (int n = 0; n < kMaxConvolutionSize; ++n)
        for (int i = 0; i < kMaxConvolutionSize; ++i) (*this)(n, i) = baseCoefficient(n, i);
    }
  };
  static inline const BaseCoefficientTable base_coefficients_{};

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
  // product of two polynomials (reference: polynomial.h:91-95, operator*)
  Polynomial operator*(const Polynomial& rhs) const { return Polynomial(convolve(coefficients_, rhs.coefficients_)); }

  // All real roots of sum c[i] t^i inside [a, b].  Same method as the device code (csrc/mtg_extrema_lane.h): walk the
  // derivative chain upwards -- between two consecutive roots of the (k+1)-th derivative the k-th derivative is monotone, so
  // every sign change brackets exactly one root, found by bisection-safeguarded Newton.  The chain is walked in t itself:
  // the device maps a segment's [0, T] to [0, 1], a pure scaling, but a general interval [a, b] with a != 0 would need a
  // Taylor shift of the monomial coefficients, which cancels catastrophically for |a| >> 1 (degree 11 on [-100, 100]:
  // shifted coefficients ~1e26 for values ~1e2 -- found by the reference's own test/test_polynomial.cpp:FindMinMax run
  // against this header, tests/ref_tests).  Only exactly-zero leading coefficients are stripped (the reference strips
  // |c| < DBL_MIN, rpoly_ak1.cpp:57-68); identically zero polynomials have no isolated roots.
  static void realRootsInInterval(std::vector<double> c, double a, double b, std::vector<double>* roots) {
    roots->clear();
    while (!c.empty() && c.back() == 0.0) c.pop_back();
    if (c.size() < 2 || !(a <= b)) return;
    const int m = (int)c.size() - 1;
    if (b - a == 0.0) return;
    auto horner = [](const std::vector<double>& p, double x) {
      double r = 0.0;
      for (size_t i = p.size(); i-- > 0;) r = r * x + p[i];
      return r;
    };
    auto bracketed = [](const std::vector<double>& p, double lo, double hi, double flo, double fhi) {
      auto eval2 = [&p](double x, double* f, double* df) {
        *f = 0.0; *df = 0.0;
        for (size_t i = p.size(); i-- > 0;) { *df = *df * x + *f; *f = *f * x + p[i]; }
      };
      double xl = flo < 0.0 ? lo : hi, xh = flo < 0.0 ? hi : lo;
      double x = lo - flo * (hi - lo) / (fhi - flo);
      if (!(x > lo && x < hi)) x = 0.5 * (lo + hi);
      double dxold = std::abs(hi - lo), dx = dxold, f, df;
      eval2(x, &f, &df);
      for (int it = 0; it < 200; ++it) {
        const bool leaves = ((x - xh) * df - f) * ((x - xl) * df - f) > 0.0;
        const bool slow = std::abs(2.0 * f) > std::abs(dxold * df);
        dxold = dx;
        if (leaves || slow || !(df != 0.0)) { dx = 0.5 * (xh - xl); x = xl + dx; } else { dx = f / df; x -= dx; }
        if (std::abs(dx) < 4e-15 * std::max(1.0, std::abs(x))) break;
        eval2(x, &f, &df);
        if (f < 0.0) xl = x; else xh = x;
      }
      return x;
    };
    std::vector<double> part, next, lvl;
    for (int k = 1; k <= m; ++k) {
      // (m-k)-th divided derivative of the polynomial: lvl[j] = c[j+s] * C(j+s, s), s = m - k
      const int s = m - k;
      lvl.assign(k + 1, 0.0);
      double binom = 1.0;
      for (int j = 0; j <= k; ++j) {
        if (j > 0) binom = binom * (double)(j + s) / (double)j;
        lvl[j] = c[j + s] * binom;
      }
      next.clear();
      double lo = a, flo = horner(lvl, a);
      for (size_t i = 0; i <= part.size(); ++i) {
        const double hi = i < part.size() ? part[i] : b;
        const double fhi = horner(lvl, hi);
        if ((flo < 0.0) != (fhi < 0.0)) next.push_back(bracketed(lvl, lo, hi, flo, fhi));
        lo = hi;
        flo = fhi;
      }
      part.swap(next);
    }
    for (double r : part) roots->push_back(r);
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

  // p_out(t) = p(scaling_factor * t): coefficient n scaled by scaling_factor^n (src/polynomial.cpp:199-205).
  void scalePolynomialInTime(double scaling_factor) {
    double scale = 1.0;
    for (int n = 0; n < N_; ++n) {
      coefficients_[n] *= scale;
      scale *= scaling_factor;
    }
  }
  void offsetPolynomial(const double offset) {   // src/polynomial.cpp:207-211
    if (coefficients_.size() == 0) return;
    coefficients_[0] += offset;
  }

 private:
  int N_;
  Eigen::VectorXd coefficients_;
};

}  // namespace mav_trajectory_generation
#endif
