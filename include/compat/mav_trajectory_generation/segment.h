// Compat veneer: Segment = D polynomials + duration (reference: segment.h:43-128, src/segment.cpp:41-81).
// plus the magnitude-extremum candidate helpers (src/segment.cpp:83-159) used by computeMaximumOfMagnitude.
#ifndef MAV_TRAJECTORY_GENERATION_SEGMENT_H_
#define MAV_TRAJECTORY_GENERATION_SEGMENT_H_
#include <cstdint>
#include <limits>
#include <vector>

#include "extremum.h"
#include "motion_defines.h"
#include "polynomial.h"

namespace mav_trajectory_generation {

class Segment {
 public:
  typedef std::vector<Segment> Vector;
  Segment(int N, int D) : time_(0.0), N_(N), D_(D) { polynomials_.resize(D_, Polynomial(N_)); }

  bool operator==(const Segment& rhs) const { return D_ == rhs.D_ && time_ == rhs.time_ && polynomials_ == rhs.polynomials_; }
  bool operator!=(const Segment& rhs) const { return !(*this == rhs); }
  int D() const { return D_; }
  int N() const { return N_; }
  double getTime() const { return time_; }
  uint64_t getTimeNSec() const { return static_cast<uint64_t>(1.0e9 * time_); }
  void setTime(double time_sec) { time_ = time_sec; }
  void setTimeNSec(uint64_t time_ns) { time_ = time_ns * 1.0e-9; }

  Polynomial& operator[](size_t idx) { CHECK_LT((int)idx, D_); return polynomials_[idx]; }
  const Polynomial& operator[](size_t idx) const { CHECK_LT((int)idx, D_); return polynomials_[idx]; }
  const Polynomial::Vector& getPolynomialsRef() const { return polynomials_; }

  Eigen::VectorXd evaluate(double t, int derivative = derivative_order::POSITION) const {
    Eigen::VectorXd result(D_);
    for (int d = 0; d < D_; ++d) result[d] = polynomials_[d].evaluate(t, derivative);
    return result;
  }

  // Times where |p^(derivative)(t)| over `dimensions` can be extremal on [t_start, t_end]: interval ends and the real
  // roots of d/dt |.|^2 = 2 sum_i p_i^(d) p_i^(d+1) (a convolution per dimension); one dimension: roots of p^(d+1).
  bool computeMinMaxMagnitudeCandidateTimes(int derivative, double t_start, double t_end, const std::vector<int>& dimensions,
                                            std::vector<double>* candidate_times) const {
    CHECK_NOTNULL(candidate_times);
    candidate_times->clear();
    if (dimensions.empty()) return false;
    for (int dim : dimensions) if (dim < 0 || dim >= D_) return false;
    if (dimensions.size() == 1) return polynomials_[dimensions[0]].computeMinMaxCandidates(t_start, t_end, derivative, candidate_times);
    const int n_d = N_ - derivative, n_dd = n_d - 1;
    Eigen::VectorXd conv = Eigen::VectorXd::Zero(Polynomial::getConvolutionLength(n_d, n_dd));
    for (int dim : dimensions) {
      const Eigen::VectorXd full_d = polynomials_[dim].getCoefficients(derivative);
      const Eigen::VectorXd full_dd = polynomials_[dim].getCoefficients(derivative + 1);
      Eigen::VectorXd d(n_d), dd(n_dd);
      for (int i = 0; i < n_d; ++i) d[i] = full_d[i];
      for (int i = 0; i < n_dd; ++i) dd[i] = full_dd[i];
      const Eigen::VectorXd c = Polynomial::convolve(d, dd);
      for (int i = 0; i < conv.size(); ++i) conv[i] += c[i];
    }
    return Polynomial(conv).computeMinMaxCandidates(t_start, t_end, -1, candidate_times);
  }
  bool computeMinMaxMagnitudeCandidates(int derivative, double t_start, double t_end, const std::vector<int>& dimensions,
                                        std::vector<Extremum>* candidates) const {
    CHECK_NOTNULL(candidates);
    std::vector<double> times;
    if (!computeMinMaxMagnitudeCandidateTimes(derivative, t_start, t_end, dimensions, &times)) return false;
    candidates->clear();
    for (double t : times) {
      double m = 0.0;
      for (int dim : dimensions) { const double v = polynomials_[dim].evaluate(t, derivative); m += v * v; }
      candidates->push_back(Extremum(t, std::sqrt(m), 0));
    }
    return true;
  }
  // segment.cpp:146-184: strict </> over the candidates inside [t_start, t_end] (std::min/std::max on Extremum)
  bool selectMinMaxMagnitudeFromCandidates(int /*derivative*/, double t_start, double t_end,
                                           const std::vector<int>& /*dimensions*/, const std::vector<Extremum>& candidates,
                                           Extremum* minimum, Extremum* maximum) const {
    CHECK_NOTNULL(minimum);
    CHECK_NOTNULL(maximum);
    if (t_start > t_end) return false;
    minimum->value = std::numeric_limits<double>::max();
    maximum->value = std::numeric_limits<double>::lowest();
    for (const Extremum& candidate : candidates) {
      if (candidate.time < t_start || candidate.time > t_end) continue;
      if (*maximum < candidate) *maximum = candidate;
      if (candidate < *minimum) *minimum = candidate;
    }
    return true;
  }

 protected:
  Polynomial::Vector polynomials_;
  double time_;

 private:
  int N_, D_;
};

// Text form of a segment (reference: segment.h:133-138): its time and, per dimension, the coefficients of the given derivative.
inline void printSegment(std::ostream& stream, const Segment& s, int derivative) {
  CHECK(derivative >= 0 && derivative < s.N());
  stream << "t: " << s.getTime() << std::endl << " coefficients for " << positionDerivativeToString(derivative) << ": " << std::endl;
  for (int d = 0; d < s.D(); ++d) {
    const Eigen::VectorXd c = s[d].getCoefficients(derivative);
    stream << "dim " << d << ": " << std::endl;
    for (int i = 0; i < (int)c.size(); ++i) stream << (i ? " " : "") << c[i];
    stream << std::endl;
  }
}
inline std::ostream& operator<<(std::ostream& stream, const Segment& s) {
  printSegment(stream, s, derivative_order::POSITION);
  return stream;
}
inline std::ostream& operator<<(std::ostream& stream, const std::vector<Segment>& segments) {
  for (const Segment& s : segments) stream << s << std::endl;
  return stream;
}

}  // namespace mav_trajectory_generation
#endif
