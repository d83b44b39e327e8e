// Compat veneer: Segment = D polynomials + duration (reference: segment.h:43-128, src/segment.cpp:41-81).
// Extremum / min-max helpers are post-solve analysis and out of scope.
#ifndef MAV_TRAJECTORY_GENERATION_SEGMENT_H_
#define MAV_TRAJECTORY_GENERATION_SEGMENT_H_
#include <cstdint>
#include <vector>

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

 protected:
  Polynomial::Vector polynomials_;
  double time_;

 private:
  int N_, D_;
};

}  // namespace mav_trajectory_generation
#endif
