// Compat veneer: minimal Trajectory = ordered list of segments (reference: trajectory.h:30-154): what
// PolynomialOptimization::getTrajectory() hands over.  Analysis helpers beyond evaluate() are out of scope.
#ifndef MAV_TRAJECTORY_GENERATION_TRAJECTORY_H_
#define MAV_TRAJECTORY_GENERATION_TRAJECTORY_H_
#include <vector>

#include "segment.h"

namespace mav_trajectory_generation {

class Trajectory {
 public:
  Trajectory() : D_(0), N_(0), max_time_(0.0) {}
  int D() const { return D_; }
  int N() const { return N_; }
  int K() const { return (int)segments_.size(); }
  bool empty() const { return segments_.empty(); }
  void clear() { segments_.clear(); D_ = N_ = 0; max_time_ = 0.0; }

  void setSegments(const Segment::Vector& segments) {
    CHECK(!segments.empty());
    D_ = segments.front().D();
    N_ = segments.front().N();
    max_time_ = 0.0;
    segments_.clear();
    addSegments(segments);
  }
  void addSegments(const Segment::Vector& segments) {
    for (const Segment& s : segments) {
      CHECK_EQ(s.D(), D_);
      CHECK_EQ(s.N(), N_);
      max_time_ += s.getTime();
    }
    segments_.insert(segments_.end(), segments.begin(), segments.end());
  }
  void getSegments(Segment::Vector* segments) const { CHECK_NOTNULL(segments); *segments = segments_; }
  const Segment::Vector& segments() const { return segments_; }
  double getMinTime() const { return 0.0; }
  double getMaxTime() const { return max_time_; }
  std::vector<double> getSegmentTimes() const {
    std::vector<double> t;
    for (const Segment& s : segments_) t.push_back(s.getTime());
    return t;
  }
  // value of the derivative at absolute time t (clamped to the last segment's end)
  Eigen::VectorXd evaluate(double t, int derivative_order = derivative_order::POSITION) const {
    CHECK(!segments_.empty());
    size_t i = 0;
    double acc = 0.0;
    while (i + 1 < segments_.size() && t > acc + segments_[i].getTime()) acc += segments_[i++].getTime();
    const double local = std::min(t - acc, segments_[i].getTime());
    return segments_[i].evaluate(local, derivative_order);
  }

 private:
  int D_, N_;
  double max_time_;
  Segment::Vector segments_;
};

}  // namespace mav_trajectory_generation
#endif
