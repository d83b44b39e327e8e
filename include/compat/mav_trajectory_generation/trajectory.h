// Compat veneer: Trajectory = ordered list of segments (reference: trajectory.h:30-154): what
// PolynomialOptimization::getTrajectory() hands over, plus the analysis helpers on the way to a feasible
// trajectory: evaluateRange, computeMinMaxMagnitude, computeMaxVelocityAndAcceleration, scaleSegmentTimes and
// scaleSegmentTimesToMeetConstraints (src/trajectory.cpp:81-141, :190-227, :343-429) -- host code, one trajectory.
// The batched, device-side forms of the same functions are in trajectory_batch.h.
#ifndef MAV_TRAJECTORY_GENERATION_TRAJECTORY_H_
#define MAV_TRAJECTORY_GENERATION_TRAJECTORY_H_
#include <algorithm>
#include <cmath>
#include <limits>
#include <numeric>
#include <vector>

#include "extremum.h"
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
  // Value of the derivative at absolute time t, with the reference's segment choice (src/trajectory.cpp:48-79): a time
  // that falls exactly on a vertex belongs to the segment RIGHT of it (derivatives of order >= N/2 jump there, and the
  // device sampler makes the same choice); the trajectory's end time evaluates the last segment at its duration; a time
  // beyond the end is an error in the reference (LOG(ERROR)) and yields the zero vector.
  Eigen::VectorXd evaluate(double t, int derivative_order = derivative_order::POSITION) const {
    CHECK(!segments_.empty());
    double accumulated = 0.0;
    size_t i = 0;
    for (; i < segments_.size(); ++i) {
      accumulated += segments_[i].getTime();
      if (accumulated > t) break;
    }
    if (t > accumulated) {
      LOG(ERROR) << "Time out of range of the trajectory!";
      return Eigen::VectorXd::Zero(D());
    }
    if (i >= segments_.size()) i = segments_.size() - 1;
    accumulated -= segments_[i].getTime();
    return segments_[i].evaluate(t - accumulated, derivative_order);
  }

  // Samples derivative_order every dt from t_start while the running time is below t_end, walking the segments
  // with a local time that accumulates dt (src/trajectory.cpp:81-141).  Reference quirks kept: the running time
  // that is compared with t_end and reported in sampling_times starts at the START of the segment containing
  // t_start (identical to t_start only when t_start is a vertex time, e.g. 0), and a start time beyond the
  // trajectory yields no samples.
  void evaluateRange(double t_start, double t_end, double dt, int derivative_order, std::vector<Eigen::VectorXd>* result,
                     std::vector<double>* sampling_times = nullptr) const {
    CHECK_NOTNULL(result);
    result->clear();
    if (sampling_times) sampling_times->clear();
    if (segments_.empty()) return;
    size_t i = 0;
    double running = 0.0;
    for (; i < segments_.size(); ++i) {
      running += segments_[i].getTime();
      if (running > t_start) break;   // a start on a vertex belongs to the segment right of it
    }
    if (t_start > running) return;    // "Start time out of range of the trajectory!"
    if (i == segments_.size()) --i;   // t_start == end of the trajectory: last segment, local time = its duration
    running -= segments_[i].getTime();
    double local = t_start - running;
    while (running < t_end) {
      if (local > segments_[i].getTime()) {
        local -= segments_[i].getTime();
        if (++i >= segments_.size()) break;
        continue;
      }
      result->push_back(segments_[i].evaluate(local, derivative_order));
      if (sampling_times) sampling_times->push_back(running);
      local += dt;
      running += dt;
    }
  }

  // src/trajectory.cpp:190-227: first segment with the strictly smallest / largest magnitude wins.
  bool computeMinMaxMagnitude(int derivative, const std::vector<int>& dimensions, Extremum* minimum,
                              Extremum* maximum) const {
    CHECK_NOTNULL(minimum);
    CHECK_NOTNULL(maximum);
    minimum->value = std::numeric_limits<double>::max();
    maximum->value = std::numeric_limits<double>::lowest();
    for (size_t i = 0; i < segments_.size(); ++i) {
      std::vector<Extremum> candidates;
      if (!segments_[i].computeMinMaxMagnitudeCandidates(derivative, 0.0, segments_[i].getTime(), dimensions, &candidates))
        return false;
      Extremum mn, mx;
      if (!segments_[i].selectMinMaxMagnitudeFromCandidates(derivative, 0.0, segments_[i].getTime(), dimensions, candidates,
                                                            &mn, &mx))
        return false;
      if (mn < *minimum) { *minimum = mn; minimum->segment_idx = static_cast<int>(i); }
      if (mx > *maximum) { *maximum = mx; maximum->segment_idx = static_cast<int>(i); }
    }
    return true;
  }

  // src/trajectory.cpp:343-361
  bool computeMaxVelocityAndAcceleration(double* v_max, double* a_max) const {
    CHECK_NOTNULL(v_max);
    CHECK_NOTNULL(a_max);
    std::vector<int> dimensions(D_);
    std::iota(dimensions.begin(), dimensions.end(), 0);
    Extremum v_min_traj, v_max_traj, a_min_traj, a_max_traj;
    bool success = computeMinMaxMagnitude(derivative_order::VELOCITY, dimensions, &v_min_traj, &v_max_traj);
    success &= computeMinMaxMagnitude(derivative_order::ACCELERATION, dimensions, &a_min_traj, &a_max_traj);
    *v_max = v_max_traj.value;
    *a_max = a_max_traj.value;
    return success;
  }

  // src/trajectory.cpp:363-381
  bool scaleSegmentTimes(double scaling) {
    if (scaling < 1.0e-6) return false;
    double new_max_time = 0.0;
    const double scaling_inverse = 1.0 / scaling;
    for (Segment& s : segments_) {
      const double new_time = s.getTime() * scaling;
      for (int d = 0; d < s.D(); ++d) s[d].scalePolynomialInTime(scaling_inverse);
      s.setTime(new_time);
      new_max_time += new_time;
    }
    max_time_ = new_max_time;
    return true;
  }

  // src/trajectory.cpp:385-429: stretches all segment times by max(1, v/v_max, sqrt(a/a_max)) until both bounds
  // hold within 1e-3 relative (at most 20 rounds; one is enough, the second only verifies).
  bool scaleSegmentTimesToMeetConstraints(double v_max, double a_max) {
    constexpr size_t kMaxCounter = 20;
    constexpr double kTolerance = 1e-3;
    bool within_range = false;
    for (size_t i = 0; i < kMaxCounter; ++i) {
      double v_max_actual, a_max_actual;
      computeMaxVelocityAndAcceleration(&v_max_actual, &a_max_actual);
      const double velocity_violation = v_max_actual / v_max;
      const double acceleration_violation = a_max_actual / a_max;
      within_range = velocity_violation <= 1.0 + kTolerance && acceleration_violation <= 1.0 + kTolerance;
      if (within_range) break;
      const double violation_scaling = std::max(1.0, std::max(velocity_violation, std::sqrt(acceleration_violation)));
      const double violation_scaling_inverse = 1.0 / violation_scaling;
      double new_max_time = 0.0;
      for (Segment& s : segments_) {
        const double new_time = s.getTime() * violation_scaling;
        for (int d = 0; d < s.D(); ++d) s[d].scalePolynomialInTime(violation_scaling_inverse);
        s.setTime(new_time);
        new_max_time += new_time;
      }
      max_time_ = new_max_time;
    }
    return within_range;
  }

 private:
  int D_, N_;
  double max_time_;
  Segment::Vector segments_;
};

}  // namespace mav_trajectory_generation
#endif
