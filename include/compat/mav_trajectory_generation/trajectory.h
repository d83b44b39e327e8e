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

  // Largest velocity and acceleration magnitude over the whole trajectory, all dimensions (reference API: trajectory.h:118).
  bool computeMaxVelocityAndAcceleration(double* v_max, double* a_max) const {
    CHECK_NOTNULL(v_max);
    CHECK_NOTNULL(a_max);
    std::vector<int> all_dimensions(D_);
    std::iota(all_dimensions.begin(), all_dimensions.end(), 0);
    bool ok = true;
    auto peak = [&](int derivative) {
      Extremum lowest, highest;
      ok = computeMinMaxMagnitude(derivative, all_dimensions, &lowest, &highest) && ok;
      return highest.value;
    };
    *v_max = peak(derivative_order::VELOCITY);
    *a_max = peak(derivative_order::ACCELERATION);
    return ok;
  }

  // Every segment time multiplied by `factor`, the polynomials re-parametrised so that the path stays the same
  // (p_new(t) = p(t / factor); reference API: trajectory.h:123).
  bool scaleSegmentTimes(double factor) {
    if (factor < 1.0e-6) return false;
    stretch(factor);
    return true;
  }

  // Stretches the trajectory in time until |v| <= v_max and |a| <= a_max hold to 1e-3 relative, at most 20 check / stretch rounds
  // (the reference's contract, trajectory.h:128-131).  Same scheme as the device path (csrc/mtg_extrema.hip, mtg_scale_loop): a
  // stretch by s re-parametrises p(t / s), so the maxima of the next round are exactly v / s and a / s^2 -- ONE root search, the
  // rounds carried out on those two numbers, the accumulated factor applied once.  (The reference searches the roots again in
  // every round and reproduces the same numbers up to round-off.)
  bool scaleSegmentTimesToMeetConstraints(double v_max, double a_max) {
    double v = 0.0, a = 0.0;
    computeMaxVelocityAndAcceleration(&v, &a);
    double total = 1.0;
    bool feasible = false;
    for (int round = 0; round < 20 && !feasible; ++round) {
      const double over_v = v / v_max, over_a = a / a_max;
      feasible = over_v <= 1.0 + 1e-3 && over_a <= 1.0 + 1e-3;
      if (feasible) break;
      const double s = std::max(1.0, std::max(over_v, std::sqrt(over_a)));
      total *= s;
      v /= s;
      a /= s * s;
    }
    if (total != 1.0) stretch(total);
    return feasible;
  }

 private:
  void stretch(double factor) {
    const double inverse = 1.0 / factor;
    max_time_ = 0.0;
    for (Segment& segment : segments_) {
      for (int d = 0; d < segment.D(); ++d) segment[d].scalePolynomialInTime(inverse);
      segment.setTime(segment.getTime() * factor);
      max_time_ += segment.getTime();
    }
  }

  int D_, N_;
  double max_time_;
  Segment::Vector segments_;
};

}  // namespace mav_trajectory_generation
#endif
