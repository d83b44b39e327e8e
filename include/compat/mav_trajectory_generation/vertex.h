// Compat veneer: Vertex (constraint map per derivative) and the input generators that define the reference's
// synthetic workloads (reference: vertex.h:42-165, src/vertex.cpp:27-82,130-163,228-290).  Header-only.
#ifndef MAV_TRAJECTORY_GENERATION_VERTEX_H_
#define MAV_TRAJECTORY_GENERATION_VERTEX_H_
#include <cmath>
#include <map>
#include <random>
#include <vector>

#include "motion_defines.h"
#include "mtg_compat_base.h"

namespace mav_trajectory_generation {

class Vertex {
 public:
  typedef std::vector<Vertex> Vector;
  typedef Eigen::VectorXd ConstraintValue;
  typedef std::pair<int, ConstraintValue> Constraint;
  typedef std::map<int, ConstraintValue> Constraints;

  explicit Vertex(size_t dimension) : D_((int)dimension) {}
  int D() const { return D_; }

  void addConstraint(int derivative_order, double value) { constraints_[derivative_order] = ConstraintValue::Constant(D_, value); }
  void addConstraint(int derivative_order, const Eigen::VectorXd& constraint) {
    CHECK_EQ((long)constraint.rows(), (long)D_);
    constraints_[derivative_order] = constraint;
  }
  bool removeConstraint(int type) { return constraints_.erase(type) > 0; }

  // position = constraint, derivatives 1..up_to_derivative = 0 (start / goal at rest)
  void makeStartOrEnd(const Eigen::VectorXd& constraint, int up_to_derivative) {
    addConstraint(derivative_order::POSITION, constraint);
    for (int i = 1; i <= up_to_derivative; ++i) constraints_[i] = ConstraintValue::Zero(D_);
  }
  void makeStartOrEnd(double value, int up_to_derivative) { makeStartOrEnd(Eigen::VectorXd::Constant(D_, value), up_to_derivative); }

  bool hasConstraint(int derivative_order) const { return constraints_.count(derivative_order) > 0; }
  bool getConstraint(int derivative_order, Eigen::VectorXd* constraint) const {
    CHECK_NOTNULL(constraint);
    Constraints::const_iterator it = constraints_.find(derivative_order);
    if (it == constraints_.end()) return false;
    *constraint = it->second;
    return true;
  }
  Constraints::const_iterator cBegin() const { return constraints_.begin(); }
  Constraints::const_iterator cEnd() const { return constraints_.end(); }
  size_t getNumberOfConstraints() const { return constraints_.size(); }

 private:
  int D_;
  Constraints constraints_;
};

// Uniform waypoints in a box, consecutive spacing > 0.2, one std::mt19937(seed), ends at rest.
// highest derivative a vertex of a polynomial with N coefficients can constrain (reference: vertex.h:147)
inline int getHighestDerivativeFromN(int N) { return N / 2 - 1; }

inline Vertex::Vector createRandomVertices(int maximum_derivative, size_t n_segments, const Eigen::VectorXd& pos_min,
                                           const Eigen::VectorXd& pos_max, size_t seed = 0) {
  CHECK_GE((int)n_segments, 1);
  CHECK_EQ(pos_min.size(), pos_max.size());
  CHECK_GE((pos_max - pos_min).norm(), 0.2);
  CHECK_GT(maximum_derivative, 0);
  const size_t dim = pos_min.size();
  std::mt19937 generator(seed);
  std::vector<std::uniform_real_distribution<double>> dist;
  for (size_t i = 0; i < dim; ++i) dist.emplace_back(pos_min[i], pos_max[i]);
  Eigen::VectorXd last(dim);
  for (size_t i = 0; i < dim; ++i) last[i] = dist[i](generator);
  Vertex::Vector vertices;
  vertices.reserve(n_segments + 1);
  vertices.push_back(Vertex(dim));
  vertices.front().makeStartOrEnd(last, maximum_derivative);
  for (size_t v = 1; v <= n_segments; ++v) {
    Eigen::VectorXd pos(dim);
    do {
      for (size_t d = 0; d < dim; ++d) pos[d] = dist[d](generator);
    } while (!((pos - last).norm() > 0.2));
    Vertex vert(dim);
    vert.addConstraint(derivative_order::POSITION, pos);
    vertices.push_back(vert);
    last = pos;
  }
  vertices.back().makeStartOrEnd(last, maximum_derivative);
  return vertices;
}

inline Vertex::Vector createRandomVertices1D(int maximum_derivative, size_t n_segments, double pos_min, double pos_max,
                                             size_t seed = 0) {
  return createRandomVertices(maximum_derivative, n_segments, Eigen::VectorXd::Constant(1, pos_min),
                              Eigen::VectorXd::Constant(1, pos_max), seed);
}

// t = 2 dist / v * (1 + c * v / a * exp(-2 dist / v))
inline std::vector<double> estimateSegmentTimesNfabian(const Vertex::Vector& vertices, double v_max, double a_max,
                                                       double magic_fabian_constant = 6.5) {
  CHECK_GE((int)vertices.size(), 2);
  std::vector<double> times;
  times.reserve(vertices.size() - 1);
  for (size_t i = 0; i + 1 < vertices.size(); ++i) {
    Eigen::VectorXd start, end;
    vertices[i].getConstraint(derivative_order::POSITION, &start);
    vertices[i + 1].getConstraint(derivative_order::POSITION, &end);
    const double distance = (end - start).norm();
    times.push_back(distance / v_max * 2 * (1.0 + magic_fabian_constant * v_max / a_max * std::exp(-distance / v_max * 2)));
  }
  return times;
}

inline double computeTimeVelocityRamp(const Eigen::VectorXd& start, const Eigen::VectorXd& goal, double v_max, double a_max) {
  const double distance = (start - goal).norm();
  const double acc_time = v_max / a_max, acc_distance = 0.5 * v_max * acc_time;
  return distance < 2.0 * acc_distance ? 2.0 * std::sqrt(distance / a_max) : 2.0 * acc_time + (distance - 2.0 * acc_distance) / v_max;
}

inline std::vector<double> estimateSegmentTimesVelocityRamp(const Vertex::Vector& vertices, double v_max, double a_max,
                                                            double time_factor = 1.0) {
  (void)time_factor;
  CHECK_GE((int)vertices.size(), 2);
  std::vector<double> times;
  for (size_t i = 0; i + 1 < vertices.size(); ++i) {
    Eigen::VectorXd start, end;
    vertices[i].getConstraint(derivative_order::POSITION, &start);
    vertices[i + 1].getConstraint(derivative_order::POSITION, &end);
    times.push_back(std::max(0.1, computeTimeVelocityRamp(start, end, v_max, a_max)));
  }
  return times;
}

inline std::vector<double> estimateSegmentTimes(const Vertex::Vector& vertices, double v_max, double a_max) {
  return estimateSegmentTimesNfabian(vertices, v_max, a_max);
}

}  // namespace mav_trajectory_generation
#endif
