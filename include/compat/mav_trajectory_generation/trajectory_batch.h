// Compat veneer, new API: TrajectoryBatch -- a device-resident batch of trajectories sharing (N, K, D), with the
// batched forms of the reference's post-solve Trajectory functions running on the MI355X through the C ABI
// (include/mtg_hip.h): evaluateRange / sampleTrajectoryInRange -> sample(), computeMinMaxMagnitude,
// computeMaxVelocityAndAcceleration, scaleSegmentTimesToMeetConstraints (src/trajectory.cpp:81-141, :190-227,
// :343-361, :385-429).  Host code only: device memory goes through mtg_device_malloc / mtg_copy_*.
// The single-trajectory functions stay on the host (trajectory.h); this class is for B >> 1.
#ifndef MAV_TRAJECTORY_GENERATION_TRAJECTORY_BATCH_H_
#define MAV_TRAJECTORY_GENERATION_TRAJECTORY_BATCH_H_
#include <cstdint>
#include <vector>

#include "../../mtg_hip.h"
#include "extremum.h"
#include "polynomial_optimization_linear.h"   // per-thread context
#include "trajectory.h"

namespace mav_trajectory_generation {

class TrajectoryBatch {
 public:
  explicit TrajectoryBatch(const std::vector<Trajectory>& trajectories) { upload(trajectories); }
  TrajectoryBatch(const TrajectoryBatch&) = delete;              // owns device memory
  TrajectoryBatch& operator=(const TrajectoryBatch&) = delete;
  ~TrajectoryBatch() { release(); }

  size_t size() const { return (size_t)B_; }
  int N() const { return N_; }
  int K() const { return K_; }
  int D() const { return D_; }

  // Per-trajectory extrema of the derivative's magnitude over `dimensions` (Trajectory::computeMinMaxMagnitude).
  bool computeMinMaxMagnitude(int derivative, const std::vector<int>& dimensions, std::vector<Extremum>* minima,
                              std::vector<Extremum>* maxima) {
    CHECK_NOTNULL(minima);
    CHECK_NOTNULL(maxima);
    if (dimensions.empty()) return false;                        // "No dimensions specified." (segment.cpp:90-93)
    uint32_t mask = 0;
    for (int dim : dimensions) {
      if (dim < 0 || dim >= D_) return false;                    // out of bounds (segment.cpp:102-107)
      mask |= 1u << dim;
    }
    if (N_ - derivative - 1 < 0) return false;                   // polynomial.cpp:70-73
    double* seg = (double*)dmalloc(sizeof(double) * 4 * B_ * K_);
    double* traj = (double*)dmalloc(sizeof(double) * 4 * B_);
    int32_t* idx = (int32_t*)dmalloc(sizeof(int32_t) * 2 * B_);
    check(mtg_minmax_magnitude(ctx(), N_, K_, D_, B_, coeffs_, times_, K_, 1, derivative, mask, seg, traj, idx));
    std::vector<double> h(4 * B_);
    std::vector<int32_t> hi(2 * B_);
    check(mtg_copy_to_host(ctx(), h.data(), traj, sizeof(double) * h.size()));
    check(mtg_copy_to_host(ctx(), hi.data(), idx, sizeof(int32_t) * hi.size()));
    mtg_device_free(ctx(), seg);
    mtg_device_free(ctx(), traj);
    mtg_device_free(ctx(), idx);
    minima->resize(B_);
    maxima->resize(B_);
    for (int64_t b = 0; b < B_; ++b) {
      (*minima)[b] = Extremum(h[4 * b + 0], h[4 * b + 1], hi[2 * b + 0]);
      (*maxima)[b] = Extremum(h[4 * b + 2], h[4 * b + 3], hi[2 * b + 1]);
    }
    return true;
  }

  bool computeMaxVelocityAndAcceleration(std::vector<double>* v_max, std::vector<double>* a_max) {
    CHECK_NOTNULL(v_max);
    CHECK_NOTNULL(a_max);
    std::vector<int> dimensions(D_);
    for (int d = 0; d < D_; ++d) dimensions[d] = d;
    std::vector<Extremum> mn, mx;
    bool ok = computeMinMaxMagnitude(derivative_order::VELOCITY, dimensions, &mn, &mx);
    v_max->resize(B_);
    for (int64_t b = 0; b < B_ && ok; ++b) (*v_max)[b] = mx[b].value;
    ok = ok && computeMinMaxMagnitude(derivative_order::ACCELERATION, dimensions, &mn, &mx);
    a_max->resize(B_);
    for (int64_t b = 0; b < B_ && ok; ++b) (*a_max)[b] = mx[b].value;
    return ok;
  }

  // In place on the device copy; returns true when every trajectory ended within range.
  bool scaleSegmentTimesToMeetConstraints(double v_max, double a_max, std::vector<char>* within_range = nullptr,
                                          std::vector<double>* scaling = nullptr) {
    double* ws = (double*)dmalloc(sizeof(double) * 8 * B_ * (K_ + 1));
    double* sc = (double*)dmalloc(sizeof(double) * B_);
    int32_t* within = (int32_t*)dmalloc(sizeof(int32_t) * B_);
    check(mtg_scale_segment_times_to_meet_constraints(ctx(), N_, K_, D_, B_, coeffs_, times_, K_, 1, v_max, a_max,
                                                      /*max_iterations=*/2, ws, sc, within));
    std::vector<int32_t> h(B_);
    check(mtg_copy_to_host(ctx(), h.data(), within, sizeof(int32_t) * h.size()));
    if (scaling) {
      scaling->resize(B_);
      check(mtg_copy_to_host(ctx(), scaling->data(), sc, sizeof(double) * B_));
    }
    mtg_device_free(ctx(), ws);
    mtg_device_free(ctx(), sc);
    mtg_device_free(ctx(), within);
    bool all = true;
    if (within_range) within_range->resize(B_);
    for (int64_t b = 0; b < B_; ++b) {
      all = all && h[b] != 0;
      if (within_range) (*within_range)[b] = (char)(h[b] != 0);
    }
    return all;
  }

  // out[b][i][derivative][dim] at t_start + i*dt, i < n_samples; samples past a trajectory's end are evaluated at
  // its end and excluded from n_valid[b] (the reference's loops simply stop there, src/trajectory.cpp:121-140).
  void sample(double t_start, double dt, int n_samples, int n_derivatives, std::vector<double>* out,
              std::vector<int>* n_valid = nullptr) {
    CHECK_NOTNULL(out);
    const size_t n = (size_t)B_ * n_samples * n_derivatives * D_;
    double* d_out = (double*)dmalloc(sizeof(double) * n);
    int32_t* d_valid = n_valid ? (int32_t*)dmalloc(sizeof(int32_t) * B_) : nullptr;
    check(mtg_sample_range(ctx(), N_, K_, D_, B_, coeffs_, times_, K_, 1, t_start, dt, n_samples, n_derivatives, d_out,
                           d_valid));
    out->resize(n);
    check(mtg_copy_to_host(ctx(), out->data(), d_out, sizeof(double) * n));
    if (n_valid) {
      std::vector<int32_t> h(B_);
      check(mtg_copy_to_host(ctx(), h.data(), d_valid, sizeof(int32_t) * B_));
      n_valid->assign(h.begin(), h.end());
      mtg_device_free(ctx(), d_valid);
    }
    mtg_device_free(ctx(), d_out);
  }

  void download(std::vector<Trajectory>* trajectories) const {
    CHECK_NOTNULL(trajectories);
    std::vector<double> c((size_t)B_ * K_ * D_ * N_), t((size_t)B_ * K_);
    check(mtg_copy_to_host(ctx(), c.data(), coeffs_, sizeof(double) * c.size()));
    check(mtg_copy_to_host(ctx(), t.data(), times_, sizeof(double) * t.size()));
    trajectories->assign(B_, Trajectory());
    for (int64_t b = 0; b < B_; ++b) {
      Segment::Vector segments(K_, Segment(N_, D_));
      for (int k = 0; k < K_; ++k) {
        segments[k].setTime(t[b * K_ + k]);
        for (int d = 0; d < D_; ++d) {
          Eigen::VectorXd v(N_);
          for (int n = 0; n < N_; ++n) v[n] = c[((b * K_ + k) * D_ + d) * N_ + n];
          segments[k][d] = Polynomial(N_, v);
        }
      }
      (*trajectories)[b].setSegments(segments);
    }
  }

 private:
  static mtg_context* ctx() { return mtg_compat_detail::context(); }
  static void check(int rc) { CHECK(rc == MTG_OK) << mtg_status_string(rc) << " " << mtg_last_error_string(ctx()); }
  static void* dmalloc(size_t bytes) {
    void* p = nullptr;
    check(mtg_device_malloc(ctx(), bytes, &p));
    return p;
  }
  void upload(const std::vector<Trajectory>& trajectories) {
    CHECK(!trajectories.empty());
    B_ = (int64_t)trajectories.size();
    N_ = trajectories[0].N();
    K_ = trajectories[0].K();
    D_ = trajectories[0].D();
    std::vector<double> c((size_t)B_ * K_ * D_ * N_), t((size_t)B_ * K_);
    for (int64_t b = 0; b < B_; ++b) {
      const Trajectory& tr = trajectories[b];
      CHECK(tr.N() == N_ && tr.K() == K_ && tr.D() == D_) << "all trajectories of a batch share (N, K, D)";
      for (int k = 0; k < K_; ++k) {
        const Segment& s = tr.segments()[k];
        t[b * K_ + k] = s.getTime();
        for (int d = 0; d < D_; ++d) {
          const Eigen::VectorXd v = s[d].getCoefficients();
          for (int n = 0; n < N_; ++n) c[((b * K_ + k) * D_ + d) * N_ + n] = v[n];
        }
      }
    }
    coeffs_ = (double*)dmalloc(sizeof(double) * c.size());
    times_ = (double*)dmalloc(sizeof(double) * t.size());
    check(mtg_copy_to_device(ctx(), coeffs_, c.data(), sizeof(double) * c.size()));
    check(mtg_copy_to_device(ctx(), times_, t.data(), sizeof(double) * t.size()));
  }
  void release() {
    if (coeffs_) mtg_device_free(ctx(), coeffs_);
    if (times_) mtg_device_free(ctx(), times_);
    coeffs_ = times_ = nullptr;
  }

  int64_t B_ = 0;
  int N_ = 0, K_ = 0, D_ = 0;
  double* coeffs_ = nullptr;   // device [B][K][D][N]
  double* times_ = nullptr;    // device [B][K]
};

}  // namespace mav_trajectory_generation
#endif
