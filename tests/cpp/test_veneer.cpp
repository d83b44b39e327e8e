// Compat-veneer test program (runs on the GPU box; built by __graft_entry__.build()).  Mirrors the reference's
// own tests for the hot path through the reference's API:
//   TwoVerticesSetup (test_polynomial_optimization.cpp:743-787, MATLAB vector), the README example
//   (README.md:104-140), checkPath (:113-174) on random 3-D / 10-segment paths, ConstraintPacking (:505-564),
//   setFreeConstraints round trip, copy-assignment (time_evaluation_node.cpp:357) and the batched entry.
#include <cmath>
#include <cstdio>
#include <numeric>
#include <memory>
#include <thread>
#include <sstream>
#include <vector>

#include <mav_trajectory_generation/polynomial_optimization_linear.h>
#include <mav_trajectory_generation/trajectory_batch.h>

using namespace mav_trajectory_generation;

static int g_fail = 0;
#define EXPECT(cond, ...) do { if (!(cond)) { ++g_fail; std::printf("FAIL %s:%d %s  ", __FILE__, __LINE__, #cond); std::printf(__VA_ARGS__); std::printf("\n"); } } while (0)

static double checkPath(const Vertex::Vector& vertices, const Segment::Vector& segments, int N) {
  double worst = 0.0;
  for (size_t i = 0; i < segments.size(); ++i) {
    for (Vertex::Constraints::const_iterator it = vertices[i].cBegin(); it != vertices[i].cEnd(); ++it) {
      const Eigen::VectorXd actual = segments[i].evaluate(0.0, it->first);
      for (int d = 0; d < actual.size(); ++d) worst = std::max(worst, std::abs(actual[d] - it->second[d]));
    }
    for (Vertex::Constraints::const_iterator it = vertices[i + 1].cBegin(); it != vertices[i + 1].cEnd(); ++it) {
      const Eigen::VectorXd actual = segments[i].evaluate(segments[i].getTime(), it->first);
      for (int d = 0; d < actual.size(); ++d) worst = std::max(worst, std::abs(actual[d] - it->second[d]));
    }
    if (i > 0) {
      for (int k = 0; k < N / 2; ++k) {
        const Eigen::VectorXd a = segments[i - 1].evaluate(segments[i - 1].getTime(), k), b = segments[i].evaluate(0.0, k);
        for (int d = 0; d < a.size(); ++d) worst = std::max(worst, std::abs(a[d] - b[d]));
      }
    }
  }
  return worst;
}

int main() {
  // --- TwoVerticesSetup ------------------------------------------------------------------------------------
  {
    Vertex start(1);
    for (int k = 0; k <= 4; ++k) start.addConstraint(k, 0.0);
    Vertex goal = start;
    goal.addConstraint(derivative_order::POSITION, 5.0);
    PolynomialOptimization<10> opt(1);
    Vertex::Vector vertices{start, goal};
    opt.setupFromVertices(vertices, {5.0}, derivative_order::SNAP);
    EXPECT(opt.getNumberFreeConstraints() == 0, "n_free=%zu", opt.getNumberFreeConstraints());
    opt.solveLinear();
    Segment::Vector segments;
    opt.getSegments(&segments);
    const double matlab[10] = {-0.000000000000004, 0.000000000000004, -0.000000000000006, 0.000000000000003,
                               -0.000000000000001, 0.201600000000015, -0.134400000000012, 0.034560000000004,
                               -0.004032000000000, 0.000179200000000};
    const Eigen::VectorXd c = segments[0].getPolynomialsRef()[0].getCoefficients();
    for (int i = 0; i < 10; ++i) EXPECT(std::abs(c[i] - matlab[i]) < 1e-12, "coeff %d: %.17g vs %.17g", i, c[i], matlab[i]);
    EXPECT(checkPath(vertices, segments, 10) < 1e-6, "checkPath");
  }
  // --- README example ----------------------------------------------------------------------------------------
  {
    const int dimension = 3;
    Vertex start(dimension), middle(dimension), end(dimension);
    start.makeStartOrEnd(Eigen::VectorXd({0, 0, 1}), derivative_order::SNAP);
    middle.addConstraint(derivative_order::POSITION, Eigen::VectorXd({1, 2, 3}));
    end.makeStartOrEnd(Eigen::VectorXd({2, 1, 5}), derivative_order::SNAP);
    Vertex::Vector vertices{start, middle, end};
    std::vector<double> segment_times = estimateSegmentTimes(vertices, 2.0, 2.0);
    EXPECT(std::abs(segment_times[0] - 3.970847833173347) < 1e-14 && std::abs(segment_times[1] - 3.8241301415334297) < 1e-14, "times");
    PolynomialOptimization<10> opt(dimension);
    opt.setupFromVertices(vertices, segment_times, derivative_order::SNAP);
    opt.solveLinear();
    Segment::Vector segments;
    opt.getSegments(&segments);
    // segment-0 x coefficients of the 50-digit solve (tests/golden)
    const double want[10] = {0, 0, 0, 0, 0, 1.339252819662407e-02, -7.845546057749868e-03, 1.954568943684918e-03,
                             -2.392908039915994e-04, 1.181394415296818e-05};
    const Eigen::VectorXd c = segments[0][0].getCoefficients();
    for (int i = 0; i < 10; ++i) EXPECT(std::abs(c[i] - want[i]) < 1e-13, "readme coeff %d: %.17g vs %.17g", i, c[i], want[i]);
    EXPECT(checkPath(vertices, segments, 10) < 1e-6, "checkPath readme");
  }
  // --- random paths: checkPath, cost, constraint packing, free-constraint round trip, copy semantics ---------
  for (int seed = 100; seed < 104; ++seed) {
    const int D = 3, K = 10;
    Vertex::Vector vertices = createRandomVertices(derivative_order::SNAP, K, Eigen::VectorXd::Constant(D, -10.0),
                                                   Eigen::VectorXd::Constant(D, 10.0), seed);
    std::vector<double> times = estimateSegmentTimes(vertices, 3.0, 5.0);
    PolynomialOptimization<10> opt(D);
    opt.setupFromVertices(vertices, times);
    EXPECT(opt.getNumberFixedConstraints() == 10 + (K - 1) && opt.getNumberFreeConstraints() == 4 * (K - 1), "counts");
    opt.solveLinear();
    Segment::Vector segments;
    opt.getSegments(&segments);
    EXPECT(checkPath(vertices, segments, 10) < 1e-6, "checkPath seed %d: %g", seed, checkPath(vertices, segments, 10));
    // cost vs numeric integral of squared snap (composite Simpson)
    double numeric = 0.0;
    for (int s = 0; s < K; ++s) {
      const int n = 2000;
      const double hstep = times[s] / n;
      for (int i = 0; i <= n; ++i) {
        const Eigen::VectorXd v = segments[s].evaluate(i * hstep, derivative_order::SNAP);
        const double w = (i == 0 || i == n) ? 1.0 : (i % 2 ? 4.0 : 2.0);
        numeric += w * (v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) * hstep / 3.0;
      }
    }
    EXPECT(std::abs(opt.computeCost() - numeric) < 1e-6 * numeric, "cost %g vs numeric %g", opt.computeCost(), numeric);
    // ConstraintPacking: [d_F; d_P] -> p = A^-1 M d -> segment coefficients; A p -> M^+ -> d again
    std::vector<Eigen::VectorXd> fixed, free_c;
    opt.getFixedConstraints(&fixed);
    opt.getFreeConstraints(&free_c);
    Eigen::MatrixXd M, A_inv, A, M_pinv;
    opt.getM(&M); opt.getAInverse(&A_inv); opt.getA(&A); opt.getMpinv(&M_pinv);
    {   // printReorderingMatrix (LIN:382-385): header line + one text row per row of M, every row with exactly one 1
      std::ostringstream os;
      opt.printReorderingMatrix(os);
      const std::string txt = os.str();
      EXPECT(txt.rfind("Mapping matrix:\n", 0) == 0, "printReorderingMatrix header");
      size_t rows = 0, ones = 0;
      for (size_t i = std::string("Mapping matrix:\n").size(); i < txt.size(); ++i) {
        if (txt[i] == '\n') ++rows;
        if (txt[i] == '1') ++ones;
      }
      EXPECT(rows == (size_t)M.rows() && ones == (size_t)M.rows(), "printReorderingMatrix: %zu rows, %zu ones for a %d-row M", rows, ones, (int)M.rows());
    }
    for (int d = 0; d < D; ++d) {
      Eigen::VectorXd d_all(fixed[d].size() + free_c[d].size());
      for (int i = 0; i < fixed[d].size(); ++i) d_all[i] = fixed[d][i];
      for (int i = 0; i < free_c[d].size(); ++i) d_all[fixed[d].size() + i] = free_c[d][i];
      const Eigen::VectorXd p = A_inv * (M * d_all);
      const Eigen::VectorXd back = M_pinv * (A * p);
      for (int i = 0; i < d_all.size(); ++i) EXPECT(std::abs(back[i] - d_all[i]) < 1e-6, "packing round trip");
      for (int s = 0; s < K; ++s) {
        const Eigen::VectorXd cs = segments[s][d].getCoefficients(0);
        for (int j = 0; j < 10; ++j) EXPECT(std::abs(cs[j] - p[s * 10 + j]) < 1e-6 * (1.0 + std::abs(cs[j])), "packing coeffs");
      }
    }
    // KKT: R_PP d_P + R_PF d_F = 0
    Eigen::MatrixXd R;
    opt.getR(&R);
    const size_t nf = opt.getNumberFixedConstraints(), np = opt.getNumberFreeConstraints();
    for (int d = 0; d < D; ++d) {
      double worst = 0.0, scale = 0.0;
      for (size_t a = 0; a < np; ++a) {
        double acc = 0.0;
        for (size_t c = 0; c < nf; ++c) { acc += R(nf + a, c) * fixed[d][c]; scale = std::max(scale, std::abs(R(nf + a, c) * fixed[d][c])); }
        for (size_t c = 0; c < np; ++c) acc += R(nf + a, nf + c) * free_c[d][c];
        worst = std::max(worst, std::abs(acc));
      }
      EXPECT(worst < 1e-7 * scale, "KKT residual %g (scale %g)", worst, scale);
    }
    // setFreeConstraints(getFreeConstraints()) reproduces the segments; a perturbed d_P costs more
    PolynomialOptimization<10> copy(D);
    copy = opt;   // copy-assignment as in time_evaluation_node.cpp:357
    const double cost0 = opt.computeCost();
    copy.setFreeConstraints(free_c);
    Segment::Vector seg2;
    copy.getSegments(&seg2);
    for (int s = 0; s < K; ++s) for (int d = 0; d < D; ++d) {
      const Eigen::VectorXd a = segments[s][d].getCoefficients(0), b = seg2[s][d].getCoefficients(0);
      for (int j = 0; j < 10; ++j) EXPECT(std::abs(a[j] - b[j]) <= 1e-12 * (1.0 + std::abs(a[j])), "setFreeConstraints round trip");
    }
    free_c[0][3] *= 1.05;
    copy.setFreeConstraints(free_c);
    EXPECT(copy.computeCost() > cost0, "perturbed cost %g <= optimal %g", copy.computeCost(), cost0);
    EXPECT(std::abs(opt.computeCost() - cost0) == 0.0, "original untouched by the copy");
  }
  // --- extrema of magnitude (test_polynomial_optimization.cpp:308-400 ExtremaOfMagnitude): analytic candidates vs 1 ms
  //     sampling, velocity and acceleration, 3-D and 1-D ---------------------------------------------------------------
  for (int D : {3, 1}) {
    Vertex::Vector vertices = createRandomVertices(derivative_order::SNAP, 6, Eigen::VectorXd::Constant(D, -10.0),
                                                   Eigen::VectorXd::Constant(D, 10.0), 321 + D);
    std::vector<double> times = estimateSegmentTimes(vertices, 3.0, 5.0);
    PolynomialOptimization<10> opt(D);
    opt.setupFromVertices(vertices, times);
    opt.solveLinear();
    Trajectory trajectory;
    opt.getTrajectory(&trajectory);
    EXPECT(trajectory.K() == 6 && std::abs(trajectory.getMaxTime() - std::accumulate(times.begin(), times.end(), 0.0)) < 1e-12, "trajectory");
    for (int derivative : {derivative_order::VELOCITY, derivative_order::ACCELERATION}) {
      std::vector<Extremum> candidates;
      const Extremum best = opt.computeMaximumOfMagnitude(derivative, &candidates);
      double sampled = 0.0;
      Segment::Vector segs;
      opt.getSegments(&segs);
      for (const Segment& s : segs) for (double t = 0.0; t <= s.getTime(); t += 1e-3) sampled = std::max(sampled, s.evaluate(t, derivative).norm());
      EXPECT(best.value >= sampled - 1e-9 && best.value <= sampled * (1.0 + 1e-3) + 1e-6, "max |d%d| analytic %.9g vs sampled %.9g (D=%d)", derivative, best.value, sampled, D);
      EXPECT(!candidates.empty(), "candidates");
    }
  }
  // --- N = 12 with free end-vertex derivatives (test_feasibility.cpp:97-99) --------------------------------------
  {
    Vertex::Vector vertices = createRandomVertices(derivative_order::SNAP, 1, Eigen::VectorXd::Constant(3, -5.0),
                                                   Eigen::VectorXd::Constant(3, 5.0), 7);
    PolynomialOptimization<12> opt(3);
    opt.setupFromVertices(vertices, {4.2}, derivative_order::SNAP);
    EXPECT(opt.getNumberFreeConstraints() == 2, "N=12 n_free=%zu", opt.getNumberFreeConstraints());
    opt.solveLinear();
    Segment::Vector segments;
    opt.getSegments(&segments);
    EXPECT(checkPath(vertices, segments, 12) < 1e-6, "checkPath N=12");
  }
  // --- batched entry vs single solves ---------------------------------------------------------------------------
  {
    const int D = 3, K = 8, B = 257;
    Vertex::Vector proto = createRandomVertices(derivative_order::SNAP, K, Eigen::VectorXd::Constant(D, -10.0),
                                                Eigen::VectorXd::Constant(D, 10.0), 0);
    PolynomialOptimizationBatch<10> batch(D, PolynomialOptimizationBatch<10>::masksFromVertices(proto));
    const size_t nf = batch.getNumberFixedConstraints();
    std::vector<double> times(B * K), d_fixed(B * D * nf), coeffs(B * K * D * 10), cost(B);
    std::vector<Vertex::Vector> all;
    for (int b = 0; b < B; ++b) {
      Vertex::Vector v = createRandomVertices(derivative_order::SNAP, K, Eigen::VectorXd::Constant(D, -10.0),
                                              Eigen::VectorXd::Constant(D, 10.0), 1000 + b);
      std::vector<double> t = estimateSegmentTimes(v, 3.0, 5.0);
      for (int k = 0; k < K; ++k) times[b * K + k] = t[k];
      PolynomialOptimization<10> one(D);
      one.setupFromVertices(v, t);
      std::vector<Eigen::VectorXd> fixed;
      one.getFixedConstraints(&fixed);
      for (int d = 0; d < D; ++d) for (size_t c = 0; c < nf; ++c) d_fixed[(b * D + d) * nf + c] = fixed[d][c];
      all.push_back(v);
    }
    batch.solveLinear(B, times.data(), d_fixed.data(), coeffs.data(), nullptr, cost.data());
    for (int b = 0; b < B; b += 64) {
      Segment::Vector segs;
      batch.getSegments(coeffs.data(), times.data(), b, &segs);
      EXPECT(checkPath(all[b], segs, 10) < 1e-6, "batch checkPath b=%d", b);
      PolynomialOptimization<10> one(D);
      std::vector<double> t(times.begin() + b * K, times.begin() + (b + 1) * K);
      one.setupFromVertices(all[b], t);
      one.solveLinear();
      EXPECT(std::abs(one.computeCost() - cost[b]) < 1e-9 * cost[b], "batch cost %g vs single %g", cost[b], one.computeCost());
      Segment::Vector s1;
      one.getSegments(&s1);
      EXPECT(s1[3][1] == segs[3][1] || true, "bitwise equality not required");
      const Eigen::VectorXd a = s1[3][1].getCoefficients(0), c = segs[3][1].getCoefficients(0);
      for (int j = 0; j < 10; ++j) EXPECT(std::abs(a[j] - c[j]) <= 1e-11 * (1e-30 + std::abs(a[j])) + 1e-300, "batch vs single coeff");
    }
  }
  // --- TrajectoryBatch (device) vs the host-side Trajectory helpers: two independent root finders ----------------
  //     (test_polynomial_optimization.cpp:690-727 TimeScaling + :176-240 analytic maxima)
  {
    const int D = 3, K = 6, B = 130;
    std::vector<Trajectory> trajs(B);
    for (int b = 0; b < B; ++b) {
      Vertex::Vector v = createRandomVertices(derivative_order::SNAP, K, Eigen::VectorXd::Constant(D, -10.0),
                                              Eigen::VectorXd::Constant(D, 10.0), 5000 + b);
      PolynomialOptimization<10> one(D);
      one.setupFromVertices(v, estimateSegmentTimes(v, 3.0, 5.0));
      one.solveLinear();
      one.getTrajectory(&trajs[b]);
    }
    TrajectoryBatch batch(trajs);
    std::vector<double> v_dev, a_dev;
    EXPECT(batch.computeMaxVelocityAndAcceleration(&v_dev, &a_dev), "batch maxima");
    std::vector<Extremum> mn, mx;
    EXPECT(batch.computeMinMaxMagnitude(derivative_order::VELOCITY, {0, 2}, &mn, &mx), "batch minmax dims {0,2}");
    EXPECT(!batch.computeMinMaxMagnitude(derivative_order::VELOCITY, {0, 3}, &mn, &mx), "dimension out of bounds");
    EXPECT(!batch.computeMinMaxMagnitude(derivative_order::VELOCITY, {}, &mn, &mx), "no dimensions");
    EXPECT(batch.computeMinMaxMagnitude(derivative_order::VELOCITY, {0, 2}, &mn, &mx), "batch minmax dims {0,2}");
    for (int b = 0; b < B; ++b) {
      double v, a;
      trajs[b].computeMaxVelocityAndAcceleration(&v, &a);
      EXPECT(std::abs(v - v_dev[b]) <= 1e-9 * v && std::abs(a - a_dev[b]) <= 1e-9 * a, "b=%d v %.15g/%.15g a %.15g/%.15g", b, v,
             v_dev[b], a, a_dev[b]);
      Extremum hmn, hmx;
      trajs[b].computeMinMaxMagnitude(derivative_order::VELOCITY, {0, 2}, &hmn, &hmx);
      EXPECT(std::abs(hmx.value - mx[b].value) <= 1e-9 * hmx.value, "b=%d partial-dimension maximum", b);
      // Extremum::segment_idx / time: the device's reported point reproduces its value
      const Segment& s = trajs[b].segments()[mx[b].segment_idx];
      const Eigen::VectorXd vel = s.evaluate(mx[b].time, derivative_order::VELOCITY);
      EXPECT(std::abs(std::sqrt(vel[0] * vel[0] + vel[2] * vel[2]) - mx[b].value) <= 1e-12 * (1 + mx[b].value), "b=%d extremum point", b);
    }
    // sampling: device grid vs host evaluate
    std::vector<double> samples;
    std::vector<int> n_valid;
    const int S = 50, ND = 3;
    const double dt = 0.37;
    batch.sample(0.0, dt, S, ND, &samples, &n_valid);
    for (int b = 0; b < B; b += 17) {
      int want_valid = 0;
      for (int i = 0; i < S; ++i) {
        const double t = 0.0 + dt * i;
        if (t <= trajs[b].getMaxTime()) ++want_valid;
        for (int der = 0; der < ND; ++der) {
          const Eigen::VectorXd h = trajs[b].evaluate(std::min(t, trajs[b].getMaxTime()), der);
          for (int d = 0; d < D; ++d) {
            const double g = samples[((size_t)(b * S + i) * ND + der) * D + d];
            EXPECT(std::abs(g - h[d]) <= 1e-10 * (1.0 + std::abs(h[d])), "sample b=%d i=%d der=%d", b, i, der);
          }
        }
      }
      EXPECT(n_valid[b] == want_valid, "n_valid %d vs %d", n_valid[b], want_valid);
    }
    // feasibility scaling: device batch vs host loop
    const double v_lim = 2.0, a_lim = 2.5;
    std::vector<char> within;
    std::vector<double> scaling;
    EXPECT(batch.scaleSegmentTimesToMeetConstraints(v_lim, a_lim, &within, &scaling), "all within range after scaling");
    std::vector<Trajectory> scaled;
    batch.download(&scaled);
    int n_scaled = 0;
    for (int b = 0; b < B; ++b) {
      Trajectory h = trajs[b];
      EXPECT(h.scaleSegmentTimesToMeetConstraints(v_lim, a_lim), "host scaling b=%d", b);
      n_scaled += scaling[b] > 1.0;
      EXPECT(std::abs(h.getMaxTime() - scaled[b].getMaxTime()) <= 1e-9 * h.getMaxTime(), "b=%d total time %.15g vs %.15g", b,
             h.getMaxTime(), scaled[b].getMaxTime());
      EXPECT(std::abs(scaling[b] - h.getMaxTime() / trajs[b].getMaxTime()) <= 1e-9 * scaling[b], "b=%d scaling", b);
      for (int k = 0; k < K; ++k)
        for (int d = 0; d < D; ++d) {
          const Eigen::VectorXd ch = h.segments()[k][d].getCoefficients(), cd = scaled[b].segments()[k][d].getCoefficients();
          double cmax = 0.0, err = 0.0;
          for (int n = 0; n < 10; ++n) { cmax = std::max(cmax, std::abs(ch[n])); err = std::max(err, std::abs(ch[n] - cd[n])); }
          EXPECT(err <= 1e-9 * cmax, "b=%d k=%d d=%d scaled coefficients", b, k, d);
        }
    }
    EXPECT(n_scaled > 0, "the case exercises the scaling branch");
  }
  {  // mixed request: problems of different length / structure in one call vs one optimiser object each
    std::vector<Vertex::Vector> problems;
    std::vector<std::vector<double>> times;
    const int Ks[6] = {2, 5, 8, 16, 8, 16};
    for (int i = 0; i < 36; ++i) {
      Vertex::Vector v = createRandomVertices(derivative_order::SNAP, Ks[i % 6], Eigen::VectorXd::Constant(3, -10.0),
                                              Eigen::VectorXd::Constant(3, 10.0), 500 + i);
      if (i % 5 == 0) {   // another structure
        Eigen::VectorXd vel(3);
        vel[0] = 0.3; vel[1] = -0.2; vel[2] = 0.1;
        v[1].addConstraint(derivative_order::VELOCITY, vel);
      }
      problems.push_back(v);
      times.push_back(estimateSegmentTimes(v, 3.0, 5.0));
    }
    std::vector<Segment::Vector> mixed;
    std::vector<double> mixed_costs;
    EXPECT(solveLinearMixed<10>(3, problems, times, derivative_order::SNAP, &mixed, &mixed_costs), "solveLinearMixed");
    for (size_t i = 0; i < problems.size(); ++i) {
      PolynomialOptimization<10> opt(3);
      opt.setupFromVertices(problems[i], times[i], derivative_order::SNAP);
      opt.solveLinear();
      Segment::Vector ref;
      opt.getSegments(&ref);
      EXPECT(ref.size() == mixed[i].size(), "problem %zu segment count", i);
      double worst = 0.0, scale = 0.0;
      for (size_t k = 0; k < ref.size(); ++k)
        for (int d = 0; d < 3; ++d) {
          const Eigen::VectorXd a = ref[k][d].getCoefficients(), b = mixed[i][k][d].getCoefficients();
          for (int n = 0; n < 10; ++n) { worst = std::max(worst, std::abs(a[n] - b[n])); scale = std::max(scale, std::abs(a[n])); }
        }
      EXPECT(worst <= 1e-9 * scale, "problem %zu mixed vs single: %.3e of %.3e", i, worst, scale);
      EXPECT(std::abs(mixed_costs[i] - opt.computeCost()) <= 1e-8 * std::abs(opt.computeCost()), "problem %zu cost", i);
      EXPECT(checkPath(problems[i], mixed[i], 10) < 1e-6, "problem %zu checkPath", i);
    }
  }
  {  // Trajectory::evaluate: the reference's segment choice (src/trajectory.cpp:48-79)
    Vertex::Vector v = createRandomVertices(derivative_order::SNAP, 3, Eigen::VectorXd::Constant(3, -10.0),
                                            Eigen::VectorXd::Constant(3, 10.0), 77);
    std::vector<double> times = estimateSegmentTimes(v, 3.0, 5.0);
    PolynomialOptimization<10> opt(3);
    opt.setupFromVertices(v, times, derivative_order::SNAP);
    EXPECT(opt.solveLinear(), "solveLinear returns true on a well-posed problem");
    Trajectory traj;
    opt.getTrajectory(&traj);
    // order-5 derivatives jump at a position-only waypoint: exactly on the vertex the RIGHT segment is evaluated
    const Eigen::VectorXd at = traj.evaluate(times[0], 5), right = traj.segments()[1].evaluate(0.0, 5),
                          left = traj.segments()[0].evaluate(times[0], 5);
    // (the local time is t - (accumulated - T_1), zero only up to rounding -- as in the reference)
    EXPECT((at - right).norm() <= 1e-9 * (1.0 + right.norm()), "vertex time evaluates the right segment");
    EXPECT((at - left).norm() > 1e-6 * (1.0 + left.norm()), "derivative 5 is discontinuous at the waypoint (otherwise the check is vacuous)");
    const double t_end = traj.getMaxTime();
    const Eigen::VectorXd end = traj.evaluate(t_end, 0), last = traj.segments()[2].evaluate(times[2], 0);
    EXPECT((end - last).norm() <= 1e-12 * (1.0 + last.norm()), "end time evaluates the last segment at its duration");
    EXPECT(traj.evaluate(t_end + 1.0, 0).norm() == 0.0, "beyond the end: zero vector, as the reference");
  }
  {  // an optimiser created on one thread, solved on another after the first thread is gone
    std::unique_ptr<PolynomialOptimization<10>> opt;
    Vertex::Vector v = createRandomVertices(derivative_order::SNAP, 5, Eigen::VectorXd::Constant(3, -10.0),
                                            Eigen::VectorXd::Constant(3, 10.0), 78);
    std::vector<double> times = estimateSegmentTimes(v, 3.0, 5.0);
    std::thread maker([&] {
      opt.reset(new PolynomialOptimization<10>(3));
      opt->setupFromVertices(v, times, derivative_order::SNAP);
    });
    maker.join();   // the creating thread (and its thread-local context handle) is gone; the plan keeps the context alive
    bool ok = false;
    Segment::Vector segs;
    std::thread solver([&] {
      PolynomialOptimization<10> copy = *opt;   // value semantics across threads
      ok = copy.solveLinear();
      copy.getSegments(&segs);
    });
    solver.join();
    EXPECT(ok, "cross-thread solveLinear");
    EXPECT(checkPath(v, segs, 10) < 1e-6, "cross-thread checkPath");
    opt.reset();
  }
  // --- rank-deficient free systems: the reference's rank-revealing SparseQR returns a basic solution and solveLinear()
  //     returns true (LIN:365-378); so does the veneer (solveLinearBasic: pivoted QR on the host).  Under-constrained
  //     problems: only positions fixed.  Checked the way the reference checks its paths (checkPath) + minimum cost.
  for (int K = 1; K <= 2; ++K) {
    for (int D = 1; D <= 3; D += 2) {
      Vertex::Vector v;
      std::vector<double> times;
      for (int i = 0; i <= K; ++i) {
        Vertex vx(D);
        Eigen::VectorXd pos(D);
        for (int d = 0; d < D; ++d) pos[d] = 1.5 * i - 0.7 * d + 0.3 * i * i * (d + 1);
        vx.addConstraint(derivative_order::POSITION, pos);
        v.push_back(vx);
        if (i > 0) times.push_back(1.3 + 0.4 * i);
      }
      PolynomialOptimization<10> opt(D);
      opt.setupFromVertices(v, times, derivative_order::SNAP);
      const bool ok = opt.solveLinear();
      EXPECT(ok, "rank-deficient K=%d D=%d: solveLinear() must return true like the reference", K, D);
      // K = 1: nullity 2, the LDL^T sweep meets a non-positive pivot and the host QR takes over.  K = 2: nullity 1 in exact
      // arithmetic, but in floating point the last pivot may come out tiny and positive -- then the device path's solution
      // stands (it satisfies the same checks: any solution of the consistent system has the minimum cost)
      if (K == 1) EXPECT(opt.getLastSolveRank() < opt.getNumberFreeConstraints(), "rank %zu of %zu", opt.getLastSolveRank(),
                         opt.getNumberFreeConstraints());
      Segment::Vector segs;
      opt.getSegments(&segs);
      EXPECT(checkPath(v, segs, 10) < 1e-6, "rank-deficient K=%d D=%d checkPath %.3g", K, D, checkPath(v, segs, 10));
      // a cubic interpolates up to four points: the minimum snap cost is zero (the reference: 1e-20 .. 3e-19)
      EXPECT(std::abs(opt.computeCost()) < 1e-9, "rank-deficient K=%d D=%d cost %.3g", K, D, opt.computeCost());
      std::vector<Eigen::VectorXd> fr;
      opt.getFreeConstraints(&fr);
      size_t zeros = 0;
      for (std::ptrdiff_t i = 0; i < fr[0].size(); ++i) zeros += fr[0][i] == 0.0;
      EXPECT(zeros >= opt.getNumberFreeConstraints() - opt.getLastSolveRank(), "basic solution: %zu exact zeros", zeros);
      (void)ok;
    }
  }
  {   // the batched entry: a batch in which SOME trajectories are under-constrained -- here all share the structure, so a
      // position-only K = 1 batch: every trajectory goes through the host fallback; the call returns true like the reference
    const std::vector<uint32_t> masks{1u, 1u};
    PolynomialOptimizationBatch<10> batch(3, masks);
    const size_t B = 5;
    std::vector<double> times(B), fixed(B * 3 * 2), coeffs(B * 1 * 3 * 10, -1.0), cost(B, -1.0);
    for (size_t b = 0; b < B; ++b) {
      times[b] = 1.0 + 0.25 * b;
      for (int d = 0; d < 3; ++d) { fixed[(b * 3 + d) * 2 + 0] = 0.5 * d - b; fixed[(b * 3 + d) * 2 + 1] = 2.0 + d + 0.1 * b; }
    }
    const bool ok = batch.solveLinear(B, times.data(), fixed.data(), coeffs.data(), nullptr, cost.data());
    EXPECT(ok, "batched rank-deficient solve returns true");
    for (size_t b = 0; b < B; ++b) {
      EXPECT(std::abs(cost[b]) < 1e-9, "batched rank-deficient cost %.3g", cost[b]);
      for (int d = 0; d < 3; ++d) {
        const double* c = &coeffs[(b * 3 + d) * 10];
        double end = 0.0, tp = 1.0;
        for (int j = 0; j < 10; ++j) { end += c[j] * tp; tp *= times[b]; }
        EXPECT(std::abs(c[0] - fixed[(b * 3 + d) * 2]) < 1e-9 && std::abs(end - fixed[(b * 3 + d) * 2 + 1]) < 1e-8,
               "batched rank-deficient end points b=%zu d=%d: %.12g %.12g", b, d, c[0], end);
      }
    }
  }
  {   // a full-rank problem still goes through the library (rank == n_free)
    Vertex::Vector v = createRandomVertices(derivative_order::SNAP, 3, Eigen::VectorXd::Constant(3, -5.0),
                                            Eigen::VectorXd::Constant(3, 5.0), 11);
    PolynomialOptimization<10> opt(3);
    opt.setupFromVertices(v, estimateSegmentTimes(v, 2.0, 2.0), derivative_order::SNAP);
    EXPECT(opt.solveLinear() && opt.getLastSolveRank() == opt.getNumberFreeConstraints(), "full rank");
  }
  std::printf(g_fail ? "VENEER TESTS FAILED: %d\n" : "VENEER TESTS PASSED%.0d\n", g_fail);
  return g_fail ? 1 : 0;
}
