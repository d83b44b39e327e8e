// Counterpart of the reference's benchmark binary (mav_trajectory_generation/src/polynomial_timing_evaluation.cpp:
// 93-128: wall clock of `PolynomialOptimization<N> opt(3); opt.setupFromVertices(...); opt.solveLinear();` for
// K in {2, 10, 50, 100}, N = 10, 1000 runs each), written against the drop-in veneer in
// include/compat/mav_trajectory_generation/ -- i.e. the reference's own source-level API, with libmtg_hip.so behind it.
//   (a) the reference's loop as is: one trajectory per call (launch + PCIe round trip per call: latency path);
//   (b) the same 1000 problems per K handed over at once through PolynomialOptimizationBatch<N> (the accelerated path).
// Build: __graft_entry__.build();  run on the GPU box: tools/cpp/polynomial_timing_evaluation
#include <chrono>
#include <cstdio>
#include <vector>

#include <mav_trajectory_generation/polynomial_optimization_linear.h>

using namespace mav_trajectory_generation;

static const int N = 10;
static const int max_derivative = derivative_order::SNAP;

static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
  const int n_segments_to_test[4] = {2, 10, 50, 100};
  const int runs = 1000;
  std::printf("%-10s %-28s %-28s %s\n", "segments", "per-call veneer [us/traj]", "batched veneer [us/traj]", "max |coeff diff|");
  for (int K : n_segments_to_test) {
    // the reference's benchmark draws a new random path per run (seeded); same here
    std::vector<Vertex::Vector> problems;
    std::vector<std::vector<double>> times;
    for (int r = 0; r < runs; ++r) {
      problems.push_back(createRandomVertices(max_derivative, K, Eigen::VectorXd::Constant(3, -5.0 * K / 2),
                                              Eigen::VectorXd::Constant(3, 5.0 * K / 2), 1 + r));
      times.push_back(estimateSegmentTimes(problems.back(), 2.0, 2.0));
    }
    // (a) the reference's loop
    std::vector<Segment::Vector> per_call(runs);
    {  // warm-up: context, plan cache
      PolynomialOptimization<N> opt(3);
      opt.setupFromVertices(problems[0], times[0], max_derivative);
      opt.solveLinear();
    }
    const double t0 = now();
    for (int r = 0; r < runs; ++r) {
      PolynomialOptimization<N> opt(3);
      opt.setupFromVertices(problems[r], times[r], max_derivative);
      opt.solveLinear();
      opt.getSegments(&per_call[r]);
    }
    const double per_call_us = (now() - t0) / runs * 1e6;
    // (b) one batched call
    const std::vector<uint32_t> masks = PolynomialOptimizationBatch<N>::masksFromVertices(problems[0]);
    PolynomialOptimizationBatch<N> batch(3, masks, max_derivative);
    const size_t nf = batch.getNumberFixedConstraints();
    std::vector<double> t_flat((size_t)runs * K), d_fixed((size_t)runs * 3 * nf), coeffs((size_t)runs * K * 3 * N);
    auto pack = [&] {
      for (int r = 0; r < runs; ++r) {
        for (int k = 0; k < K; ++k) t_flat[(size_t)r * K + k] = times[r][k];
        size_t col = 0;
        for (size_t v = 0; v < problems[r].size(); ++v)
          for (int p = 0; p < N / 2; ++p) {
            Eigen::VectorXd c;
            if (!problems[r][v].getConstraint(p, &c)) continue;
            for (int d = 0; d < 3; ++d) d_fixed[((size_t)r * 3 + d) * nf + col] = c[d];
            ++col;
          }
      }
    };
    pack();
    batch.solveLinear(runs, t_flat.data(), d_fixed.data(), coeffs.data());   // warm-up
    const double t1 = now();
    pack();                                                                   // host-side packing is part of the cost
    batch.solveLinear(runs, t_flat.data(), d_fixed.data(), coeffs.data());
    const double batched_us = (now() - t1) / runs * 1e6;
    double worst = 0.0;
    for (int r = 0; r < runs; r += 97)
      for (int k = 0; k < K; ++k)
        for (int d = 0; d < 3; ++d) {
          const Eigen::VectorXd c = per_call[r][k][d].getCoefficients(0);
          for (int j = 0; j < N; ++j)
            worst = std::max(worst, std::abs(c[j] - coeffs[(((size_t)r * K + k) * 3 + d) * N + j]));
        }
    std::printf("%-10d %-28.2f %-28.3f %.2e\n", K, per_call_us, batched_us, worst);
  }
  return 0;
}
