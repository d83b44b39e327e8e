// Host-only test program for the veneer's trajectory file format (include/compat/.../io.h) and the host-side
// Trajectory analysis helpers (trajectory.h).  No GPU, no library: plain g++.
//   test_io write <file>          write a fixed 2-segment trajectory
//   test_io read <file>           read a file, print "K N D", times and coefficients with %.17g
//   test_io selftest <tmpfile>    round trip + malformed inputs + analysis-helper checks
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>

#include <mav_trajectory_generation/io.h>

using namespace mav_trajectory_generation;

static int g_fail = 0;
#define EXPECT(cond) do { if (!(cond)) { ++g_fail; std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); } } while (0)

static Segment::Vector fixture() {
  Segment::Vector segs(2, Segment(6, 2));
  const double t[2] = {1.2345678901234, 0.75};
  for (int k = 0; k < 2; ++k) {
    segs[k].setTime(t[k]);
    for (int d = 0; d < 2; ++d) {
      Eigen::VectorXd c(6);
      for (int n = 0; n < 6; ++n) c[n] = std::sin(1.0 + 7 * k + 3 * d + n) * std::pow(10.0, n - 3) / 3.0;
      segs[k][d] = Polynomial(6, c);
    }
  }
  segs[1][1] = Polynomial(6, [] { Eigen::VectorXd c(6); c[0] = -1.5; c[1] = 0.0; c[2] = 1e-300; c[3] = -2.5e17; c[4] = 1.0 / 3.0; c[5] = 42.0; return c; }());
  return segs;
}

static void dump(const Segment::Vector& segs) {
  std::printf("%zu %d %d\n", segs.size(), segs.empty() ? 0 : segs[0].N(), segs.empty() ? 0 : segs[0].D());
  for (const Segment& s : segs) {
    std::printf("%llu\n", (unsigned long long)s.getTimeNSec());
    for (int d = 0; d < s.D(); ++d) {
      const Eigen::VectorXd c = s[d].getCoefficients();
      for (int n = 0; n < s.N(); ++n) std::printf("%.17g%c", c[n], n + 1 == s.N() ? '\n' : ' ');
    }
  }
}

static bool write_text(const std::string& f, const char* text) {
  std::ofstream o(f);
  o << text;
  return (bool)o;
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const std::string mode = argv[1], file = argv[2];
  if (mode == "write") return segmentsToFile(file, fixture()) ? 0 : 1;
  if (mode == "read") {
    Segment::Vector segs;
    if (!segmentsFromFile(file, &segs)) { std::printf("READ FAILED\n"); return 1; }
    dump(segs);
    return 0;
  }
  // ---- selftest
  const Segment::Vector segs = fixture();
  EXPECT(segmentsToFile(file, segs));
  Segment::Vector back;
  EXPECT(segmentsFromFile(file, &back));
  EXPECT(back.size() == segs.size());
  for (size_t k = 0; k < back.size() && k < segs.size(); ++k) {
    EXPECT(back[k].getTimeNSec() == segs[k].getTimeNSec());       // times survive as truncated ns
    EXPECT(std::abs(back[k].getTime() - segs[k].getTime()) < 1e-9);
    for (int d = 0; d < 2; ++d) EXPECT(back[k][d] == segs[k][d]);  // coefficients bit-exact
  }
  Trajectory tr, tr2;
  tr.setSegments(segs);
  EXPECT(trajectoryToFile(file, tr) && trajectoryFromFile(file, &tr2) && tr2.K() == 2 && tr2.N() == 6 && tr2.D() == 2);
  EXPECT(!segmentsFromFile(file + ".does_not_exist", &back));
  // yaml-cpp style variations: other key order, wrapped flow sequence, comments
  EXPECT(write_text(file, "# header\nsegments:\n  - time: 500000000 # [ns]\n    coefficients:\n      - [1, 2,\n         3]\n"
                          "      - [-4, 5e0, .5]\n    D: 2\n    N: 3\n"));
  EXPECT(segmentsFromFile(file, &back) && back.size() == 1 && back[0].N() == 3 && back[0].D() == 2);
  if (back.size() == 1) {
    EXPECT(back[0].getTime() == 0.5 && back[0][0].getCoefficients()[2] == 3.0 && back[0][1].getCoefficients()[0] == -4.0 &&
           back[0][1].getCoefficients()[2] == 0.5);
  }
  // malformed documents (io.cpp:183-215 return false)
  EXPECT(write_text(file, "foo: 1\n") && !segmentsFromFile(file, &back));                                       // no segments
  EXPECT(write_text(file, "segments:\n  - N: 3\n    D: 1\n    coefficients:\n      - [1, 2, 3]\n") && !segmentsFromFile(file, &back));   // no time
  EXPECT(write_text(file, "segments:\n  - N: 3\n    D: 2\n    time: 1\n    coefficients:\n      - [1, 2, 3]\n") && !segmentsFromFile(file, &back));   // D mismatch
  EXPECT(write_text(file, "segments:\n  - N: 4\n    D: 1\n    time: 1\n    coefficients:\n      - [1, 2, 3]\n") && !segmentsFromFile(file, &back));   // N mismatch

  // ---- host-side analysis helpers on a known polynomial: p(t) = t^3 - 3 t on [0, 2], 1-D
  {
    Segment s(4, 1);
    Eigen::VectorXd c(4);
    c[0] = 0; c[1] = -3; c[2] = 0; c[3] = 1;
    s[0] = Polynomial(4, c);
    s.setTime(2.0);
    Trajectory t1;
    t1.setSegments(Segment::Vector(1, s));
    Extremum mn, mx;
    EXPECT(t1.computeMinMaxMagnitude(derivative_order::POSITION, {0}, &mn, &mx));
    EXPECT(std::abs(mx.value - 2.0) < 1e-12);                 // |p| max = 2 at t = 1 and t = 2
    double v, a;
    EXPECT(t1.computeMaxVelocityAndAcceleration(&v, &a));
    EXPECT(std::abs(v - 9.0) < 1e-12 && std::abs(a - 12.0) < 1e-12);     // p' = 3t^2 - 3, p'' = 6t on [0, 2]
    EXPECT(t1.scaleSegmentTimesToMeetConstraints(3.0, 3.0));
    EXPECT(t1.computeMaxVelocityAndAcceleration(&v, &a) && v <= 3.0 * (1 + 1e-3) && a <= 3.0 * (1 + 1e-3));
    EXPECT(std::abs(t1.getMaxTime() - 2.0 * 3.0) < 1e-9);     // s = max(9/3, sqrt(12/3)) = 3
    std::vector<Eigen::VectorXd> samples;
    std::vector<double> ts;
    t1.evaluateRange(0.0, t1.getMaxTime(), 0.5, derivative_order::POSITION, &samples, &ts);
    EXPECT(samples.size() == 12 && ts.size() == 12 && ts[3] == 1.5);
    EXPECT(!t1.scaleSegmentTimes(1e-7) && t1.scaleSegmentTimes(0.5) && std::abs(t1.getMaxTime() - 3.0) < 1e-9);
  }
  // ---- host real-root finder (Polynomial::realRootsInInterval): known roots, shifted interval, badly scaled leading term
  {
    auto from_roots = [](const std::vector<double>& r, double lead) {
      std::vector<double> c{lead};
      for (double x : r) {
        std::vector<double> n(c.size() + 1, 0.0);
        for (size_t i = 0; i < c.size(); ++i) { n[i + 1] += c[i]; n[i] -= x * c[i]; }
        c = n;
      }
      return c;
    };
    std::vector<double> roots;
    Polynomial::realRootsInInterval(from_roots({-3.0, 0.5, 2.0, 7.25, 11.0}, 1.0), -5.0, 8.0, &roots);
    EXPECT(roots.size() == 4);
    const double want[4] = {-3.0, 0.5, 2.0, 7.25};
    for (size_t i = 0; i < roots.size() && i < 4; ++i) EXPECT(std::abs(roots[i] - want[i]) < 1e-10);
    // leading coefficient 1e-18 relative to the constant term: still a genuine degree-6 polynomial on [0, 40]
    Polynomial::realRootsInInterval(from_roots({5.0, 10.0, 15.0, 20.0, 25.0, 30.0}, 1e-12), 0.0, 40.0, &roots);
    EXPECT(roots.size() == 6);
    for (size_t i = 0; i < roots.size(); ++i) EXPECT(std::abs(roots[i] - 5.0 * (i + 1)) < 1e-6);
    Polynomial::realRootsInInterval({1.0, 0.0, 1.0}, -10.0, 10.0, &roots);     // t^2 + 1: no real roots
    EXPECT(roots.empty());
    Polynomial::realRootsInInterval({0.0, 0.0, 0.0}, 0.0, 1.0, &roots);        // zero polynomial
    EXPECT(roots.empty());
  }
  if (g_fail == 0) std::printf("IO TESTS PASSED\n");
  return g_fail ? 1 : 0;
}
