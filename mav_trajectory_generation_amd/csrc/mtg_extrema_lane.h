// mtg_extrema_lane.h -- per-lane algorithm of the batched magnitude-extrema search.
//
// What it replaces, per segment: Segment::computeMinMaxMagnitudeCandidateTimes / ...Candidates /
// selectMinMaxMagnitudeFromCandidates (src/segment.cpp:83-184), i.e. the extrema of
//   m(t) = sqrt( sum_dim (p_dim^(der)(t))^2 )  on [0, T]
// found from the real roots of  g(t) = sum_dim p^(der)(t) * p^(der+1)(t)  (= m(t)^2' / 2, the "convolved"
// polynomial of segment.cpp:96-115) plus the two end points.  The reference finds ALL complex roots of g with
// Jenkins-Traub (src/rpoly/rpoly_ak1.cpp via polynomial.cpp:28-30, 65-83) and filters the real ones inside the
// interval (polynomial.cpp:32-63).  A GPU lane cannot afford that branchy deflation scheme, and does not need the
// complex roots: this file isolates the real roots inside the interval directly.
//
// Method (compile-time unrolled over the polynomial degree, no dynamic register indexing):
//   * map the segment to tau = t/T in [0, 1] (coefficient j scaled by T^j) -- same roots, better scaling;
//   * derivative chain: g_k = (m-k)-th divided derivative of g (degree k).  Between two consecutive roots of
//     g_{k-1} the polynomial g_k is monotone, so every sign change of g_k over that partition brackets exactly one
//     root, found by bisection-safeguarded Newton.  Level k's roots partition [0, 1] for level k+1; the root list
//     is updated in place (at most one new root per interval).  Identically-zero leading levels (zero-padded
//     coefficients, trailing zeros -- the reference strips them, rpoly_ak1.cpp:57-68) produce no sign changes and
//     fall through.
//   * evaluate the magnitude at tau = 0, 1 and at every root in that order, keeping strict </> like
//     std::min/std::max over Extremum::operator< (extremum.h:37-38; segment.cpp:175-181).
// Same code runs on the host for the emulation tests (tests/extrema_emu.cpp).
#pragma once

#include <cfloat>
#include <cmath>

#if defined(__HIPCC__)
#define MTGX_HD __host__ __device__ inline
#else
#define MTGX_HD inline
#endif

namespace mtgx {

constexpr int kMaxCoeffs = 12;   // Polynomial::kMaxN (polynomial.h:44)

// base(der, i) = i (i-1) ... (i-der+1): coefficient multiplier of the der-th derivative (src/polynomial.cpp:145-160)
MTGX_HD double falling_factorial(int i, int der) {
  double r = 1.0;
  for (int k = 0; k < der; ++k) r *= (double)(i - k);
  return r;
}

template <int K>
MTGX_HD double horner(const double* a, double x) {   // degree K, a[0..K]
  double r = a[K];
#pragma unroll
  for (int j = K - 1; j >= 0; --j) r = fma(r, x, a[j]);
  return r;
}

template <int K>
MTGX_HD void horner2(const double* a, double x, double& f, double& df) {   // value and first derivative
  f = a[K];
  df = 0.0;
#pragma unroll
  for (int j = K - 1; j >= 0; --j) {
    df = fma(df, x, f);
    f = fma(f, x, a[j]);
  }
}

constexpr double kRootTol = 4e-15;        // absolute, in tau in [0, 1]: roots of g itself (the candidates)
constexpr double kPartitionTol = 1e-7;    // roots of the derivative levels only partition [0, 1] for the next level
constexpr int kRootMaxIter = 100;         // pure bisection needs ~48

// One root of the degree-K polynomial a in [lo, hi], given f(lo), f(hi) of opposite sign and a monotone there.
template <int K>
MTGX_HD double bracketed_root(const double* a, double lo, double hi, double flo, double fhi, double tol) {
  double xl = flo < 0.0 ? lo : hi;   // f(xl) < 0 <= f(xh)
  double xh = flo < 0.0 ? hi : lo;
  // start from the chord's zero (inside the bracket by construction), nudged off the end points
  double x = lo - flo * (hi - lo) / (fhi - flo);
  if (!(x > lo && x < hi)) x = 0.5 * (lo + hi);
  double dxold = fabs(hi - lo), dx = dxold;
  double f, df;
  horner2<K>(a, x, f, df);
  for (int it = 0; it < kRootMaxIter; ++it) {
    const bool newton_leaves = ((x - xh) * df - f) * ((x - xl) * df - f) > 0.0;
    const bool newton_slow = fabs(2.0 * f) > fabs(dxold * df);
    dxold = dx;
    if (newton_leaves || newton_slow || !(df != 0.0)) {
      dx = 0.5 * (xh - xl);
      x = xl + dx;
    } else {
      dx = f / df;
      x -= dx;
    }
    if (fabs(dx) < tol) break;
    horner2<K>(a, x, f, df);
    if (f < 0.0) xl = x; else xh = x;
  }
  return x;
}

// Derivative-chain level K (compile time): a[0..K] holds the (M-K)-th divided derivative of g on entry.
template <int M, int K, class Roots>
struct Level {
  static MTGX_HD void run(const double* g, double* a, Roots& roots, int& cnt) {
    // roots of this level between the partition points left by level K-1
    int cnt_new = 0;
    double lo = 0.0, flo = a[0];
    for (int i = 0; i <= cnt; ++i) {
      const double hi = i < cnt ? roots[i] : 1.0;
      const double fhi = horner<K>(a, hi);
      if ((flo < 0.0) != (fhi < 0.0)) {
        const double r = bracketed_root<K>(a, lo, hi, flo, fhi, K < M ? kPartitionTol : kRootTol);
        roots[cnt_new] = r;   // cnt_new <= i and roots[i] was already read: in-place is safe
        ++cnt_new;
      }
      lo = hi;
      flo = fhi;
    }
    cnt = cnt_new;
    if constexpr (K < M) {
      // integrate once: divided derivative of order s-1 from order s, s = M-K:
      //   a'_j = g[j+s-1] C(j+s-1, s-1) = a_{j-1} * s / j   (j >= 1),   a'_0 = g[s-1]
      constexpr int s = M - K;
#pragma unroll
      for (int j = K + 1; j >= 1; --j) a[j] = a[j - 1] * ((double)s / (double)j);
      a[0] = g[s - 1];
      Level<M, K + 1, Roots>::run(g, a, roots, cnt);
    }
  }
};

// Real roots in [0, 1] of g(tau) = sum_j g[j] tau^j, j < L; ascending in roots[0..return).  L >= 2.
template <int L, class Roots>
MTGX_HD int real_roots_unit(const double* g, Roots& roots) {
  constexpr int M = L - 1;   // degree
  double a[L];
#pragma unroll
  for (int j = 0; j < L; ++j) a[j] = 0.0;
  a[0] = g[M - 1];
  a[1] = g[M] * (double)M;   // level 1 = (M-1)-th divided derivative: g[M-1] + M g[M] tau
  int cnt = 0;
  Level<M, 1, Roots>::run(g, a, roots, cnt);
  return cnt;
}

struct MinMax {
  double t_min, v_min, t_max, v_max;
};

// magnitude of the der-th derivative over the selected dimensions at local time t (segment.cpp:133-141;
// Polynomial::evaluate polynomial.h:137-149: Horner over base(der, i) * c_i from the highest power down)
MTGX_HD double magnitude_at(const double* c, int N, int D, unsigned dim_mask, int der, double t) {
  double acc = 0.0;
  for (int d = 0; d < D; ++d) {
    if (!((dim_mask >> d) & 1u)) continue;
    const double* cd = c + (long long)d * N;
    double r = 0.0;
    for (int i = N - 1; i >= der; --i) r = fma(r, t, falling_factorial(i, der) * cd[i]);
    acc = fma(r, r, acc);
  }
  return sqrt(acc);
}

// Extrema of one segment.  c = [D][N] coefficients (increasing powers), T = segment time, der = derivative whose
// magnitude is searched (N - der - 1 >= 0, polynomial.cpp:70-73), dim_mask = dimensions entering the magnitude.
// NMAX >= N - der (compile-time bound on the derivative polynomial's coefficient count).
template <int NMAX, class Roots>
MTGX_HD MinMax segment_minmax(const double* c, int N, int D, unsigned dim_mask, int der, double T, Roots& roots) {
  constexpr int L = 2 * NMAX - 2 >= 2 ? 2 * NMAX - 2 : 2;   // coefficient count of g (getConvolutionLength, polynomial.h:230-232)
  const int n_d = N - der;
  double g[L];
#pragma unroll
  for (int j = 0; j < L; ++j) g[j] = 0.0;
  int n_dims = 0;
  for (int d = 0; d < D; ++d) n_dims += (dim_mask >> d) & 1u;
  for (int d = 0; d < D; ++d) {
    if (!((dim_mask >> d) & 1u)) continue;
    const double* cd = c + (long long)d * N;
    // u = der-th derivative in tau (times a positive constant T^der), w = u'
    double u[NMAX];
    double tp = 1.0;
#pragma unroll
    for (int i = 0; i < NMAX; ++i) {
      u[i] = i < n_d ? cd[i + der] * falling_factorial(i + der, der) * tp : 0.0;
      tp *= T;
    }
    if (n_dims > 1) {
      // g += u * u'  (Polynomial::convolve of d and dd, segment.cpp:108-113)
#pragma unroll
      for (int i = 0; i < NMAX; ++i)
#pragma unroll
        for (int j = 0; j + 1 < NMAX; ++j) g[i + j] = fma(u[i], (double)(j + 1) * u[j + 1], g[i + j]);
    } else {
      // single dimension: critical points of p^(der) itself (segment.cpp:124-131 -> polynomial.cpp:65-83)
#pragma unroll
      for (int j = 0; j + 1 < NMAX; ++j) g[j] = (double)(j + 1) * u[j + 1];
    }
  }
  const int cnt = real_roots_unit<L>(g, roots);

  MinMax mm;
  mm.v_min = DBL_MAX;     // segment.cpp:172-173
  mm.v_max = -DBL_MAX;
  mm.t_min = 0.0;
  mm.t_max = 0.0;
  for (int i = -2; i < cnt; ++i) {   // candidate order of polynomial.cpp:43-45: t_start, t_end, then the roots
    const double t = i == -2 ? 0.0 : (i == -1 ? T : roots[i] * T);
    const double v = magnitude_at(c, N, D, dim_mask, der, t);
    if (v > mm.v_max) { mm.v_max = v; mm.t_max = t; }
    if (v < mm.v_min) { mm.v_min = v; mm.t_min = t; }
  }
  return mm;
}

// Violation scaling of Trajectory::scaleSegmentTimesToMeetConstraints (src/trajectory.cpp:385-429, one iteration):
// returns the factor to stretch the segment times by; within_range per :403-404 with kTolerance = 1e-3.
MTGX_HD double violation_scaling(double v_max_actual, double a_max_actual, double v_max, double a_max, bool& within_range) {
  const double kTolerance = 1e-3;
  const double velocity_violation = v_max_actual / v_max;
  const double acceleration_violation = a_max_actual / a_max;
  within_range = velocity_violation <= 1.0 + kTolerance && acceleration_violation <= 1.0 + kTolerance;
  const double s = fmax(1.0, fmax(velocity_violation, sqrt(acceleration_violation)));
  return s;
}

}  // namespace mtgx
