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
//     root, found by bisection-safeguarded Newton.  Level k's roots partition [0, 1] for level k+1 (two root buffers,
//     the levels alternate).  Identically-zero leading levels (zero-padded coefficients, trailing zeros -- the reference
//     strips them, rpoly_ak1.cpp:57-68) produce no sign changes and fall through.
//     Round 4: every level runs in two phases -- (A) the level's values at all partition points, TWO Horner chains at a
//     time, sign changes noted in a bit mask; (B) the brackets refined TWO at a time, a lane's j-th pair together with every
//     other lane's j-th pair.  Round 3 refined each bracket where the scan met it: one dependent chain at a time (8 cycles
//     per dependent FP64 operation for a lone wave) and, across the wave, one refinement per interval with a sign change in
//     ANY lane.
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

// (an extremum's VALUE is second order in the root's error: 1e-12 in tau moves it by ~1e-24 relative; its instant by 1e-12 T)
constexpr double kRootTol = 1e-12;        // absolute, in tau in [0, 1]: roots of g itself (the candidates)
constexpr double kPartitionTol = 1e-3;    // roots of the derivative levels only partition [0, 1] for the next level: the last
                                          // Newton step that is smaller than this leaves ~ C 1e-6 (C = |f'' / 2 f'|).  A partition
                                          // point off by e loses a root PAIR of the next level only if both lie within e of it, and a
                                          // pair that close changes an extremum value by ~ e^2 -- below the 1e-9 the tests hold the values
                                          // to.  (Round 6: 1e-4 -> 1e-3, 14 % fewer refinement rounds per wavefront on the bench workload:
                                          // 10k x 8 velocity 182 -> 172 us, acceleration 155 -> 142, 100k x 8 1074 -> 983.)
constexpr double kBisectTol = 1e-7;       // a bracket whose last step was a bisection is refined to at least this
constexpr int kRootMaxIter = 100;         // pure bisection needs ~48
#ifndef MTGX_NOISE_ULPS
#define MTGX_NOISE_ULPS 4.0
#endif
constexpr double kNoiseUlps = MTGX_NOISE_ULPS;        // |f| below this many ulps of sum |a_k|: converged (evaluation noise)
#if defined(MTGX_COUNT_ITERATIONS)
static long long mtgx_iteration_count = 0;   // host diagnostics (tests/extrema_emu.cpp): refinement rounds executed
static int mtgx_trace[32][16];               // [level][pair slot of the level]: rounds of the last root search (one lane)
static int mtgx_trace_slot = 0;
#endif

// 1 / x for a Newton STEP (24 bits are plenty: the step's error is second order in the iteration, and convergence is judged by
// |dx| and |f|, not by the step's last bits).  v_rcp_f64 is one instruction; the IEEE division sequence is ~12 dependent ones,
// and there were two per refinement round.
MTGX_HD double fast_rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcp(x);
#else
  return 1.0 / x;
#endif
}

// Several chains at once: a lane alone on its SIMD pays ~8 cycles per DEPENDENT FP64 operation and 4 per independent one, and the
// extrema kernels run at about one wave per SIMD at the sizes that matter (10k trajectories x 8 segments = 1250 waves) -- the
// round-3 form (one Horner chain after the other) was bound by exactly that latency.
// NCH chains at once (MTGX_CHAINS); x / f / d are arrays with compile-time indices
template <int K, int NCH>
MTGX_HD void horner_multi(const double* a, const double (&x)[NCH], double (&f)[NCH]) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) f[c] = a[K];
#pragma unroll
  for (int j = K - 1; j >= 0; --j) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) f[c] = fma(f[c], x[c], a[j]);
  }
}
template <int K, int NCH>
MTGX_HD void horner2_multi(const double* a, const double (&x)[NCH], double (&f)[NCH], double (&d)[NCH]) {
  // (the first derivative step of the textbook loop is fma(0, x, a[K]) = a[K]: written out, one FMA per chain and call less)
#pragma unroll
  for (int c = 0; c < NCH; ++c) { f[c] = a[K]; d[c] = K >= 1 ? a[K] : 0.0; }
  if (K >= 1) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) f[c] = fma(f[c], x[c], a[K - 1]);
  }
#pragma unroll
  for (int j = K - 2; j >= 0; --j) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) d[c] = fma(d[c], x[c], f[c]);
#pragma unroll
    for (int c = 0; c < NCH; ++c) f[c] = fma(f[c], x[c], a[j]);
  }
}
// brackets a lane refines at once (and partition points it evaluates at once).  2: with value and derivative that is four
// independent FMAs per Horner step -- the 4-cycle issue already covers the 8-cycle dependent latency.  4 measured 20 % SLOWER at
// every launch size (round 5, profiles/r05_extrema_lanes_per_search.jsonl: 10k x 8 segments 237 vs 192 us): most levels have one
// or two brackets per lane, the idle chains are pure instructions.
constexpr int kChains = 2;

// Bisection-safeguarded Newton on kChains brackets at once (each: f of opposite sign at its ends, monotone inside); a bracket that
// has converged keeps its x while the others finish.  Same step rule as the one-bracket form of round 3.
struct Bracket {
  double xl, xh, x, dx, dxold;
  bool done;
};
MTGX_HD void bracket_init(Bracket& b, double lo, double hi, double flo, double fhi) {
  b.xl = flo < 0.0 ? lo : hi;   // f(xl) < 0 <= f(xh)
  b.xh = flo < 0.0 ? hi : lo;
  // start from the chord's zero (inside the bracket by construction), nudged off the end points
  b.x = lo - flo * (hi - lo) * fast_rcp(fhi - flo);
  if (!(b.x > lo && b.x < hi)) b.x = 0.5 * (lo + hi);
  b.dxold = fabs(hi - lo);
  b.dx = b.dxold;
  b.done = false;
}
// fnoise: evaluation noise of the level's polynomial on [0, 1] (a few ulps of sum |a_k|).  Below it the sign of f carries no
// information: the iteration has reached what float64 can resolve (a root of multiplicity m -- the rest-to-rest end segments have
// a 7-fold root of g at the trajectory end -- is resolvable only to ~eps^(1/m); bisecting such a cluster down to `tol` cost ~45
// rounds per bracket and level, and it was the slowest lane of every wave: 250 us per 10k x 8 segments in round 3).
MTGX_HD void bracket_step(Bracket& b, double f, double df, double tol, double fnoise) {
  // branch-free (selects only): the brackets of a lane interleave, and a finished bracket costs no exec-mask detour.  Only x is
  // guarded: once a bracket has stopped (`done` is sticky) its other fields are never looked at again, so they may drift --
  // guarding all five cost 6 of the step's 16 v_cndmask, in a loop whose bookkeeping outweighs its Horner chains up to degree ~20
  // (85 of the 98 + 4 K instructions of a level-K iteration of two brackets).
  const bool stop = b.done || fabs(f) <= fnoise;
  const bool neg = f < 0.0;
  const double xl = neg ? b.x : b.xl;
  const double xh = neg ? b.xh : b.x;
  const bool newton_leaves = ((b.x - xh) * df - f) * ((b.x - xl) * df - f) > 0.0;
  const bool newton_slow = fabs(2.0 * f) > fabs(b.dxold * df);
  // (df == 0 needs no test of its own: then newton_leaves is f * f > 0, and newton_slow is |2 f| > 0 should f * f underflow;
  // f == 0 has stopped the bracket above)
  const bool bisect = newton_leaves || newton_slow;
  const double dx_b = 0.5 * (xh - xl);
  const double dx_n = f * fast_rcp(df);
  const double dx = bisect ? dx_b : dx_n;
  const double x = bisect ? xl + dx_b : b.x - dx_n;
  b.xl = xl;
  b.xh = xh;
  b.dxold = b.dx;
  b.dx = dx;
  b.x = stop ? b.x : x;
  // (a NEWTON step smaller than the loose partition tolerance leaves an error ~ its square; a bisection step of that size leaves
  // the error it has -- those continue to kBisectTol, so that the next level's intervals stay monotone: ADVICE round 4)
  b.done = stop || fabs(dx) < (bisect ? fmin(tol, kBisectTol) : tol);
}
template <int K, int NCH>
MTGX_HD void bracketed_root_multi(const double* a, Bracket (&b)[NCH], double tol, double fnoise) {
#if defined(MTGX_COUNT_ITERATIONS)
  const long long before = mtgx_iteration_count;
  struct Rec { long long b; ~Rec() { if (mtgx_trace_slot < 16) mtgx_trace[K][mtgx_trace_slot++] = (int)(mtgx_iteration_count - b); } } rec{before};
#endif
  for (int it = 0; it < kRootMaxIter; ++it) {
    double x[NCH], f[NCH], d[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) x[c] = b[c].x;
    horner2_multi<K, NCH>(a, x, f, d);
#if defined(MTGX_COUNT_ITERATIONS)
    ++mtgx_iteration_count;
#endif
    bool all_done = true;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      bracket_step(b[c], f[c], d[c], tol, fnoise);
      all_done = all_done && b[c].done;
    }
    if (all_done) break;
  }
}

// Derivative-chain level K (compile time): a[0..K] holds the (M-K)-th divided derivative of g on entry.  The partition points
// left by level K-1 are in buffer (K-1) & 1 of `roots`, this level's roots go to buffer K & 1 (element i of buffer b:
// roots[b * M + i]).  Two phases, so that the lanes of a wave refine their brackets TOGETHER (a lane's j-th bracket with every
// other lane's j-th bracket -- the wave runs max-over-lanes(brackets) / 2 refinements per level, not one per interval that has
// a sign change in ANY lane): (A) evaluate this level at every partition point, two points at a time, and note the
// intervals with a sign change in a bit mask; (B) refine them two at a time.
// Several lanes may SHARE one root search (small launches: the per-lane chain of refinements is the latency of the whole
// kernel).  All of them run phase A -- identical arithmetic, identical bit mask -- and refine only the brackets whose rank
// (position among the level's sign changes) falls to them: rank mod nparts in [part_begin, part_end).  Every root is written to
// its rank's slot of the shared root buffer, which the lanes of a group address as ONE column.  A lone lane is {0, 1, 1}; the
// host emulation runs a shared search as one lane that takes every part, {0, n, n}: a bracket's refinement does not depend on
// which lane (or which pair) it is refined in, so the results are bit-identical by construction.
struct Share { int part_begin, part_end, nparts; };   // nparts <= 4

// KH: degree the Horner chains run over.  The fully unrolled chain (KH = K, one body per level) is ~60 KB of straight-line code
// for degree 15, each wavefront walks through it once.  ROLLED: ONE body for all levels, the chains run over the full length M
// with the coefficients above the level's degree zero (~1.6x the FMAs, 1/15 of the code) -- built to test whether instruction
// fetch bounds small launches (a lone wave runs at ~12 cycles per instruction): it does not, the rolled form is 15-20 % SLOWER
// (profiles/r04f_extrema_variants.jsonl); kept selectable (measurement knob), the per-level bodies are the default.
template <int M, int K, class Roots, bool ROLLED = false>
struct Level {
  static MTGX_HD void run(const double* g, double* a, Roots& roots, int& cnt, const Share& sh, int kr = K) {
    constexpr int KH = ROLLED ? M : K;
    const int SRC = ROLLED ? ((kr - 1) & 1) * M : ((K - 1) & 1) * M, DST = ROLLED ? (kr & 1) * M : (K & 1) * M;
    const double tol = (ROLLED ? kr < M : K < M) ? kPartitionTol : kRootTol;
    double anorm = 0.0;
#pragma unroll
    for (int j = 0; j <= KH; ++j) anorm += fabs(a[j]);
    const double fnoise = kNoiseUlps * DBL_EPSILON * anorm;
    // partition points P_0 = 0, P_i = roots[SRC + i - 1] (i = 1 .. cnt), P_(cnt+1) = 1; interval i = [P_i, P_(i+1)]
    // (Round 5, measured and not adopted: keeping this level's VALUES at the partition points in a third LDS buffer instead of
    // evaluating both ends of every bracket again in phase B -- a fifth of a refinement's arithmetic on paper -- gave 4 % on the
    // one-derivative launch (214 -> 205 / 176 -> 170 us per 10k x 8) and LOST 35 % on the velocity + acceleration launch of the
    // time-scaling path (311 -> 418 us): 48 instead of 30 KB of LDS per 128-lane workgroup, three instead of five workgroups
    // per CU, a second round of workgroups.  profiles/r05h_extrema_partition_values_in_lds_not_adopted.txt)
    unsigned mask = 0;
    {
      double flo = a[0];   // value at 0
      for (int i = 0; i <= cnt; i += kChains) {
        double h[kChains], f[kChains];
#pragma unroll
        for (int c = 0; c < kChains; ++c) h[c] = i + c < cnt ? roots[SRC + i + c] : 1.0;
        horner_multi<KH, kChains>(a, h, f);
#pragma unroll
        for (int c = 0; c < kChains; ++c) {
          if (i + c <= cnt && (flo < 0.0) != (f[c] < 0.0)) mask |= 1u << (i + c);
          flo = f[c];
        }
      }
    }
#if defined(MTGX_COUNT_ITERATIONS)
    mtgx_trace_slot = 0;
#endif
    // The brackets are taken in GROUPS of kChains x nparts consecutive ranks: every lane of a shared search pops the same bits in
    // the same iteration (uniform control flow -- the lanes are neighbours in one wavefront) and picks its own kChains, ranks
    // part, part + nparts, ... of the group; all of them then refine at once.
    int rank = 0;
    while (mask != 0u) {
      int idx[4 * kChains], n_in_group = 0;
      const int base = rank;
      for (int u = 0; u < kChains * sh.nparts && u < 4 * kChains && mask != 0u; ++u) {
        idx[u] = __builtin_ctz(mask);
        mask &= mask - 1u;
        ++n_in_group;
        ++rank;
      }
      for (int part = sh.part_begin; part < sh.part_end; ++part) {
        // (a chain without a bracket of its own in this group runs along on the lane's first one -- or the group's -- and writes nothing)
        int u[kChains];
        bool w[kChains];
        double lo[kChains], hi[kChains], ends[2 * kChains], fe[2 * kChains];
#pragma unroll
        for (int c = 0; c < kChains; ++c) {
          w[c] = part + c * sh.nparts < n_in_group;
          u[c] = w[c] ? part + c * sh.nparts : (c > 0 ? u[0] : 0);
          const int ia = idx[u[c]];
          lo[c] = ia > 0 ? roots[SRC + ia - 1] : 0.0;
          hi[c] = ia < cnt ? roots[SRC + ia] : 1.0;
          ends[2 * c] = lo[c];
          ends[2 * c + 1] = hi[c];
        }
        horner_multi<KH, 2 * kChains>(a, ends, fe);
        Bracket b[kChains];
#pragma unroll
        for (int c = 0; c < kChains; ++c) bracket_init(b[c], lo[c], hi[c], fe[2 * c], fe[2 * c + 1]);
        bracketed_root_multi<KH, kChains>(a, b, tol, fnoise);
#pragma unroll
        for (int c = 0; c < kChains; ++c)
          if (w[c]) roots[DST + base + u[c]] = b[c].x;
      }
    }
    const int cnt_new = rank;
    cnt = cnt_new;
    if constexpr (ROLLED) {
      // integrate once (run-time level kr): a'_j = a_(j-1) s / j, s = M - kr; the entries above the new degree stay zero
      if (kr < M) {
        const double sd = (double)(M - kr);
#pragma unroll
        for (int j = M; j >= 1; --j) a[j] = a[j - 1] * (sd * (1.0 / (double)j));
        a[0] = g[M - kr - 1];
      }
    } else if constexpr (K < M) {
      // integrate once: divided derivative of order s-1 from order s, s = M-K:
      //   a'_j = g[j+s-1] C(j+s-1, s-1) = a_{j-1} * s / j   (j >= 1),   a'_0 = g[s-1]
      constexpr int s = M - K;
#pragma unroll
      for (int j = K + 1; j >= 1; --j) a[j] = a[j - 1] * ((double)s / (double)j);
      a[0] = g[s - 1];
      Level<M, K + 1, Roots>::run(g, a, roots, cnt, sh);
    }
  }
};

// Real roots in [0, 1] of g(tau) = sum_j g[j] tau^j, j < L; ascending.  L >= 2.  `roots` holds 2 * (L - 1) elements (two
// buffers the levels alternate between); returns the count and, in `base`, the offset of the buffer that holds the result.
template <int L, class Roots, bool ROLLED = false>
MTGX_HD int real_roots_unit(const double* g, Roots& roots, int& base, const Share& sh = Share{0, 1, 1}) {
  constexpr int M = L - 1;   // degree
  static_assert(M <= 31, "interval bit mask");
  double a[L];
#pragma unroll
  for (int j = 0; j < L; ++j) a[j] = 0.0;
  a[0] = g[M - 1];
  a[1] = g[M] * (double)M;   // level 1 = (M-1)-th divided derivative: g[M-1] + M g[M] tau
  int cnt = 0;
  if constexpr (ROLLED) {
    for (int kr = 1; kr <= M; ++kr) Level<M, 1, Roots, true>::run(g, a, roots, cnt, sh, kr);
  } else {
    Level<M, 1, Roots>::run(g, a, roots, cnt, sh);
  }
  base = (M & 1) * M;
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
template <int NMAX, class Roots, bool ROLLED = false>
MTGX_HD MinMax segment_minmax(const double* c, int N, int D, unsigned dim_mask, int der, double T, Roots& roots,
                              const Share& sh = Share{0, 1, 1}) {
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
  int base = 0;
  const int cnt = real_roots_unit<L, Roots, ROLLED>(g, roots, base, sh);

  MinMax mm;
  mm.v_min = DBL_MAX;     // segment.cpp:172-173
  mm.v_max = -DBL_MAX;
  mm.t_min = 0.0;
  mm.t_max = 0.0;
  for (int i = -2; i < cnt; ++i) {   // candidate order of polynomial.cpp:43-45: t_start, t_end, then the roots
    const double t = i == -2 ? 0.0 : (i == -1 ? T : roots[base + i] * T);
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
