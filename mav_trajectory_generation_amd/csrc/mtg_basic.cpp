// mtg_basic.cpp -- basic solution of a RANK-DEFICIENT free-constraint system (host code, plain g++).
//
// The reference factorises R_PP with a rank-revealing SparseQR and always returns a solution and `true`
// (impl/polynomial_optimization_linear_impl.h:365-378): on an under-constrained problem (e.g. one segment with only the
// end positions fixed) that is a BASIC solution -- free variables beyond the numerical rank are zero.  The kernels' block
// LDL^T sweep flags such trajectories (non-positive pivot, MTG_FLAG_SINGULAR) instead of solving them; with
// MTG_FLAG_BASIC_SOLUTION the C ABI solves exactly those trajectories here: dense R = M^T H M of ONE trajectory from the
// same constant tables and scaling identity as the kernels (H(T) = T^(1-2d) S H(1) S, impl/...:318, :334-335), a
// column-pivoted Householder QR of R_PP (Eigen's rank threshold, SparseQR::factorize: 20 (rows + cols) max column norm eps),
// back-substitution with the variables beyond the rank at zero.  Such problems are tiny and rare; nothing here is hot.
// The pivot order (largest remaining column norm) differs from Eigen's COLAMD order, so on rank-deficient problems the
// coefficients may differ from the reference's while cost and constraints agree (the minimum is unique, the minimiser is not).
#include <algorithm>
#include <cmath>
#include <limits>
#include <numeric>
#include <vector>

#include "mtg_lane.h"

// H: N / 2; mask / offF / offP: the plan's tables ([K+1], [K+2], [K+2]); times [K]; dfix [D][n_fixed]; dfree out [D][n_free].
// Returns the numerical rank of R_PP (n_free for a regular system), or -1 for an unsupported shape.
extern "C" int mtg_basic_solution_one(int H, int K, int D, int deriv, const int* mask, const int* offF, const int* offP,
                                      const double* times, const double* dfix, double* dfree) {
  if (H < 1 || H > 6 || K < 1 || D < 1) return -1;
  const int N = 2 * H;
  const size_t nf = (size_t)offF[K + 1], np = (size_t)offP[K + 1];
  if (np == 0) return 0;
  const double* h1 = kH1 + mtg_h1_offset(N, deriv);
  // global column of (vertex v, derivative p): fixed slots first, then free slots, each ordered by (vertex, derivative)
  auto column = [&](int v, int p) -> size_t {
    const int below = __builtin_popcount((unsigned)(mask[v] & ((1 << p) - 1)));
    return ((mask[v] >> p) & 1) ? (size_t)(offF[v] + below) : nf + (size_t)(offP[v] + (p - below));
  };
  const size_t nc = nf + np;
  std::vector<double> R(nc * nc, 0.0);                    // row-major, symmetric
  for (int i = 0; i < K; ++i) {
    const double T = times[i];
    double s[6];
    s[0] = 1.0;
    for (int p = 1; p < H; ++p) s[p] = s[p - 1] * T;
    const double base = std::pow(T, 1 - 2 * deriv);
    for (int a = 0; a < N; ++a) {
      const size_t ga = column(a < H ? i : i + 1, a % H);
      for (int b = 0; b < N; ++b) {
        const size_t gb = column(b < H ? i : i + 1, b % H);
        R[ga * nc + gb] += base * s[a % H] * s[b % H] * h1[a * N + b];
      }
    }
  }
  std::vector<double> A(np * np);                         // R_PP, column-major
  for (size_t c = 0; c < np; ++c) for (size_t r = 0; r < np; ++r) A[c * np + r] = R[(nf + r) * nc + nf + c];
  std::vector<double> rhs((size_t)D * np, 0.0);           // -R_PF d_F per dimension (impl/...:371-372)
  for (int d = 0; d < D; ++d)
    for (size_t r = 0; r < np; ++r) {
      double acc = 0.0;
      for (size_t c = 0; c < nf; ++c) acc += R[(nf + r) * nc + c] * dfix[(size_t)d * nf + c];
      rhs[(size_t)d * np + r] = -acc;
    }
  // Householder QR with column pivoting (Businger-Golub), reflectors applied to the right-hand sides on the fly
  std::vector<size_t> perm(np);
  std::iota(perm.begin(), perm.end(), (size_t)0);
  std::vector<double> cn(np), v(np);
  double max_norm = 0.0;
  for (size_t c = 0; c < np; ++c) {
    double s2 = 0.0;
    for (size_t r = 0; r < np; ++r) s2 += A[c * np + r] * A[c * np + r];
    max_norm = std::max(max_norm, std::sqrt(s2));
  }
  const double threshold = 20.0 * double(2 * np) * max_norm * std::numeric_limits<double>::epsilon();
  size_t rank = 0;
  for (size_t k = 0; k < np; ++k) {
    size_t piv = k;
    for (size_t c = k; c < np; ++c) {                     // remaining column norms, recomputed (tiny matrices)
      double s2 = 0.0;
      for (size_t r = k; r < np; ++r) s2 += A[c * np + r] * A[c * np + r];
      cn[c] = s2;
      if (s2 > cn[piv]) piv = c;
    }
    if (std::sqrt(cn[piv]) <= threshold) break;           // every remaining column is numerically dependent
    if (piv != k) {
      for (size_t r = 0; r < np; ++r) std::swap(A[k * np + r], A[piv * np + r]);
      std::swap(perm[k], perm[piv]);
    }
    const double alpha = A[k * np + k] > 0.0 ? -std::sqrt(cn[piv]) : std::sqrt(cn[piv]);
    double vnorm2 = 0.0;
    for (size_t r = k; r < np; ++r) { v[r] = A[k * np + r]; if (r == k) v[r] -= alpha; vnorm2 += v[r] * v[r]; }
    if (vnorm2 > 0.0) {
      auto reflect = [&](double* x) {
        double dot = 0.0;
        for (size_t r = k; r < np; ++r) dot += v[r] * x[r];
        const double f = 2.0 * dot / vnorm2;
        for (size_t r = k; r < np; ++r) x[r] -= f * v[r];
      };
      for (size_t c = k; c < np; ++c) reflect(&A[c * np]);
      for (int d = 0; d < D; ++d) reflect(&rhs[(size_t)d * np]);
    }
    ++rank;
  }
  std::vector<double> y(np);
  for (int d = 0; d < D; ++d) {
    std::fill(y.begin(), y.end(), 0.0);                   // basic solution: variables beyond the rank stay zero
    for (size_t i = rank; i-- > 0;) {
      double acc = rhs[(size_t)d * np + i];
      for (size_t c = i + 1; c < rank; ++c) acc -= A[c * np + i] * y[c];
      y[i] = acc / A[i * np + i];
    }
    for (size_t i = 0; i < np; ++i) dfree[(size_t)d * np + perm[i]] = y[i];
  }
  return (int)rank;
}
