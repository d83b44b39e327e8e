// cpu_ref.cpp -- ORACLE / CPU BASELINE (test infrastructure, NOT product code).
//
// Plain C++17 restatement (no Eigen: the reference's Eigen/glog dependencies are not vendored and absent
// from this image) of the reference's hot path, step by step:
//   LIN  = mav_trajectory_generation/include/mav_trajectory_generation/impl/polynomial_optimization_linear_impl.h
//   POLY = mav_trajectory_generation/include/mav_trajectory_generation/polynomial.h, src/polynomial.cpp
//   VERT = mav_trajectory_generation/src/vertex.cpp
// It performs the SAME arithmetic steps as the reference (pow-based Q, Schur-complement A^-1 with a dense LU of
// the h x h block, H = A^-T Q A^-1 as two N^3 products, R = M^T H M, Householder-QR solve of R_PP, A^-1 M d),
// but none of Eigen's sparse bookkeeping or per-Polynomial heap allocation -- i.e. it is a conservative
// (faster-than-real-Eigen) stand-in for "the Eigen path on the host cores".  Parity status: pinned against the reference's own code
// (oracle/_ref/libmtg_ref.so) in tests/test_reference_build.py; see oracle_np.py.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

namespace {

constexpr int kMaxN = 12;   // POLY polynomial.h:44
constexpr int kTab = 22;    // kMaxConvolutionSize, polynomial.h:47

struct BaseCoeffs {
  double v[kTab][kTab];
  BaseCoeffs() {  // src/polynomial.cpp:145-160
    std::memset(v, 0, sizeof(v));
    for (int i = 0; i < kTab; ++i) v[0][i] = 1.0;
    const int deg = kTab - 1;
    int order = deg;
    for (int n = 1; n < kTab; ++n) {
      for (int i = deg - order; i < kTab; ++i) v[n][i] = (order - deg + i) * v[n - 1][i];
      order--;
    }
  }
};
const BaseCoeffs kBase;

// polynomial.h:201-219
void base_coeffs_with_time(int n, int derivative, double t, double* c) {
  for (int i = 0; i < n; ++i) c[i] = 0.0;
  c[derivative] = kBase.v[derivative][derivative];
  if (std::abs(t) < std::numeric_limits<double>::epsilon()) return;
  double t_power = t;
  for (int j = derivative + 1; j < n; ++j) {
    c[j] = kBase.v[derivative][j] * t_power;
    t_power = t_power * t;
  }
}

// dense LU inverse with partial pivoting (stand-in for Eigen's fixed-size .inverse(), LIN:170-171)
void lu_inverse(int n, const double* a, double* inv) {
  double m[6 * 12];
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < n; ++j) { m[i * 2 * n + j] = a[i * n + j]; m[i * 2 * n + n + j] = (i == j); }
  }
  for (int c = 0; c < n; ++c) {
    int p = c;
    for (int r = c + 1; r < n; ++r) if (std::abs(m[r * 2 * n + c]) > std::abs(m[p * 2 * n + c])) p = r;
    if (p != c) for (int j = 0; j < 2 * n; ++j) std::swap(m[c * 2 * n + j], m[p * 2 * n + j]);
    const double piv = 1.0 / m[c * 2 * n + c];
    for (int j = 0; j < 2 * n; ++j) m[c * 2 * n + j] *= piv;
    for (int r = 0; r < n; ++r) {
      if (r == c) continue;
      const double f = m[r * 2 * n + c];
      if (f != 0.0) for (int j = 0; j < 2 * n; ++j) m[r * 2 * n + j] -= f * m[c * 2 * n + j];
    }
  }
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) inv[i * n + j] = m[i * 2 * n + n + j];
}

struct SegmentMats {
  double ainv[kMaxN * kMaxN];
  double q[kMaxN * kMaxN];
};

void setup_segment(int n, int deriv, double t, SegmentMats* s) {
  const int h = n / 2;
  // LIN:568-583 computeQuadraticCostJacobian
  std::memset(s->q, 0, sizeof(s->q));
  for (int col = 0; col < n - deriv; ++col) {
    for (int row = 0; row < n - deriv; ++row) {
      const double exponent = (n - 1 - deriv) * 2 + 1 - row - col;
      s->q[(n - 1 - row) * n + (n - 1 - col)] =
          kBase.v[deriv][n - 1 - row] * kBase.v[deriv][n - 1 - col] * std::pow(t, exponent) * 2.0 / exponent;
    }
  }
  // LIN:112-121 setupMappingMatrix
  double a[kMaxN * kMaxN];
  for (int i = 0; i < h; ++i) {
    base_coeffs_with_time(n, i, 0.0, a + i * n);
    base_coeffs_with_time(n, i, t, a + (i + h) * n);
  }
  // LIN:143-179 invertMappingMatrix (Schur complement)
  double a_inv_diag[6], c[36], d[36], d_inv[36];
  for (int i = 0; i < h; ++i) a_inv_diag[i] = 1.0 / a[i * n + i];
  for (int i = 0; i < h; ++i) for (int j = 0; j < h; ++j) { c[i * h + j] = a[(h + i) * n + j]; d[i * h + j] = a[(h + i) * n + h + j]; }
  lu_inverse(h, d, d_inv);
  std::memset(s->ainv, 0, sizeof(s->ainv));
  for (int i = 0; i < h; ++i) s->ainv[i * n + i] = a_inv_diag[i];
  for (int i = 0; i < h; ++i) {
    for (int j = 0; j < h; ++j) {
      double acc = 0.0;
      for (int k = 0; k < h; ++k) acc += d_inv[i * h + k] * c[k * h + j];
      s->ainv[(h + i) * n + j] = -acc * a_inv_diag[j];
      s->ainv[(h + i) * n + h + j] = d_inv[i * h + j];
    }
  }
}

// Householder QR solve of the square system a x = b for nrhs right-hand sides (stand-in for SparseQR, LIN:365-374)
void qr_solve(int n, std::vector<double>& a, std::vector<double>& b, int nrhs) {
  for (int k = 0; k < n; ++k) {
    double norm = 0.0;
    for (int i = k; i < n; ++i) norm += a[i * n + k] * a[i * n + k];
    norm = std::sqrt(norm);
    if (norm == 0.0) continue;
    const double alpha = a[k * n + k] > 0 ? -norm : norm;
    thread_local std::vector<double> v;
    v.resize(n - k);
    for (int i = k; i < n; ++i) v[i - k] = a[i * n + k];
    v[0] -= alpha;
    double vnorm2 = 0.0;
    for (double x : v) vnorm2 += x * x;
    if (vnorm2 == 0.0) continue;
    for (int j = k; j < n; ++j) {
      double dot = 0.0;
      for (int i = k; i < n; ++i) dot += v[i - k] * a[i * n + j];
      const double f = 2.0 * dot / vnorm2;
      for (int i = k; i < n; ++i) a[i * n + j] -= f * v[i - k];
    }
    for (int r = 0; r < nrhs; ++r) {
      double dot = 0.0;
      for (int i = k; i < n; ++i) dot += v[i - k] * b[r * n + i];
      const double f = 2.0 * dot / vnorm2;
      for (int i = k; i < n; ++i) b[r * n + i] -= f * v[i - k];
    }
  }
  for (int r = 0; r < nrhs; ++r) {
    for (int i = n - 1; i >= 0; --i) {
      double acc = b[r * n + i];
      for (int j = i + 1; j < n; ++j) acc -= a[i * n + j] * b[r * n + j];
      b[r * n + i] = acc / a[i * n + i];
    }
  }
}

struct Plan {
  int n, h, k, dim, deriv, n_fixed, n_free;
  std::vector<int> col_of;  // [(k+1)*h] column of (vertex, derivative) in [d_F; d_P] order (LINH:288-295)
};

Plan make_plan(int n, int deriv, int k, int dim, const int* mask) {
  Plan p;
  p.n = n; p.h = n / 2; p.k = k; p.dim = dim; p.deriv = deriv;
  p.col_of.assign((k + 1) * p.h, -1);
  int nf = 0, np = 0;
  for (int v = 0; v <= k; ++v) for (int q = 0; q < p.h; ++q) if ((mask[v] >> q) & 1) p.col_of[v * p.h + q] = nf++;
  p.n_fixed = nf;
  for (int v = 0; v <= k; ++v) for (int q = 0; q < p.h; ++q) if (!((mask[v] >> q) & 1)) p.col_of[v * p.h + q] = nf + np++;
  p.n_free = np;
  return p;
}

// setupFromVertices + solveLinear + computeCost for one trajectory (LIN:57-109, :339-379, :124-140)
void solve_one(const Plan& p, const double* times, const double* dfix, double* coeffs, double* dfree, double* cost) {
  const int n = p.n, h = p.h, k = p.k, dim = p.dim, nf = p.n_fixed, np = p.n_free, na = nf + np;
  thread_local std::vector<SegmentMats> seg;
  seg.resize(k);
  for (int i = 0; i < k; ++i) setup_segment(n, p.deriv, times[i], &seg[i]);   // LIN:286-305
  thread_local std::vector<double> d_all;
  d_all.assign((size_t)dim * na, 0.0);
  for (int d = 0; d < dim; ++d) for (int c = 0; c < nf; ++c) d_all[(size_t)d * na + c] = dfix[(size_t)d * nf + c];
  if (np > 0) {
    // LIN:308-336 constructR: H_i = A_i^-T Q_i A_i^-1, R = M^T blkdiag(H) M (M is a 0/1 selection)
    thread_local std::vector<double> r;
    r.assign((size_t)na * na, 0.0);
    for (int i = 0; i < k; ++i) {
      double qa[kMaxN * kMaxN], hm[kMaxN * kMaxN];
      for (int a = 0; a < n; ++a) for (int b = 0; b < n; ++b) {
        double acc = 0.0;
        for (int c = 0; c < n; ++c) acc += seg[i].q[a * n + c] * seg[i].ainv[c * n + b];
        qa[a * n + b] = acc;
      }
      for (int a = 0; a < n; ++a) for (int b = 0; b < n; ++b) {
        double acc = 0.0;
        for (int c = 0; c < n; ++c) acc += seg[i].ainv[c * n + a] * qa[c * n + b];
        hm[a * n + b] = acc;
      }
      for (int a = 0; a < n; ++a) {
        const int ca = p.col_of[(a < h ? i : i + 1) * h + (a % h)];
        for (int b = 0; b < n; ++b) {
          const int cb = p.col_of[(b < h ? i : i + 1) * h + (b % h)];
          r[(size_t)ca * na + cb] += hm[a * n + b];
        }
      }
    }
    thread_local std::vector<double> rpp, rhs;
    rpp.resize((size_t)np * np);
    rhs.resize((size_t)dim * np);
    for (int a = 0; a < np; ++a) for (int b = 0; b < np; ++b) rpp[(size_t)a * np + b] = r[(size_t)(nf + a) * na + nf + b];
    for (int d = 0; d < dim; ++d) {
      for (int a = 0; a < np; ++a) {
        double acc = 0.0;
        for (int c = 0; c < nf; ++c) acc += r[(size_t)(nf + a) * na + c] * d_all[(size_t)d * na + c];
        rhs[(size_t)d * np + a] = -acc;   // LIN:371-372
      }
    }
    qr_solve(np, rpp, rhs, dim);          // LIN:365-374
    for (int d = 0; d < dim; ++d) for (int a = 0; a < np; ++a) {
      d_all[(size_t)d * na + nf + a] = rhs[(size_t)d * np + a];
      if (dfree) dfree[(size_t)d * np + a] = rhs[(size_t)d * np + a];
    }
  }
  // LIN:263-283 updateSegmentsFromCompactConstraints, LIN:124-140 computeCost
  double total = 0.0;
  for (int d = 0; d < dim; ++d) {
    for (int i = 0; i < k; ++i) {
      double nd[kMaxN], c[kMaxN];
      for (int a = 0; a < n; ++a) nd[a] = d_all[(size_t)d * na + p.col_of[(a < h ? i : i + 1) * h + (a % h)]];
      for (int a = 0; a < n; ++a) {
        double acc = 0.0;
        for (int b = 0; b < n; ++b) acc += seg[i].ainv[a * n + b] * nd[b];
        c[a] = acc;
        coeffs[((size_t)i * dim + d) * n + a] = acc;
      }
      if (cost) {
        for (int a = 0; a < n; ++a) {
          double acc = 0.0;
          for (int b = 0; b < n; ++b) acc += seg[i].q[a * n + b] * c[b];
          total += c[a] * acc;
        }
      }
    }
  }
  if (cost) *cost = 0.5 * total;
}

}  // namespace

extern "C" {

// times [B][K], dfix [B][D][n_fixed] -> coeffs [B][K][D][N], dfree [B][D][n_free] (optional), cost [B] (optional).
// Returns wall-clock seconds of the solve loop (threads = nthreads, contiguous batch split).
// `repeat` > 1 re-solves each thread's slice that many times (timing runs: one thread spawn for the whole sample).
double cpu_ref_solve_batch_repeat(int n, int deriv, int k, int dim, const int* mask, long long bsz, const double* times,
                                  const double* dfix, double* coeffs, double* dfree, double* cost, int nthreads,
                                  int repeat) {
  const Plan p = make_plan(n, deriv, k, dim, mask);
  if (nthreads < 1) nthreads = 1;
  const auto t0 = std::chrono::steady_clock::now();
  auto work = [&](long long lo, long long hi) {
    for (int rep = 0; rep < repeat; ++rep)
    for (long long b = lo; b < hi; ++b) {
      solve_one(p, times + b * k, dfix + b * (long long)dim * p.n_fixed, coeffs + b * (long long)k * dim * n,
                dfree ? dfree + b * (long long)dim * p.n_free : nullptr, cost ? cost + b : nullptr);
    }
  };
  if (nthreads == 1) {
    work(0, bsz);
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) th.emplace_back(work, bsz * t / nthreads, bsz * (t + 1) / nthreads);
    for (auto& x : th) x.join();
  }
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

double cpu_ref_solve_batch(int n, int deriv, int k, int dim, const int* mask, long long bsz, const double* times,
                           const double* dfix, double* coeffs, double* dfree, double* cost, int nthreads) {
  return cpu_ref_solve_batch_repeat(n, deriv, k, dim, mask, bsz, times, dfix, coeffs, dfree, cost, nthreads, 1);
}

// createRandomVertices (VERT:27-82, real std::mt19937 / std::uniform_real_distribution of this libstdc++) +
// estimateSegmentTimesNfabian (VERT:255-272) for trajectory seeds seed0 .. seed0+B-1.  Ends fully fixed
// (makeStartOrEnd, derivatives 1..max zero), interior position only.  positions [B][K+1][D], times [B][K].
// (fp-contract off: the reference's packages build without -march=native, so a*(b-a)+a is two roundings.)
__attribute__((optimize("fp-contract=off")))
void cpu_ref_generate(int k, int dim, long long bsz, unsigned long long seed0, double box, double v_max, double a_max,
                      double magic, double* positions, double* times) {
  for (long long b = 0; b < bsz; ++b) {
    std::mt19937 generator(seed0 + b);
    std::vector<std::uniform_real_distribution<double>> dist(dim, std::uniform_real_distribution<double>(-box, box));
    double* pos = positions + b * (long long)(k + 1) * dim;
    for (int d = 0; d < dim; ++d) pos[d] = dist[d](generator);
    for (int v = 1; v <= k; ++v) {
      while (true) {
        double n2 = 0.0;
        for (int d = 0; d < dim; ++d) {
          pos[v * dim + d] = dist[d](generator);
          const double diff = pos[v * dim + d] - pos[(v - 1) * dim + d];
          n2 += diff * diff;
        }
        if (std::sqrt(n2) > 0.2) break;
      }
    }
    for (int i = 0; i < k; ++i) {
      double n2 = 0.0;
      for (int d = 0; d < dim; ++d) {
        const double diff = pos[(i + 1) * dim + d] - pos[i * dim + d];
        n2 += diff * diff;
      }
      const double distance = std::sqrt(n2);
      times[b * k + i] = distance / v_max * 2 * (1.0 + magic * v_max / a_max * std::exp(-distance / v_max * 2));
    }
  }
}

int cpu_ref_hardware_threads() { return (int)std::thread::hardware_concurrency(); }

}  // extern "C"
