// CPU check of the Eigen stand-in the compiled reference (oracle/_ref) is built against -- test infrastructure testing
// test infrastructure: oracle/ref_shim/mini_eigen.h's inverse() (cofactors <= 4x4, partial-pivoting LU above: the classes of
// algorithm Eigen uses at LIN:171) and SparseQR (minimum-degree column order, left-looking Householder, Eigen's pivot
// threshold and basic solution, LIN:365-375).  Prints lines the Python test (tests/test_mini_eigen.py) parses.
#include <cstdio>
#include <random>
#include <vector>

#include "mini_eigen.h"

int main() {
  std::mt19937 rng(7);
  std::uniform_real_distribution<double> u(-1.0, 1.0);
  // inverse: || A inv(A) - I ||_max for sizes 1 .. 6 (4x4 and below: cofactors; 5, 6: LU)
  for (int n = 1; n <= 6; ++n) {
    Eigen::MatrixXd a(n, n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) a(i, j) = u(rng) + (i == j ? 2.0 : 0.0);
    Eigen::MatrixXd inv = a.inverse(), p = a * inv;
    double err = 0;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) err = std::max(err, std::abs(p(i, j) - (i == j ? 1.0 : 0.0)));
    std::printf("inverse n=%d residual=%.3e\n", n, err);
  }
  // a permutation-needing LU: zero leading pivot
  {
    Eigen::MatrixXd a(5, 5);
    for (int i = 0; i < 5; ++i) for (int j = 0; j < 5; ++j) a(i, j) = u(rng);
    a(0, 0) = 0.0;
    Eigen::MatrixXd p = a * a.inverse();
    double err = 0;
    for (int i = 0; i < 5; ++i) for (int j = 0; j < 5; ++j) err = std::max(err, std::abs(p(i, j) - (i == j ? 1.0 : 0.0)));
    std::printf("inverse zero-pivot residual=%.3e\n", err);
  }
  // QR column order on the block-tridiagonal SPD pattern of R_PP: nb blocks of size f
  for (int f : {3, 4, 5}) {
    const int nb = 7, n = nb * f;
    Eigen::MatrixXd m(n, n);
    m.setZero();
    for (int b = 0; b < nb; ++b)
      for (int i = 0; i < f; ++i)
        for (int j = 0; j < f; ++j) {
          m(b * f + i, b * f + j) = (i == j ? 6.0 : 0.0) + 0.3 * u(rng);
          if (b + 1 < nb) { const double v = 0.5 * u(rng); m(b * f + i, (b + 1) * f + j) = v; m((b + 1) * f + j, b * f + i) = v; }
        }
    for (int i = 0; i < n; ++i) for (int j = 0; j < i; ++j) m(i, j) = m(j, i);
    Eigen::SparseMatrix<double> sm(m);
    Eigen::SparseQR<Eigen::SparseMatrix<double>, Eigen::COLAMDOrdering<int>> qr(sm);
    std::printf("qr f=%d rank=%d order=", f, (int)qr.rank());
    for (Eigen::Index c : qr.acceptedColumns()) std::printf("%d,", (int)c);
    Eigen::VectorXd b(n);
    for (int i = 0; i < n; ++i) b[i] = u(rng);
    Eigen::VectorXd x = qr.solve(b), r = m * x - b;
    std::printf(" residual=%.3e\n", r.norm());
  }
  // rank-deficient system: column 2 = column 0 + column 1 -> rank 3, basic solution has one exact zero, A x = b (b in range)
  {
    Eigen::MatrixXd a(4, 4);
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) a(i, j) = u(rng);
    for (int i = 0; i < 4; ++i) a(i, 2) = a(i, 0) + a(i, 1);
    Eigen::VectorXd x0(4);
    for (int i = 0; i < 4; ++i) x0[i] = u(rng);
    Eigen::VectorXd b = a * x0;
    Eigen::SparseMatrix<double> sm(a);
    Eigen::SparseQR<Eigen::SparseMatrix<double>, Eigen::COLAMDOrdering<int>> qr(sm);
    Eigen::VectorXd x = qr.solve(b), r = a * x - b;
    int zeros = 0;
    for (int i = 0; i < 4; ++i) zeros += x[i] == 0.0;
    std::printf("rankdef rank=%d zeros=%d residual=%.3e\n", (int)qr.rank(), zeros, r.norm());
  }
  return 0;
}
