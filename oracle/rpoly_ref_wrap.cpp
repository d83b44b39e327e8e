// C entry around the REFERENCE's Jenkins-Traub root finder, compiled from the reference sources where they lie
// (oracle/Makefile target _ref/librpoly_ref.so).  Test infrastructure only.
#include <mav_trajectory_generation/rpoly/rpoly_ak1.h>

extern "C" int rpoly_ref_find_roots(const double* coefficients_increasing, int n, double* roots_re, double* roots_im) {
  Eigen::VectorXd c(n);
  for (int i = 0; i < n; ++i) c[i] = coefficients_increasing[i];
  Eigen::VectorXcd roots;
  const bool ok = mav_trajectory_generation::findRootsJenkinsTraub(c, &roots);
  const int m = (int)roots.size();
  for (int i = 0; i < m; ++i) {
    roots_re[i] = roots[i].real();
    roots_im[i] = roots[i].imag();
  }
  return ok ? m : -1 - m;
}
