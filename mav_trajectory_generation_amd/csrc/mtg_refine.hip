// mtg_refine.hip -- MTG_FLAG_REFINE: the residual of one step of iterative refinement, in DOUBLE-DOUBLE.
//
// The free system R_PP d_P = -R_PF d_F (impl/polynomial_optimization_linear_impl.h:360-375) of an N = 12 trajectory with a
// segment-time ratio of ~17 has a condition number of ~2e9: the float64 sweep -- and every float64 evaluation of the reference's
// formulas, the reference's own QR included -- returns d_P good to ~2e-7.  The error is NOT in the float64 entries of R: a
// correctly rounded R solved in float64 is just as far off (2.2e-7), and refinement with a residual formed from float64 entries
// converges to the solution of the rounded matrix (round 5's study, tools/n12_refinement_study.py).  What converges to the true
// solution is refinement whose residual  r = -(R_PP x + R_PF d_F)  is formed from entries that are exact to ~1e-32:
//     R = sum over segments of  T^(1-2d) S H(1) S   (the scaling identity)
// with H(1) a double-double constant (kH1 + kH1Lo: the exact rational to 2^-106), the powers of T and T^(1-2d) in double-double,
// and the sums over a segment's twelve columns and over a vertex's two segments accumulated in double-double (two-sum / two-prod
// with FMA).  One such step takes the worst of 400 N = 12 / K = 16 trajectories from 2e-7 to 5e-15 (profiles/r06_n12_refinement.txt):
// the float64 factorisation is an excellent preconditioner (cond x eps ~ 2e-7 is the contraction per step).
// The correction solve R_PP delta = r and the coefficient recovery are the ordinary float64 kernels (mtg_abi.hip: solve_refined).
//
// One thread per (trajectory, dimension); walks the segments once, keeps the double-double partial sums of the vertex it shares
// with the next segment.  Generic over masks / strides (it reads the plan's device tables): this is an accuracy mode, not the
// throughput path -- ~2e5 flops per N = 12 / K = 32 trajectory.
#include <hip/hip_runtime.h>

#include <cstdint>

#define MTG_TABLE_QUAL __constant__ const
#include "mtg_tables.inc"
#include "mtg_tables_dd.inc"

#include "mtg_refine_dd.h"

namespace {
using mtg_refine::RefineArgs;

template <int H>
__global__ __launch_bounds__(64) void mtg_residual_dd_kernel(RefineArgs A) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= A.B * A.D) return;
  const long long b = i / A.D;
  mtg_refine::residual_dd_one<H>(A, b, (int)(i - b * A.D), kH1 + A.h1off, kH1Lo + A.h1off);
}

// x <- x + delta over the free slots ([B][D][n_free]; x with strides, delta contiguous)
__global__ void mtg_refine_axpy_kernel(double* x, long long ps_b, long long ps_d, long long ps_c, const double* delta, long long B, int D, int np) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * D * np) return;
  const int j = (int)(i % np);
  const int dm = (int)((i / np) % D);
  const long long b = i / ((long long)np * D);
  x[b * ps_b + dm * ps_d + (long long)j * ps_c] += delta[i];
}

}  // namespace

// (declared in mtg_kernels.h)
int mtg_refine_residual_launch(void* stream, int H, int K, int D, int deriv, int h1off, const int* d_vmask, const int* d_offF,
                               const int* d_offP, long long B, const double* times, long long ts_b, long long ts_k,
                               const double* dfix, long long fs_b, long long fs_d, long long fs_c, const double* dfree,
                               long long ps_b, long long ps_d, long long ps_c, double* rhs, int n_free) {
  RefineArgs A{times, ts_b, ts_k, dfix, fs_b, fs_d, fs_c, dfree, ps_b, ps_d, ps_c, rhs, d_vmask, d_offF, d_offP, B, K, D, deriv, n_free, h1off};
  const long long n = B * D;
  if (n <= 0 || n_free <= 0) return 0;
  const dim3 grid((unsigned)((n + 63) / 64)), block(64);
  hipStream_t st = (hipStream_t)stream;
  switch (H) {
    case 1: hipLaunchKernelGGL(mtg_residual_dd_kernel<1>, grid, block, 0, st, A); break;
    case 2: hipLaunchKernelGGL(mtg_residual_dd_kernel<2>, grid, block, 0, st, A); break;
    case 3: hipLaunchKernelGGL(mtg_residual_dd_kernel<3>, grid, block, 0, st, A); break;
    case 4: hipLaunchKernelGGL(mtg_residual_dd_kernel<4>, grid, block, 0, st, A); break;
    case 5: hipLaunchKernelGGL(mtg_residual_dd_kernel<5>, grid, block, 0, st, A); break;
    case 6: hipLaunchKernelGGL(mtg_residual_dd_kernel<6>, grid, block, 0, st, A); break;
    default: return -1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int mtg_refine_axpy_launch(void* stream, double* x, long long ps_b, long long ps_d, long long ps_c, const double* delta, long long B, int D, int np) {
  const long long n = B * D * np;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(mtg_refine_axpy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ps_b, ps_d, ps_c, delta, B, D, np);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
