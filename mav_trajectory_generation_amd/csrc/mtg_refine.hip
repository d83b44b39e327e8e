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

namespace {

struct dd { double hi, lo; };

__device__ __forceinline__ dd two_sum(double a, double b) {
  const double s = a + b, bb = s - a;
  return dd{s, (a - (s - bb)) + (b - bb)};
}
__device__ __forceinline__ dd quick_two_sum(double a, double b) {   // |a| >= |b|
  const double s = a + b;
  return dd{s, b - (s - a)};
}
__device__ __forceinline__ dd two_prod(double a, double b) {
  const double p = a * b;
  return dd{p, __builtin_fma(a, b, -p)};
}
__device__ __forceinline__ dd dd_add(dd a, dd b) {                  // accurate (IEEE-style) double-double addition
  dd s = two_sum(a.hi, b.hi);
  const dd t = two_sum(a.lo, b.lo);
  s.lo += t.hi;
  s = quick_two_sum(s.hi, s.lo);
  s.lo += t.lo;
  return quick_two_sum(s.hi, s.lo);
}
__device__ __forceinline__ dd dd_mul(dd a, dd b) {
  dd p = two_prod(a.hi, b.hi);
  p.lo += a.hi * b.lo + a.lo * b.hi;
  return quick_two_sum(p.hi, p.lo);
}
__device__ __forceinline__ dd dd_mul_d(dd a, double b) {
  dd p = two_prod(a.hi, b);
  p.lo = __builtin_fma(a.lo, b, p.lo);
  return quick_two_sum(p.hi, p.lo);
}
__device__ __forceinline__ dd dd_recip(dd a) {                      // 1 / a to ~1e-31
  const double q0 = 1.0 / a.hi;
  // r = 1 - a q0 in double-double, q = q0 + q0 r
  dd e = two_prod(a.hi, q0);
  const double r = (1.0 - e.hi) - e.lo - a.lo * q0;
  const double q1 = q0 * r;
  // one more correction term: r2 = 1 - a (q0 + q1)
  dd aq = dd_add(dd_mul_d(a, q0), dd_mul_d(a, q1));
  const double r2 = (1.0 - aq.hi) - aq.lo;
  dd q = quick_two_sum(q0, q1);
  q.lo += q0 * r2;
  return quick_two_sum(q.hi, q.lo);
}

struct RefineArgs {
  const double* times;  long long ts_b, ts_k;
  const double* dfix;   long long fs_b, fs_d, fs_c;
  const double* dfree;  long long ps_b, ps_d, ps_c;
  double* rhs;          // [B][D][n_free], contiguous
  const int* vmask;     // [K + 1]
  const int* offF;      // [K + 2]
  const int* offP;      // [K + 2]
  long long B;
  int K, D, deriv, n_free, h1off;
};

template <int H>
__global__ __launch_bounds__(64) void mtg_residual_dd_kernel(RefineArgs A) {
  constexpr int N = 2 * H;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= A.B * A.D) return;
  const long long b = i / A.D;
  const int dm = (int)(i - b * A.D);
  const double* hh = kH1 + A.h1off;
  const double* hl = kH1Lo + A.h1off;
  double* out = A.rhs + (b * A.D + dm) * (long long)A.n_free;
  auto value = [&](int v, int p, int mask) -> double {      // derivative p at vertex v: fixed value or current free value
    const int below = __builtin_popcount((unsigned)(mask & ((1 << p) - 1)));
    if ((mask >> p) & 1) return A.dfix[b * A.fs_b + dm * A.fs_d + (long long)(A.offF[v] + below) * A.fs_c];
    return A.dfree[b * A.ps_b + dm * A.ps_d + (long long)(A.offP[v] + (p - below)) * A.ps_c];
  };
  dd carry[H];                                               // (R d) rows of the current left vertex from the PREVIOUS segment
#pragma unroll
  for (int p = 0; p < H; ++p) carry[p] = dd{0.0, 0.0};
  int ml = A.vmask[0];
  double dv[N];
#pragma unroll
  for (int p = 0; p < H; ++p) dv[p] = value(0, p, ml);
  for (int k = 0; k < A.K; ++k) {
    const int mr = A.vmask[k + 1];
#pragma unroll
    for (int p = 0; p < H; ++p) dv[H + p] = value(k + 1, p, mr);
    const double T = A.times[b * A.ts_b + (long long)k * A.ts_k];
    dd s[H];                                                 // T^p
    s[0] = dd{1.0, 0.0};
    if constexpr (H > 1) s[1] = dd{T, 0.0};
#pragma unroll
    for (int p = 2; p < H; ++p) s[p] = dd_mul_d(s[p - 1], T);
    dd base;                                                 // T^(1 - 2 d)
    if (A.deriv == 0) {
      base = dd{T, 0.0};
    } else {
      dd tp = dd{T, 0.0};
      for (int e = 1; e < 2 * A.deriv - 1; ++e) tp = dd_mul_d(tp, T);
      base = dd_recip(tp);
    }
    dd y[N];                                                 // S d
#pragma unroll
    for (int q = 0; q < N; ++q) y[q] = dd_mul_d(s[q % H], dv[q]);
    dd z[N];
#pragma unroll
    for (int p = 0; p < N; ++p) {
      const bool want = !(((p < H ? ml : mr) >> (p % H)) & 1);      // free rows only
      dd acc{0.0, 0.0};
      if (want) {
#pragma unroll
        for (int q = 0; q < N; ++q) acc = dd_add(acc, dd_mul(dd{hh[p * N + q], hl[p * N + q]}, y[q]));
        acc = dd_mul(dd_mul(base, s[p % H]), acc);
      }
      z[p] = acc;
    }
    // the left vertex is complete: previous segment's end rows + this segment's start rows
    {
      const int off = A.offP[k];
      int col = 0;
#pragma unroll
      for (int p = 0; p < H; ++p) {
        if ((ml >> p) & 1) continue;
        const dd r = dd_add(carry[p], z[p]);
        out[off + col] = -(r.hi + r.lo);
        ++col;
      }
    }
#pragma unroll
    for (int p = 0; p < H; ++p) { carry[p] = z[H + p]; dv[p] = dv[H + p]; }
    ml = mr;
  }
  {
    const int off = A.offP[A.K];
    int col = 0;
#pragma unroll
    for (int p = 0; p < H; ++p) {
      if ((ml >> p) & 1) continue;
      out[off + col] = -(carry[p].hi + carry[p].lo);
      ++col;
    }
  }
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
