// mtg_refine_dd.h -- the double-double residual of MTG_FLAG_REFINE, one (trajectory, dimension) at a time: plain C++17 (device:
// mtg_refine.hip; host: tests/refine_emu.cpp, which the CPU suite compares with the 50-digit residual).  See mtg_refine.hip.
#ifndef MTG_REFINE_DD_H_
#define MTG_REFINE_DD_H_
#if defined(__HIPCC__)
#define MTG_RD __device__ __forceinline__
#else
#include <cmath>
#define MTG_RD inline
#endif

// NO floating-point contraction in this file.  hipcc's default is -ffp-contract=fast, and the AMDGPU back end fuses multiplies that
// have SEVERAL uses: in two_prod's  p = a * b; e = fma(a, b, -p)  followed by  s = p + l  it rewrites the sum as fma(a, b, l) -- the
// exact product plus l -- while e still carries the product's rounding error: that error is then counted twice and every
// double-double product is only double-accurate.  (Found in round 6 on the device: the residual came out 77-150 % wrong in the
// entries that are differences of 1e-8-relative size, while the same source built with g++ -- which contracts single-use
// multiplies only -- matched the 50-digit residual to 2e-16.)
#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace mtg_refine {

struct dd { double hi, lo; };

MTG_RD double rd_fma(double a, double b, double c) {
#if defined(__HIPCC__)
  return __builtin_fma(a, b, c);
#else
  return std::fma(a, b, c);
#endif
}

MTG_RD dd two_sum(double a, double b) {
  const double s = a + b, bb = s - a;
  return dd{s, (a - (s - bb)) + (b - bb)};
}
MTG_RD dd quick_two_sum(double a, double b) {   // |a| >= |b|
  const double s = a + b;
  return dd{s, b - (s - a)};
}
MTG_RD dd two_prod(double a, double b) {
  const double p = a * b;
  return dd{p, rd_fma(a, b, -p)};
}
MTG_RD dd dd_add(dd a, dd b) {                  // accurate (IEEE-style) double-double addition
  dd s = two_sum(a.hi, b.hi);
  const dd t = two_sum(a.lo, b.lo);
  s.lo += t.hi;
  s = quick_two_sum(s.hi, s.lo);
  s.lo += t.lo;
  return quick_two_sum(s.hi, s.lo);
}
MTG_RD dd dd_mul(dd a, dd b) {
  dd p = two_prod(a.hi, b.hi);
  p.lo += a.hi * b.lo + a.lo * b.hi;
  return quick_two_sum(p.hi, p.lo);
}
MTG_RD dd dd_mul_d(dd a, double b) {
  dd p = two_prod(a.hi, b);
  p.lo = rd_fma(a.lo, b, p.lo);
  return quick_two_sum(p.hi, p.lo);
}
MTG_RD dd dd_recip(dd a) {                      // 1 / a to ~1e-31
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

// one (trajectory b, dimension dm): r = -(R_PP x + R_PF d_F) over the trajectory's free slots; hh / hl: H(1) high / low words
template <int H>
MTG_RD void residual_dd_one(const RefineArgs& A, long long b, int dm, const double* hh, const double* hl) {
  constexpr int N = 2 * H;
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

}  // namespace mtg_refine
#endif  // MTG_REFINE_DD_H_
