// mtg_coop.hip -- the row-cooperative solve kernel (algorithm: mtg_coop.h): 16 lanes per trajectory-half, workgroup = two
// wavefronts (chain direction A / B) x four trajectories.  The LATENCY form: small launches of long chains.
// Standard shapes only (end vertices fix all h derivatives, interior vertices the position; D = 3; N = 8 / 10 / 12; any K >= 2),
// any input strides, coefficient output only.
#include <hip/hip_runtime.h>

#include "../../include/mtg_hip.h"
#include "mtg_coop.h"

namespace {

constexpr int kWave = 64;

struct CoopParams {
  const double* times; long long ts_b, ts_k;
  const double* dfix;  long long fs_b, fs_d, fs_c;
  double* coeffs;
  int* status; int* tstatus;
  long long B;
  int K, deriv;
};

// device back end of mtg_coop.h: a value is this lane's double; the cross-lane reads are DPP row_newbcast sources of v_fmac_f64
struct CoopDev {
  using V = double;
  using P = bool;
  static __device__ __forceinline__ V splat(double x) { return x; }
  static __device__ __forceinline__ V add(V a, V b) { return a + b; }
  static __device__ __forceinline__ V mul(V a, V b) { return a * b; }
  static __device__ __forceinline__ V fma(V a, V b, V c) { return __builtin_fma(a, b, c); }
  static __device__ __forceinline__ V neg(V a) { return -a; }
  static __device__ __forceinline__ V rcp(V a) { return mtg_rcp(a); }
  static __device__ __forceinline__ V sel(P c, V a, V b) { return c ? a : b; }
  static __device__ __forceinline__ V powi(V x, int e) { return mtg_powi<11>(x, e); }
  template <int E> static __device__ __forceinline__ V powc(V x) { return mtg_powi<E>(x, E); }
  static __device__ __forceinline__ P pand(P a, P b) { return a && b; }
  static __device__ __forceinline__ P por(P a, P b) { return a || b; }
  static __device__ __forceinline__ P pfalse() { return false; }
  static __device__ __forceinline__ P not_gt0(V a) { return !(a > 0.0); }
  // "VALU writes a VGPR -> a DPP instruction reads it" needs two wait states, and the hazard recogniser does not look inside
  // inline asm: every value a DPP source is read from is first pinned by an (empty) asm, then ONE s_nop 1 follows -- volatile
  // asm statements keep their order, so the producers sit in front of the wait states.  tools/check_dpp_hazards.py verifies the
  // disassembly (no VALU write of a DPP source within the two preceding instructions).
  static __device__ __forceinline__ void settle(V& x) { asm volatile("s_nop 1" : "+v"(x)); }
  template <class A, class B, class C>
  static __device__ __forceinline__ void settle_rows(A& a, B& b, C& c) {
    for (auto& e : a) asm volatile("" : "+v"(e));
    for (auto& e : b) asm volatile("" : "+v"(e));
    for (auto& e : c) asm volatile("" : "+v"(e));
    asm volatile("s_nop 1");
  }
  template <class A>
  static __device__ __forceinline__ void settle_vec(A& a) {
    for (auto& e : a) asm volatile("" : "+v"(e));
    asm volatile("s_nop 1");
  }
  template <int L>
  static __device__ __forceinline__ void fmac_bcast(V& acc, V src, V m) {
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(m), "n"(L));
  }
};

struct DevLanes {
  int l16;
  template <class F> __device__ __forceinline__ double make(F f) const { return f(l16); }
  template <class F> __device__ __forceinline__ bool pred(F f) const { return f(l16); }
};

template <int H, int D>
__host__ __device__ constexpr size_t coop_step_doubles() { return (size_t)(H - 1 + D) * (4 * (H - 1) + 1); }   // per wave and step
template <int H, int D>
__host__ __device__ constexpr size_t coop_xch_doubles() { return (size_t)(H - 1 + D) * kWave; }                // per wave

// Inputs through 32-bit byte offsets from the two wave-uniform base pointers (global_load with an SGPR base): a lane's offset is
// its row's trajectory part plus a scalar (segment / column) part.  The launcher checks that every offset fits 32 bits.
template <int H, int D>
struct DevIO {
  static constexpr int N = 2 * H, F = H - 1, COLS = 4 * F + 1;
  const CoopParams& P;
  long long b;
  bool active;
  int l16;
  bool store_lane;        // active trajectory and lane < N: this lane stores a coefficient
  unsigned cb;            // byte offset of (this row's trajectory, coefficient `lane`) in coeffs
  unsigned tb, fb;        // byte offsets of this row's trajectory in times / d_fixed
  unsigned tsk, fsd, fsc; // byte strides (segment; dimension, column)
  int colE, colO;         // this lane's column of the step area at even / odd step parity (COLS - 1: the dump column)
  int p0;                 // parity of chain step 0
  double* steps;          // this wave's step area: [j][k][COLS]
  __device__ __forceinline__ double time(int seg) const {
    return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(P.times) + (tb + (unsigned)seg * tsk));
  }
  __device__ __forceinline__ double fixed(int dm, int col) const {
    return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(P.dfix) + (fb + (unsigned)dm * fsd + (unsigned)col * fsc));
  }
  __device__ __forceinline__ double* slot(int j, int k) const {
    return steps + ((unsigned)(j * (F + D) + k) * (unsigned)COLS + (unsigned)(((j + p0) & 1) ? colO : colE));
  }
  __device__ __forceinline__ void save(int j, int k, double v) const { *slot(j, k) = v; }
  __device__ __forceinline__ double load(int j, int k) const { return *slot(j, k); }
  // (32-bit byte offset from the wave-uniform coefficient base: the launcher checks B K D N 8 < 4 GiB)
  __device__ __forceinline__ void store(int seg, const double (&v)[D]) const {
    if (!store_lane) return;
    char* base = reinterpret_cast<char*>(P.coeffs) + (cb + (unsigned)seg * (unsigned)(D * N * 8));
#pragma unroll
    for (int dm = 0; dm < D; ++dm) *reinterpret_cast<double*>(base + dm * N * 8) = v[dm];
  }
};

template <int H, int D>
__global__ __launch_bounds__(2 * kWave) void mtg_solve_coop_kernel(CoopParams P) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int N = 2 * H, F = H - 1;
  const int lane = threadIdx.x & (kWave - 1);
  const int dir = threadIdx.x >> 6;          // wave-uniform: 0 = forward chain from vertex 0, 1 = backward chain from vertex K
  const int row = lane >> 4, l16 = lane & 15;
  const int K = P.K, KA = (K + 1) / 2, KB = K / 2;
  const int kc = dir == 0 ? KA : KB;
  long long b = (long long)blockIdx.x * 4 + row;
  const bool active = b < P.B;
  if (!active) b = P.B - 1;
  // LDS: [exchange A | exchange B | steps A (KA) | steps B (KB)]
  double* xch_mine = lds + (size_t)dir * coop_xch_doubles<H, D>();
  const double* xch_other = lds + (size_t)(1 - dir) * coop_xch_doubles<H, D>();
  double* steps = lds + 2 * coop_xch_doubles<H, D>() + (dir == 0 ? 0 : (size_t)KA * coop_step_doubles<H, D>());
  const int g = l16 >> 3, i = l16 & 7;
  const int col = row * F + i;
  DevIO<H, D> io{P, b, active, l16, active && l16 < N, (unsigned)((b * K * D * N + l16) * 8), (unsigned)(b * P.ts_b * 8), (unsigned)(b * P.fs_b * 8), (unsigned)(P.ts_k * 8), (unsigned)(P.fs_d * 8),
                 (unsigned)(P.fs_c * 8), (g == 0 && i < F) ? col : 4 * F, (g == 1 && i < F) ? col : 4 * F, (kc - 1) & 1, steps};
  mtgc::Coop<CoopDev, H, D> cp;
  const double* h1 = kH1 + mtg_h1_offset(N, P.deriv);
  cp.init(h1, DevLanes{l16});
  double posm[D];
  if (dir == 0) mtgc::coop_forward<CoopDev, H, D, 1>(cp, io, K, kc, P.deriv, posm);
  else mtgc::coop_forward<CoopDev, H, D, -1>(cp, io, K, kc, P.deriv, posm);
  // middle vertex: both directions publish their rows (group 1, block B1), add the other's, solve
#pragma unroll
  for (int q = 0; q < F; ++q) xch_mine[q * kWave + lane] = cp.B1[q];
#pragma unroll
  for (int dm = 0; dm < D; ++dm) xch_mine[(F + dm) * kWave + lane] = cp.R[dm];
  __syncthreads();
  double other[F + D];
#pragma unroll
  for (int k = 0; k < F + D; ++k) other[k] = xch_other[k * kWave + lane];
  cp.solve_middle(other);
  if (dir == 0) mtgc::coop_backward<CoopDev, H, D, 1>(cp, io, K, kc, P.deriv, posm);
  else mtgc::coop_backward<CoopDev, H, D, -1>(cp, io, K, kc, P.deriv, posm);
  const int flags = (cp.flag_time ? MTG_FLAG_BAD_TIME : 0) | (cp.flag_singular ? MTG_FLAG_SINGULAR : 0);
  if (flags != 0 && active) {
    atomicOr(P.status, flags);
    if (P.tstatus) atomicOr(P.tstatus + b, flags);
  }
}

template <int H, int D>
int launch(hipStream_t st, const CoopParams& P) {
  const int KA = (P.K + 1) / 2, KB = P.K / 2;
  const size_t lds = (2 * coop_xch_doubles<H, D>() + (size_t)(KA + KB) * coop_step_doubles<H, D>()) * sizeof(double);
  if (lds > 160 * 1024) return 1;
  // (set on every launch that needs it: the attribute belongs to the CURRENT device's copy of the kernel -- a device group runs
  // this on several devices -- and the call is a host-side table update)
  if (lds > 48 * 1024 &&
      hipFuncSetAttribute((const void*)mtg_solve_coop_kernel<H, D>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
    return 2;
  const unsigned grid = (unsigned)((P.B + 3) / 4);
  hipLaunchKernelGGL((mtg_solve_coop_kernel<H, D>), dim3(grid), dim3(2 * kWave), lds, st, P);
  return 0;
}

}  // namespace

// 0: launched; 1: shape not covered (the caller takes another form); 2: runtime error.  Strides in elements.
int mtg_coop_launch(void* stream, int H, int D, int K, int deriv, long long B, const double* times, long long ts_b, long long ts_k,
                    const double* dfix, long long fs_b, long long fs_d, long long fs_c, double* coeffs, int* status, int* tstatus) {
  if (D != 3 || K < 2 || B <= 0) return 1;
  {   // 32-bit byte offsets into times / d_fixed (the form is for small launches; larger ones take another form)
    const long long n_fixed = 2 * H + (K - 1);
    const long long tmax = (B - 1) * ts_b + (K - 1) * ts_k, fmax = (B - 1) * fs_b + (D - 1) * fs_d + (n_fixed - 1) * fs_c;
    if (ts_b < 0 || ts_k < 0 || fs_b < 0 || fs_d < 0 || fs_c < 0 || tmax * 8 >= (1ll << 32) || fmax * 8 >= (1ll << 32)) return 1;
    if (B * K * D * (2ll * H) * 8 >= (1ll << 32)) return 1;
  }
  CoopParams P{times, ts_b, ts_k, dfix, fs_b, fs_d, fs_c, coeffs, status, tstatus, B, K, deriv};
  switch (H) {
    case 4: return launch<4, 3>((hipStream_t)stream, P);
    case 5: return launch<5, 3>((hipStream_t)stream, P);
    case 6: return launch<6, 3>((hipStream_t)stream, P);
  }
  return 1;
}

// LDS bytes a launch of chain length K needs (0: shape not covered) -- the form is eligible while this fits one CU
size_t mtg_coop_lds_bytes(int H, int D, int K) {
  if (D != 3 || K < 2 || H < 4 || H > 6) return 0;
  const size_t f = H - 1, step = (f + D) * (4 * f + 1), xch = (f + D) * kWave;
  return (2 * xch + (size_t)K * step) * sizeof(double);
}
