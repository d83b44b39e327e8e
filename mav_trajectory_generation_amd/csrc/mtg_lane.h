// mtg_lane.h -- per-lane algorithm of the batched solveLinear() kernels (gfx950).
//
// One trajectory is solved by TWO lanes that live in two different wavefronts of one
// workgroup: wave A walks the vertex chain forward from vertex 0, wave B walks it backward
// from vertex K; they meet at the middle vertex (a "twisted" block-LDL^T factorisation of
// the block-tridiagonal R_PP of impl/polynomial_optimization_linear_impl.h:360-367), swap
// their Schur complements through LDS, and back-substitute outwards, recovering polynomial
// coefficients segment by segment (impl/...:263-283) as the free derivatives become known.
// Direction is wave-uniform, so every mask test / table lookup below is scalar.
//
// Maths.  With S(T) = diag(T^0..T^(h-1)) on both segment ends,
//   H(T) = A(T)^-T Q(T) A(T)^-1 = T^(1-2d) S H(1) S        (impl/...:318 without the N^3 products)
//   coeffs_j = T^-j * sum_k A(1)^-1[j][k] * (S d)_k         (impl/...:276-277)
// H(1), A(1)^-1 are exact rational constants (mtg_tables.inc).  Time reversal maps the
// end/end block onto the start/start block with signs (-1)^(a+b); wave B therefore runs the
// very same code with the signed scale vector s_p = (-T)^p.
// The chain is solved in the variables s_p x_p of the current segment (see "Scaled-variable chain" at MtgLane): there
// the blocks of H(T) are the constant blocks of H(1), and only the carried Schur complement is converted from one
// segment's scaling to the next.
//
// The file is plain C++17 so that tests/host_emu.cpp can run the identical code on the CPU
// (test infrastructure only; the product path is the HIP build).
#ifndef MTG_LANE_H_
#define MTG_LANE_H_

#include <stdint.h>

#if defined(__HIPCC__)
#define MTG_HD __device__ __forceinline__
#define MTG_TABLE_QUAL __constant__ const
#else
#include <cmath>
#define MTG_HD inline
#define MTG_TABLE_QUAL static const
#endif

#include "mtg_tables.inc"

enum { MTG_FLAG_BAD_TIME = 1, MTG_FLAG_SINGULAR = 2 };

struct MtgParams {
  const double* times;  long long ts_b, ts_k;
  const double* dfix;   long long fs_b, fs_d, fs_c;
  double* coeffs;                                   // [B][K][Dtot][N]
  double* dfree;        long long ps_b, ps_d, ps_c; // optional output (solve) / input (update)
  double* cost;                                     // optional, pre-zeroed, atomically accumulated
  double* ws;           long long ws_stride;        // generic mode back-substitution store
  long long ws_share;   // dimension-in-lane long chains (MtgCfg::DLW): element offset from a lane's own workspace column to
                        // the column of its trajectory's dimension-0 lane
  unsigned lds_steps;   // MtgCfg::LSJ: LDS byte address of this lane's column in its wave's step area (row stride 64 doubles)
  int* status;                                      // OR of MTG_FLAG_* over the batch
  int* tstatus;                                     // optional [B]: OR of MTG_FLAG_* per trajectory (pre-zeroed)
  const int* vmask;                                 // [K+1] fixed masks        (generic mode)
  const int* offF;                                  // [K+2] prefix of fixed slots
  const int* offP;                                  // [K+2] prefix of free slots
  long long B;
  int K, Dtot, dim0, deriv, h1off, ainvoff;
  // Virtual batch of perturbed-time problems (mtg_mellinger_cost_gradient): variant n of trajectory b uses the times of b
  // with T_(n-1) += pert_h and every other T_k -= pert_corr, clamped from below (variant 0: unperturbed).  The kernels map
  // tile -> (variant, tile of the real batch); pert_seg is that tile's perturbed segment (-1: none).
  int pert_on, pert_seg, pert_tpv;   // pert_tpv: tiles per variant
  double pert_h, pert_corr, pert_lo;
  // Generic mode only (MTG_FLAG_REFINE, the correction solve R_PP delta = r of one step of iterative refinement): an EXPLICIT
  // right-hand side over the free slots, [B][D][n_free] with these strides, ADDED to the one the fixed values produce (which the
  // caller makes zero); null otherwise
  const double* rhs;    long long rh_b, rh_d, rh_c;
};

// segment time as the virtual problem sees it (identity unless a perturbed-time launch)
MTG_HD double mtg_perturb(const MtgParams& P, int seg, double T) {
  if (!P.pert_on || P.pert_seg < 0) return T;
  T += seg == P.pert_seg ? P.pert_h : -P.pert_corr;
  return T < P.pert_lo ? P.pert_lo : T;
}

// offsets into the generated tables (same formulas as gen_tables.py)
constexpr int mtg_h1_offset(int n, int d) {
  int off = 0;
  for (int m = 2; m < n; m += 2) off += (m / 2) * m * m;
  return off + d * n * n;
}
constexpr int mtg_ainv_offset(int n) {
  int off = 0;
  for (int m = 2; m < n; m += 2) off += (m / 2) * m;
  return off;
}

#ifndef MTG_PARTIAL_ALL
// 1: EVERY kernel's forward step eliminates partially (Schur update from W = L^-1 U, the back-substitution for G after it).  Built and
// measured in round 4 (profiles/r04v_*): the K = 8 kernels are unchanged at 100k (38.0 / 49.5 / 93.0 us), the bench line 0.668-0.671
// (inside the box-to-box spread), and it exposed that a rank-deficient system is recognised by the SIGN of a pivot that is pure
// round-off (mtg_ldl: !(d > 0)): with this association the under-constrained case of tests/cpp/test_veneer.cpp:424 gets a tiny
// positive pivot and is not flagged.  Off until the pivot test is a relative threshold.
// (Round 5: rank deficiency is structural now -- see mtg_ldl -- so the association no longer decides what is flagged.)
#define MTG_PARTIAL_ALL 0
#endif
template <int H_, int D_, int KT_, int MS_, int MI_, int ME_, int DV_ = 0, int PT_ = 0, int WS_ = 0, int DLW_ = 0, int LS_ = 0, int RS_ = 0>
struct MtgCfg {
  // PT_ != 0: the kernel serves perturbed-time virtual batches (mtg_mellinger_cost_gradient); only the cost-only
  // instantiations carry that code
  static constexpr bool kPert = PT_ != 0;
  static constexpr int H = H_, N = 2 * H_, D = D_, KT = KT_;
  // KT_ > 0: "static"  -- compile-time K, masks, derivative; chain loops unrolled, back-substitution data in registers
  // KT_ < 0: "rolled"  -- compile-time masks and derivative, run-time K >= 2 (uniform interior mask MI); the chain
  //                       is a real loop (small code), back-substitution data in the coalesced global workspace
  // KT_ == 0: "generic" -- everything at run time (per-vertex mask table)
  static constexpr bool kStatic = KT_ > 0;
  static constexpr bool kRolled = KT_ < 0;
  static constexpr bool kCT = KT_ != 0;   // compile-time masks / tables
  static constexpr int DV = DV_;   // static mode: derivative_to_optimize (table offsets fold to immediates)
  static constexpr int H1OFF = mtg_h1_offset(2 * H_, DV_), AINVOFF = mtg_ainv_offset(2 * H_);
  static constexpr int MS = MS_, MI = MI_, ME = ME_;
  static constexpr int KA = (KT_ + 1) / 2, KB = KT_ / 2;
  static constexpr int popc(int x) { int c = 0; for (; x; x &= x - 1) ++c; return c; }
  static constexpr int KCS = kStatic ? ((KT_ + 1) / 2) : 1;
  // WS_ > 0 (static mode, long chains): the back-substitution data of the first WS_ steps of each half-chain (the ones
  // consumed LAST) go through the lane-coalesced global workspace, the remaining steps' stay in registers -- a K = 32
  // half-chain (16 steps x (f*f + D*f) doubles) does not fit the 512-register budget, its second half does.
  static constexpr int WSJ = kStatic ? WS_ : 0;
  static constexpr int KREG = (KCS - WSJ) > 0 ? KCS - WSJ : 1;
  static constexpr int FMAXW = H_ - (popc(MS_) < popc(MI_) ? (popc(MS_) < popc(ME_) ? popc(MS_) : popc(ME_))
                                                           : (popc(MI_) < popc(ME_) ? popc(MI_) : popc(ME_)));
  // DLW_ > 0 (dimension-in-lane form, D_ == 1): the DLW_ dimension lanes of a trajectory compute identical G; each stores
  // every DLW_-th element (one row of the workspace holds DLW_ consecutive elements, one per dimension lane) and reads
  // the others from its sibling lanes' columns: 1/DLW_ of the G traffic.  g stays per lane.
  static constexpr int DLW = (kStatic || kRolled) ? DLW_ : 0;   // (rolled: the run-time-K dimension-in-lane body, mtg_dimlane_rt.h)
  // LS_ > 0 (with DLW_): the LAST LS_ of the WS_ workspace steps (the ones next to the register steps) are kept in the
  // wave's LDS instead of global memory -- same row layout, row stride 64 lanes (MtgParams::lds_steps).
  static constexpr int LSJ = (kStatic && DLW_ > 0) ? LS_ : 0;
  // Factor store (round 4; shared-G configurations whose steps leave the registers): what a step keeps for the back-substitution
  // is the LDL^T factor of its pivot block (strict lower triangle of L + 1 / d: f (f + 1) / 2 numbers) instead of
  // G = Dtilde^-1 U (f * f numbers).  The back-substitution then forms x_l = g - Dtilde^-1 (U x_r) with U read from the
  // constant table (mtg_bwd_backsub_fs; in the segment's scaling U IS the table): ~2 f^2 more FP64 operations per step and
  // dimension lane, on a kernel whose time is the workspace round trip -- 14 -> 10 rows per step for N = 12, 10 -> 8 for N = 10,
  // 6 -> 5 for N = 8.
  static constexpr bool kFS = DLW_ > 0 && ((kStatic && WS_ > 0) || kRolled);
  // (factor-store entries are standard shapes: with fully fixed trajectory ends R_PP is positive definite for every T > 0)
  static_assert(!kFS || (MS_ == (1 << H_) - 1 && ME_ == (1 << H_) - 1), "factor-store kernels: trajectory ends fully fixed");
  static constexpr int FCNT = kFS ? FMAXW * (FMAXW + 1) / 2 : FMAXW * FMAXW;   // kept numbers per step besides g
  static constexpr int WSE = DLW > 0 ? (FCNT + DLW - 1) / DLW + FMAXW
                                     : FMAXW * FMAXW + D_ * FMAXW;   // workspace rows per step (free x free of G, free of g)
  // RS_ != 0 (with DLW_): the REGISTER steps keep G shared as well -- lane of dimension k holds elements k, DLW + k, ... of
  // G (GROWS doubles instead of up to H * H) and fetches its siblings' elements with ds_bpermute at back-substitution time
  // (the three lanes of a trajectory compute identical G): half the registers per step, i.e. twice the steps on chip.
  static constexpr bool kRegShared = (kStatic || kRolled) && DLW_ > 0 && RS_ != 0;
  static constexpr int GROWS = DLW > 0 ? (FCNT + DLW - 1) / DLW : 1;
  static constexpr int FULL = (1 << H_) - 1;
  // static mode: fixed-slot column prefix and the column range each direction touches
  static constexpr int offF(int v) { return v == 0 ? 0 : popc(MS_) + (v - 1) * popc(MI_); }
  static constexpr int offFEnd = kStatic ? offF(KT_) + popc(ME_) : 0;
  static constexpr int colBeginA = 0, colEndA = kStatic ? offF(KA + 1 > KT_ ? KT_ : KA + 1) + (KA + 1 > KT_ ? popc(ME_) : 0) : 0;
  static constexpr int colBeginB = kStatic ? offF(KA) : 0, colEndB = offFEnd;
  static constexpr int NCA = colEndA - colBeginA, NCB = colEndB - colBeginB;
  static constexpr int NC = kStatic ? (NCA > NCB ? NCA : NCB) : 1;
};

// Opaque re-definition of a (wave-uniform) table pointer.  Without it LICM hoists every scalar
// table load out of the tile loop, the ~250 live constants overflow the SGPR file and come back as
// v_readlane traffic (measured: 2231 v_readlane in the K=8 kernel).  With it the s_loads stay
// inside the phase that uses them.
// (The offset, not the pointer, is laundered: a pointer that went through inline asm loses its
// constant address space and the loads degrade to per-lane flat_load.)
MTG_HD int mtg_launder(int off) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+s"(off));
#endif
  return off;
}

// 16-byte store of two doubles (ds_write_b128 / global_store_dwordx4 on the device)
MTG_HD void mtg_store2(double* p, double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef double mtg_d2 __attribute__((ext_vector_type(2)));
  mtg_d2 v; v.x = a; v.y = b;
  *reinterpret_cast<mtg_d2*>(p) = v;
#else
  p[0] = a; p[1] = b;
#endif
}

// Workspace pointers as GLOBAL-address-space pointers.  The kernels re-define their workspace column pointer opaquely per tile (an
// empty asm, so that the per-step addresses are not hoisted and spilled); a pointer that went through an asm has lost its address
// space, and every workspace access came out as flat_load / flat_store.  A FLAT access may hit LDS or memory, which return out of
// order: the compiler has to wait for it with vmcnt(0) lgkmcnt(0) -- i.e. each backward step waited for the acknowledgement of the
// coefficient stores issued AFTER its workspace loads (found in round 6: the N = 12 / K = 32 kernel had 20 such waits per tile and
// direction).  As global_load / global_store the waits are counted (vmcnt(n): only what was issued before the loads).
// Measured (profiles/r06a_flat_vs_global_ab.jsonl, r06a_other_k_ws_global.jsonl): bit-identical; the static bodies inside the noise,
// the run-time-K body (most steps through the workspace) 1-6 % faster at K = 100.
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(1))) double mtg_glb_double;
__device__ __forceinline__ mtg_glb_double* mtg_glb(double* p) { return (mtg_glb_double*)p; }
__device__ __forceinline__ const mtg_glb_double* mtg_glb(const double* p) { return (const mtg_glb_double*)p; }
#else
typedef double mtg_glb_double;
MTG_HD double* mtg_glb(double* p) { return p; }
MTG_HD const double* mtg_glb(const double* p) { return p; }
#endif

MTG_HD int mtg_popc(int x) {
#if defined(__HIPCC__)
  return __builtin_popcount(x);
#else
  return __builtin_popcount((unsigned)x);
#endif
}

MTG_HD double mtg_fma(double a, double b, double c) {
#if defined(__HIPCC__)
  return __builtin_fma(a, b, c);
#else
  return std::fma(a, b, c);
#endif
}

// a * b.  (Measured and rejected: multiplying through v_fma_f64 with a zero addend -- a lone wave issues independent
// v_mul_f64 every 5.5 cycles against 4.4 for v_fma_f64 in the microbenchmark, but in the kernels the three-operand
// encoding costs more than it gains: B = 10k 7.46 -> 7.74 us.)
MTG_HD double mtg_mul(double a, double b) { return a * b; }

// The value, opaque to the optimiser from here on (no instruction): a product that went through it cannot be contracted into
// a later addition.
MTG_HD double mtg_pin(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("" : "+v"(x));
#endif
  return x;
}

// 1/x for the LDL^T pivots and segment times.  v_rcp_f64 seed + Newton steps (no IEEE
// division sequence); accuracy is checked on the device by mtg_selftest_rcp().
MTG_HD double mtg_rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  // v_rcp_f64 is good to ~4.6e-8 (24 bits; mtg_selftest_rcp with 0 steps).  One CUBIC correction
  //   e = 1 - x r,  r <- r + r (e + e^2)        (error e^3 ~ 1e-22 before the final rounding)
  // reaches the same result as two Newton steps (error e^4) with three dependent operations instead of four: the pivots
  // sit on the kernels' longest dependency chain (8 cycles per dependent FP64 operation for a lone wave).
  const double r0 = __builtin_amdgcn_rcp(x);
  const double e = mtg_fma(-x, r0, 1.0);
  return mtg_fma(r0, mtg_fma(e, e, e), r0);
#else
  return 1.0 / x;
#endif
}

// Table bases.  Static mode: compile-time offset => every entry folds to an immediate (no SGPR
// pressure, trivially rematerialisable).  Generic mode: runtime offset, laundered (see above).
// (Round 6, looked at and not built further: the immediates cost two s_mov_b32 per constant whose low word is not zero -- 2658 in
// the N = 12 / K = 32 body, 10 % of its instructions.  Reading the static bodies' constants with s_load instead (the laundered
// offset) removes them and puts 1465 s_add_u32 / s_addc_u32 pairs, 1685 s_load, 730 more s_waitcnt and 350 more spilled SGPRs
// (2170 v_readlane) in their place.)
template <class C> MTG_HD const double* mtg_h1(const MtgParams& P) {
  if constexpr (C::kCT) return kH1 + C::H1OFF; else return kH1 + mtg_launder(P.h1off);
}
template <class C> MTG_HD const double* mtg_q1(const MtgParams& P) {
  if constexpr (C::kCT) return kQ1 + C::H1OFF; else return kQ1 + mtg_launder(P.h1off);
}
template <class C> MTG_HD const double* mtg_ainv(const MtgParams& P) {
  if constexpr (C::kCT) return kAinvLo + C::AINVOFF; else return kAinvLo + mtg_launder(P.ainvoff);
}
template <class C> MTG_HD int mtg_deriv(const MtgParams& P) {
  if constexpr (C::kCT) return C::DV; else return P.deriv;
}

template <class C> MTG_HD int mtg_nseg(const MtgParams& P) { if constexpr (C::kStatic) return C::KT; else return P.K; }

template <class C> MTG_HD int mtg_mask(const MtgParams& P, int v) {
  if constexpr (C::kCT) return v == 0 ? C::MS : (v == mtg_nseg<C>(P) ? C::ME : C::MI);
  else return P.vmask[v];
}
template <class C> MTG_HD int mtg_offF(const MtgParams& P, int v) {
  if constexpr (C::kCT) return v == 0 ? 0 : mtg_popc(C::MS) + (v - 1) * mtg_popc(C::MI);
  else return P.offF[v];
}
template <class C> MTG_HD int mtg_offP(const MtgParams& P, int v) {
  if constexpr (C::kCT) return v == 0 ? 0 : (C::H - mtg_popc(C::MS)) + (v - 1) * (C::H - mtg_popc(C::MI));
  else return P.offP[v];
}

// chain step j of direction DIR: actual segment index, left (already reached) and right vertex
template <int DIR> MTG_HD int mtg_seg(int K, int j) { return DIR > 0 ? j : K - 1 - j; }
template <int DIR> MTG_HD int mtg_vl(int K, int j) { return DIR > 0 ? j : K - j; }
template <int DIR> MTG_HD int mtg_vr(int K, int j) { return DIR > 0 ? j + 1 : K - 1 - j; }

// Scaled-variable chain (round 6).  The forward sweep carries the Schur complement and its right-hand side in the SCALING OF THE
// CURRENT SEGMENT: with tau = DIR T, s_p = tau^p, t = T^(1-2d) it holds  Sc^ = Sc / (t s_p s_q),  rc^ = rc / (t s_p)  and solves for
// x^ = s_p x_p.  In these variables the segment's blocks are the CONSTANT tables: pivot block D^ = Sc^ + H1_ss, coupling
// U^ = H1_se, the right vertex's diagonal H1_ee -- no products with powers of T for U (2 f_l f_r multiplications per step) nor
// for the diagonal block (f_r (f_r + 1)); what is left of the assembly is the conversion of the carried block from the previous
// segment's scaling, one FMA per entry: D^_pq = kappa sigma^(p+q) Sc^'_pq + H1_ss,pq  with sigma = tau' / tau and
// kappa = (T' / T)^(1-2d) (mtg_step_scales: for d = h - 1 all 2h - 1 factors are powers of T / T' and one T' / T).  The
// back-substitution works on x^ as well and hands the scaled vertex values s_p x_p, which the coefficient recovery needs anyway
// (impl/...:276-277 through the scaling identity), to mtg_recover.  Unscaled quantities cross the function boundaries: the two
// directions exchange Sc, rc (mtg_unscale_carried at the end of the forward phase), xl / xr / d_P are plain derivatives.
// (Rounds 1-5 formed every block as T^(1-2d) S H1 S: the last commit with that assembly, selectable as MTG_SCALED_CHAIN=0, is 13d7019.)

template <class C>
struct MtgLane {
  double G[C::kRegShared ? 1 : C::KREG][C::H][C::H];  // G_v = Dtilde_v^-1 U_v          (static mode: registers; steps >= C::WSJ)
  double Gs[C::kRegShared ? C::KREG : 1][C::kRegShared ? C::GROWS : 1];   // MtgCfg::kRegShared: this lane's share of G_v
  double g[C::KREG][C::D][C::H];  // g_v = Dtilde_v^-1 rtilde_v
  double Sc[C::H][C::H];          // Schur complement carried onto the next vertex (lower tri)
  double rc[C::D][C::H];          // its right-hand side
  double T[C::KCS];               // static mode: this lane's segment times, chain order
  double fx[C::D][C::NC];         // static mode: this lane's fixed values (columns colBegin..colEnd)
  double cT, cTinv;               // signed time DIR T (and its reciprocal) of the segment whose scaling Sc / rc are in
  int flags;
};

// start of a half-chain: nothing carried, unit scaling
template <class C, int DIR>
MTG_HD void mtg_lane_reset(MtgLane<C>& ln) {
  ln.flags = 0;
#pragma unroll
  for (int p = 0; p < C::H; ++p) {
#pragma unroll
    for (int q = 0; q < C::H; ++q) ln.Sc[p][q] = 0.0;
  }
#pragma unroll
  for (int dm = 0; dm < C::D; ++dm) {
#pragma unroll
    for (int p = 0; p < C::H; ++p) ln.rc[dm][p] = 0.0;
  }
  ln.cT = DIR > 0 ? 1.0 : -1.0;
  ln.cTinv = ln.cT;
}

// Static mode: issue every input load of this lane's half-chain up front with incrementally
// advanced per-lane pointers (one v_lshl_add_u64 per load, no hoistable 64-bit stride products).
template <class C, int DIR>
MTG_HD void mtg_preload_into(const MtgParams& P, long long b, double (&T)[C::KCS], double (&fx)[C::D][C::NC]);

template <class C, int DIR>
MTG_HD void mtg_preload(const MtgParams& P, long long b, MtgLane<C>& ln) {
  mtg_preload_into<C, DIR>(P, b, ln.T, ln.fx);
}

template <class C, int DIR>
MTG_HD void mtg_preload_into(const MtgParams& P, long long b, double (&T)[C::KCS], double (&fx)[C::D][C::NC]) {
  if constexpr (C::kStatic) {
    constexpr int KC = DIR > 0 ? C::KA : C::KB;
    constexpr int c0 = DIR > 0 ? C::colBeginA : C::colBeginB;
    constexpr int nc = DIR > 0 ? C::NCA : C::NCB;
    const double* pt = P.times + b * P.ts_b + (long long)(DIR > 0 ? 0 : C::KT - 1) * P.ts_k;
    const long long tstep = DIR > 0 ? P.ts_k : -P.ts_k;
    // (both directions write every element, zeros beyond their own count: see mtg_dl_preload)
#pragma unroll
    for (int j = 0; j < C::KCS; ++j) {
      if (j < KC) {
        T[j] = *pt;
        if constexpr (C::kPert) T[j] = mtg_perturb(P, mtg_seg<DIR>(C::KT, j), T[j]);
        pt += tstep;
      } else {
        T[j] = 0.0;
      }
    }
    const double* pd = P.dfix + b * P.fs_b + (long long)P.dim0 * P.fs_d + (long long)c0 * P.fs_c;
#pragma unroll
    for (int dm = 0; dm < C::D; ++dm) {
      const double* pc = pd;
#pragma unroll
      for (int c = 0; c < C::NC; ++c) {
        if (c < nc) {
          fx[dm][c] = *pc;
          pc += P.fs_c;
        } else {
          fx[dm][c] = 0.0;
        }
      }
      pd += P.fs_d;
    }
  }
}

// fixed values of vertex v (zeros at free slots)
template <class C, int DIR>
MTG_HD void mtg_load_vals(const MtgParams& P, long long b, int v, int mask, const MtgLane<C>& ln,
                          double (&out)[C::D][C::H]) {
  const int off = mtg_offF<C>(P, v);
#pragma unroll
  for (int dm = 0; dm < C::D; ++dm) {
#pragma unroll
    for (int p = 0; p < C::H; ++p) {
      if ((mask >> p) & 1) {
        const int col = off + mtg_popc(mask & ((1 << p) - 1));
        if constexpr (C::kStatic) {
          out[dm][p] = ln.fx[dm][col - (DIR > 0 ? C::colBeginA : C::colBeginB)];
        } else {
          out[dm][p] = P.dfix[b * P.fs_b + (long long)(P.dim0 + dm) * P.fs_d + (long long)col * P.fs_c];
        }
      } else {
        out[dm][p] = 0.0;
      }
    }
  }
}

// in-place LDL^T on the index set {p : bit p of `fixed` clear}; A lower triangle in, L (strict
// lower) out, dinv = 1/d.
// The pivot test `!(d > 0)` is a BREAKDOWN guard (the matrix is numerically not positive definite: NaN inputs, problems beyond
// float64 such as N = 12 chains with segment-time ratios of 400), not the rank decision.  Whether the free system is
// rank-deficient -- where the reference's rank-revealing SparseQR returns a basic solution, LIN:365-378 -- is a property of
// the constraint PATTERN and is decided once per plan on the host (mtg_abi.hip: structural_null_dim; every trajectory of such
// a plan is flagged, and MTG_FLAG_BASIC_SOLUTION solves it through the plan's pinned shadow).  Round 5 built and measured the
// alternative the round-4 review asked for, a relative threshold d_j <= 20 (n_free + n_free) eps R_PP[j][j] (SparseQR's default
// form, the diagonal carried through the chain: +2 % on small launches), and took it out again: on chains of free vertices
// the structurally zero pivot comes out at 1e-12 ... 1e0 of its diagonal, of either sign (false negatives: profiles/
// r05_pivot_ratio_study.txt), and a REGULAR 50-segment chain of free vertices has legitimate pivots 1e-12 of their diagonal
// while its solution is good to 6e-13 (false positive: the case of tests/test_pivot_threshold.py::
// test_rank_deficient_long_chains_through_every_launch_route[...-10-50], trajectory 2).  No pivot threshold separates the two.
template <int H>
MTG_HD void mtg_ldl(double (&A)[H][H], double (&dinv)[H], int fixed, int& flags) {
#pragma unroll
  for (int j = 0; j < H; ++j) {
    if ((fixed >> j) & 1) continue;
    const double dj = A[j][j];
    if (!(dj > 0.0)) flags |= MTG_FLAG_SINGULAR;
    const double r = mtg_rcp(dj);
    dinv[j] = r;
    double l[H];
#pragma unroll
    for (int i = j + 1; i < H; ++i) {
      if ((fixed >> i) & 1) continue;
      l[i] = mtg_mul(A[i][j], r);
    }
#pragma unroll
    for (int i = j + 1; i < H; ++i) {
      if ((fixed >> i) & 1) continue;
#pragma unroll
      for (int k = j + 1; k <= i; ++k) {
        if ((fixed >> k) & 1) continue;
        A[i][k] = mtg_fma(-l[i], A[k][j], A[i][k]);
      }
    }
#pragma unroll
    for (int i = j + 1; i < H; ++i) {
      if ((fixed >> i) & 1) continue;
      A[i][j] = l[i];
    }
  }
}

// X <- (L D L^T)^-1 X for NR right-hand sides at once, X[i][c] (row i, column c).  The column loop
// is innermost so that consecutive instructions are independent (ILP = number of columns in use).
template <int H, int NR>
MTG_HD void mtg_ldl_solve_multi(const double (&L)[H][H], const double (&dinv)[H], int fixed, double (&x)[H][NR],
                                int colskip) {
#pragma unroll
  for (int i = 0; i < H; ++i) {
    if ((fixed >> i) & 1) continue;
#pragma unroll
    for (int k = 0; k < i; ++k) {
      if ((fixed >> k) & 1) continue;
#pragma unroll
      for (int c = 0; c < NR; ++c) {
        if ((colskip >> c) & 1) continue;
        x[i][c] = mtg_fma(-L[i][k], x[k][c], x[i][c]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < H; ++i) {
    if ((fixed >> i) & 1) continue;
#pragma unroll
    for (int c = 0; c < NR; ++c) {
      if ((colskip >> c) & 1) continue;
      x[i][c] = mtg_mul(x[i][c], dinv[i]);
    }
  }
#pragma unroll
  for (int i = H - 1; i >= 0; --i) {
    if ((fixed >> i) & 1) continue;
#pragma unroll
    for (int k = i + 1; k < H; ++k) {
      if ((fixed >> k) & 1) continue;
#pragma unroll
      for (int c = 0; c < NR; ++c) {
        if ((colskip >> c) & 1) continue;
        x[i][c] = mtg_fma(-L[k][i], x[k][c], x[i][c]);
      }
    }
  }
}

// x^e for a small compile-time-bounded exponent by squaring (dependency depth ~log2 e; the
// wave runs alone on its SIMD, so chain depth -- 8 cycles per dependent FP64 op -- is what costs).
template <int EMAX>
MTG_HD double mtg_powi(double x, int e) {
  double r = 1.0, p = x;
#pragma unroll
  for (int bit = 0; (1 << bit) <= EMAX; ++bit) {
    if ((e >> bit) & 1) r = mtg_mul(r, p);
    p = mtg_mul(p, p);
  }
  return r;
}

// per-segment scale vectors: s[p] = (DIR*T)^p, bs[p] = T^(1-2d) * s[p]
template <int H, int DIR>
MTG_HD void mtg_scales(double T, int deriv, double (&s)[H], double (&bs)[H], double& tinv, int& flags) {
  if (!(T > 0.0)) flags |= MTG_FLAG_BAD_TIME;
  tinv = mtg_rcp(T);
  const double base = deriv == 0 ? T : mtg_powi<2 * H - 1>(tinv, 2 * deriv - 1);
  const double ts = DIR > 0 ? T : -T;
  s[0] = 1.0;
  if constexpr (H > 1) s[1] = ts;
#pragma unroll
  for (int p = 2; p < H; ++p) s[p] = mtg_mul(s[p / 2], s[p - p / 2]);   // depth log2(p)
#pragma unroll
  for (int p = 0; p < H; ++p) bs[p] = mtg_mul(base, s[p]);
}

// pw[m] = rho^(2 DV - 1 - m), m = 0 .. 2H - 2, from the power tables of rho and sigma = 1 / rho
template <int H, int DV>
MTG_HD void mtg_pw_table(double rho, double sigma, double (&pw)[2 * H - 1]) {
  constexpr int E = 2 * DV - 1;
  constexpr int NR = E > 0 ? E + 1 : 1;                           // rho^0 .. rho^E
  constexpr int NS = 2 * H - 2 - E > 0 ? 2 * H - 2 - E + 1 : 1;   // sigma^0 .. sigma^(2H - 2 - E)
  double rp[NR], sg[NS];
  rp[0] = 1.0; sg[0] = 1.0;
  if constexpr (NR > 1) rp[1] = rho;
  if constexpr (NS > 1) sg[1] = sigma;
#pragma unroll
  for (int e = 2; e < NR; ++e) rp[e] = mtg_mul(rp[e / 2], rp[e - e / 2]);
#pragma unroll
  for (int e = 2; e < NS; ++e) sg[e] = mtg_mul(sg[e / 2], sg[e - e / 2]);
#pragma unroll
  for (int m = 0; m < 2 * H - 1; ++m) pw[m] = E - m >= 0 ? rp[E - m >= 0 ? E - m : 0] : sg[m - E > 0 ? m - E : 0];
}
// run-time derivative (generic kernels): the same table, selected by a wave-uniform branch -- the generic and the specialised
// kernels of a shape then form identical factors (a one-ulp difference in them is round-off x cond(R_PP) in the solution)
template <int H, int DV>
MTG_HD void mtg_pw_dispatch(int deriv, double rho, double sigma, double (&pw)[2 * H - 1]) {
  if (deriv == DV) mtg_pw_table<H, DV>(rho, sigma, pw);
  else if constexpr (DV + 1 < H) mtg_pw_dispatch<H, DV + 1>(deriv, rho, sigma, pw);
  else mtg_pw_table<H, DV>(rho, sigma, pw);     // (not reached: derivative_to_optimize < h)
}

// Scaled-variable chain: scale vector of the step, s[p] = (DIR T)^p, and the factors that bring the carried block from the previous
// step's scaling (ln.cT) into this one:  pw[m] = kappa sigma^m,  sigma = tau_prev / tau,  kappa = (T_prev / T)^(1 - 2 d)
// (entry (p, q) of the Schur complement takes pw[p + q], entry p of its right-hand side pw[p]).  With rho = 1 / sigma the factor
// is rho^(2d - 1 - m): compile-time derivative => the non-negative exponents come from one power table of rho, the negative ones
// from one of sigma (d = h - 1: 2h - 2 powers of rho and sigma itself).
template <class C, int DIR>
MTG_HD void mtg_step_scales(const MtgParams& P, double T, const MtgLane<C>& ln, double (&s)[C::H], double (&pw)[2 * C::H - 1],
                            double& tinv, int& flags) {
  constexpr int H = C::H;
  if (!(T > 0.0)) flags |= MTG_FLAG_BAD_TIME;
  tinv = mtg_rcp(T);
  const double ts = DIR > 0 ? T : -T;
  s[0] = 1.0;
  if constexpr (H > 1) s[1] = ts;
#pragma unroll
  for (int p = 2; p < H; ++p) s[p] = mtg_mul(s[p / 2], s[p - p / 2]);   // depth log2(p)
  const double sigma = mtg_mul(ln.cT, DIR > 0 ? tinv : -tinv);          // tau_prev / tau  (> 0: both carry DIR's sign)
  const double rho = mtg_mul(ts, ln.cTinv);                             // tau / tau_prev
  if constexpr (C::kCT) mtg_pw_table<H, C::DV>(rho, sigma, pw);
  else mtg_pw_dispatch<H, 0>(P.deriv, rho, sigma, pw);
}

// End of the forward phase (scaled-variable chain): the carried Schur complement and right-hand side leave the last segment's
// scaling -- the two directions add theirs at the middle vertex (mtg_pack_mid / mtg_solve_mid work on plain quantities).
template <class C, int DIR>
MTG_HD void mtg_unscale_carried(const MtgParams& P, MtgLane<C>& ln) {
  constexpr int H = C::H;
  const int deriv = mtg_deriv<C>(P);
  const double T = ln.cT < 0.0 ? -ln.cT : ln.cT, tinv = ln.cTinv < 0.0 ? -ln.cTinv : ln.cTinv;
  const double base = deriv == 0 ? T : mtg_powi<2 * H - 1>(tinv, 2 * deriv - 1);
  double s[H], bs[H];
  s[0] = 1.0;
  if constexpr (H > 1) s[1] = ln.cT;
#pragma unroll
  for (int p = 2; p < H; ++p) s[p] = mtg_mul(s[p / 2], s[p - p / 2]);
#pragma unroll
  for (int p = 0; p < H; ++p) bs[p] = mtg_mul(base, s[p]);
#pragma unroll
  for (int p = 0; p < H; ++p) {
#pragma unroll
    for (int q = 0; q <= p; ++q) ln.Sc[p][q] = mtg_mul(mtg_mul(bs[p], s[q]), ln.Sc[p][q]);
#pragma unroll
    for (int dm = 0; dm < C::D; ++dm) ln.rc[dm][p] = mtg_mul(bs[p], ln.rc[dm][p]);
  }
  ln.cT = 1.0;       // plain quantities from here on
  ln.cTinv = 1.0;
}

// One forward elimination step (chain step j): completes the left vertex, produces
// (G, g) for back-substitution and the carried Schur complement for the right vertex.
// The arithmetic of one step on inputs that are already in registers: segment time T and the (unscaled) fixed values of
// the left / right vertex, zeros at free slots.  The rolled chain loops request the inputs of step j+1 before step j's
// back-substitution data is stored (see mtg_lane_forward).
template <class C, int DIR>
MTG_HD void mtg_fwd_step_core(const MtgParams& P, int ml, int mr, MtgLane<C>& ln, double T,
                              const double (&fix_l)[C::D][C::H], const double (&fix_r)[C::D][C::H],
                              double (&G)[C::H][C::H], double (&g)[C::D][C::H]) {
  constexpr int H = C::H, D = C::D, N = C::N;
  double s[H], pw[2 * H - 1], tinv;
  mtg_step_scales<C, DIR>(P, T, ln, s, pw, tinv, ln.flags);

  // scaled fixed values
  double val_l[D][H], val_r[D][H];
#pragma unroll
  for (int dm = 0; dm < D; ++dm) {
#pragma unroll
    for (int p = 0; p < H; ++p) {
      val_l[dm][p] = mtg_mul(fix_l[dm][p], s[p]);
      val_r[dm][p] = mtg_mul(fix_r[dm][p], s[p]);
    }
  }

  // rhs of the left vertex and of the right vertex (fixed-value couplings).  Loop order: source column q
  // outermost, the independent accumulators (row p, dimension) innermost, so consecutive FMAs never depend
  // on each other (a wave alone on its SIMD pays 8 cycles per dependent FP64 op, 4 per independent one).
  double rv[D][H], rnext[D][H];
  {
    const double* hc = mtg_h1<C>(P);
    double accl[H][D], accr[H][D];
#pragma unroll
    for (int p = 0; p < H; ++p) {
#pragma unroll
      for (int dm = 0; dm < D; ++dm) { accl[p][dm] = 0.0; accr[p][dm] = 0.0; }
    }
#pragma unroll
    for (int q = 0; q < H; ++q) {
#pragma unroll
      for (int p = 0; p < H; ++p) {
        if (!((ml >> p) & 1)) {
          if ((ml >> q) & 1) {
            const double c = hc[p * N + q];
#pragma unroll
            for (int dm = 0; dm < D; ++dm) accl[p][dm] = mtg_fma(c, val_l[dm][q], accl[p][dm]);
          }
          if ((mr >> q) & 1) {
            const double c = hc[p * N + H + q];
#pragma unroll
            for (int dm = 0; dm < D; ++dm) accl[p][dm] = mtg_fma(c, val_r[dm][q], accl[p][dm]);
          }
        }
        if (!((mr >> p) & 1)) {
          if ((mr >> q) & 1) {
            const double c = hc[(H + p) * N + H + q];
#pragma unroll
            for (int dm = 0; dm < D; ++dm) accr[p][dm] = mtg_fma(c, val_r[dm][q], accr[p][dm]);
          }
          if ((ml >> q) & 1) {
            const double c = hc[q * N + H + p];
#pragma unroll
            for (int dm = 0; dm < D; ++dm) accr[p][dm] = mtg_fma(c, val_l[dm][q], accr[p][dm]);
          }
        }
      }
    }
#pragma unroll
    for (int p = 0; p < H; ++p) {
#pragma unroll
      for (int dm = 0; dm < D; ++dm) {
        rv[dm][p] = ((ml >> p) & 1) ? 0.0 : mtg_fma(pw[p], ln.rc[dm][p], -accl[p][dm]);
        rnext[dm][p] = ((mr >> p) & 1) ? 0.0 : -accr[p][dm];
      }
    }
  }

  // Dtilde_l = Sc + a_ll (free x free, lower), U = a_lr (free_l x free_r)
  double A[H][H], U[H][H], dinv[H];
  {
    const double* hc = mtg_h1<C>(P);
#pragma unroll
    for (int p = 0; p < H; ++p) {
      dinv[p] = 0.0;
#pragma unroll
      for (int q = 0; q < H; ++q) {
        A[p][q] = 0.0;
        U[p][q] = 0.0;
        if (!((ml >> p) & 1)) {
          if (q <= p && !((ml >> q) & 1)) A[p][q] = mtg_fma(pw[p + q], ln.Sc[p][q], hc[p * N + q]);
          if (!((mr >> q) & 1)) U[p][q] = hc[p * N + H + q];
        }
      }
    }
  }
  mtg_ldl<H>(A, dinv, ml, ln.flags);

  const double* hrr = mtg_h1<C>(P);
  if constexpr (C::kFS || MTG_PARTIAL_ALL != 0) {
    // Factor store, partial elimination: nothing in the FORWARD sweep needs G = Dtilde^-1 U itself.  With W = L^-1 U and
    // z = L^-1 rv (forward substitution only), U^T G = W^T D^-1 W and U^T g = W^T D^-1 z: the carried Schur complement is what
    // f pivots of the 2f x 2f block [Dtilde U; U^T a_rr] leave behind.  The back-substitution half of the solve is needed for
    // g alone (D columns instead of f + D), and no later forward step depends on it: it leaves the dependent chain
    // pivot -> column -> update -> next pivot, which is what a lone wave per SIMD waits on.
    double X[H][H + D], Y[H][H + D];
#pragma unroll
    for (int p = 0; p < H; ++p) {
#pragma unroll
      for (int q = 0; q < H; ++q) X[p][q] = U[p][q];
#pragma unroll
      for (int dm = 0; dm < D; ++dm) X[p][H + dm] = rv[dm][p];
    }
    // (scaled-variable chain: U is the constant table, so the first free row of X and the initial value of every other row are
    // literals; the rows are eliminated from the nearest one down, so that only the LAST product of a row meets two literals'
    // worth of operands -- an FMA takes one)
#pragma unroll
    for (int i = 0; i < H; ++i) {
      if ((ml >> i) & 1) continue;
#pragma unroll
      for (int kk = 0; kk < i; ++kk) {
        const int k = i - 1 - kk;
        if ((ml >> k) & 1) continue;
#pragma unroll
        for (int c = 0; c < H + D; ++c) {
          if (c < H && ((mr >> c) & 1)) continue;
          X[i][c] = mtg_fma(-A[i][k], X[k][c], X[i][c]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < H; ++i) {
#pragma unroll
      for (int c = 0; c < H + D; ++c) Y[i][c] = ((ml >> i) & 1) ? 0.0 : mtg_mul(X[i][c], dinv[i]);
    }
#pragma unroll
    for (int p = 0; p < H; ++p) {
#pragma unroll
      for (int q = 0; q < H; ++q) {
        ln.Sc[p][q] = 0.0;
        if (q <= p && !((mr >> p) & 1) && !((mr >> q) & 1)) ln.Sc[p][q] = hrr[(H + p) * N + H + q];
      }
    }
#pragma unroll
    for (int dm = 0; dm < D; ++dm) {
#pragma unroll
      for (int p = 0; p < H; ++p) ln.rc[dm][p] = rnext[dm][p];
    }
#pragma unroll
    for (int mm = 0; mm < H; ++mm) {
      const int m = H - 1 - mm;     // (the literal row of X -- the first free one -- last)
      if ((ml >> m) & 1) continue;
#pragma unroll
      for (int p = 0; p < H; ++p) {
        if ((mr >> p) & 1) continue;
#pragma unroll
        for (int q = 0; q <= p; ++q) {
          if ((mr >> q) & 1) continue;
          ln.Sc[p][q] = mtg_fma(-X[m][p], Y[m][q], ln.Sc[p][q]);
        }
#pragma unroll
        for (int dm = 0; dm < D; ++dm) ln.rc[dm][p] = mtg_fma(-X[m][p], Y[m][H + dm], ln.rc[dm][p]);
      }
    }
    // g = L^-T D^-1 z (kept for the back-substitution; off the forward chain) -- and, where the step keeps G itself rather than the
    // factor (MTG_PARTIAL_ALL: every other kernel), G = L^-T D^-1 W as well: the same operations as before, after the Schur update
    // instead of in front of it
    constexpr int C0 = C::kFS ? H : 0;       // first column that is back-substituted
#pragma unroll
    for (int i = H - 1; i >= 0; --i) {
      if ((ml >> i) & 1) continue;
#pragma unroll
      for (int k = i + 1; k < H; ++k) {
        if ((ml >> k) & 1) continue;
#pragma unroll
        for (int c = C0; c < H + D; ++c) {
          if (c < H && ((mr >> c) & 1)) continue;
          Y[i][c] = mtg_fma(-A[k][i], Y[k][c], Y[i][c]);
        }
      }
    }
#pragma unroll
    for (int p = 0; p < H; ++p) {
      if constexpr (!C::kFS) {
#pragma unroll
        for (int q = 0; q < H; ++q) G[p][q] = (((ml >> p) & 1) || ((mr >> q) & 1)) ? 0.0 : Y[p][q];
      }
#pragma unroll
      for (int dm = 0; dm < D; ++dm) g[dm][p] = ((ml >> p) & 1) ? 0.0 : Y[p][H + dm];
    }
  } else {
  // [G | g] = Dtilde^-1 [U | rv]: H + D right-hand sides solved together
  {
    double X[H][H + D];
#pragma unroll
    for (int p = 0; p < H; ++p) {
#pragma unroll
      for (int q = 0; q < H; ++q) X[p][q] = U[p][q];
#pragma unroll
      for (int dm = 0; dm < D; ++dm) X[p][H + dm] = rv[dm][p];
    }
    mtg_ldl_solve_multi<H, H + D>(A, dinv, ml, X, mr);   // columns of fixed right-vertex slots are zero: skip
#pragma unroll
    for (int p = 0; p < H; ++p) {
#pragma unroll
      for (int q = 0; q < H; ++q) G[p][q] = (((ml >> p) & 1) || ((mr >> q) & 1)) ? 0.0 : X[p][q];
#pragma unroll
      for (int dm = 0; dm < D; ++dm) g[dm][p] = ((ml >> p) & 1) ? 0.0 : X[p][H + dm];
    }
  }

  // carried onto the right vertex: Sc' = a_rr - U^T G,  rc' = rnext - U^T g   (m outermost => independent FMAs adjacent)
#pragma unroll
  for (int p = 0; p < H; ++p) {
#pragma unroll
    for (int q = 0; q < H; ++q) {
      ln.Sc[p][q] = 0.0;
      if (q <= p && !((mr >> p) & 1) && !((mr >> q) & 1)) ln.Sc[p][q] = hrr[(H + p) * N + H + q];
    }
  }
#pragma unroll
  for (int dm = 0; dm < D; ++dm) {
#pragma unroll
    for (int p = 0; p < H; ++p) ln.rc[dm][p] = rnext[dm][p];
  }
#pragma unroll
  for (int m = 0; m < H; ++m) {
    if ((ml >> m) & 1) continue;
#pragma unroll
    for (int p = 0; p < H; ++p) {
      if ((mr >> p) & 1) continue;
#pragma unroll
      for (int q = 0; q <= p; ++q) {
        if ((mr >> q) & 1) continue;
        ln.Sc[p][q] = mtg_fma(-U[m][p], G[m][q], ln.Sc[p][q]);
      }
#pragma unroll
      for (int dm = 0; dm < D; ++dm) ln.rc[dm][p] = mtg_fma(-U[m][p], g[dm][m], ln.rc[dm][p]);
    }
  }
  }
  if constexpr (C::kFS) {
    // factor store: the caller keeps the pivot block's factor in place of G (strict lower triangle: L, diagonal: 1 / d)
#pragma unroll
    for (int p = 0; p < H; ++p) {
#pragma unroll
      for (int q = 0; q < H; ++q)
        G[p][q] = (((ml >> p) & 1) || ((ml >> q) & 1) || q > p) ? 0.0 : (q == p ? dinv[p] : A[p][q]);
    }
  }
  ln.cT = DIR > 0 ? T : -T;          // what is carried on is in this segment's scaling
  ln.cTinv = DIR > 0 ? tinv : -tinv;
}

// segment time of chain step j (static mode: preloaded register; otherwise a global load)
template <class C, int DIR>
MTG_HD double mtg_step_time(const MtgParams& P, long long b, int j, const MtgLane<C>& ln) {
  if constexpr (C::kStatic) return ln.T[j];
  else {
    const double T = P.times[b * P.ts_b + (long long)mtg_seg<DIR>(mtg_nseg<C>(P), j) * P.ts_k];
    if constexpr (C::kPert) return mtg_perturb(P, mtg_seg<DIR>(mtg_nseg<C>(P), j), T);
    else return T;
  }
}

// generic mode: the explicit right-hand side (MtgParams::rhs) of vertex v's free slots joins the carried one
template <class C>
MTG_HD void mtg_add_explicit_rhs(const MtgParams& P, long long b, int v, int mask, MtgLane<C>& ln) {
  if constexpr (!C::kCT) {
    if (P.rhs == nullptr) return;
    // the carried right-hand side is in the scaling of ln.cT: entry p divided by t s_p, t = |cT|^(1-2d), s_p = cT^p
    double rs[C::H];
    {
      const double Ta = ln.cT < 0.0 ? -ln.cT : ln.cT, Tai = ln.cTinv < 0.0 ? -ln.cTinv : ln.cTinv;
      rs[0] = P.deriv == 0 ? Tai : mtg_powi<2 * C::H - 1>(Ta, 2 * P.deriv - 1);
#pragma unroll
      for (int p = 1; p < C::H; ++p) rs[p] = rs[p - 1] * ln.cTinv;
    }
    const int off = mtg_offP<C>(P, v);
#pragma unroll
    for (int dm = 0; dm < C::D; ++dm) {
#pragma unroll
      for (int p = 0; p < C::H; ++p) {
        if ((mask >> p) & 1) continue;
        const int col = off + (p - mtg_popc(mask & ((1 << p) - 1)));
        ln.rc[dm][p] += P.rhs[b * P.rh_b + (long long)(P.dim0 + dm) * P.rh_d + (long long)col * P.rh_c] * rs[p];
      }
    }
  }
}

template <class C, int DIR>
MTG_HD void mtg_fwd_step(const MtgParams& P, long long b, int j, int ml, int mr, MtgLane<C>& ln,
                         double (&G)[C::H][C::H], double (&g)[C::D][C::H]) {
  const int K = mtg_nseg<C>(P);
  mtg_add_explicit_rhs<C>(P, b, mtg_vl<DIR>(K, j), ml, ln);      // (the step completes its LEFT vertex)
  const double T = mtg_step_time<C, DIR>(P, b, j, ln);
  double fix_l[C::D][C::H], fix_r[C::D][C::H];
  mtg_load_vals<C, DIR>(P, b, mtg_vl<DIR>(K, j), ml, ln, fix_l);
  mtg_load_vals<C, DIR>(P, b, mtg_vr<DIR>(K, j), mr, ln, fix_r);
  mtg_fwd_step_core<C, DIR>(P, ml, mr, ln, T, fix_l, fix_r, G, g);
}

// Output policy that stores one lane's D*N coefficients of a segment straight to global memory
// (host emulation, and the one-lane-per-trajectory update kernel).  The solve kernels use the
// LDS-staged, coalesced policy MtgLdsOut in mtg_kernels.h instead.
template <class C>
struct MtgDirectOut {
  double buf[C::D * C::N];
  long long b;
  MTG_HD double* row(int) { return buf; }
  MTG_HD void drain(const MtgParams&) {}
  MTG_HD void flush(const MtgParams&) {}
  MTG_HD void commit(const MtgParams& P, int seg) {
    const int K = mtg_nseg<C>(P);
    double* out = P.coeffs + (((long long)b * K + seg) * P.Dtot + P.dim0) * C::N;
#pragma unroll
    for (int i = 0; i < C::D * C::N; ++i) out[i] = buf[i];
  }
};

// SGN != 0 (scaled-variable chain): the back-substitution hands over yS / yE = (SGN T)^p x_p of the segment's start / end vertex,
// which is dl up to the sign of the odd derivatives in the backward direction; SGN == 0: dl is formed here.
template <class C, int OUT, int SGN = 0, class IO>
MTG_HD double mtg_recover(const MtgParams& P, long long b, int seg, double T,
                          const double (&xS)[C::D][C::H], const double (&xE)[C::D][C::H], IO& io,
                          const double (*yS)[C::H] = nullptr, const double (*yE)[C::H] = nullptr) {
  constexpr int H = C::H, D = C::D, N = C::N;
  int dummy = 0;
  double s[H], bs[H], tinv;
  mtg_scales<H, 1>(T, mtg_deriv<C>(P), s, bs, tinv, dummy);
  double tp[H];                    // T^-(H+j)
  {
    double ti[H];                  // tinv^j, depth log2(j)
    ti[0] = 1.0;
    if constexpr (H > 1) ti[1] = tinv;
#pragma unroll
    for (int p = 2; p < H; ++p) ti[p] = mtg_mul(ti[p / 2], ti[p - p / 2]);
    double th = tinv;   // tinv^H
    if constexpr (H > 1) th = mtg_mul(ti[H / 2], ti[H - H / 2]);
#pragma unroll
    for (int p = 0; p < H; ++p) tp[p] = mtg_mul(th, ti[p]);
  }
  double invfact[H];
  {
    double f = 1.0;
#pragma unroll
    for (int p = 0; p < H; ++p) {
      if (p > 0) f *= (double)p;
      invfact[p] = 1.0 / f;   // compile-time constant after unrolling
    }
  }
  double dl[D][N], qs[D][N], park[D];   // park: c_(h-1) when h is odd (pairs with c_h in the 16-byte store)
  constexpr bool kStore = (OUT & 8) == 0;   // OUT bit 3: cost-only launch, no coefficient output at all
  double dummy_row[2];
  double* row = dummy_row;
  if constexpr (kStore) {
    io.drain(P);               // stream out the previously committed segment before reusing the staging row
    row = io.row(seg);
  }
#pragma unroll
  for (int dm = 0; dm < D; ++dm) {
    double clo[H + 1];
#pragma unroll
    for (int p = 0; p < H; ++p) {
      if constexpr (SGN == 0) {
        dl[dm][p] = mtg_mul(s[p], xS[dm][p]);
        dl[dm][H + p] = mtg_mul(s[p], xE[dm][p]);
      } else {
        dl[dm][p] = (SGN < 0 && (p & 1)) ? -yS[dm][p] : yS[dm][p];
        dl[dm][H + p] = (SGN < 0 && (p & 1)) ? -yE[dm][p] : yE[dm][p];
      }
      clo[p] = mtg_mul(xS[dm][p], invfact[p]);
      qs[dm][p] = mtg_mul(dl[dm][p], invfact[p]);
    }
    // low half of the coefficients: c_p = d_p / p!  (pairs that lie entirely in the low half)
    if constexpr (kStore) {
#pragma unroll
      for (int p = 0; p + 1 < H; p += 2) mtg_store2(row + dm * N + p, clo[p], clo[p + 1]);
    }
    park[dm] = clo[H - 1];
  }
  {
    const double* ai = mtg_ainv<C>(P);   // [H][N]
    double acc[H][D];
#pragma unroll
    for (int jj = 0; jj < H; ++jj) {
#pragma unroll
      for (int dm = 0; dm < D; ++dm) acc[jj][dm] = 0.0;
    }
    // source index k outermost, the H*D independent accumulators innermost (no back-to-back dependent FMAs)
#pragma unroll
    for (int k = 0; k < N; ++k) {
#pragma unroll
      for (int jj = 0; jj < H; ++jj) {
        const double a = ai[jj * N + k];
#pragma unroll
        for (int dm = 0; dm < D; ++dm) acc[jj][dm] = mtg_fma(a, dl[dm][k], acc[jj][dm]);
      }
    }
    double prev[D];
#pragma unroll
    for (int dm = 0; dm < D; ++dm) prev[dm] = park[dm];
#pragma unroll
    for (int jj = 0; jj < H; ++jj) {
#pragma unroll
      for (int dm = 0; dm < D; ++dm) {
        qs[dm][H + jj] = acc[jj][dm];
        const double cj = mtg_mul(acc[jj][dm], tp[jj]);
        // coefficient index H + jj; store in aligned pairs (even index first)
        if constexpr (kStore) {
          if (((H + jj) & 1) != 0) mtg_store2(row + dm * N + H + jj - 1, prev[dm], cj);
          else prev[dm] = cj;
        }
      }
    }
  }
  if constexpr (kStore) io.commit(P, seg);
  double cost = 0.0;
  if constexpr ((OUT & 1) != 0) {
    // 0.5 c^T Q(T) c = 0.5 T^(1-2d) q^T Q(1) q with q_j = c_j T^j   (impl/...:124-140).
    // Q(1) is symmetric and zero below row/column d: static full loops, no runtime indexing.
    const double* q1 = mtg_q1<C>(P);
    double acc[D];
#pragma unroll
    for (int dm = 0; dm < D; ++dm) acc[dm] = 0.0;
#pragma unroll
    for (int r = 0; r < N; ++r) {
      double row[D];
      {
        const double qd = 0.5 * q1[r * N + r];
#pragma unroll
        for (int dm = 0; dm < D; ++dm) row[dm] = qd * qs[dm][r];
      }
#pragma unroll
      for (int cc = 0; cc < r; ++cc) {
        const double qv = q1[r * N + cc];
#pragma unroll
        for (int dm = 0; dm < D; ++dm) row[dm] = mtg_fma(qv, qs[dm][cc], row[dm]);
      }
#pragma unroll
      for (int dm = 0; dm < D; ++dm) acc[dm] = mtg_fma(row[dm], qs[dm][r], acc[dm]);
    }
#pragma unroll
    for (int dm = 0; dm < D; ++dm) cost = mtg_fma(bs[0], acc[dm], cost);
  }
  return cost;
}

// number of doubles a lane publishes for the middle vertex
template <class C> MTG_HD int mtg_mid_slots(int mm) {
  const int f = C::H - mtg_popc(mm);
  return f * (f + 1) / 2 + C::D * f;
}

template <class C>
MTG_HD void mtg_pack_mid(const MtgLane<C>& ln, int mm, double* buf, int stride) {
  int slot = 0;
#pragma unroll
  for (int p = 0; p < C::H; ++p) {
    if ((mm >> p) & 1) continue;
#pragma unroll
    for (int q = 0; q <= p; ++q) {
      if ((mm >> q) & 1) continue;
      buf[slot * stride] = ln.Sc[p][q];
      ++slot;
    }
  }
#pragma unroll
  for (int dm = 0; dm < C::D; ++dm) {
#pragma unroll
    for (int p = 0; p < C::H; ++p) {
      if ((mm >> p) & 1) continue;
      buf[slot * stride] = ln.rc[dm][p];
      ++slot;
    }
  }
}

// merge with the other direction's contribution, solve the middle vertex -> xm (all h slots)
template <class C, int DIR>
MTG_HD void mtg_solve_mid(const MtgParams& P, long long b, MtgLane<C>& ln, int vm, int mm,
                          const double* other, int stride, double (&xm)[C::D][C::H]) {
  constexpr int H = C::H, D = C::D;
  double A[H][H], dinv[H];
  int slot = 0;
#pragma unroll
  for (int p = 0; p < H; ++p) {
    dinv[p] = 0.0;
#pragma unroll
    for (int q = 0; q < H; ++q) A[p][q] = 0.0;
  }
#pragma unroll
  for (int p = 0; p < H; ++p) {
    if ((mm >> p) & 1) continue;
#pragma unroll
    for (int q = 0; q <= p; ++q) {
      if ((mm >> q) & 1) continue;
      A[p][q] = ln.Sc[p][q] + other[slot * stride];
      ++slot;
    }
  }
  mtg_load_vals<C, DIR>(P, b, vm, mm, ln, xm);
  double r[D][H];
#pragma unroll
  for (int dm = 0; dm < D; ++dm) {
#pragma unroll
    for (int p = 0; p < H; ++p) {
      r[dm][p] = 0.0;
      if ((mm >> p) & 1) continue;
      r[dm][p] = ln.rc[dm][p] + other[slot * stride];
      ++slot;
    }
  }
  mtg_ldl<H>(A, dinv, mm, ln.flags);
  {
    double X[H][D];
#pragma unroll
    for (int p = 0; p < H; ++p) {
#pragma unroll
      for (int dm = 0; dm < D; ++dm) X[p][dm] = r[dm][p];
    }
    mtg_ldl_solve_multi<H, D>(A, dinv, mm, X, 0);
#pragma unroll
    for (int dm = 0; dm < D; ++dm) {
#pragma unroll
      for (int p = 0; p < H; ++p) {
        if (!((mm >> p) & 1)) xm[dm][p] = X[p][dm];
      }
    }
  }
}

template <class C>
MTG_HD void mtg_store_free_impl(const MtgParams& P, long long b, int v, int mask, const double (&x)[C::D][C::H]) {
  if (P.dfree == nullptr) return;
  const int off = mtg_offP<C>(P, v);
#pragma unroll
  for (int dm = 0; dm < C::D; ++dm) {
#pragma unroll
    for (int p = 0; p < C::H; ++p) {
      if ((mask >> p) & 1) continue;
      const int col = off + (p - mtg_popc(mask & ((1 << p) - 1)));
      P.dfree[b * P.ps_b + (long long)(P.dim0 + dm) * P.ps_d + (long long)col * P.ps_c] = x[dm][p];
    }
  }
}
template <class C, int OUT>
MTG_HD void mtg_store_free(const MtgParams& P, long long b, int v, int mask, const double (&x)[C::D][C::H]) {
  if constexpr ((OUT & 2) == 0) return;
  if constexpr (C::H == 1) {
    // N = 2: a vertex has one slot.  Written without the generic slot loop: with that loop inlined hipcc
    // (ROCm 7.2) generates wrong code for the <h = 1, D = 2> instantiation (fixed values read back as zero at
    // -O1 and -O3; D = 1 / 3 / 4 unaffected; tools/debug_case3.py reproduces it).
    if (P.dfree == nullptr || (mask & 1)) return;
    double* dst = P.dfree + b * P.ps_b + (long long)P.dim0 * P.ps_d + (long long)mtg_offP<C>(P, v) * P.ps_c;
#pragma unroll
    for (int dm = 0; dm < C::D; ++dm) {
      *dst = x[dm][0];
      dst += P.ps_d;
    }
    return;
  }
  mtg_store_free_impl<C>(P, b, v, mask, x);
}

// Which entries (p, q) of a step's kept matrix exist: G's free x free block, or (factor store) the lower triangle of the
// left vertex's free x free pivot block.
template <class C>
MTG_HD constexpr bool mtg_keep(int p, int q, int ml, int mr) {
  if ((ml >> p) & 1) return false;
  if (C::kFS) return q <= p && !((ml >> q) & 1);
  return !((mr >> q) & 1);
}

// s_p x_p of a chain step's two vertices (every slot, s = (DIR T)^p): what the back-substitution of the scaled-variable chain
// works on, and what the coefficient recovery multiplies A(1)^-1 with
template <class C>
struct MtgScaledEnds { double l[C::D][C::H], r[C::D][C::H]; };

// s[p] = (DIR T)^p, si[p] = (DIR T)^-p
template <int H, int DIR>
MTG_HD void mtg_bwd_scales(double T, double (&s)[H], double (&si)[H]) {
  const double tinv = mtg_rcp(T);
  s[0] = 1.0; si[0] = 1.0;
  if constexpr (H > 1) { s[1] = DIR > 0 ? T : -T; si[1] = DIR > 0 ? tinv : -tinv; }
#pragma unroll
  for (int p = 2; p < H; ++p) {
    s[p] = mtg_mul(s[p / 2], s[p - p / 2]);
    si[p] = mtg_mul(si[p / 2], si[p - p / 2]);
  }
}

// Back-substitution of one chain step from the FACTOR of its pivot block (MtgCfg::kFS): F strict lower = L, diagonal = 1 / d,
// in the segment's scaling: x^_l = g^ - (L D L^T)^-1 (H1_lr x^_r) with x^_r = s x_r, then xl = [fixed values | x^_l / s]; the scaled
// values of both vertices go on to the coefficient recovery (ye).
template <class C, int DIR>
MTG_HD void mtg_bwd_backsub_fs(const MtgParams& P, int ml, int mr, double T, const double (&fix_l)[C::D][C::H],
                               const double (&F)[C::H][C::H], const double (&g)[C::D][C::H], const double (&xr)[C::D][C::H],
                               double (&xl)[C::D][C::H], MtgScaledEnds<C>& ye) {
  constexpr int H = C::H, D = C::D, N = C::N;
  double s[H], si[H];
  mtg_bwd_scales<H, DIR>(T, s, si);
  const double* hc = mtg_h1<C>(P);
  double (&y)[D][H] = ye.r;
  double w[H][D], dinv[H];
#pragma unroll
  for (int q = 0; q < H; ++q) {
#pragma unroll
    for (int dm = 0; dm < D; ++dm) y[dm][q] = mtg_pin(mtg_mul(s[q], mtg_pin(xr[dm][q])));
  }
#pragma unroll
  for (int p = 0; p < H; ++p) {
    dinv[p] = F[p][p];
#pragma unroll
    for (int dm = 0; dm < D; ++dm) w[p][dm] = 0.0;
  }
#pragma unroll
  for (int q = 0; q < H; ++q) {          // source column outermost: independent accumulators adjacent
    if ((mr >> q) & 1) continue;
#pragma unroll
    for (int p = 0; p < H; ++p) {
      if ((ml >> p) & 1) continue;
      const double c = hc[p * N + H + q];
#pragma unroll
      for (int dm = 0; dm < D; ++dm) w[p][dm] = mtg_fma(c, y[dm][q], w[p][dm]);
    }
  }
  // (y above is formed from a pinned COPY of xr: the same product s x_r is formed by mtg_recover for the segment's coefficients
  // (dl), where the A^-1 row of coefficient h has entries +-1, fma(1, dl, acc) folds to an addition and the compiler contracts
  // the product into it or not depending on the product's other uses.  Sharing the product with this function changed that
  // decision in the coefficient-only kernel of N = 10 / K = 32 but not in its extra-output twin: c_h off by 4 - 140 ulps
  // between two kernels that must agree bit for bit.)
  // (L D L^T)^-1 w in place.  Every product that feeds an addition or subtraction below is pinned (an empty asm the value
  // passes through): with -ffp-contract=fast the compiler may otherwise fuse a product into the final subtraction in one
  // kernel and not in another instantiation of the same step -- observed: g of a REGISTER step is the forward phase's
  // x * (1 / d), still visible as a product when "g - w" is formed here, and the coefficient-only and extra-output kernels of
  // N = 10 / K = 32 differed in the last bit of 3 % of their coefficients -- and the coefficient-only / extra-output / queue
  // kernels of a shape are required to be bit-identical (tests/test_gpu_dimlane.py).
#pragma unroll
  for (int i = 0; i < H; ++i) {
    if ((ml >> i) & 1) continue;
#pragma unroll
    for (int k = 0; k < i; ++k) {
      if ((ml >> k) & 1) continue;
#pragma unroll
      for (int dm = 0; dm < D; ++dm) w[i][dm] = mtg_fma(-F[i][k], w[k][dm], w[i][dm]);
    }
  }
#pragma unroll
  for (int i = 0; i < H; ++i) {
    if ((ml >> i) & 1) continue;
#pragma unroll
    for (int dm = 0; dm < D; ++dm) w[i][dm] = mtg_pin(mtg_mul(w[i][dm], dinv[i]));
  }
#pragma unroll
  for (int i = H - 1; i >= 0; --i) {
    if ((ml >> i) & 1) continue;
#pragma unroll
    for (int k = i + 1; k < H; ++k) {
      if ((ml >> k) & 1) continue;
#pragma unroll
      for (int dm = 0; dm < D; ++dm) w[i][dm] = mtg_fma(-F[k][i], w[k][dm], w[i][dm]);
    }
  }
#pragma unroll
  for (int dm = 0; dm < D; ++dm) {
#pragma unroll
    for (int p = 0; p < H; ++p) {
      ye.l[dm][p] = ((ml >> p) & 1) ? mtg_pin(mtg_mul(s[p], fix_l[dm][p])) : mtg_pin(mtg_pin(g[dm][p]) - w[p][dm]);
      xl[dm][p] = ((ml >> p) & 1) ? fix_l[dm][p] : mtg_pin(mtg_mul(ye.l[dm][p], si[p]));
    }
  }
}

// One back-substitution step + coefficient recovery of the segment it completes.
// xr: solution (all slots) at the right vertex on entry, at the left vertex on exit.
// Back-substitution of one chain step in the segment's scaling (s = (DIR T)^p): x^_l = g^ - G^ (s xr) over the free slots,
// xl = [fixed values | x^_l / s] (all slots of the left vertex).  fix_l: the (unscaled) fixed values of the left vertex, already in
// registers.  G and g are dead afterwards; ye: s x of both vertices for the coefficient recovery.
template <class C, int DIR>
MTG_HD void mtg_bwd_backsub(int ml, int mr, double T, const double (&fix_l)[C::D][C::H],
                            const double (&G)[C::H][C::H], const double (&g)[C::D][C::H], const double (&xr)[C::D][C::H],
                            double (&xl)[C::D][C::H], MtgScaledEnds<C>& ye) {
  constexpr int H = C::H, D = C::D;
  double s[H], si[H];
  mtg_bwd_scales<H, DIR>(T, s, si);
#pragma unroll
  for (int dm = 0; dm < D; ++dm) {
#pragma unroll
    for (int p = 0; p < H; ++p) {
      ye.r[dm][p] = mtg_pin(mtg_mul(s[p], mtg_pin(xr[dm][p])));
      ye.l[dm][p] = ((ml >> p) & 1) ? mtg_pin(mtg_mul(s[p], fix_l[dm][p])) : g[dm][p];
    }
  }
#pragma unroll
  for (int q = 0; q < H; ++q) {
    if ((mr >> q) & 1) continue;
#pragma unroll
    for (int p = 0; p < H; ++p) {
      if ((ml >> p) & 1) continue;
#pragma unroll
      for (int dm = 0; dm < D; ++dm) ye.l[dm][p] = mtg_fma(-G[p][q], ye.r[dm][q], ye.l[dm][p]);
    }
  }
#pragma unroll
  for (int dm = 0; dm < D; ++dm) {
#pragma unroll
    for (int p = 0; p < H; ++p) {
      if (!((ml >> p) & 1)) ye.l[dm][p] = mtg_pin(ye.l[dm][p]);
      xl[dm][p] = ((ml >> p) & 1) ? fix_l[dm][p] : mtg_pin(mtg_mul(ye.l[dm][p], si[p]));
    }
  }
}

// Second half of a backward step: optional d_P output, coefficient recovery of the segment the step completes
// (T: its time), xr <- xl.  ye: the scaled vertex values the back-substitution formed (scaled-variable chain); the overload
// without it forms them in mtg_recover (a step whose left vertex is copied, not solved).
// No `active` guard: tail lanes are clamped duplicates of the last trajectory and store identical values to identical
// addresses.  A per-lane condition here makes the compiler unswitch the chain loop on it, and the wave-cooperative
// coefficient drain (all 64 lanes must take part together) would run in two halves.
template <class C, int DIR, int OUT, bool PRE, class IO>
MTG_HD double mtg_bwd_finish_impl(const MtgParams& P, long long b, int j, int ml, double T, const double (&xl)[C::D][C::H],
                                  double (&xr)[C::D][C::H], IO& io, const MtgScaledEnds<C>* ye) {
  constexpr int H = C::H, D = C::D;
  const int K = mtg_nseg<C>(P);
  const int seg = mtg_seg<DIR>(K, j), vl = mtg_vl<DIR>(K, j);
  mtg_store_free<C, OUT>(P, b, vl, ml, xl);
  double cost;
  if constexpr (PRE) {
    if (DIR > 0) cost = mtg_recover<C, OUT, 1>(P, b, seg, T, xl, xr, io, ye->l, ye->r);
    else cost = mtg_recover<C, OUT, -1>(P, b, seg, T, xr, xl, io, ye->r, ye->l);
  } else {
    if (DIR > 0) cost = mtg_recover<C, OUT>(P, b, seg, T, xl, xr, io);
    else cost = mtg_recover<C, OUT>(P, b, seg, T, xr, xl, io);
  }
#pragma unroll
  for (int dm = 0; dm < D; ++dm) {
#pragma unroll
    for (int p = 0; p < H; ++p) xr[dm][p] = xl[dm][p];
  }
  return cost;
}
template <class C, int DIR, int OUT, class IO>
MTG_HD double mtg_bwd_finish(const MtgParams& P, long long b, int j, int ml, double T, const double (&xl)[C::D][C::H],
                             double (&xr)[C::D][C::H], IO& io) {
  return mtg_bwd_finish_impl<C, DIR, OUT, false>(P, b, j, ml, T, xl, xr, io, nullptr);
}
template <class C, int DIR, int OUT, class IO>
MTG_HD double mtg_bwd_finish(const MtgParams& P, long long b, int j, int ml, double T, const double (&xl)[C::D][C::H],
                             double (&xr)[C::D][C::H], IO& io, const MtgScaledEnds<C>& ye) {
  return mtg_bwd_finish_impl<C, DIR, OUT, true>(P, b, j, ml, T, xl, xr, io, &ye);
}

template <class C, int DIR, int OUT, class IO>
MTG_HD double mtg_bwd_step(const MtgParams& P, long long b, int j, int ml, int mr, const MtgLane<C>& ln,
                           const double (&G)[C::H][C::H], const double (&g)[C::D][C::H],
                           double (&xr)[C::D][C::H], IO& io, bool active) {
  (void)active;
  double fix_l[C::D][C::H], xl[C::D][C::H];
  MtgScaledEnds<C> ye;
  const double T = mtg_step_time<C, DIR>(P, b, j, ln);
  mtg_load_vals<C, DIR>(P, b, mtg_vl<DIR>(mtg_nseg<C>(P), j), ml, ln, fix_l);
  mtg_bwd_backsub<C, DIR>(ml, mr, T, fix_l, G, g, xr, xl, ye);
  return mtg_bwd_finish<C, DIR, OUT>(P, b, j, ml, T, xl, xr, io, ye);
}

// Back-substitution data of one chain step in the lane-coalesced workspace (element stride = number of lanes):
// only free x free entries of G and free entries of g exist; they are packed in traversal order and addressed by
// walking a pointer (one 64-bit add per element) -- indexed addressing made the compiler keep ~40 hoisted
// 64-bit slot offsets live across the chain loop (450-900 SGPR spills in the rolled kernels).
template <class C, class PTR>
MTG_HD void mtg_ws_store(PTR w, long long stride, const double (&G)[C::H][C::H], const double (&g)[C::D][C::H],
                         int ml, int mr) {
  constexpr int H = C::H, D = C::D;
#pragma unroll
  for (int p = 0; p < H; ++p) {
    if ((ml >> p) & 1) continue;
#pragma unroll
    for (int q = 0; q < H; ++q) {
      if ((mr >> q) & 1) continue;
      *w = G[p][q];
      w += stride;
    }
#pragma unroll
    for (int dm = 0; dm < D; ++dm) {
      *w = g[dm][p];
      w += stride;
    }
  }
}
template <class C, class PTR>
MTG_HD void mtg_ws_load(PTR w, long long stride, double (&G)[C::H][C::H], double (&g)[C::D][C::H], int ml,
                        int mr) {
  constexpr int H = C::H, D = C::D;
#pragma unroll
  for (int p = 0; p < H; ++p) {
#pragma unroll
    for (int q = 0; q < H; ++q) {
      G[p][q] = 0.0;
      if (((ml >> p) & 1) || ((mr >> q) & 1)) continue;
      G[p][q] = *w;
      w += stride;
    }
#pragma unroll
    for (int dm = 0; dm < D; ++dm) {
      g[dm][p] = 0.0;
      if ((ml >> p) & 1) continue;
      g[dm][p] = *w;
      w += stride;
    }
  }
}

// Shared-G workspace layout (MtgCfg::DLW lanes per trajectory, lane = dim * TPW + trajectory; C::D == 1): row r of a step
// holds G's free elements DLW*r .. DLW*r + DLW-1 (traversal order), element DLW*r + k in the column of dimension lane k.
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) double mtg_lds_double;
// this lane's column of workspace step j in the wave's LDS step area (steps WSJ - LSJ .. WSJ - 1, WSE rows of 64 doubles each)
template <class C>
__device__ __forceinline__ mtg_lds_double* mtg_lds_step(const MtgParams& P, int j) {
  return (mtg_lds_double*)(size_t)(P.lds_steps + (unsigned)(j - (C::WSJ - C::LSJ)) * (unsigned)(C::WSE * 64 * sizeof(double)));
}
#endif
template <int DL>
MTG_HD double mtg_pick(int d, const double (&c)[DL]) {
  double v = c[0];
#pragma unroll
  for (int k = 1; k < DL; ++k) {
    int hit = d == k;
#if defined(__HIP_DEVICE_COMPILE__)
    // opaque flag: otherwise the select chain is canonicalised into a dynamically indexed load from a private array of
    // the candidates (scratch store + load per workspace row)
    asm volatile("" : "+v"(hit));
#endif
    v = hit ? c[k] : v;
  }
  return v;
}
template <class C, class PTR>
MTG_HD void mtg_ws_store_shared(PTR w, long long stride, int d, const double (&G)[C::H][C::H],
                                const double (&g)[C::D][C::H], int ml, int mr) {
  constexpr int H = C::H, DL = C::DLW > 0 ? C::DLW : 1;
  double cand[DL];
  int cnt = 0;
#pragma unroll
  for (int p = 0; p < H; ++p) {
#pragma unroll
    for (int q = 0; q < H; ++q) {
      if (!mtg_keep<C>(p, q, ml, mr)) continue;
      cand[cnt % DL] = G[p][q];
      ++cnt;
      if (cnt % DL == 0) {
        *w = mtg_pick<DL>(d, cand);
        w += stride;
      }
    }
  }
  if (cnt % DL != 0) {   // last, partial row: the lanes beyond it store a duplicate that is never read
#pragma unroll
    for (int k = 1; k < DL; ++k)
      if (k >= cnt % DL) cand[k] = cand[0];
    *w = mtg_pick<DL>(d, cand);
    w += stride;
  }
#pragma unroll
  for (int p = 0; p < H; ++p) {
    if ((ml >> p) & 1) continue;
    *w = g[0][p];
    w += stride;
  }
}
// w: this lane's own column; share: element offset to the trajectory's dimension-0 column; TPW lanes between dimension columns
template <class C, class PTR>
MTG_HD void mtg_ws_load_shared(PTR w, long long stride, long long share, double (&G)[C::H][C::H],
                               double (&g)[C::D][C::H], int ml, int mr) {
  constexpr int H = C::H, DL = C::DLW > 0 ? C::DLW : 1, TPW = 64 / DL;
  PTR col[DL];
#pragma unroll
  for (int k = 0; k < DL; ++k) col[k] = w + share + k * TPW;
  int cnt = 0;
#pragma unroll
  for (int p = 0; p < H; ++p) {
#pragma unroll
    for (int q = 0; q < H; ++q) {
      G[p][q] = 0.0;
      if (!mtg_keep<C>(p, q, ml, mr)) continue;
      G[p][q] = *col[cnt % DL];
      col[cnt % DL] += stride;
      ++cnt;
    }
  }
  PTR wg = w + (long long)((cnt + DL - 1) / DL) * stride;
#pragma unroll
  for (int p = 0; p < H; ++p) {
    g[0][p] = 0.0;
    if ((ml >> p) & 1) continue;
    g[0][p] = *wg;
    wg += stride;
  }
}

// MtgCfg::kRegShared: register steps.  pack: this dimension lane's share of G (row r = elements DLW r .. DLW r + DLW - 1 in
// traversal order, element DLW r + k kept by the lane of dimension k).  unpack: all elements back, element DLW r + k from
// the sibling lane of dimension k (byte address perm[k] = 4 * that lane) -- a ds_bpermute pair per double, no LDS memory.
template <class C>
MTG_HD void mtg_rs_pack(int d, const double (&G)[C::H][C::H], int ml, int mr, double (&Gs)[C::kRegShared ? C::GROWS : 1]) {
  constexpr int H = C::H, DL = C::DLW > 0 ? C::DLW : 1;
  double cand[DL];
  int cnt = 0;
#pragma unroll
  for (int p = 0; p < H; ++p) {
#pragma unroll
    for (int q = 0; q < H; ++q) {
      if (!mtg_keep<C>(p, q, ml, mr)) continue;
      cand[cnt % DL] = G[p][q];
      ++cnt;
      if (cnt % DL == 0) {
        Gs[cnt / DL - 1] = mtg_pick<DL>(d, cand);
#if defined(__HIP_DEVICE_COMPILE__)
        // the share is materialised HERE: without the opaque use the selects are sunk to the back-substitution, and all of
        // G stays live in between (measured in the run-time-K body: ~50 instead of ~20 registers per register step)
        asm volatile("" : "+v"(Gs[cnt / DL - 1]));
#endif
      }
    }
  }
  if (cnt % DL != 0) {   // last, partial row: the lanes beyond it keep a duplicate that is never read
#pragma unroll
    for (int k = 1; k < DL; ++k)
      if (k >= cnt % DL) cand[k] = cand[0];
    Gs[cnt / DL] = mtg_pick<DL>(d, cand);
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(Gs[cnt / DL]));
#endif
  }
}
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double mtg_bperm(int addr, double v) {
  const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
template <class C>
__device__ __forceinline__ void mtg_rs_unpack(const int (&perm)[C::DLW > 0 ? C::DLW : 1], const double (&Gs)[C::kRegShared ? C::GROWS : 1],
                                              int ml, int mr, double (&G)[C::H][C::H]) {
  constexpr int H = C::H, DL = C::DLW > 0 ? C::DLW : 1;
  int cnt = 0;
#pragma unroll
  for (int p = 0; p < H; ++p) {
#pragma unroll
    for (int q = 0; q < H; ++q) {
      G[p][q] = 0.0;
      if (!mtg_keep<C>(p, q, ml, mr)) continue;
      G[p][q] = mtg_bperm(perm[cnt % DL], Gs[cnt / DL]);
      ++cnt;
    }
  }
}
#endif

// ---- whole-lane phases -----------------------------------------------------------------
// wsl: this lane's slab of the generic-mode workspace (element stride P.ws_stride)
// do_preload = false: the caller already filled ln.T / ln.fx (the kernel prefetches the next tile's inputs
// while the current tile is being solved).
template <class C, int DIR>
MTG_HD void mtg_lane_forward(const MtgParams& P, long long b, MtgLane<C>& ln, double* wsl, bool do_preload = true) {
  constexpr int H = C::H, D = C::D;
  if (do_preload) mtg_preload<C, DIR>(P, b, ln);
  mtg_lane_reset<C, DIR>(ln);
  if constexpr (C::kStatic) {
    constexpr int KC = DIR > 0 ? C::KA : C::KB;
#pragma unroll
    for (int j = 0; j < KC; ++j) {
      const int ml = mtg_mask<C>(P, mtg_vl<DIR>(C::KT, j)), mr = mtg_mask<C>(P, mtg_vr<DIR>(C::KT, j));
      if (j < C::WSJ) {
        double G[H][H], g[D][H];
        mtg_fwd_step<C, DIR>(P, b, j, ml, mr, ln, G, g);
        if constexpr (C::DLW > 0) {
          if (j >= C::WSJ - C::LSJ) {
#if defined(__HIP_DEVICE_COMPILE__)
            mtg_ws_store_shared<C>(mtg_lds_step<C>(P, j), 64, P.dim0, G, g, ml, mr);
#endif
          } else {
            mtg_ws_store_shared<C>(mtg_glb(wsl + (long long)j * C::WSE * P.ws_stride), P.ws_stride, P.dim0, G, g, ml, mr);
          }
        } else {
          mtg_ws_store<C>(mtg_glb(wsl + (long long)j * C::WSE * P.ws_stride), P.ws_stride, G, g, ml, mr);
        }
      } else if constexpr (C::kRegShared) {
        constexpr int JR0 = C::WSJ;
        double G[H][H];
        mtg_fwd_step<C, DIR>(P, b, j, ml, mr, ln, G, ln.g[j < JR0 ? 0 : j - JR0]);
        mtg_rs_pack<C>(P.dim0, G, ml, mr, ln.Gs[j < JR0 ? 0 : j - JR0]);
      } else {
        constexpr int JR0 = C::WSJ;
        mtg_fwd_step<C, DIR>(P, b, j, ml, mr, ln, ln.G[j < JR0 ? 0 : j - JR0], ln.g[j < JR0 ? 0 : j - JR0]);
      }
    }
  } else {
    const int K = P.K;
    const int kc = DIR > 0 ? (K + 1) / 2 : K / 2;
    if constexpr (C::kRolled) {
      // K >= 2: step 0 starts at a trajectory end, every other vertex of the half-chain (incl. the middle one)
      // carries the interior mask => the loop body is compiled with constant masks.
      // Issue order per step: inputs of step j+1 (segment time, fixed values of its right vertex), arithmetic of step
      // j, then step j's back-substitution stores.  Loads and stores share the in-order vmcnt counter on this
      // hardware: with the loads at the top of the NEXT iteration instead, waiting for them meant waiting for every
      // one of the ~H*H + D*H workspace stores of the previous step to be acknowledged -- one full store round trip
      // per chain step, serialised (measured: ~10 us per step at B = 100k, K = 32).
      constexpr int M0 = DIR > 0 ? C::MS : C::ME;
      double T_cur = mtg_step_time<C, DIR>(P, b, 0, ln);
      double fl[D][H], fr[D][H];
      mtg_load_vals<C, DIR>(P, b, mtg_vl<DIR>(K, 0), M0, ln, fl);
      mtg_load_vals<C, DIR>(P, b, mtg_vr<DIR>(K, 0), C::MI, ln, fr);
      for (int j = 0; j < kc; ++j) {
        double G[H][H], g[D][H];
        const int jn = j + 1 < kc ? j + 1 : j;     // last step: harmless reload, keeps the issue pattern uniform
        const double T_nxt = mtg_step_time<C, DIR>(P, b, jn, ln);
        double fn[D][H];
        mtg_load_vals<C, DIR>(P, b, mtg_vr<DIR>(K, jn), C::MI, ln, fn);
        if (j == 0) mtg_fwd_step_core<C, DIR>(P, M0, C::MI, ln, T_cur, fl, fr, G, g);
        else mtg_fwd_step_core<C, DIR>(P, C::MI, C::MI, ln, T_cur, fl, fr, G, g);
        mtg_ws_store<C>(mtg_glb(wsl + (long long)j * (H * H + D * H) * P.ws_stride), P.ws_stride, G, g, j == 0 ? M0 : C::MI,
                        C::MI);
        T_cur = T_nxt;
#pragma unroll
        for (int dm = 0; dm < D; ++dm) {
#pragma unroll
          for (int p = 0; p < H; ++p) {
            fl[dm][p] = fr[dm][p];
            fr[dm][p] = fn[dm][p];
          }
        }
      }
    } else {
      for (int j = 0; j < kc; ++j) {
        double G[H][H], g[D][H];
        const int ml = mtg_mask<C>(P, mtg_vl<DIR>(K, j));
        const int mr = mtg_mask<C>(P, mtg_vr<DIR>(K, j));
        mtg_fwd_step<C, DIR>(P, b, j, ml, mr, ln, G, g);
        mtg_ws_store<C>(mtg_glb(wsl + (long long)j * (H * H + D * H) * P.ws_stride), P.ws_stride, G, g, ml, mr);
      }
      // the middle vertex's explicit right-hand side: once, by the forward direction (both directions sum the two carried ones)
      mtg_unscale_carried<C, DIR>(P, ln);
      if (DIR > 0) mtg_add_explicit_rhs<C>(P, b, (K + 1) / 2, mtg_mask<C>(P, (K + 1) / 2), ln);
    }
  }
  if constexpr (C::kCT) mtg_unscale_carried<C, DIR>(P, ln);     // (the generic branch above has already left the scaling)
}

// `active` = this lane owns a real trajectory (tail tiles run clamped duplicates whose outputs
// are suppressed; every lane still takes part in the cooperative coefficient flush).
template <class C, int DIR, int OUT, class IO>
MTG_HD void mtg_lane_finish(const MtgParams& P, long long b, MtgLane<C>& ln, const double* wsl,
                            const double* other, int stride, IO& io, bool active, double* cost_out = nullptr) {
  constexpr int H = C::H, D = C::D;
  const int K = mtg_nseg<C>(P);
  const int vm = (K + 1) / 2;
  const int mm = C::kRolled ? C::MI : mtg_mask<C>(P, vm);   // rolled: K >= 2, the middle vertex is interior
  double xr[D][H];
  mtg_solve_mid<C, DIR>(P, b, ln, vm, mm, other, stride, xr);
  if (DIR > 0) mtg_store_free<C, OUT>(P, b, vm, mm, xr);
  double cost = 0.0;
  if constexpr (C::kStatic) {
    constexpr int KC = DIR > 0 ? C::KA : C::KB;
    // steps below C::WSJ: (G, g) come back from the workspace, requested one step ahead -- right after the previous
    // step's back-substitution and BEFORE its coefficient stores (loads and stores retire through one in-order counter)
    double Gw[H][H], gw[D][H];
    [[maybe_unused]] int perm[C::DLW > 0 ? C::DLW : 1];
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (C::kRegShared) {   // byte addresses (4 * lane) of this trajectory's dimension lanes, for ds_bpermute
#pragma unroll
      for (int k = 0; k < C::DLW; ++k) perm[k] = 4 * ((int)(threadIdx.x & 63) + (int)P.ws_share + k * (64 / C::DLW));
    }
#endif
    auto request = [&](int j) {
      if constexpr (C::kRegShared) {
        if (j >= C::WSJ) {   // register step: G back from the three dimension lanes' shares (g is per lane)
#if defined(__HIP_DEVICE_COMPILE__)
          mtg_rs_unpack<C>(perm, ln.Gs[j - C::WSJ < 0 ? 0 : j - C::WSJ], mtg_mask<C>(P, mtg_vl<DIR>(C::KT, j)),
                           mtg_mask<C>(P, mtg_vr<DIR>(C::KT, j)), Gw);
#endif
          return;
        }
      }
      if constexpr (C::DLW > 0) {
        if (j >= C::WSJ - C::LSJ) {
#if defined(__HIP_DEVICE_COMPILE__)
          mtg_ws_load_shared<C>(mtg_lds_step<C>(P, j), 64, P.ws_share, Gw, gw, mtg_mask<C>(P, mtg_vl<DIR>(C::KT, j)),
                                mtg_mask<C>(P, mtg_vr<DIR>(C::KT, j)));
#endif
        } else {
          mtg_ws_load_shared<C>(mtg_glb(wsl + (long long)j * C::WSE * P.ws_stride), P.ws_stride, P.ws_share, Gw, gw,
                                mtg_mask<C>(P, mtg_vl<DIR>(C::KT, j)), mtg_mask<C>(P, mtg_vr<DIR>(C::KT, j)));
        }
      } else
        mtg_ws_load<C>(mtg_glb(wsl + (long long)j * C::WSE * P.ws_stride), P.ws_stride, Gw, gw, mtg_mask<C>(P, mtg_vl<DIR>(C::KT, j)),
                       mtg_mask<C>(P, mtg_vr<DIR>(C::KT, j)));
    };
    if ((C::WSJ > 0 && KC <= C::WSJ) || C::kRegShared) request(KC - 1);
#pragma unroll
    for (int j = KC - 1; j >= 0; --j) {
      const int ml = mtg_mask<C>(P, mtg_vl<DIR>(C::KT, j)), mr = mtg_mask<C>(P, mtg_vr<DIR>(C::KT, j));
      if constexpr (C::WSJ > 0 || C::kRegShared) {
        double fix_l[D][H], xl[D][H];
        mtg_load_vals<C, DIR>(P, b, mtg_vl<DIR>(C::KT, j), ml, ln, fix_l);
        const int jr = j - C::WSJ < 0 ? 0 : j - C::WSJ;
        MtgScaledEnds<C> ye;
        const double Tj = mtg_step_time<C, DIR>(P, b, j, ln);
        if constexpr (C::kFS) {
          if (j < C::WSJ) mtg_bwd_backsub_fs<C, DIR>(P, ml, mr, Tj, fix_l, Gw, gw, xr, xl, ye);
          else if constexpr (C::kRegShared) mtg_bwd_backsub_fs<C, DIR>(P, ml, mr, Tj, fix_l, Gw, ln.g[jr], xr, xl, ye);
          else mtg_bwd_backsub_fs<C, DIR>(P, ml, mr, Tj, fix_l, ln.G[jr], ln.g[jr], xr, xl, ye);
        } else {
          if (j < C::WSJ) mtg_bwd_backsub<C, DIR>(ml, mr, Tj, fix_l, Gw, gw, xr, xl, ye);
          else if constexpr (C::kRegShared) mtg_bwd_backsub<C, DIR>(ml, mr, Tj, fix_l, Gw, ln.g[jr], xr, xl, ye);
          else mtg_bwd_backsub<C, DIR>(ml, mr, Tj, fix_l, ln.G[jr], ln.g[jr], xr, xl, ye);
        }
        // the next step's data is requested right after this step's back-substitution and BEFORE its coefficient
        // stores (loads and stores retire through one in-order counter; the ds_bpermute round trip overlaps the recovery)
        if (j >= 1 && (j - 1 < C::WSJ || C::kRegShared)) request(j - 1);
        cost += mtg_bwd_finish<C, DIR, OUT>(P, b, j, ml, Tj, xl, xr, io, ye);
      } else {
        cost += mtg_bwd_step<C, DIR, OUT>(P, b, j, ml, mr, ln, ln.G[j], ln.g[j], xr, io, active);
      }
    }
  } else {
    const int kc = DIR > 0 ? (K + 1) / 2 : K / 2;
    if constexpr (C::kRolled) {
      // Same discipline as the forward loop: the back-substitution data (G, g), segment time and fixed values of step
      // j-1 are requested right after step j's back-substitution has consumed its own (same registers), i.e. BEFORE
      // step j recovers its segment and streams out the previous one -- waiting for them never includes a store.
      constexpr int M0 = DIR > 0 ? C::MS : C::ME;
      double G[H][H], g[D][H], fl[D][H];
      double T_cur = 0.0;
      auto request = [&](int j) {
        const auto w = mtg_glb(wsl + (long long)j * (H * H + D * H) * P.ws_stride);
        if (j == 0) {
          mtg_ws_load<C>(w, P.ws_stride, G, g, M0, C::MI);
          mtg_load_vals<C, DIR>(P, b, mtg_vl<DIR>(K, j), M0, ln, fl);
        } else {
          mtg_ws_load<C>(w, P.ws_stride, G, g, C::MI, C::MI);
          mtg_load_vals<C, DIR>(P, b, mtg_vl<DIR>(K, j), C::MI, ln, fl);
        }
        T_cur = mtg_step_time<C, DIR>(P, b, j, ln);
      };
      if (kc > 0) request(kc - 1);
      for (int j = kc - 1; j >= 0; --j) {
        double xl[D][H];
        const double T_use = T_cur;
        MtgScaledEnds<C> ye;
        if (j == 0) mtg_bwd_backsub<C, DIR>(M0, C::MI, T_use, fl, G, g, xr, xl, ye);
        else mtg_bwd_backsub<C, DIR>(C::MI, C::MI, T_use, fl, G, g, xr, xl, ye);
        if (j > 0) request(j - 1);
        cost += mtg_bwd_finish<C, DIR, OUT>(P, b, j, j == 0 ? M0 : C::MI, T_use, xl, xr, io, ye);
      }
    } else {
      for (int j = kc - 1; j >= 0; --j) {
        double G[H][H], g[D][H];
        const auto w = mtg_glb(wsl + (long long)j * (H * H + D * H) * P.ws_stride);
        const int ml = mtg_mask<C>(P, mtg_vl<DIR>(K, j)), mr = mtg_mask<C>(P, mtg_vr<DIR>(K, j));
        mtg_ws_load<C>(w, P.ws_stride, G, g, ml, mr);
        cost += mtg_bwd_step<C, DIR, OUT>(P, b, j, ml, mr, ln, G, g, xr, io, active);
      }
    }
  }
  io.flush(P);
  if constexpr ((OUT & 1) != 0) {
    if (cost_out != nullptr) {        // the caller combines the lanes' partial sums (dimension-in-lane form)
      *cost_out = cost;
    } else if (P.cost != nullptr && active) {
#if defined(__HIP_DEVICE_COMPILE__)
      atomicAdd(P.cost + b, cost);
#else
      P.cost[b] += cost;
#endif
    }
  }
  if (ln.flags && active) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicOr(P.status, ln.flags);
    if (P.tstatus != nullptr) atomicOr(P.tstatus + b, ln.flags);
#else
    *P.status |= ln.flags;
    if (P.tstatus != nullptr) P.tstatus[b] |= ln.flags;
#endif
  }
}

// ---- setFreeConstraints path: coefficients from given d_free (no solve) ------------------
// One lane per trajectory; `io` is the coefficient output policy (LDS-staged in the kernel, direct on the host).
template <class C, int OUT, class IO>
MTG_HD void mtg_lane_update(const MtgParams& P, long long b, IO& io, bool active) {
  constexpr int H = C::H, D = C::D;
  const int K = P.K;
  double xa[D][H], xb[D][H];
  double cost = 0.0;
  int flags = 0;
  static_assert(!C::kStatic, "update path: generic or rolled (compile-time masks) configurations");
  MtgLane<C> dummy_lane;
  auto load_vertex = [&](int v, double (&x)[D][H]) {
    const int m = mtg_mask<C>(P, v);
    mtg_load_vals<C, 1>(P, b, v, m, dummy_lane, x);
    const int off = mtg_offP<C>(P, v);
#pragma unroll
    for (int dm = 0; dm < D; ++dm) {
#pragma unroll
      for (int p = 0; p < H; ++p) {
        if ((m >> p) & 1) continue;
        const int col = off + (p - mtg_popc(m & ((1 << p) - 1)));
        x[dm][p] = P.dfree[b * P.ps_b + (long long)(P.dim0 + dm) * P.ps_d + (long long)col * P.ps_c];
      }
    }
  };
  // Issue order per segment: vertex i+2 and time i+1 are requested BEFORE segment i is recovered (which streams out
  // segment i-1): loads and stores retire through one in-order counter, so a load issued after a store cannot be
  // consumed before that store is acknowledged.
  double xc[D][H];
  load_vertex(0, xa);
  load_vertex(1, xb);
  double T = P.times[b * P.ts_b];
  for (int i = 0; i < K; ++i) {
    const int vn = i + 2 <= K ? i + 2 : K;         // last segment: harmless reload of the end vertex
    const int tn = i + 1 < K ? i + 1 : i;
    load_vertex(vn, xc);
    const double T_nxt = P.times[b * P.ts_b + (long long)tn * P.ts_k];
    if (!(T > 0.0)) flags |= MTG_FLAG_BAD_TIME;
    cost += mtg_recover<C, OUT>(P, b, i, T, xa, xb, io);
    T = T_nxt;
#pragma unroll
    for (int dm = 0; dm < D; ++dm) {
#pragma unroll
      for (int p = 0; p < H; ++p) {
        xa[dm][p] = xb[dm][p];
        xb[dm][p] = xc[dm][p];
      }
    }
  }
  io.flush(P);
  if constexpr ((OUT & 1) != 0) {
    if (active) {
#if defined(__HIP_DEVICE_COMPILE__)
      atomicAdd(P.cost + b, cost);
#else
      P.cost[b] += cost;
#endif
    }
  }
  if (flags && active) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicOr(P.status, flags);
    if (P.tstatus != nullptr) atomicOr(P.tstatus + b, flags);
#else
    *P.status |= flags;
    if (P.tstatus != nullptr) P.tstatus[b] |= flags;
#endif
  }
}

#endif  // MTG_LANE_H_
