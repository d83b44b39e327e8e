// mtg_coop.h -- the ROW-COOPERATIVE form of the solve: 16 lanes work on ONE trajectory-half.
//
// Every other kernel of this library gives a trajectory-half (and dimension) to ONE lane: ~300-560 FP64 instructions per chain
// step in one dependent stream, the back-substitution data of all steps in that lane's registers / LDS / workspace.  That is the
// throughput mapping -- and the latency of a chain is steps x ~3000 cycles whatever the batch: 2 500 trajectories of 100
// segments occupy 240 of 1024 SIMDs for 100 us.  Here a 16-lane DPP row owns the half-chain:
//
//   * lane = one ROW of the step's block system.  Chain step j couples the free derivatives of its left vertex (already reached)
//     and its right vertex: rows [ A  U | rv ; U^T a_rr | rnext ] (impl/polynomial_optimization_linear_impl.h:308-336 restricted to
//     two neighbouring vertices; f = h - 1 free derivatives per interior vertex, D right-hand sides).  Lanes 0 .. f-1 of the row
//     ("group 0") and lanes 8 .. 8+f-1 ("group 1") hold the rows of the two vertices; the groups swap roles from step to step, so
//     the Schur complement a step leaves in the right vertex's lanes IS the next step's left block -- nothing moves.
//   * the elimination of a pivot is a rank-1 update: row_i -= (a_ij / a_jj) row_j.  With v_fmac_f64 and a DPP row_newbcast source
//     (gfx90a+: D[lane] += S0[lane n of the row] * S1[lane]) the pivot row's element arrives inside the FMA: ONE instruction per
//     column updates every row -- Gauss-Jordan (rows above the pivot too) costs nothing extra, and leaves G = A^-1 U and
//     g = A^-1 rv row by row in the left vertex's lanes (f + D doubles per lane and step, kept in LDS).
//   * back-substitution x_l = g - G x_r and the coefficient recovery c = diag(T^-i) A(1)^-1 S d (impl/...:263-283) are the same
//     broadcast-FMAs: lane i of the row ends up with coefficient i of all D dimensions and stores it.
//   * the two chain directions of a trajectory (twisted factorisation, as everywhere in this library) are two wavefronts of one
//     workgroup; they meet at the middle vertex through LDS.
//
// Per step ~190 instructions per WAVE (four trajectory-halves), i.e. ~2.4x the lane-instructions of the lane-per-half forms:
// this form does not win throughput, it wins LATENCY -- small launches of long chains (mtg_plan_launch_form 7).
//
// The file is written once for two "value" back ends: CoopDev (device: a double per lane, DPP through inline asm) and a host
// emulation of ONE 16-lane row in lock step (tests/coop_emu.cpp), so that the algorithm is checked against the oracle on the CPU.
#ifndef MTG_COOP_H_
#define MTG_COOP_H_

#include "mtg_lane.h"

namespace mtgc {

constexpr int kRow = 16;     // lanes that cooperate (one DPP row)
constexpr int kGroup = 8;    // lane offset of group 1

// Full A(1)^-1 (N x N): rows 0 .. h-1 are e_i / i! (c_i = d_i / i!), rows h .. N-1 from the generated table.
MTG_HD double ainv_entry(int H, int i, int k) {
  const int N = 2 * H;
  if (i < H) {
    if (k != i) return 0.0;
    double f = 1.0;
    for (int p = 2; p <= i; ++p) f *= (double)p;
    return 1.0 / f;
  }
  return kAinvLo[mtg_ainv_offset(N) + (i - H) * N + k];
}

// ---- the algorithm, generic over the value back end O -----------------------------------------------------------------
// O::V value (one per lane of the row), O::P predicate; O::lane16() = lane index inside the row as an "int per lane" only through
// the predicates below (no per-lane integer arithmetic is needed: everything lane-dependent is a table row or a predicate).
template <class O, int H_, int D_>
struct Coop {
  using V = typename O::V;
  using P = typename O::P;
  static constexpr int H = H_, D = D_, N = 2 * H_, F = H_ - 1;
  static_assert(F >= 1 && F <= kGroup && N <= kRow, "row layout: f <= 8 rows per vertex, N <= 16 coefficient lanes");

  // per-lane constants (built once per kernel)
  V tabE[N], tabO[N];   // H(1) row of this lane's role at even / odd step parity (LEFT role: row p, RIGHT role: row h + p)
  V ai[N];              // A(1)^-1 row of coefficient i = lane
  V keepE, keepO;       // 0 in the lanes that are LEFT at even / odd parity, else 1 (their block is cleared after a step)
  P in_g0, in_g1;       // lane belongs to group 0 / 1 (and i < f)
  P is_i[F];            // lane's row index inside its group is i
  P bit[4];             // bits of the lane index (coefficient power T^-i)
  P coef_lane;          // lane < N: holds a coefficient in the recovery
  // state
  V B0[F], B1[F], R[D]; // the lane's row: column blocks of the even / odd vertex, right-hand sides
  V x[D];               // backward: solution at the lane's own (vertex, derivative)
  P flag_singular, flag_time;

  // ---- one-time set-up: the lane-dependent constants -------------------------------------------------------------------
  // h1: H(1) table of (N, derivative) [N][N]
  template <class LaneInfo>
  MTG_HD void init(const double* h1, const LaneInfo& li) {
    // li.row_value(fn): V whose lane l is fn(l)  (device: evaluates fn for this lane; host: for all 16)
#pragma unroll
    for (int k = 0; k < N; ++k) {
      tabE[k] = li.make([&](int l) { const int g = l >> 3, i = l & 7; return i < F ? h1[(g == 0 ? i + 1 : H + i + 1) * N + k] : 0.0; });
      tabO[k] = li.make([&](int l) { const int g = l >> 3, i = l & 7; return i < F ? h1[(g == 1 ? i + 1 : H + i + 1) * N + k] : 0.0; });
      ai[k] = li.make([&](int l) { return l < N ? ainv_entry(H, l, k) : 0.0; });
    }
    keepE = li.make([&](int l) { return (l >> 3) == 0 ? 0.0 : 1.0; });
    keepO = li.make([&](int l) { return (l >> 3) == 1 ? 0.0 : 1.0; });
    in_g0 = li.pred([&](int l) { return (l >> 3) == 0 && (l & 7) < F; });
    in_g1 = li.pred([&](int l) { return (l >> 3) == 1 && (l & 7) < F; });
#pragma unroll
    for (int i = 0; i < F; ++i) is_i[i] = li.pred([&](int l) { return (l & 7) == i; });
#pragma unroll
    for (int b = 0; b < 4; ++b) bit[b] = li.pred([&](int l) { return ((l >> b) & 1) != 0; });
    coef_lane = li.pred([&](int l) { return l < N; });
#pragma unroll
    for (int q = 0; q < F; ++q) { B0[q] = O::splat(0.0); B1[q] = O::splat(0.0); }
#pragma unroll
    for (int dm = 0; dm < D; ++dm) { R[dm] = O::splat(0.0); x[dm] = O::splat(0.0); }
    flag_singular = O::pfalse();
    flag_time = O::pfalse();
  }

  // this lane's own power (sign^p T^p of its derivative index p = i + 1), and T^-lane for the recovery
  MTG_HD V own_of(const V (&s)[H]) const {
    V r = s[1];
#pragma unroll
    for (int i = 1; i < F; ++i) r = O::sel(is_i[i], s[i + 1], r);
    return r;
  }

  // ---- scales of a segment (mtg_scales of mtg_lane.h, per row): s[p] = (sign T)^p, base = T^(1 - 2 d) ---------------------------
  // (deriv == H - 1, the standard shapes' derivative, takes the compile-time exponent: no bit loop over a run-time value)
  MTG_HD void scales(V T, double sign, int deriv, V (&s)[H], V& tinv, V& base) {
    flag_time = O::por(flag_time, O::not_gt0(T));
    tinv = O::rcp(T);
    if (deriv == H - 1) base = H == 1 ? T : O::template powc<2 * (H - 1) - 1>(tinv);
    else base = deriv == 0 ? T : O::powi(tinv, 2 * deriv - 1);
    s[0] = O::splat(1.0);
    if constexpr (H > 1) s[1] = sign > 0 ? T : O::neg(T);
#pragma unroll
    for (int p = 2; p < H; ++p) s[p] = O::mul(s[p / 2], s[p - p / 2]);
  }

  // ---- elimination of pivot J of the PI-group rows from every row (Gauss-Jordan) -------------------------------------------------
  // BL: the pivot vertex's column block, BR: the other vertex's (NR = F columns, or 0 at the middle vertex).
  // r: 1 / (own element of the pivot column), computed by the caller (for pivot J + 1 right after the FIRST update of pivot J's
  // batch has made that column final: the reciprocal's dependent chain runs under the rest of the batch); rnext: the same for
  // the next pivot, on return.
  template <int PI, int J, int NR>
  MTG_HD void pivot(V (&BL)[F], V (&BR)[F], V& rkeep, V r, V& rnext) {
    constexpr int L = PI * kGroup + J;
    const P mine = O::pand(PI == 0 ? in_g0 : in_g1, is_i[J]);
    flag_singular = O::por(flag_singular, O::pand(mine, O::not_gt0(BL[J])));
    rkeep = O::sel(mine, r, rkeep);
    O::settle(r);                          // (device: the DPP read below must not follow r's VALU write within two wait states)
    V m = O::splat(0.0);
    O::template fmac_bcast<L>(m, r, BL[J]);   // m = (1 / pivot) * own element of the pivot column
    m = O::sel(mine, O::splat(0.0), O::neg(m));
    rnext = r;
    if constexpr (J + 1 < F) {
      O::template fmac_bcast<L>(BL[J + 1], BL[J + 1], m);
      rnext = O::rcp(BL[J + 1]);             // the next pivot's column is final now
    }
#pragma unroll
    for (int c = J + 2; c < F; ++c) O::template fmac_bcast<L>(BL[c], BL[c], m);
#pragma unroll
    for (int c = 0; c < NR; ++c) O::template fmac_bcast<L>(BR[c], BR[c], m);
#pragma unroll
    for (int dm = 0; dm < D; ++dm) O::template fmac_bcast<L>(R[dm], R[dm], m);
  }
  template <int PI, int NR, int J = 0>
  MTG_HD void eliminate(V (&BL)[F], V (&BR)[F], V& rkeep, V r) {
    if constexpr (J < F) {
      V rnext;
      pivot<PI, J, NR>(BL, BR, rkeep, r, rnext);
      eliminate<PI, NR, J + 1>(BL, BR, rkeep, rnext);
    }
  }
  template <int PI, int NR>
  MTG_HD void eliminate(V (&BL)[F], V (&BR)[F], V& rkeep) { eliminate<PI, NR, 0>(BL, BR, rkeep, O::rcp(BL[0])); }

  // ---- one forward step.  PI: parity of the step (LEFT vertex rows = group PI, its columns = block PI).
  // FIRST: the left vertex is the trajectory's end vertex (all h derivatives fixed: no rows, right-hand-side terms only).
  // T: segment time; sign: +1 / -1 (chain direction); fixl: the left vertex's fixed values (FIRST: all h per dimension, else
  // [dm][0] = its position); posr: the right vertex's position.  save(k, value): keeps value k (0 .. F + D - 1) of this step's
  // back-substitution data (-G row, g) -- meaningful in the LEFT lanes.
  template <int PI, bool FIRST, class Save>
  MTG_HD void forward_step(V T, double sign, int deriv, const V (&fixl)[D][H], const V (&posr)[D], Save&& save) {
    V(&BL)[F] = PI == 0 ? B0 : B1;
    V(&BR)[F] = PI == 0 ? B1 : B0;
    const V(&c)[N] = PI == 0 ? tabE : tabO;
    V s[H], tinv, base;
    scales(T, sign, deriv, s, tinv, base);
    const V bs = O::mul(base, own_of(s));
    // rows: block columns and right-hand sides
#pragma unroll
    for (int q = 1; q <= F; ++q) {
      const V t = O::mul(bs, s[q]);
      if constexpr (FIRST) BL[q - 1] = O::splat(0.0);
      else BL[q - 1] = O::fma(t, c[q], BL[q - 1]);
      BR[q - 1] = O::mul(t, c[H + q]);
    }
    const V nbs = O::neg(bs);
#pragma unroll
    for (int dm = 0; dm < D; ++dm) {
      V acc = O::mul(c[H], posr[dm]);
      if constexpr (FIRST) {
#pragma unroll
        for (int k = 0; k < H; ++k) acc = O::fma(O::mul(c[k], s[k]), fixl[dm][k], acc);
        R[dm] = O::mul(nbs, acc);
      } else {
        acc = O::fma(c[0], fixl[dm][0], acc);
        R[dm] = O::fma(nbs, acc, R[dm]);
      }
    }
    if constexpr (!FIRST) {
      V rkeep = O::splat(0.0);
      O::settle_rows(BL, BR, R);
      eliminate<PI, F>(BL, BR, rkeep);
      const V nr = O::neg(rkeep);
#pragma unroll
      for (int q = 0; q < F; ++q) save(q, O::mul(BR[q], nr));          // -G row
#pragma unroll
      for (int dm = 0; dm < D; ++dm) save(F + dm, O::mul(R[dm], rkeep));   // g
    }
    // the LEFT lanes' row is finished (saved): they are the next step's RIGHT rows and start from zero
    const V keep = PI == 0 ? keepE : keepO;
#pragma unroll
    for (int q = 0; q < F; ++q) BR[q] = O::mul(BR[q], keep);
#pragma unroll
    for (int dm = 0; dm < D; ++dm) R[dm] = O::mul(R[dm], keep);
  }

  // ---- middle vertex: the forward sweeps of both directions end with parity 0, i.e. the middle vertex's rows are the group-1
  // lanes, block B1.  other[k]: the other direction's row (k < F: block, then right-hand sides).  Leaves x = solution in group 1.
  MTG_HD void solve_middle(const V (&other)[F + D]) {
#pragma unroll
    for (int q = 0; q < F; ++q) B1[q] = O::add(B1[q], other[q]);
#pragma unroll
    for (int dm = 0; dm < D; ++dm) R[dm] = O::add(R[dm], other[F + dm]);
    V rkeep = O::splat(0.0);
    O::settle_rows(B1, B0, R);
    eliminate<1, 0>(B1, B0, rkeep);
#pragma unroll
    for (int dm = 0; dm < D; ++dm) x[dm] = O::mul(R[dm], rkeep);
  }

  // ---- one backward step (parity PI as in the forward step): back-substitution of the LEFT vertex, then the coefficients of the
  // step's segment.  load(k): value k of the step's saved data (LEFT lanes).  DIR > 0: the LEFT vertex is the segment's start.
  // store(values[D]): coefficient `lane` of every dimension (lanes < N).
  template <int PI, bool FIRST, int DIR, class Load, class Store>
  MTG_HD void backward_step(V T, int deriv, const V (&fixl)[D][H], const V (&posr)[D], Load&& load, Store&& store) {
    constexpr int GL = PI, GR = 1 - PI;
    if constexpr (!FIRST) {
      V xl[D];
#pragma unroll
      for (int dm = 0; dm < D; ++dm) xl[dm] = load(F + dm);
      O::settle_vec(x);
#pragma unroll
      for (int q = 0; q < F; ++q) {
        const V ng = load(q);
        pivot_unused();
#pragma unroll
        for (int dm = 0; dm < D; ++dm) fmac_from<GR>(q, xl[dm], x[dm], ng);
      }
      const P left = GL == 0 ? in_g0 : in_g1;
#pragma unroll
      for (int dm = 0; dm < D; ++dm) x[dm] = O::sel(left, xl[dm], x[dm]);
    }
    // coefficients: c_i = T^-i sum_k A(1)^-1[i][k] (S d)_k,  d = [derivatives at the segment start ; at its end]
    V s[H], tinv, base;
    scales(T, 1.0, deriv, s, tinv, base);   // (flags were raised by the forward step already)
    const V so = own_of(s);
    V sx[D];
#pragma unroll
    for (int dm = 0; dm < D; ++dm) sx[dm] = O::mul(so, x[dm]);
    O::settle_vec(sx);
    constexpr int KL = DIR > 0 ? 0 : H, KR = DIR > 0 ? H : 0;   // table columns of the LEFT / RIGHT vertex's derivatives
    // T^-lane (the coefficient's power), once for all dimensions
    V tp = O::sel(bit[0], tinv, O::splat(1.0));
    const V t2 = O::mul(tinv, tinv);
    tp = O::mul(tp, O::sel(bit[1], t2, O::splat(1.0)));
    const V t4 = O::mul(t2, t2);
    tp = O::mul(tp, O::sel(bit[2], t4, O::splat(1.0)));
    if constexpr (N > 8) {
      const V t8 = O::mul(t4, t4);
      tp = O::mul(tp, O::sel(bit[3], t8, O::splat(1.0)));
    }
    // (source lane outermost, the D accumulators innermost: the broadcast-FMAs are volatile asm and issue in program order --
    // D independent chains interleave instead of one dependent chain per dimension)
    V cf[D];
#pragma unroll
    for (int dm = 0; dm < D; ++dm) {
      cf[dm] = O::mul(ai[KR], posr[dm]);
      if constexpr (FIRST) {
#pragma unroll
        for (int k = 0; k < H; ++k) cf[dm] = O::fma(O::mul(ai[KL + k], s[k]), fixl[dm][k], cf[dm]);
      } else {
        cf[dm] = O::fma(ai[KL], fixl[dm][0], cf[dm]);
      }
    }
#pragma unroll
    for (int q = 0; q < F; ++q) {
      if constexpr (!FIRST) {
#pragma unroll
        for (int dm = 0; dm < D; ++dm) fmac_from<GL>(q, cf[dm], sx[dm], ai[KL + q + 1]);
      }
#pragma unroll
      for (int dm = 0; dm < D; ++dm) fmac_from<GR>(q, cf[dm], sx[dm], ai[KR + q + 1]);
    }
#pragma unroll
    for (int dm = 0; dm < D; ++dm) cf[dm] = O::mul(cf[dm], tp);
    store(cf);
  }

 private:
  MTG_HD void pivot_unused() {}
  // acc += bcast(src from lane G * 8 + q) * mul, q a loop variable of an unrolled loop (compile-time after unrolling is not
  // enough for an asm immediate: dispatch over the F possible lanes)
  template <int G>
  MTG_HD void fmac_from(int q, V& acc, V src, V mul) {
    switch (q) {
      case 0: O::template fmac_bcast<G * kGroup + 0>(acc, src, mul); break;
      case 1: if constexpr (F > 1) O::template fmac_bcast<G * kGroup + 1>(acc, src, mul); break;
      case 2: if constexpr (F > 2) O::template fmac_bcast<G * kGroup + 2>(acc, src, mul); break;
      case 3: if constexpr (F > 3) O::template fmac_bcast<G * kGroup + 3>(acc, src, mul); break;
      case 4: if constexpr (F > 4) O::template fmac_bcast<G * kGroup + 4>(acc, src, mul); break;
      case 5: if constexpr (F > 5) O::template fmac_bcast<G * kGroup + 5>(acc, src, mul); break;
      case 6: if constexpr (F > 6) O::template fmac_bcast<G * kGroup + 6>(acc, src, mul); break;
      default: if constexpr (F > 7) O::template fmac_bcast<G * kGroup + 7>(acc, src, mul); break;
    }
  }
};

// ---- half-chain drivers, generic over the back end and an IO policy ---------------------------------------------------
// IO (one 16-lane row = one trajectory-half):
//   V time(int seg); V fixed(int dm, int col);                  inputs (the same value in every lane of the row)
//   void save(int j, int k, V); V load(int j, int k);           step storage
//   void store(int seg, const V (&)[D]);                        coefficient `lane` of every dimension of segment seg
// Standard shapes: end vertices fix all h derivatives (columns 0 .. h-1 and the last h), interior vertices the position.
template <int H>
MTG_HD int coop_col_of_vertex(int K, int v) { return v == 0 ? 0 : (v == K ? H + (K - 1) : H + (v - 1)); }

// Forward sweep of a half-chain of kc steps.  On return `posm` holds the middle vertex's position (the backward sweep starts
// from it) -- a step's right vertex is the next step's left vertex, so every position is loaded once per sweep.
template <class O, int H, int D, int DIR, class IO>
MTG_HD void coop_forward(Coop<O, H, D>& cp, IO& io, int K, int kc, int deriv, typename O::V (&posm)[D]) {
  using V = typename O::V;
  // parity schedule: the LAST step has parity 0 (the middle vertex's rows end up in group 1 / block B1 for both directions)
  const int p0 = (kc - 1) & 1;
  V fixl[D][H], posr[D];
  {   // step 0: the left vertex is the trajectory's end vertex, all h derivatives fixed
    const int cl = coop_col_of_vertex<H>(K, mtg_vl<DIR>(K, 0)), cr = coop_col_of_vertex<H>(K, mtg_vr<DIR>(K, 0));
    const V T = io.time(mtg_seg<DIR>(K, 0));
#pragma unroll
    for (int dm = 0; dm < D; ++dm) {
      posr[dm] = io.fixed(dm, cr);
#pragma unroll
      for (int k = 0; k < H; ++k) fixl[dm][k] = io.fixed(dm, cl + k);
    }
    auto save = [&](int, V) {};
    if (p0 == 0) cp.template forward_step<0, true>(T, (double)DIR, deriv, fixl, posr, save);
    else cp.template forward_step<1, true>(T, (double)DIR, deriv, fixl, posr, save);
  }
  // the inputs of step j + 1 (segment time, right vertex position) are requested while step j is eliminated: a step is ~1000
  // cycles of dependent arithmetic, a load from L2 / HBM 500-900 -- and the form exists for launches too small to hide it
  V Tn = O::splat(1.0), posn[D];
#pragma unroll
  for (int dm = 0; dm < D; ++dm) posn[dm] = O::splat(0.0);
  if (kc > 1) {
    Tn = io.time(mtg_seg<DIR>(K, 1));
#pragma unroll
    for (int dm = 0; dm < D; ++dm) posn[dm] = io.fixed(dm, coop_col_of_vertex<H>(K, mtg_vr<DIR>(K, 1)));
  }
  for (int j = 1; j < kc; ++j) {
    const V T = Tn;
#pragma unroll
    for (int dm = 0; dm < D; ++dm) {
      fixl[dm][0] = posr[dm];
      posr[dm] = posn[dm];
    }
    if (j + 1 < kc) {
      Tn = io.time(mtg_seg<DIR>(K, j + 1));
      const int cn = coop_col_of_vertex<H>(K, mtg_vr<DIR>(K, j + 1));
#pragma unroll
      for (int dm = 0; dm < D; ++dm) posn[dm] = io.fixed(dm, cn);
    }
    auto save = [&](int k, V v) { io.save(j, k, v); };
    if (((j + p0) & 1) == 0) cp.template forward_step<0, false>(T, (double)DIR, deriv, fixl, posr, save);
    else cp.template forward_step<1, false>(T, (double)DIR, deriv, fixl, posr, save);
  }
#pragma unroll
  for (int dm = 0; dm < D; ++dm) posm[dm] = posr[dm];
}

template <class O, int H, int D, int DIR, class IO>
MTG_HD void coop_backward(Coop<O, H, D>& cp, IO& io, int K, int kc, int deriv, const typename O::V (&posm)[D]) {
  using V = typename O::V;
  const int p0 = (kc - 1) & 1;
  V fixl[D][H], posr[D];
#pragma unroll
  for (int dm = 0; dm < D; ++dm) posr[dm] = posm[dm];
  // (inputs one step ahead, as in the forward sweep; the last prefetch is step 0's time -- its end-vertex values follow below)
  V Tn = io.time(mtg_seg<DIR>(K, kc - 1)), posn[D];
#pragma unroll
  for (int dm = 0; dm < D; ++dm) posn[dm] = kc > 1 ? io.fixed(dm, coop_col_of_vertex<H>(K, mtg_vl<DIR>(K, kc - 1))) : O::splat(0.0);
  for (int j = kc - 1; j >= 1; --j) {
    const int seg = mtg_seg<DIR>(K, j);
    const V T = Tn;
#pragma unroll
    for (int dm = 0; dm < D; ++dm) fixl[dm][0] = posn[dm];
    Tn = io.time(mtg_seg<DIR>(K, j - 1));
    if (j > 1) {
      const int cn = coop_col_of_vertex<H>(K, mtg_vl<DIR>(K, j - 1));
#pragma unroll
      for (int dm = 0; dm < D; ++dm) posn[dm] = io.fixed(dm, cn);
    }
    auto load = [&](int k) { return io.load(j, k); };
    auto store = [&](const V (&v)[D]) { io.store(seg, v); };
    if (((j + p0) & 1) == 0) cp.template backward_step<0, false, DIR>(T, deriv, fixl, posr, load, store);
    else cp.template backward_step<1, false, DIR>(T, deriv, fixl, posr, load, store);
#pragma unroll
    for (int dm = 0; dm < D; ++dm) posr[dm] = fixl[dm][0];
  }
  {   // step 0: the trajectory's end vertex
    const int seg = mtg_seg<DIR>(K, 0);
    const V T = Tn;
    const int cl = coop_col_of_vertex<H>(K, mtg_vl<DIR>(K, 0));
#pragma unroll
    for (int dm = 0; dm < D; ++dm) {
#pragma unroll
      for (int k = 0; k < H; ++k) fixl[dm][k] = io.fixed(dm, cl + k);
    }
    auto load = [&](int) { return O::splat(0.0); };
    auto store = [&](const V (&v)[D]) { io.store(seg, v); };
    if (p0 == 0) cp.template backward_step<0, true, DIR>(T, deriv, fixl, posr, load, store);
    else cp.template backward_step<1, true, DIR>(T, deriv, fixl, posr, load, store);
  }
}

}  // namespace mtgc
#endif  // MTG_COOP_H_
