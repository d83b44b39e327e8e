// mtg_dimlane_rt.h -- the dimension-in-lane form for a RUN-TIME number of segments: one kernel body per polynomial order
// for every chain length K >= 2 of the standard shapes (trajectory ends fully fixed or any compile-time end masks, one
// compile-time interior mask), where mtg_dimlane.h has one fully unrolled body per (N, K).
//
// Same per-lane algorithm and the same per-step arithmetic (mtg_lane.h: mtg_fwd_step_core, mtg_bwd_backsub, mtg_recover --
// bit-identical results), same lane mapping (lane = dimension * TPW + trajectory, two waves = the two chain directions of a
// tile) and the same whole-sector coefficient output (MtgSlabOutRt, mtg_slab.h).  What is organised differently is where a
// half-chain of kc = K/2 steps keeps its back-substitution data (G_v, g_v):
//   * step 0 (the trajectory end) has none when that vertex is fully fixed -- nothing is stored for it;
//   * the LAST R steps (the ones back-substitution consumes first) stay in registers, G shared between the dimension lanes
//     (MtgCfg::kRegShared: every lane keeps each DL-th element and fetches its siblings' with ds_bpermute) -- an unrolled
//     tail whose positions are skipped wave-uniformly when the chain is shorter;
//   * the L steps before them live in the wave's LDS step area, everything earlier in the lane-coalesced global workspace
//     (shared-G rows as in mtg_dimlane.h) -- a real loop, the code does not grow with K.
// Segment times and fixed values are loaded when a step needs them, one step ahead of their use and ahead of the previous
// step's stores (loads and stores retire through one in-order counter), instead of being preloaded into registers: that is
// what limited the static long-chain variants (K = 32: 34 doubles per lane).
#ifndef MTG_DIMLANE_RT_H_
#define MTG_DIMLANE_RT_H_
#ifndef MTG_RT_INPUT_DEPTH
#define MTG_RT_INPUT_DEPTH 3      // steps the on-demand input loads run ahead of their use (beyond the next step)
#endif
// (Round 6, built, measured and removed: the back-substitution rows of a head step requested TWO steps ahead through a second
// register set -- paid with 0 / 2 / 2 register steps for N = 10 / 8 / 12.  K = 100 at B = 100k: 873 / 1338 / 1965 us against
// 843 / 1283 / 1791 us one step ahead (N = 8 / 10 / 12; profiles/r06g_other_k_rows_two_steps_ahead_not_adopted.jsonl): the lead of
// the workspace loads is not what the head steps wait for.)
#include "mtg_kernels.h"

// C: rolled configuration (KT_ < 0) with D == 1, DLW = DL, RS = 1.  R: register steps, L: LDS steps.
template <class C, int R>
struct MtgTailRt {
  double Gs[R][C::GROWS];   // this dimension lane's share of G of the tail steps
  double g[R][1][C::H];     // g of the tail steps (per lane)
};

template <class C>
__host__ __device__ constexpr size_t mtg_rt_step_bytes() { return (size_t)C::WSE * kWave * sizeof(double); }

#if defined(__HIP_DEVICE_COMPILE__)   // (device-only pieces: LDS pointers, ds_bpermute)
// Where head step j (1 <= j < nh; nh = kc - R) keeps its rows: the last L of them in the wave's LDS step area, the others
// in the global workspace (slot j - 1).
template <class C, int L>
struct MtgRtStore {
  double* wsl;              // this lane's column of the global workspace
  long long ws_stride;      // elements between workspace rows
  unsigned lds_col;         // LDS byte address of this lane's column in the wave's step area (row stride 64 doubles)
  long long share;          // lane offset to the trajectory's dimension-0 lane (both areas are lane-coalesced)
  int nh;                   // head steps are j = 1 .. nh - 1
  __device__ __forceinline__ bool in_lds(int j) const { return L > 0 && j >= nh - L; }
  __device__ __forceinline__ mtg_lds_double* lds_ptr(int j) const {
    return (mtg_lds_double*)(size_t)(lds_col + (unsigned)(j - (nh - L)) * (unsigned)mtg_rt_step_bytes<C>());
  }
  __device__ __forceinline__ mtg_glb_double* ws_ptr(int j) const { return mtg_glb(wsl + (long long)(j - 1) * C::WSE * ws_stride); }   // (global address space: see mtg_glb)
};

// The trajectory index with an opaque zero added that "depends" on `dep`: a load addressed through it cannot be issued before
// `dep` exists.  The unrolled tail positions' input loads have addresses that are known up front; without the tie the
// scheduler hoists all of them (and their 64-bit address arithmetic) in front of the tail and keeps them live -- measured:
// ~50 instead of ~20 registers per register step.
__device__ __forceinline__ long long mtg_rt_tie(long long b, double dep) {
  int z = 0;
  asm volatile("" : "+v"(z) : "v"(dep));
  return b + z;
}

template <class C, int R, int L, int DIR>
__device__ __forceinline__ void mtg_lane_forward_rt(const MtgParams& P, long long b, int kc, MtgLane<C>& ln, MtgTailRt<C, R>& tail,
                                                    const MtgRtStore<C, L>& st) {
  constexpr int H = C::H;
  static_assert(C::kRolled && C::D == 1 && C::kRegShared, "run-time-K dimension-in-lane body: rolled one-dimension configuration, shared G");
  constexpr int M0 = DIR > 0 ? C::MS : C::ME;
  static_assert(C::popc(M0) == H, "run-time-K body: the trajectory ends are fully fixed (no back-substitution data for step 0)");
  const int K = P.K;
  mtg_lane_reset<C, DIR>(ln);
  // Issue order per step: inputs of step j + 1 + PD, arithmetic of step j, then step j's back-substitution stores.  The segment
  // time and the fixed values of the right vertex are requested PD + 1 steps ahead (a ring of PD entries: 1 + popc(MI) doubles
  // each): one step of arithmetic (~0.7 us) does not cover the latency of a load that goes to HBM behind a store stream.
  constexpr int PD = MTG_RT_INPUT_DEPTH;
  double T_cur = mtg_step_time<C, DIR>(P, b, 0, ln);
  double fl[1][H], fr[1][H], fn[1][H];
  double Tq[PD], fq[PD][1][H];
  mtg_load_vals<C, DIR>(P, b, mtg_vl<DIR>(K, 0), M0, ln, fl);
  mtg_load_vals<C, DIR>(P, b, mtg_vr<DIR>(K, 0), C::MI, ln, fr);
#pragma unroll
  for (int i = 0; i < PD; ++i) {
    const int jn = i + 1 < kc ? i + 1 : kc - 1;
    Tq[i] = mtg_step_time<C, DIR>(P, b, jn, ln);
    mtg_load_vals<C, DIR>(P, b, mtg_vr<DIR>(K, jn), C::MI, ln, fq[i]);
  }
  auto prefetch = [&](int j, double& T_nxt) {       // segment time and right vertex of step j + 1 + PD (chain end: harmless reload)
    const int jn = j + 1 + PD < kc ? j + 1 + PD : kc - 1;
    const long long bt = mtg_rt_tie(b, ln.Sc[H - 1][H - 1]);     // not before the previous step's arithmetic
    T_nxt = mtg_step_time<C, DIR>(P, bt, jn, ln);
    mtg_load_vals<C, DIR>(P, bt, mtg_vr<DIR>(K, jn), C::MI, ln, fn);
  };
  auto shift = [&](double T_nxt) {
    T_cur = Tq[0];
#pragma unroll
    for (int p = 0; p < H; ++p) { fl[0][p] = fr[0][p]; fr[0][p] = fq[0][0][p]; }
#pragma unroll
    for (int i = 0; i + 1 < PD; ++i) {
      Tq[i] = Tq[i + 1];
#pragma unroll
      for (int p = 0; p < H; ++p) fq[i][0][p] = fq[i + 1][0][p];
    }
    Tq[PD - 1] = T_nxt;
#pragma unroll
    for (int p = 0; p < H; ++p) fq[PD - 1][0][p] = fn[0][p];
  };
  {   // step 0: the trajectory end (fully fixed: nothing to keep)
    double G[H][H], g[1][H], T_nxt;
    prefetch(0, T_nxt);
    mtg_fwd_step_core<C, DIR>(P, M0, C::MI, ln, T_cur, fl, fr, G, g);
    shift(T_nxt);
  }
  const int nh = st.nh;
  for (int j = 1; j < nh; ++j) {                    // head: rows to LDS / workspace
    double G[H][H], g[1][H], T_nxt;
    prefetch(j, T_nxt);
    mtg_fwd_step_core<C, DIR>(P, C::MI, C::MI, ln, T_cur, fl, fr, G, g);
    if (st.in_lds(j)) mtg_ws_store_shared<C>(st.lds_ptr(j), 64, P.dim0, G, g, C::MI, C::MI);
    else mtg_ws_store_shared<C>(st.ws_ptr(j), st.ws_stride, P.dim0, G, g, C::MI, C::MI);
    shift(T_nxt);
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {                     // tail: registers; positions in front of the chain are skipped
    const int j = kc - R + r;
    if (j >= 1) {
      double G[H][H], T_nxt;
      prefetch(j, T_nxt);
      mtg_fwd_step_core<C, DIR>(P, C::MI, C::MI, ln, T_cur, fl, fr, G, tail.g[r]);
      mtg_rs_pack<C>(P.dim0, G, C::MI, C::MI, tail.Gs[r]);
      shift(T_nxt);
    }
  }
  mtg_unscale_carried<C, DIR>(P, ln);      // (scaled-variable chain: plain Schur complement / right-hand side for the middle vertex)
}

template <class C, int R, int L, int DIR, class IO>
__device__ __forceinline__ void mtg_lane_finish_rt(const MtgParams& P, long long b, int kc, MtgLane<C>& ln, MtgTailRt<C, R>& tail,
                                                   const MtgRtStore<C, L>& st, const double* other, int stride, IO& io, bool active) {
  constexpr int H = C::H;
  constexpr int M0 = DIR > 0 ? C::MS : C::ME;
  const int K = P.K;
  const int vm = (K + 1) / 2;
  double xr[1][H];
  mtg_solve_mid<C, DIR>(P, b, ln, vm, C::MI, other, stride, xr);
  int perm[C::DLW];      // byte addresses (4 * lane) of this trajectory's dimension lanes, for ds_bpermute
#pragma unroll
  for (int k = 0; k < C::DLW; ++k) perm[k] = 4 * ((int)(threadIdx.x & 63) + (int)st.share + k * (64 / C::DLW));
  // Per step: back-substitution with the data requested one step earlier, then the request for step j - 1 (its G / g, its
  // segment time, the fixed values of its left vertex) -- BEFORE step j recovers its segment and streams out the previous
  // one --, then recovery.
  double Gw[H][H], gw[1][H], fl[1][H], T_cur = 0.0;
  double tie_on = 0.0;                 // (set to the latest back-substitution result before every request)
  // Segment time and left-vertex fixed values: requested PD + 1 steps ahead (ring of PD entries, as in the forward phase); the
  // fully fixed end vertex (step 0, mask M0) has its own H values f0, requested when the ring reaches step 0.
  constexpr int PD = MTG_RT_INPUT_DEPTH;
  double Tq[PD], fq[PD][1][H], f0[1][H];
#pragma unroll
  for (int p = 0; p < H; ++p) f0[0][p] = 0.0;
  auto fetch = [&](int j, double& T, double (&f)[1][H]) {      // inputs of step j (j < 0: nothing left to request)
    const long long bt = mtg_rt_tie(b, tie_on);
    T = mtg_step_time<C, DIR>(P, bt, j > 0 ? j : 0, ln);
    mtg_load_vals<C, DIR>(P, bt, mtg_vl<DIR>(K, j > 0 ? j : 1 < kc ? 1 : 0), C::MI, ln, f);      // (j <= 0: harmless reload, unused)
    if (j == 0) mtg_load_vals<C, DIR>(P, bt, mtg_vl<DIR>(K, 0), M0, ln, f0);
  };
  auto request_inputs = [&](int j) {   // step j becomes the current one: its inputs leave the ring, step j - PD's are requested
    double Tn, fn_[1][H];
    fetch(j - PD, Tn, fn_);
    T_cur = Tq[0];
#pragma unroll
    for (int p = 0; p < H; ++p) fl[0][p] = j == 0 ? f0[0][p] : fq[0][0][p];
#pragma unroll
    for (int i = 0; i + 1 < PD; ++i) {
      Tq[i] = Tq[i + 1];
#pragma unroll
      for (int p = 0; p < H; ++p) fq[i][0][p] = fq[i + 1][0][p];
    }
    Tq[PD - 1] = Tn;
#pragma unroll
    for (int p = 0; p < H; ++p) fq[PD - 1][0][p] = fn_[0][p];
  };
  auto request_head = [&](int j) {     // j < nh: a head step (rows in LDS / workspace) or step 0 (no rows)
    if (j >= 1) {
      if (st.in_lds(j)) mtg_ws_load_shared<C>(st.lds_ptr(j), 64, st.share, Gw, gw, C::MI, C::MI);
      else mtg_ws_load_shared<C>((const mtg_glb_double*)st.ws_ptr(j), st.ws_stride, st.share, Gw, gw, C::MI, C::MI);
    }
    request_inputs(j);
  };
  const int nh = st.nh;
  // the first step processed: j = kc - 1 (a tail step when kc >= 2, else step 0)
  tie_on = xr[0][H - 1];
#pragma unroll
  for (int i = 0; i < PD; ++i) fetch(kc - 1 - i, Tq[i], fq[i]);     // ring <- steps kc - 1, ..., kc - PD
  if (kc - 1 >= 1) {
    if (kc - 1 >= nh) { /* tail position R - 1: its G is unpacked in the loop below */ request_inputs(kc - 1); }
    else request_head(kc - 1);     // (R == 0 only)
  } else {
    request_inputs(0);
  }
#pragma unroll
  for (int r = R - 1; r >= 0; --r) {
    const int j = kc - R + r;
    if (j >= 1) {
      double xl[1][H];
      const double T_use = T_cur;
      {   // (tied to the previous step's result: the ds_bpermute fetches of ALL tail positions depend only on forward-phase
          // values, and hoisted to the start of the backward phase each keeps a full G live)
        int zt = 0, pt[C::DLW];
        asm volatile("" : "+v"(zt) : "v"(xr[0][H - 1]));
#pragma unroll
        for (int k = 0; k < C::DLW; ++k) pt[k] = perm[k] + zt;
        mtg_rs_unpack<C>(pt, tail.Gs[r], C::MI, C::MI, Gw);
      }
      MtgScaledEnds<C> ye;
      if constexpr (C::kFS) mtg_bwd_backsub_fs<C, DIR>(P, C::MI, C::MI, T_use, fl, Gw, tail.g[r], xr, xl, ye);
      else mtg_bwd_backsub<C, DIR>(C::MI, C::MI, T_use, fl, Gw, tail.g[r], xr, xl, ye);
      tie_on = xl[0][H - 1];
      if (r > 0 && j - 1 >= 1) request_inputs(j - 1);    // next: tail position r - 1 (its G comes from registers)
      else request_head(j - 1);                          // next: the last head step, or step 0
      mtg_bwd_finish<C, DIR, 0>(P, b, j, C::MI, T_use, xl, xr, io, ye);
    }
  }
  for (int j = (nh - 1 < kc - 1 ? nh - 1 : kc - 1); j >= 1; --j) {     // head steps nh - 1 .. 1
    double xl[1][H];
    const double T_use = T_cur;
    MtgScaledEnds<C> ye;
    if constexpr (C::kFS) mtg_bwd_backsub_fs<C, DIR>(P, C::MI, C::MI, T_use, fl, Gw, gw, xr, xl, ye);
    else mtg_bwd_backsub<C, DIR>(C::MI, C::MI, T_use, fl, Gw, gw, xr, xl, ye);
    tie_on = xl[0][H - 1];
    request_head(j - 1);
    mtg_bwd_finish<C, DIR, 0>(P, b, j, C::MI, T_use, xl, xr, io, ye);
  }
  {   // step 0: every slot of the end vertex is fixed
    double xl[1][H];
#pragma unroll
    for (int p = 0; p < H; ++p) xl[0][p] = fl[0][p];
    mtg_bwd_finish<C, DIR, 0>(P, b, 0, M0, T_cur, xl, xr, io);
  }
  io.flush(P);
  if (ln.flags && active) {
    atomicOr(P.status, ln.flags);
    if (P.tstatus != nullptr) atomicOr(P.tstatus + b, ln.flags);
  }
}

#endif  // __HIP_DEVICE_COMPILE__

// One workgroup = two waves = the two chain directions of one tile of TPW = 64 / DL trajectories; persistent over the tiles.
// Dynamic LDS: [slab A][slab B][steps A][steps B]; the exchange buffer a direction publishes lives in the OTHER direction's
// slab (read before that direction writes its first coefficient row).
template <class C, int DL>
__host__ __device__ constexpr size_t mtg_rt_half_bytes() {
  constexpr int fmid = C::H - C::popc(C::MI);
  constexpr size_t xch = (size_t)(fmid * (fmid + 1) / 2 + fmid) * kWave * sizeof(double);
  constexpr size_t slab = ((size_t)MtgSlabOutRt<C::N, DL, 1, 0, false>::TPW * MtgSlabOutRt<C::N, DL, 1, 0, false>::ROWB + 15) / 16 * 16;
  return slab > xch ? slab : xch;
}
template <class C, int DL, int L>
__host__ __device__ constexpr size_t mtg_rt_lds_bytes() { return 2 * mtg_rt_half_bytes<C, DL>() + 2 * (size_t)L * mtg_rt_step_bytes<C>(); }

// PHASE: instantiation for chain lengths whose K * (DL N 8)-byte pieces are not a multiple of 64 bytes (MtgSlabOutRt)
template <class C, int DL, int R, int L, int AUX, bool PHASE>
__global__ __launch_bounds__(2 * kWave, 1) void mtg_solve_dl_rt_kernel(const double* __restrict__ times, const double* __restrict__ dfix,
                                                                     double* __restrict__ coeffs, int* status, int* traj_status,
                                                                     int B, int K, int ntiles, int aos, double* ws) {
  static_assert(C::DLW == DL && DL >= 1 && DL <= 4, "lanes per trajectory");
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  constexpr int TPW = kWave / DL;
  const int lane = threadIdx.x & (kWave - 1);
  const int dir = threadIdx.x >> 6;      // wave-uniform
  int d = lane / TPW, t = lane - d * TPW;
  const bool dup = d >= DL;              // surplus lanes (64 % DL) duplicate the last lane's work, outputs suppressed
  if (dup) { d = DL - 1; t = TPW - 1; }
  const int n_fixed = C::popc(C::MS) + (K - 1) * C::popc(C::MI) + C::popc(C::ME);
  MtgParams P;
  // inputs: canonical SoA (times[K][B], d_fixed[DL][n_fixed][B]) or, aos, canonical AoS (times[B][K], d_fixed[B][DL][n_fixed])
  P.times = times; P.ts_b = aos ? K : 1; P.ts_k = aos ? 1 : B;
  P.dfix = dfix; P.fs_b = aos ? DL * n_fixed : 1; P.fs_c = aos ? 1 : B; P.fs_d = aos ? n_fixed : (long long)n_fixed * B;
  P.coeffs = coeffs;
  P.dfree = nullptr; P.ps_b = P.ps_d = P.ps_c = 0;
  P.cost = nullptr; P.ws = ws; P.ws_stride = (long long)gridDim.x * (2 * kWave);
  P.ws_share = (long long)t - lane;
  P.lds_steps = 0;
  P.status = status; P.tstatus = traj_status;
  P.vmask = nullptr; P.offF = nullptr; P.offP = nullptr;
  P.B = B; P.K = K; P.Dtot = DL; P.dim0 = d;     // dim0 is a per-lane value here
  P.deriv = C::DV; P.h1off = C::H1OFF; P.ainvoff = C::AINVOFF;
  P.pert_on = 0; P.pert_seg = -1; P.pert_tpv = 1; P.pert_h = P.pert_corr = P.pert_lo = 0.0;
  P.rhs = nullptr; P.rh_b = P.rh_d = P.rh_c = 0;
  const int kc = dir == 0 ? (K + 1) / 2 : K / 2;
  constexpr size_t half = mtg_rt_half_bytes<C, DL>();
  char* my_slab = lds_raw + (size_t)dir * half;
  MtgRtStore<C, L> st;
  st.wsl = ws + (size_t)blockIdx.x * (2 * kWave) + threadIdx.x;
  st.ws_stride = P.ws_stride;
  st.lds_col = (unsigned)(size_t)(lds_raw + 2 * half + (size_t)dir * L * mtg_rt_step_bytes<C>()) + (unsigned)lane * 8u;
  st.share = P.ws_share;
  st.nh = kc - R;
  double* mine = reinterpret_cast<double*>(lds_raw + (size_t)(1 - dir) * half) + lane;
  const double* other = reinterpret_cast<const double*>(my_slab) + lane;
  MtgSlabOutRt<C::N, DL, 1, AUX, PHASE> ioA;
  MtgSlabOutRt<C::N, DL, -1, AUX, PHASE> ioB;
  ioA.init(my_slab, lane, t, d, K);
  ioB.init(my_slab, lane, t, d, K);
  MtgLane<C> ln;
  MtgTailRt<C, R> tail;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long b0 = (long long)tile * TPW;
    const long long bl = b0 + t;
    const bool active = bl < B && !dup;
    const long long b = bl < B ? bl : B - 1;
    // The chain length and the per-lane workspace pointer are re-defined opaquely per tile: the input / workspace addresses
    // of the R unrolled tail positions (segment index x stride: 64-bit products) are otherwise invariant in this loop, get
    // hoisted in front of it and stay live throughout (measured: ~50 instead of ~20 registers per register step).
    int kc_t = __builtin_amdgcn_readfirstlane(kc);     // (direction = wave index: uniform, but derived from threadIdx)
    MtgParams Pt = P;
    asm volatile("" : "+s"(kc_t));
    asm volatile("" : "+s"(Pt.K));
    asm volatile("" : "+v"(st.wsl));
    st.nh = kc_t - R;
    if (dir == 0) mtg_lane_forward_rt<C, R, L, 1>(Pt, b, kc_t, ln, tail, st);
    else mtg_lane_forward_rt<C, R, L, -1>(Pt, b, kc_t, ln, tail, st);
    mtg_pack_mid<C>(ln, C::MI, mine, kWave);
    __syncthreads();
    if (dir == 0) {
      ioA.begin_tile(coeffs, b0, B);
      mtg_lane_finish_rt<C, R, L, 1>(Pt, b, kc_t, ln, tail, st, other, kWave, ioA, active);
    } else {
      ioB.begin_tile(coeffs, b0, B);
      mtg_lane_finish_rt<C, R, L, -1>(Pt, b, kc_t, ln, tail, st, other, kWave, ioB, active);
    }
    __syncthreads();
  }
#endif  // __HIP_DEVICE_COMPILE__
}

// table entry of the run-time-K bodies (one per polynomial order / dimension count): mtg_dimlane_rt.hip
struct MtgDimlaneRtEntry {
  int h, ms, mi, me, dv, dl;
  int tpw, r_steps, l_steps;
  size_t lds;
  size_t step_bytes_per_lane;   // workspace bytes per head step and resident lane
  int (*launch)(void* stream, int grid, const double* times, const double* dfix, double* coeffs, int* status, int* traj_status,
                int B, int K, int ntiles, double* ws, int aos);
};
const MtgDimlaneRtEntry* mtg_find_dimlane_rt(int h, int dl, int k, int deriv, const int* mask);
#endif  // MTG_DIMLANE_RT_H_
