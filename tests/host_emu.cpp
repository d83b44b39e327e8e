// host_emu.cpp -- TEST INFRASTRUCTURE ONLY.  Compiles the device lane code of
// mav_trajectory_generation_amd/csrc/mtg_lane.h for the host and runs it sequentially
// (lane A, lane B, exchange, finish) so that the exact kernel arithmetic can be checked
// against the oracle without a GPU.  Never linked into libmtg_hip.so; the product path has
// no CPU fallback.
#include <cstring>
#include <vector>

#include "../mav_trajectory_generation_amd/csrc/mtg_lane.h"

namespace {

template <class C, int WC>
void emu_solve(MtgParams P) {
  constexpr int H = C::H, D = C::D;
  const int K = mtg_nseg<C>(P);
  const int vm = (K + 1) / 2;
  const int mm = C::kRolled ? C::MI : mtg_mask<C>(P, vm);
  const int nslots = mtg_mid_slots<C>(mm);
  const int kc = (K + 1) / 2;
  const size_t E = (size_t)H * H + (size_t)D * H;
  std::vector<double> wsa(kc * E + 1), wsb(kc * E + 1), bufa(nslots + 1), bufb(nslots + 1);
  P.ws_stride = 1;
  for (long long b = 0; b < P.B; ++b) {
    MtgLane<C> la, lb;
    mtg_lane_forward<C, 1>(P, b, la, wsa.data());
    mtg_lane_forward<C, -1>(P, b, lb, wsb.data());
    mtg_pack_mid<C>(la, mm, bufa.data(), 1);
    mtg_pack_mid<C>(lb, mm, bufb.data(), 1);
    MtgDirectOut<C> io;
    io.b = b;
    mtg_lane_finish<C, 1, WC>(P, b, la, wsa.data(), bufb.data(), 1, io, true);
    mtg_lane_finish<C, -1, WC>(P, b, lb, wsb.data(), bufa.data(), 1, io, true);
  }
}

template <class C, int WC>
void emu_update(MtgParams P) {
  for (long long b = 0; b < P.B; ++b) {
    MtgDirectOut<C> io;
    io.b = b;
    mtg_lane_update<C, WC>(P, b, io, true);
  }
}

// Dimension-in-lane form with the SHARED step storage (MtgCfg::DLW = 3) and every step of the half-chain in the lane-coalesced
// workspace (WS = all steps, no LDS / register steps: those branches exist on the device only): three dimension lanes of one
// trajectory share a step's kept matrix -- G, or with MtgCfg::kFS the LDL^T factor of the pivot block -- through a 64-column
// workspace exactly as a wavefront does (lane = dimension * 21 + trajectory, trajectory 0 here).
template <class C>
void emu_solve_shared(MtgParams P) {
  constexpr int DL = 3, TPW = 64 / DL;
  static_assert(C::D == 1 && C::DLW == DL && C::WSJ >= C::KCS && C::LSJ == 0 && !C::kRegShared, "all steps through the shared workspace");
  const int K = C::KT, vm = (K + 1) / 2, mm = mtg_mask<C>(P, vm);
  const int nslots = mtg_mid_slots<C>(mm);
  std::vector<double> wsa((size_t)C::KCS * C::WSE * 64 + 64, 0.0), wsb(wsa.size(), 0.0);
  std::vector<double> bufa[DL], bufb[DL];
  P.ws_stride = 64;
  P.Dtot = DL;
  for (long long b = 0; b < P.B; ++b) {
    MtgLane<C> la[DL], lb[DL];
    for (int d = 0; d < DL; ++d) {
      MtgParams Q = P;
      Q.dim0 = d;
      Q.ws_share = -(long long)(d * TPW);
      mtg_lane_forward<C, 1>(Q, b, la[d], wsa.data() + d * TPW);
      mtg_lane_forward<C, -1>(Q, b, lb[d], wsb.data() + d * TPW);
      bufa[d].assign(nslots + 1, 0.0);
      bufb[d].assign(nslots + 1, 0.0);
      mtg_pack_mid<C>(la[d], mm, bufa[d].data(), 1);
      mtg_pack_mid<C>(lb[d], mm, bufb[d].data(), 1);
    }
    for (int d = 0; d < DL; ++d) {
      MtgParams Q = P;
      Q.dim0 = d;
      Q.ws_share = -(long long)(d * TPW);
      MtgDirectOut<C> io;
      io.b = b;
      mtg_lane_finish<C, 1, 0>(Q, b, la[d], wsa.data() + d * TPW, bufb[d].data(), 1, io, true);
      mtg_lane_finish<C, -1, 0>(Q, b, lb[d], wsb.data() + d * TPW, bufa[d].data(), 1, io, true);
    }
  }
}

using Fn = void (*)(MtgParams);
template <int H, int D> using GenericCfg = MtgCfg<H, D, 0, 0, 0, 0>;

template <int H>
Fn pick_h(int d, bool wc, bool upd) {
#define CASE(DD)                                                                         \
  case DD:                                                                               \
    if (upd) return wc ? (Fn)emu_update<GenericCfg<H, DD>, 1> : (Fn)emu_update<GenericCfg<H, DD>, 0>; \
    return wc ? (Fn)emu_solve<GenericCfg<H, DD>, 3> : (Fn)emu_solve<GenericCfg<H, DD>, 0>;
  switch (d) { CASE(1) CASE(2) CASE(3) CASE(4) }
#undef CASE
  return nullptr;
}
Fn pick(int h, int d, bool wc, bool upd) {
  switch (h) {
    case 1: return pick_h<1>(d, wc, upd);
    case 2: return pick_h<2>(d, wc, upd);
    case 3: return pick_h<3>(d, wc, upd);
    case 4: return pick_h<4>(d, wc, upd);
    case 5: return pick_h<5>(d, wc, upd);
    case 6: return pick_h<6>(d, wc, upd);
  }
  return nullptr;
}

struct StaticEntry { int h, d, k, ms, mi, me, dv; Fn fn[2]; };
#define MTG_STATIC(H, D, K, MS, MI, ME, DV) \
  {H, D, K, MS, MI, ME, DV, {(Fn)emu_solve<MtgCfg<H, D, K, MS, MI, ME, DV>, 0>, (Fn)emu_solve<MtgCfg<H, D, K, MS, MI, ME, DV>, 3>}},
#define MTG_STATIC_HEAVY(H, D, K, MS, MI, ME, DV) MTG_STATIC(H, D, K, MS, MI, ME, DV)
#define MTG_ROLLED(H, D, MS, MI, ME, DV) \
  {H, D, -1, MS, MI, ME, DV, {(Fn)emu_solve<MtgCfg<H, D, -1, MS, MI, ME, DV>, 0>, (Fn)emu_solve<MtgCfg<H, D, -1, MS, MI, ME, DV>, 3>}},
const StaticEntry kStatic[] = {
#include "../mav_trajectory_generation_amd/csrc/mtg_variants.inc"
};
#undef MTG_STATIC
#undef MTG_STATIC_HEAVY
#undef MTG_ROLLED

}  // namespace

// AoS layouts: times[B][K], dfix[B][D][n_fixed], dfree[B][D][n_free], coeffs[B][K][D][N].
// mode: 0 = generic solve, 1 = static variant if one matches (returns -2 if none), 2 = update-from-free,
// 3 = rolled variant if one matches (returns -2 if none).
static const double* g_explicit_rhs = nullptr;   // mtg_emu_run_rhs: [B][D][n_free], added to the right-hand side (generic mode)

extern "C" int mtg_emu_run(int N, int D, int K, int deriv, const int* mask, long long B, const double* times,
                           const double* dfix, double* coeffs, double* dfree, double* cost, int mode, int* status);
// the generic solve with an EXPLICIT right-hand side over the free slots (MtgParams::rhs: the correction solve of
// MTG_FLAG_REFINE) -- same arguments as mtg_emu_run with mode 0
extern "C" int mtg_emu_run_rhs(int N, int D, int K, int deriv, const int* mask, long long B, const double* times, const double* dfix,
                               double* coeffs, double* dfree, const double* rhs, int* status) {
  g_explicit_rhs = rhs;
  const int rc = mtg_emu_run(N, D, K, deriv, mask, B, times, dfix, coeffs, dfree, nullptr, 0, status);
  g_explicit_rhs = nullptr;
  return rc;
}

extern "C" int mtg_emu_run(int N, int D, int K, int deriv, const int* mask, long long B, const double* times,
                           const double* dfix, double* coeffs, double* dfree, double* cost, int mode, int* status) {
  const int H = N / 2;
  std::vector<int> offF(K + 2, 0), offP(K + 2, 0);
  for (int v = 0; v <= K; ++v) {
    const int nf = __builtin_popcount((unsigned)mask[v]);
    offF[v + 1] = offF[v] + nf;
    offP[v + 1] = offP[v] + H - nf;
  }
  const int n_fixed = offF[K + 1], n_free = offP[K + 1];
  MtgParams P;
  std::memset(&P, 0, sizeof(P));
  P.times = times; P.ts_b = K; P.ts_k = 1;
  P.dfix = dfix; P.fs_b = (long long)D * n_fixed; P.fs_d = n_fixed; P.fs_c = 1;
  P.coeffs = coeffs;
  P.dfree = n_free ? dfree : nullptr; P.ps_b = (long long)D * n_free; P.ps_d = n_free; P.ps_c = 1;
  P.cost = cost;
  int st = 0;
  P.status = &st;
  P.vmask = mask; P.offF = offF.data(); P.offP = offP.data();
  P.B = B; P.K = K; P.Dtot = D; P.deriv = deriv;
  P.ainvoff = kAinvLoOff[H];
  P.h1off = kH1Off[H][deriv];
  if (g_explicit_rhs && mode == 0) { P.rhs = g_explicit_rhs; P.rh_b = (long long)D * n_free; P.rh_d = n_free; P.rh_c = 1; }
  if (cost) for (long long b = 0; b < B; ++b) cost[b] = 0.0;
  const bool wc = cost != nullptr || (mode != 2 && n_free > 0 && dfree != nullptr);
  if (mode == 1 || mode == 3) {
    for (const StaticEntry& e : kStatic) {
      if (e.h != H || e.d != D || e.dv != deriv) continue;
      if (mode == 1 ? e.k != K : !(e.k < 0 && K >= 2)) continue;
      bool ok = mask[0] == e.ms && mask[K] == e.me;
      for (int v = 1; v < K && ok; ++v) ok = mask[v] == e.mi;
      if (!ok) continue;
      e.fn[wc ? 1 : 0](P);
      if (status) *status = st;
      return 0;
    }
    return -2;
  }
  for (int dim0 = 0; dim0 < D; dim0 += 4) {
    const int dc = D - dim0 < 4 ? D - dim0 : 4;
    Fn fn = pick(H, dc, wc, mode == 2);
    if (!fn) return -1;
    MtgParams Q = P;
    Q.dim0 = dim0;
    fn(Q);
  }
  if (status) *status = st;
  return 0;
}

// Shared-workspace dimension-in-lane emulation (emu_solve_shared): D = 3, trajectory ends fully fixed, every interior vertex with
// the mask `mi` (1 = position only; 3 / 7 = + velocity / + acceleration, N = 10 only: the run-time-K table's other patterns),
// derivative N / 2 - 1; K in {4, 8}.  Returns -2 for any other shape.  kept_is_factor: MtgCfg::kFS of the build.
extern "C" int mtg_emu_run_shared(int N, int K, int mi, long long B, const double* times, const double* dfix, double* coeffs,
                                  int* status, int* kept_is_factor) {
  const int H = N / 2, D = 3;
  const int n_fixed = 2 * H + (K - 1) * __builtin_popcount((unsigned)mi);
  MtgParams P;
  std::memset(&P, 0, sizeof(P));
  P.times = times; P.ts_b = K; P.ts_k = 1;
  P.dfix = dfix; P.fs_b = (long long)D * n_fixed; P.fs_d = n_fixed; P.fs_c = 1;
  P.coeffs = coeffs;
  int st = 0;
  P.status = &st;
  P.B = B; P.K = K; P.Dtot = D; P.deriv = H - 1;
  P.ainvoff = kAinvLoOff[H];
  P.h1off = kH1Off[H][H - 1];
#define SHARED(HH, KK, MI)                                                                               \
  if (H == HH && K == KK && mi == MI) {                                                                  \
    using C = MtgCfg<HH, 1, KK, (1 << HH) - 1, MI, (1 << HH) - 1, HH - 1, 0, (KK + 1) / 2, 3, 0, 0>;      \
    if (kept_is_factor) *kept_is_factor = C::kFS ? 1 : 0;                                                \
    emu_solve_shared<C>(P);                                                                              \
    if (status) *status = st;                                                                            \
    return 0;                                                                                            \
  }
  SHARED(4, 4, 1) SHARED(4, 8, 1) SHARED(5, 4, 1) SHARED(5, 8, 1) SHARED(6, 4, 1) SHARED(6, 8, 1)
  SHARED(5, 8, 3) SHARED(5, 8, 7) SHARED(5, 5, 3) SHARED(6, 5, 1)
#undef SHARED
  return -2;
}
