// mtg_host.cpp -- host build of the per-lane algorithm (mtg_lane.h) for latency-bound callers.
//
// The reference's nlopt-style callers solve ONE trajectory at a time, thousands of times in a row
// (polynomial_optimization_nonlinear_impl.h:569-571, :632-633): a kernel launch plus a PCIe round trip (~25 us) is 5-10x
// the arithmetic of such a solve.  With MTG_FLAG_HOST_POINTERS | MTG_FLAG_HOST_BACKEND and a batch of at most
// MTG_HOST_BACKEND_MAX_BATCH trajectories, mtg_solve_linear / mtg_update_segments_from_free run THIS code on the calling
// thread instead: the very same lane functions the kernels are made of (twisted block-LDL^T, scaling identities, constant
// tables), compiled for the host -- one "lane" per trajectory-half, executed one after the other.  It is product code,
// independent of the oracle directory (which restates the REFERENCE's algorithm); it is never chosen for device pointers, never
// without the explicit flag, and the library still refuses to create a context without a HIP device.
#include <cstring>
#include <vector>

#include "mtg_lane.h"

namespace {

template <class C, int OUT>
void host_solve(MtgParams P) {
  constexpr int H = C::H, D = C::D;
  const int K = P.K;
  const int vm = (K + 1) / 2;
  const int mm = mtg_mask<C>(P, vm);
  const int nslots = mtg_mid_slots<C>(mm);
  const int kc = (K + 1) / 2;
  const size_t E = (size_t)H * H + (size_t)D * H;
  // per-thread scratch, grown on demand: a single-trajectory call must not pay four heap allocations
  static thread_local std::vector<double> wsa, wsb, bufa, bufb;
  if (wsa.size() < kc * E + 1) { wsa.resize(kc * E + 1); wsb.resize(kc * E + 1); }
  if (bufa.size() < (size_t)nslots + 1) { bufa.resize(nslots + 1); bufb.resize(nslots + 1); }
  P.ws_stride = 1;
  for (long long b = 0; b < P.B; ++b) {
    MtgLane<C> la, lb;
    mtg_lane_forward<C, 1>(P, b, la, wsa.data());
    mtg_lane_forward<C, -1>(P, b, lb, wsb.data());
    mtg_pack_mid<C>(la, mm, bufa.data(), 1);
    mtg_pack_mid<C>(lb, mm, bufb.data(), 1);
    MtgDirectOut<C> io;
    io.b = b;
    mtg_lane_finish<C, 1, OUT>(P, b, la, wsa.data(), bufb.data(), 1, io, true);
    mtg_lane_finish<C, -1, OUT>(P, b, lb, wsb.data(), bufa.data(), 1, io, true);
  }
}

template <class C, int OUT>
void host_update(MtgParams P) {
  for (long long b = 0; b < P.B; ++b) {
    MtgDirectOut<C> io;
    io.b = b;
    mtg_lane_update<C, OUT>(P, b, io, true);
  }
}

using Fn = void (*)(MtgParams);
template <int H, int D> using Cfg = MtgCfg<H, D, 0, 0, 0, 0>;   // everything at run time: any masks, any K

template <int H>
Fn pick_h(int d, bool extra, bool update) {
#define MTG_CASE(DD)                                                                       \
  case DD:                                                                                 \
    if (update) return extra ? (Fn)host_update<Cfg<H, DD>, 1> : (Fn)host_update<Cfg<H, DD>, 0>; \
    return extra ? (Fn)host_solve<Cfg<H, DD>, 3> : (Fn)host_solve<Cfg<H, DD>, 0>;
  switch (d) { MTG_CASE(1) MTG_CASE(2) MTG_CASE(3) MTG_CASE(4) }
#undef MTG_CASE
  return nullptr;
}

Fn pick(int h, int d, bool extra, bool update) {
  switch (h) {
    case 1: return pick_h<1>(d, extra, update);
    case 2: return pick_h<2>(d, extra, update);
    case 3: return pick_h<3>(d, extra, update);
    case 4: return pick_h<4>(d, extra, update);
    case 5: return pick_h<5>(d, extra, update);
    case 6: return pick_h<6>(d, extra, update);
  }
  return nullptr;
}

}  // namespace

// P: fully populated MtgParams with HOST pointers (times, dfix, coeffs, dfree, cost, status, tstatus, vmask, offF, offP);
// Dtot dimensions are processed in groups of at most 4.  Returns 0, or -1 for an unsupported shape.
int mtg_host_run(const MtgParams& P0, int H, bool update) {
  MtgParams P = P0;
  if (P.cost) for (long long b = 0; b < P.B; ++b) P.cost[b] = 0.0;
  const bool extra = P.cost != nullptr || (!update && P.dfree != nullptr);
  for (int dim0 = 0; dim0 < P.Dtot; dim0 += 4) {
    const int dc = P.Dtot - dim0 < 4 ? P.Dtot - dim0 : 4;
    Fn fn = pick(H, dc, extra, update);
    if (!fn) return -1;
    MtgParams Q = P;
    Q.dim0 = dim0;
    fn(Q);
  }
  return 0;
}
