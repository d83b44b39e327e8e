// Run-time-K dimension-in-lane bodies (mtg_dimlane_rt.h): one per polynomial order of the standard shapes (trajectory ends fully
// fixed, position-only interior vertices, D = 3).  R (register steps) and L (LDS steps) per order: the largest values that
// compile without scratch spills / fit the LDS next to the output slabs with two workgroups per CU.  D = 3 and D = 4.
#include "mtg_dimlane_rt.h"

namespace {
constexpr int kMaxDevices = 64;
template <class C, int DL, int R, int L>
int launch_rt(void* stream, int grid, const double* times, const double* dfix, double* coeffs, int* status, int* traj_status,
              int B, int K, int ntiles, double* ws, int aos) {
  constexpr size_t lds = mtg_rt_lds_bytes<C, DL, L>();
  static bool attr_set[2][kMaxDevices] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return -1;
  // two instantiations: pieces of K * DL * N * 8 bytes that are a multiple of 64 bytes, and the others (per-row phase maps)
  const bool phase = ((long long)K * DL * C::N * 8) % 64 != 0;
  auto go = [&](auto kern, bool& done) -> int {
    if (!done) {
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
      done = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(2 * kWave), lds, (hipStream_t)stream, times, dfix, coeffs, status, traj_status, B, K,
                       ntiles, aos, ws);
    return 0;
  };
  if constexpr ((DL * C::N * 8) % 64 != 0) {
    if (phase) return go(mtg_solve_dl_rt_kernel<C, DL, R, L, 18, true>, attr_set[1][dev]);
  }
  return go(mtg_solve_dl_rt_kernel<C, DL, R, L, 18, false>, attr_set[0][dev]);
}
#define MTG_RTCFG(H, MS, MI, ME, DV, DL) MtgCfg<H, 1, -1, MS, MI, ME, DV, 0, 0, DL, 0, 1>
#define MTG_RT(H, MS, MI, ME, DV, DL, R, L)                                                                      \
  {H, MS, MI, ME, DV, DL, 64 / DL, R, L, mtg_rt_lds_bytes<MTG_RTCFG(H, MS, MI, ME, DV, DL), DL, L>(),          \
   (size_t)MTG_RTCFG(H, MS, MI, ME, DV, DL)::WSE * sizeof(double), launch_rt<MTG_RTCFG(H, MS, MI, ME, DV, DL), DL, R, L>},
const MtgDimlaneRtEntry kRtTable[] = {
    // (R, L) = the largest register / LDS step counts without scratch spills / with two workgroups per CU (probed with
    // hipcc -Rpass-analysis=kernel-resource-usage): half-chains of up to 1 + R + L steps stay on chip
    // (round 4, factor store: a kept step is f (f + 1) / 2 + DL f numbers per trajectory instead of f * f + DL f -- was (24, 8) / (11, 5) /
    // (5, 3) and K <= 66 / 34 / 18; now K <= 74 / 40 / 26)
    MTG_RT(4, 15, 1, 15, 3, 3, 27, 9)
    MTG_RT(5, 31, 1, 31, 4, 3, 13, 6)
    MTG_RT(6, 63, 1, 63, 5, 3, 8, 4)
    // the same with a yaw dimension (x, y, z, yaw; 16 trajectories per wave)
    MTG_RT(4, 15, 1, 15, 3, 4, 24, 9)
    MTG_RT(5, 31, 1, 31, 4, 4, 15, 6)
    MTG_RT(6, 63, 1, 63, 5, 4, 8, 4)
    // N = 10 with more fixed at the interior vertices: position + velocity (+ acceleration: BASELINE config 5's pattern, any K)
    MTG_RT(5, 31, 7, 31, 4, 4, 24, 8)
    MTG_RT(5, 31, 7, 31, 4, 3, 24, 8)
    MTG_RT(5, 31, 3, 31, 4, 3, 20, 6)
};
}  // namespace

const MtgDimlaneRtEntry* mtg_find_dimlane_rt(int h, int dl, int k, int deriv, const int* mask) {
  if (k < 2) return nullptr;
  for (const MtgDimlaneRtEntry& e : kRtTable) {
    if (e.h != h || e.dl != dl || e.dv != deriv) continue;
    bool ok = mask[0] == e.ms && mask[k] == e.me;
    for (int v = 1; v < k && ok; ++v) ok = mask[v] == e.mi;
    if (ok) return &e;
  }
  return nullptr;
}
