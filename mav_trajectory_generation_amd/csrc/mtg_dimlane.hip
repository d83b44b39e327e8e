// Dimension-in-lane kernel instantiations (mtg_dimlane.h, mtg_dimlane_variants.inc).  This translation unit is
// compiled with -mllvm -amdgpu-kernarg-preload-count=14: the kernels' arguments arrive in user SGPRs at wave launch.
#include <algorithm>
#include "mtg_dimlane.h"

namespace {
template <class C, int DL, int NP>
int launch_dl(void* stream, int grid, const double* times, const double* dfix, double* coeffs, int* status,
              int* traj_status, int B, int ntiles, int policy, double* ws) {
  constexpr size_t lds = mtg_dl_lds_bytes<C, DL, NP>();
  static bool attr_set[3] = {false, false, false};
  hipStream_t st = (hipStream_t)stream;
  // store policy: 0 = nt sc1 (small launches, resident or not), 1 = sc1, 2 = plain write-back
  auto go = [&](auto kern, int slot) -> int {
    if (!attr_set[slot]) {
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
      attr_set[slot] = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NP * 2 * kWave), lds, st, times, dfix, coeffs, status, traj_status, B, ntiles, grid, ws);
    return 0;
  };
  if (policy == 1) return go(mtg_solve_dl_kernel<C, DL, NP, 0, 16>, 1);
  if (policy == 2) return go(mtg_solve_dl_kernel<C, DL, NP, 0, 0>, 2);
  return go(mtg_solve_dl_kernel<C, DL, NP, 0, 18>, 0);
}
}  // namespace

#define MTG_DLCFG(H, K, MS, MI, ME, DV, DL, WS, LS) MtgCfg<H, 1, K, MS, MI, ME, DV, 0, WS, (WS > 0 ? DL : 0), LS>
#define MTG_DLW(H, K, MS, MI, ME, DV, DL, NP, LO, HI, WS, LS)                                                              \
  {H, K, MS, MI, ME, DV, DL, NP, 64 / DL, LO, HI, mtg_dl_lds_bytes<MTG_DLCFG(H, K, MS, MI, ME, DV, DL, WS, LS), DL, NP>(),    \
   (size_t)(MTG_DLCFG(H, K, MS, MI, ME, DV, DL, WS, LS)::WSJ - MTG_DLCFG(H, K, MS, MI, ME, DV, DL, WS, LS)::LSJ) *           \
       MTG_DLCFG(H, K, MS, MI, ME, DV, DL, WS, LS)::WSE * sizeof(double),                                                   \
   launch_dl<MTG_DLCFG(H, K, MS, MI, ME, DV, DL, WS, LS), DL, NP>},
#define MTG_DL(H, K, MS, MI, ME, DV, DL, NP, LO, HI) MTG_DLW(H, K, MS, MI, ME, DV, DL, NP, LO, HI, 0, 0)
static const MtgDimlaneEntry kDimlaneTable[] = {
#include "mtg_dimlane_variants.inc"
};
#undef MTG_DL
#undef MTG_DLW
#undef MTG_DLCFG

const MtgDimlaneEntry* mtg_find_dimlane(int h, int dl, int k, int deriv, const int* mask) {
  for (const MtgDimlaneEntry& e : kDimlaneTable) {
    if (e.h != h || e.dl != dl || e.k != k || e.dv != deriv) continue;
    bool ok = mask[0] == e.ms && mask[k] == e.me;
    for (int v = 1; v < k && ok; ++v) ok = mask[v] == e.mi;
    if (ok) return &e;
  }
  return nullptr;
}

// ---- cross-structure launches (mtg_solve_dl_any_kernel) ----
int mtg_dl_any_index(const MtgDimlaneEntry* e) {
  if (!e || e->dl != 3) return -1;
#define MTG_X(I, H, K, MS, MI, ME, DV, WS, LS) \
  if (e->h == H && e->k == K && e->ms == MS && e->mi == MI && e->me == ME && e->dv == DV) return I;
  MTG_DL_ANY_LIST(MTG_X)
#undef MTG_X
  return -1;
}
size_t mtg_dl_any_lds_bytes() {
  size_t m = 0;
#define MTG_X(I, H, K, MS, MI, ME, DV, WS, LS) \
  m = std::max(m, mtg_dl_pair_bytes<MtgCfg<H, 1, K, MS, MI, ME, DV, 0, WS, (WS > 0 ? 3 : 0), LS>, 3>());
  MTG_DL_ANY_LIST(MTG_X)
#undef MTG_X
  return m;
}
size_t mtg_dl_any_ws_per_lane() {
  size_t m = 0;
#define MTG_X(I, H, K, MS, MI, ME, DV, WS, LS) \
  m = std::max(m, (size_t)(WS - LS) * MtgCfg<H, 1, K, MS, MI, ME, DV, 0, WS, (WS > 0 ? 3 : 0), LS>::WSE * sizeof(double));
  MTG_DL_ANY_LIST(MTG_X)
#undef MTG_X
  return m;
}
int mtg_dl_any_launch(void* stream, int grid, const MtgDlAnyItem* items, const MtgDlAnyUnit* units, int nunits, int* status,
                      double* ws) {
  static bool attr_set = false;
  const size_t lds = mtg_dl_any_lds_bytes();
  auto kern = mtg_solve_dl_any_kernel<18>;   // nt sc1 coefficient stores, as for single-plan launches
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(2 * kWave), lds, (hipStream_t)stream, items, units, nunits, status, ws);
  return 0;
}
