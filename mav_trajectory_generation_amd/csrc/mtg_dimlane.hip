// Dimension-in-lane kernel instantiations (mtg_dimlane.h, mtg_dimlane_variants.inc; further chain lengths in
// mtg_dimlane_h4 / h5 / h6.hip (K <= 15) and mtg_dimlane_h4b / h5b / h6b.hip (K = 17 .. 31)).  These translation units are compiled with -mllvm -amdgpu-kernarg-preload-count=14: the
// kernels' arguments arrive in user SGPRs at wave launch.
#define MTG_DL_TABLE_FN mtg_dimlane_main
#define MTG_DL_TABLE_INC "mtg_dimlane_variants.inc"
#include "mtg_dimlane_table.h"

const MtgDimlaneEntry* mtg_dimlane_more_h4(int* count);
const MtgDimlaneEntry* mtg_dimlane_more_h5(int* count);
const MtgDimlaneEntry* mtg_dimlane_more_h6(int* count);
const MtgDimlaneEntry* mtg_dimlane_more_h4b(int* count);
const MtgDimlaneEntry* mtg_dimlane_more_h5b(int* count);
const MtgDimlaneEntry* mtg_dimlane_more_h6b(int* count);

const MtgDimlaneEntry* mtg_find_dimlane(int h, int dl, int k, int deriv, const int* mask) {
  typedef const MtgDimlaneEntry* (*TableFn)(int*);
  static const TableFn tables[] = {mtg_dimlane_main,    mtg_dimlane_more_h4,  mtg_dimlane_more_h5, mtg_dimlane_more_h6,
                                   mtg_dimlane_more_h4b, mtg_dimlane_more_h5b, mtg_dimlane_more_h6b};
  for (TableFn fn : tables) {
    int n = 0;
    const MtgDimlaneEntry* tab = fn(&n);
    for (int i = 0; i < n; ++i) {
      const MtgDimlaneEntry& e = tab[i];
      if (e.h != h || e.dl != dl || e.k != k || e.dv != deriv) continue;
      bool ok = mask[0] == e.ms && mask[k] == e.me;
      for (int v = 1; v < k && ok; ++v) ok = mask[v] == e.mi;
      if (ok) return &e;
    }
  }
  return nullptr;
}

// ---- cross-structure launches (mtg_solve_dl_any_kernel) ----
int mtg_dl_any_index(const MtgDimlaneEntry* e) {
  if (!e || e->dl != 3) return -1;
#define MTG_X(I, H, K, MS, MI, ME, DV, WS, LS, RS) \
  if (e->h == H && e->k == K && e->ms == MS && e->mi == MI && e->me == ME && e->dv == DV) return I;
  MTG_DL_ANY_LIST(MTG_X)
#undef MTG_X
  return -1;
}
size_t mtg_dl_any_lds_bytes() {
  size_t m = 0;
#define MTG_X(I, H, K, MS, MI, ME, DV, WS, LS, RS) \
  m = std::max(m, mtg_dl_pair_bytes<MtgCfg<H, 1, K, MS, MI, ME, DV, 0, WS, ((WS > 0 || RS) ? 3 : 0), LS, RS>, 3>());
  MTG_DL_ANY_LIST(MTG_X)
#undef MTG_X
  return m;
}
size_t mtg_dl_any_ws_per_lane() {
  size_t m = 0;
#define MTG_X(I, H, K, MS, MI, ME, DV, WS, LS, RS) \
  m = std::max(m, (size_t)(WS - LS) * MtgCfg<H, 1, K, MS, MI, ME, DV, 0, WS, ((WS > 0 || RS) ? 3 : 0), LS, RS>::WSE * sizeof(double));
  MTG_DL_ANY_LIST(MTG_X)
#undef MTG_X
  return m;
}
int mtg_dl_any_launch(void* stream, int grid, const MtgDlAnyItem* items, const MtgDlAnyUnit* units, const int* wg_begin,
                      int* status, double* ws) {
  static bool attr_set[kMaxDevices] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return -1;
  const size_t lds = mtg_dl_any_lds_bytes();
  auto kern = mtg_solve_dl_any_kernel<18>;   // nt sc1 coefficient stores, as for single-plan launches
  if (!attr_set[dev]) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(2 * kWave), lds, (hipStream_t)stream, items, units, wg_begin, status, ws);
  return 0;
}
