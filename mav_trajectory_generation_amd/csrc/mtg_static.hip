// Specialised register-resident kernel variants (mtg_variants.inc).
#include "mtg_kernels.h"

#define MTG_STATIC(H, D, K, MS, MI, ME, DV)                                 \
  {H, D, K, MS, MI, ME, DV,                                                 \
   {(SolveFn)mtg_solve_kernel<MtgCfg<H, D, K, MS, MI, ME, DV>, 0>,          \
    (SolveFn)mtg_solve_kernel<MtgCfg<H, D, K, MS, MI, ME, DV>, 3>,          \
    (SolveFn)mtg_solve_kernel<MtgCfg<H, D, K, MS, MI, ME, DV>, 4>,          \
    (SolveFn)mtg_solve_kernel<MtgCfg<H, D, K, MS, MI, ME, DV>, 7>}},
#define MTG_ROLLED(H, D, MS, MI, ME, DV)                                    \
  {H, D, -1, MS, MI, ME, DV,                                                \
   {(SolveFn)mtg_solve_kernel<MtgCfg<H, D, -1, MS, MI, ME, DV>, 0>,         \
    (SolveFn)mtg_solve_kernel<MtgCfg<H, D, -1, MS, MI, ME, DV>, 3>,         \
    (SolveFn)mtg_solve_kernel<MtgCfg<H, D, -1, MS, MI, ME, DV>, 4>,         \
    (SolveFn)mtg_solve_kernel<MtgCfg<H, D, -1, MS, MI, ME, DV>, 7>}},
static const MtgStaticEntry kStaticTable[] = {
#include "mtg_variants.inc"
};
#undef MTG_STATIC
#undef MTG_ROLLED

const MtgStaticEntry* mtg_find_static(int h, int d, int k, int deriv, const int* mask) {
  for (const MtgStaticEntry& e : kStaticTable) {
    if (e.h != h || e.d != d || e.dv != deriv) continue;
    if (e.k != k && !(e.k < 0 && k >= 2)) continue;   // k < 0: rolled variant, any K >= 2
    bool ok = mask[0] == e.ms && mask[k] == e.me;
    for (int v = 1; v < k && ok; ++v) ok = mask[v] == e.mi;
    if (ok) return &e;
  }
  return nullptr;
}
