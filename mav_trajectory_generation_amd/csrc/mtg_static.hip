// Specialised register-resident kernel variants (mtg_variants.inc).
#include "mtg_kernels.h"

#define MTG_STATIC_HEAVY(H, D, K, MS, MI, ME, DV) MTG_STATIC_(H, D, K, MS, MI, ME, DV, 1)
#define MTG_STATIC(H, D, K, MS, MI, ME, DV) MTG_STATIC_(H, D, K, MS, MI, ME, DV, 0)
#define MTG_STATIC_(H, D, K, MS, MI, ME, DV, HEAVY)                         \
  {H, D, K, MS, MI, ME, DV, HEAVY,                                          \
   {(SolveFn)mtg_solve_kernel<MtgCfg<H, D, K, MS, MI, ME, DV>, 0>,          \
    (SolveFn)mtg_solve_kernel<MtgCfg<H, D, K, MS, MI, ME, DV>, 3>,          \
    (SolveFn)mtg_solve_kernel<MtgCfg<H, D, K, MS, MI, ME, DV>, 4>,          \
    (SolveFn)mtg_solve_kernel<MtgCfg<H, D, K, MS, MI, ME, DV>, 7>,          \
    (SolveFn)mtg_solve_kernel<MtgCfg<H, D, K, MS, MI, ME, DV, 1>, 9>},         \
   {nullptr, nullptr},                                                      \
   {{nullptr, nullptr}, {nullptr, nullptr}}, 0,                             \
   {nullptr, nullptr, nullptr, nullptr}},
#define MTG_ROLLED(H, D, MS, MI, ME, DV)                                    \
  {H, D, -1, MS, MI, ME, DV, 0,                                             \
   {(SolveFn)mtg_solve_kernel<MtgCfg<H, D, -1, MS, MI, ME, DV>, 0>,         \
    (SolveFn)mtg_solve_kernel<MtgCfg<H, D, -1, MS, MI, ME, DV>, 3>,         \
    (SolveFn)mtg_solve_kernel<MtgCfg<H, D, -1, MS, MI, ME, DV>, 4>,         \
    (SolveFn)mtg_solve_kernel<MtgCfg<H, D, -1, MS, MI, ME, DV>, 7>,         \
    (SolveFn)mtg_solve_kernel<MtgCfg<H, D, -1, MS, MI, ME, DV, 1>, 9>},        \
   {(UpdateFn)mtg_update_kernel<MtgCfg<H, D, -1, MS, MI, ME, DV>, 0>,       \
    (UpdateFn)mtg_update_kernel<MtgCfg<H, D, -1, MS, MI, ME, DV>, 1>},      \
   {{(UpdateFn)mtg_update_slab_kernel<MtgCfg<H, D, -1, MS, MI, ME, DV>, 0, false>,  \
     (UpdateFn)mtg_update_slab_kernel<MtgCfg<H, D, -1, MS, MI, ME, DV>, 0, true>},  \
    {(UpdateFn)mtg_update_slab_kernel<MtgCfg<H, D, -1, MS, MI, ME, DV>, 1, false>,  \
     (UpdateFn)mtg_update_slab_kernel<MtgCfg<H, D, -1, MS, MI, ME, DV>, 1, true>}}, \
   mtg_update_slab_lds_bytes<MtgCfg<H, D, -1, MS, MI, ME, DV>>(),           \
   {(SolveMultiFn)mtg_solve_multi_kernel<MtgCfg<H, D, -1, MS, MI, ME, DV>, 0>, \
    (SolveMultiFn)mtg_solve_multi_kernel<MtgCfg<H, D, -1, MS, MI, ME, DV>, 3>, \
    (SolveMultiFn)mtg_solve_multi_kernel<MtgCfg<H, D, -1, MS, MI, ME, DV>, 4>, \
    (SolveMultiFn)mtg_solve_multi_kernel<MtgCfg<H, D, -1, MS, MI, ME, DV>, 7>}},
static const MtgStaticEntry kStaticTable[] = {
#include "mtg_variants.inc"
};
#undef MTG_STATIC
#undef MTG_STATIC_
#undef MTG_STATIC_HEAVY
#undef MTG_ROLLED

const MtgStaticEntry* mtg_find_static(int h, int d, int k, int deriv, const int* mask, bool rolled_only) {
  for (const MtgStaticEntry& e : kStaticTable) {
    if (e.h != h || e.d != d || e.dv != deriv) continue;
    if (rolled_only && e.k >= 0) continue;
    if (e.k != k && !(e.k < 0 && k >= 2)) continue;   // k < 0: rolled variant, any K >= 2
    bool ok = mask[0] == e.ms && mask[k] == e.me;
    for (int v = 1; v < k && ok; ++v) ok = mask[v] == e.mi;
    if (ok) return &e;
  }
  return nullptr;
}

int mtg_any_cfg_index(const MtgStaticEntry* e) {
  if (!e || e->k >= 0 || (e->d != 1 && e->d != 3) || e->mi != 1) return -1;
  if (e->h == 4 && e->ms == 15 && e->me == 15 && e->dv == 3) return 0;
  if (e->h == 5 && e->ms == 31 && e->me == 31 && e->dv == 4) return 1;
  if (e->h == 6 && e->ms == 63 && e->me == 63 && e->dv == 5) return 2;
  return -1;
}
SolveMultiFn mtg_multi_any_fn(int dg, int variant) {
  static const SolveMultiFn fused[4] = {mtg_solve_multi_any_kernel<3, 0>, mtg_solve_multi_any_kernel<3, 3>,
                                        mtg_solve_multi_any_kernel<3, 4>, mtg_solve_multi_any_kernel<3, 7>};
  static const SolveMultiFn split[4] = {mtg_solve_multi_any_kernel<1, 0>, mtg_solve_multi_any_kernel<1, 3>,
                                        mtg_solve_multi_any_kernel<1, 4>, mtg_solve_multi_any_kernel<1, 7>};
  return dg == 3 ? fused[variant & 3] : dg == 1 ? split[variant & 3] : nullptr;
}

// Slab-output instantiations of the fused form (mtg_solve_slab_kernel): the shapes that run large batches.
#define MTG_SLAB(H, D, K, MS, MI, ME, DV)                                              \
  {H, D, K, MS, MI, ME, DV, mtg_slab_lds_bytes<MtgCfg<H, D, K, MS, MI, ME, DV>>(),      \
   {(SolveFn)mtg_solve_slab_kernel<MtgCfg<H, D, K, MS, MI, ME, DV>, 0>,                \
    (SolveFn)mtg_solve_slab_kernel<MtgCfg<H, D, K, MS, MI, ME, DV>, 18>},               \
   (SolveQueueFn)mtg_solve_slab_queue_kernel<MtgCfg<H, D, K, MS, MI, ME, DV>, 18>,     \
   (SolveFn)mtg_solve_slab_kernel<MtgCfg<H, D, K, MS, MI, ME, DV>, 18, 3>},
static const MtgSlabEntry kSlabTable[] = {
    MTG_SLAB(5, 3, 8, 31, 1, 31, 4)   // BASELINE configs 2/3
    MTG_SLAB(4, 3, 8, 15, 1, 15, 3)
    MTG_SLAB(5, 3, 4, 31, 1, 31, 4)
    MTG_SLAB(4, 3, 4, 15, 1, 15, 3)
    MTG_SLAB(6, 3, 4, 63, 1, 63, 5)
};
#undef MTG_SLAB

const MtgSlabEntry* mtg_find_slab(int h, int d, int k, int deriv, const int* mask) {
  for (const MtgSlabEntry& e : kSlabTable) {
    if (e.h != h || e.d != d || e.k != k || e.dv != deriv) continue;
    bool ok = mask[0] == e.ms && mask[k] == e.me;
    for (int v = 1; v < k && ok; ++v) ok = mask[v] == e.mi;
    if (ok) return &e;
  }
  return nullptr;
}
