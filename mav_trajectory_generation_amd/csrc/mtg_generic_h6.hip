// Generic (runtime K / masks / derivative) kernels for N = 12: explicit instantiations for D = 1..4.
#include "mtg_kernels.h"

SolveFn mtg_pick_generic_solve_h6(int d, int mode) {
#define MTG_CASE(DD)                                                                                   \
  case DD:                                                                                             \
    return mode == 2 ? (SolveFn)mtg_solve_kernel<GenericCostCfg<6, DD>, 9>                                 \
                     : (mode == 1 ? (SolveFn)mtg_solve_kernel<GenericCfg<6, DD>, 3>                    \
                                  : (SolveFn)mtg_solve_kernel<GenericCfg<6, DD>, 0>);
  switch (d) { MTG_CASE(1) MTG_CASE(2) MTG_CASE(3) MTG_CASE(4) }
#undef MTG_CASE
  return nullptr;
}
UpdateFn mtg_pick_generic_update_h6(int d, bool wc) {
  switch (d) {
    case 1: return wc ? (UpdateFn)mtg_update_kernel<GenericCfg<6, 1>, 1> : (UpdateFn)mtg_update_kernel<GenericCfg<6, 1>, 0>;
    case 2: return wc ? (UpdateFn)mtg_update_kernel<GenericCfg<6, 2>, 1> : (UpdateFn)mtg_update_kernel<GenericCfg<6, 2>, 0>;
    case 3: return wc ? (UpdateFn)mtg_update_kernel<GenericCfg<6, 3>, 1> : (UpdateFn)mtg_update_kernel<GenericCfg<6, 3>, 0>;
    case 4: return wc ? (UpdateFn)mtg_update_kernel<GenericCfg<6, 4>, 1> : (UpdateFn)mtg_update_kernel<GenericCfg<6, 4>, 0>;
  }
  return nullptr;
}
