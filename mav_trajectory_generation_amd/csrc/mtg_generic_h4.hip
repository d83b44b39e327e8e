// Generic (runtime K / masks / derivative) kernels for N = 8: explicit instantiations for D = 1..4.
#include "mtg_kernels.h"

SolveFn mtg_pick_generic_solve_h4(int d, bool extra) {
  switch (d) {
    case 1: return extra ? (SolveFn)mtg_solve_kernel<GenericCfg<4, 1>, 3> : (SolveFn)mtg_solve_kernel<GenericCfg<4, 1>, 0>;
    case 2: return extra ? (SolveFn)mtg_solve_kernel<GenericCfg<4, 2>, 3> : (SolveFn)mtg_solve_kernel<GenericCfg<4, 2>, 0>;
    case 3: return extra ? (SolveFn)mtg_solve_kernel<GenericCfg<4, 3>, 3> : (SolveFn)mtg_solve_kernel<GenericCfg<4, 3>, 0>;
    case 4: return extra ? (SolveFn)mtg_solve_kernel<GenericCfg<4, 4>, 3> : (SolveFn)mtg_solve_kernel<GenericCfg<4, 4>, 0>;
  }
  return nullptr;
}
UpdateFn mtg_pick_generic_update_h4(int d, bool wc) {
  switch (d) {
    case 1: return wc ? (UpdateFn)mtg_update_kernel<GenericCfg<4, 1>, 1> : (UpdateFn)mtg_update_kernel<GenericCfg<4, 1>, 0>;
    case 2: return wc ? (UpdateFn)mtg_update_kernel<GenericCfg<4, 2>, 1> : (UpdateFn)mtg_update_kernel<GenericCfg<4, 2>, 0>;
    case 3: return wc ? (UpdateFn)mtg_update_kernel<GenericCfg<4, 3>, 1> : (UpdateFn)mtg_update_kernel<GenericCfg<4, 3>, 0>;
    case 4: return wc ? (UpdateFn)mtg_update_kernel<GenericCfg<4, 4>, 1> : (UpdateFn)mtg_update_kernel<GenericCfg<4, 4>, 0>;
  }
  return nullptr;
}
