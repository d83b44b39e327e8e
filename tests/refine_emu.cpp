// Host build of the double-double residual of MTG_FLAG_REFINE (csrc/mtg_refine_dd.h -- the code the device kernel runs), for the
// CPU suite: tests/test_refine.py compares it with the residual formed at 50 digits.  AoS layouts as tests/host_emu.cpp.
#include <cstring>
#include <vector>
#define MTG_TABLE_QUAL static const
#include "../mav_trajectory_generation_amd/csrc/mtg_tables.inc"
#include "../mav_trajectory_generation_amd/csrc/mtg_tables_dd.inc"
#include "../mav_trajectory_generation_amd/csrc/mtg_refine_dd.h"

extern "C" int mtg_refine_emu_residual(int N, int D, int K, int deriv, const int* mask, long long B, const double* times,
                                       const double* dfix, const double* dfree, double* rhs) {
  const int H = N / 2;
  std::vector<int> offF(K + 2, 0), offP(K + 2, 0);
  for (int v = 0; v <= K; ++v) {
    const int nf = __builtin_popcount((unsigned)mask[v]);
    offF[v + 1] = offF[v] + nf;
    offP[v + 1] = offP[v] + H - nf;
  }
  const int n_fixed = offF[K + 1], n_free = offP[K + 1];
  mtg_refine::RefineArgs A{times, K, 1, dfix, (long long)D * n_fixed, n_fixed, 1, dfree, (long long)D * n_free, n_free, 1, rhs,
                           mask, offF.data(), offP.data(), B, K, D, deriv, n_free, kH1Off[H][deriv]};
  for (long long b = 0; b < B; ++b)
    for (int dm = 0; dm < D; ++dm) {
      const double *hh = kH1 + A.h1off, *hl = kH1Lo + A.h1off;
      switch (H) {
        case 1: mtg_refine::residual_dd_one<1>(A, b, dm, hh, hl); break;
        case 2: mtg_refine::residual_dd_one<2>(A, b, dm, hh, hl); break;
        case 3: mtg_refine::residual_dd_one<3>(A, b, dm, hh, hl); break;
        case 4: mtg_refine::residual_dd_one<4>(A, b, dm, hh, hl); break;
        case 5: mtg_refine::residual_dd_one<5>(A, b, dm, hh, hl); break;
        case 6: mtg_refine::residual_dd_one<6>(A, b, dm, hh, hl); break;
        default: return -1;
      }
    }
  return 0;
}
