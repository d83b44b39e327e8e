// mtg_dimlane_table.h -- launcher + table boilerplate shared by the dimension-in-lane translation units.  The including
// file defines MTG_DL_TABLE_INC (the variant list), MTG_DL_TABLE_FN (name of the function that returns the table) and,
// for the secondary units, MTG_DL_SINGLE_POLICY (no queue / extra-output twins: coefficient-only launches).
#include <algorithm>
#include "mtg_dimlane.h"

namespace {
constexpr int kMaxDevices = 64;
template <class C, int DL, int NP>
int launch_dl(void* stream, int grid, const double* times, const double* dfix, double* coeffs, int* status,
              int* traj_status, int B, int ntiles, double* ws, int aos) {
  constexpr size_t lds = mtg_dl_lds_bytes<C, DL, NP>();
  // (the attribute is a property of the function ON A DEVICE: one flag per device for processes that drive several GPUs)
  static bool attr_set[kMaxDevices] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return -1;
  auto kern = mtg_solve_dl_kernel<C, DL, NP, 0, 18>;      // coefficient stores nt sc1 (round 2 / 5 measured sc1 and write-back: no gain)
  if (!attr_set[dev]) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NP * 2 * kWave), lds, (hipStream_t)stream, times, dfix, coeffs, status, traj_status, B, ntiles, grid, aos, ws);
  return 0;
}
#if !defined(MTG_DL_SINGLE_POLICY)
// the queue form (mtg_solve_linear_sequence): nt sc1 stores, main table only
template <class C, int DL, int NP>
int launch_dl_queue(void* stream, int grid, const MtgSeqQueue* q, int* status, int B, int ntiles, double* ws, int aos) {
  constexpr size_t lds = mtg_dl_lds_bytes<C, DL, NP>();
  static bool attr_set[kMaxDevices] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return -1;
  auto kern = mtg_solve_dl_queue_kernel<C, DL, NP, 0, 18>;
  if (!attr_set[dev]) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NP * 2 * kWave), lds, (hipStream_t)stream, status, B, ntiles, grid, aos, ws, *q);
  return 0;
}
#define MTG_DL_QUEUE_FN(...) launch_dl_queue<__VA_ARGS__>
// solves with extra outputs (cost / d_P): nt sc1 stores, main table only
template <class C, int DL, int NP>
int launch_dl_extra(void* stream, int grid, const double* times, const double* dfix, double* coeffs, int* status,
                    int* traj_status, int B, int ntiles, double* ws, int aos, double* dfree, double* cost, long long ps_b,
                    long long ps_d, long long ps_c) {
  constexpr size_t lds = mtg_dl_lds_bytes<C, DL, NP>();
  static bool attr_set[kMaxDevices] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return -1;
  auto kern = mtg_solve_dl_extra_kernel<C, DL, NP, 18>;
  if (!attr_set[dev]) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NP * 2 * kWave), lds, (hipStream_t)stream, times, dfix, coeffs, status, traj_status, B,
                     ntiles, grid, aos, ws, MtgDlExtra{dfree, cost, ps_b, ps_d, ps_c});
  return 0;
}
#define MTG_DL_EXTRA_FN(...) launch_dl_extra<__VA_ARGS__>
#else
#define MTG_DL_QUEUE_FN(...) nullptr
#define MTG_DL_EXTRA_FN(...) nullptr
#endif
}  // namespace

// MTG_DLR: as MTG_DLW with the register steps' G shared between the dimension lanes as well (MtgCfg::kRegShared)
#define MTG_DLCFG(H, K, MS, MI, ME, DV, DL, WS, LS, RS) MtgCfg<H, 1, K, MS, MI, ME, DV, 0, WS, ((WS > 0 || RS) ? DL : 0), LS, RS>
#define MTG_DLX(H, K, MS, MI, ME, DV, DL, NP, LO, HI, WS, LS, RS)                                                               \
  {H, K, MS, MI, ME, DV, DL, NP, 64 / DL, LO, HI, mtg_dl_lds_bytes<MTG_DLCFG(H, K, MS, MI, ME, DV, DL, WS, LS, RS), DL, NP>(),    \
   (size_t)(MTG_DLCFG(H, K, MS, MI, ME, DV, DL, WS, LS, RS)::WSJ - MTG_DLCFG(H, K, MS, MI, ME, DV, DL, WS, LS, RS)::LSJ) *           \
       MTG_DLCFG(H, K, MS, MI, ME, DV, DL, WS, LS, RS)::WSE * sizeof(double),                                                   \
   launch_dl<MTG_DLCFG(H, K, MS, MI, ME, DV, DL, WS, LS, RS), DL, NP>,                                                         \
   MTG_DL_QUEUE_FN(MTG_DLCFG(H, K, MS, MI, ME, DV, DL, WS, LS, RS), DL, NP),                                                    \
   MTG_DL_EXTRA_FN(MTG_DLCFG(H, K, MS, MI, ME, DV, DL, WS, LS, RS), DL, NP)},
#define MTG_DLW(H, K, MS, MI, ME, DV, DL, NP, LO, HI, WS, LS) MTG_DLX(H, K, MS, MI, ME, DV, DL, NP, LO, HI, WS, LS, 0)
#define MTG_DLR(H, K, MS, MI, ME, DV, DL, NP, LO, HI, WS, LS) MTG_DLX(H, K, MS, MI, ME, DV, DL, NP, LO, HI, WS, LS, 1)
#define MTG_DL(H, K, MS, MI, ME, DV, DL, NP, LO, HI) MTG_DLX(H, K, MS, MI, ME, DV, DL, NP, LO, HI, 0, 0, 0)
static const MtgDimlaneEntry kDimlaneTable[] = {
#include MTG_DL_TABLE_INC
};
#undef MTG_DL
#undef MTG_DLW
#undef MTG_DLR
#undef MTG_DLX
#undef MTG_DLCFG
#undef MTG_DL_QUEUE_FN

const MtgDimlaneEntry* MTG_DL_TABLE_FN(int* count) {
  *count = (int)(sizeof(kDimlaneTable) / sizeof(kDimlaneTable[0]));
  return kDimlaneTable;
}
