// mtg_small.h -- launch form of the solve kernel for SMALL batches (a few hundred tiles: BASELINE config 2).
//
// Same per-lane algorithm as mtg_solve_kernel (mtg_lane.h: twisted block-LDL^T, one lane per trajectory-half, static
// configurations only); what differs is everything around it, because at B = 10k every wave solves exactly one tile and
// the kernel time is the serial latency chain of a single wave plus the launch overhead:
//   * workgroup = 4 wavefronts = two (tile, dimension-group) units x two chain directions: one workgroup per CU puts
//     exactly one wave on every SIMD (the 2-wave workgroups of mtg_solve_kernel landed on SIMDs {s, s+1} / {s+1, s+2}:
//     215 SIMDs doubly occupied, 297 idle -- profiles/r02_phase_timing.txt);
//   * all arguments are scalars / pointers that fit the 14 user SGPRs the hardware can PRELOAD at wave launch
//     (-mllvm -amdgpu-kernarg-preload-count): no s_load round trip to the kernarg segment before the first input load
//     can be addressed; canonical SoA layout only (times[K][B], d_fixed[D][n_fixed][B]) -- every stride is B;
//   * only the coefficient output (what BASELINE config 2 asks for); other outputs / layouts use mtg_solve_kernel.
#ifndef MTG_SMALL_H_
#define MTG_SMALL_H_
#include "mtg_kernels.h"

constexpr int kSmallBlock = 4 * kWave;
#ifndef MTG_SMALL_OCC
#define MTG_SMALL_OCC 1
#endif

template <class C, int OUT>
__global__ __launch_bounds__(kSmallBlock, MTG_SMALL_OCC) void mtg_solve_small_kernel(const double* __restrict__ times,
                                                                          const double* __restrict__ dfix,
                                                                          double* __restrict__ coeffs, int* status,
                                                                          int B, int ntiles, long long* tdbg_base) {
  static_assert(C::kStatic, "small-launch form: static configurations");
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x & (kWave - 1);
  const int w = threadIdx.x >> 6;      // wave-uniform
  const int pair = w >> 1, dir = w & 1;
#if defined(MTG_LAB_TIMING)
  long long* tdbg = tdbg_base + ((long long)((blockIdx.y * gridDim.x + blockIdx.x) * 4 + w)) * 16;
  if (lane == 0) { tdbg[0] = clock64(); tdbg[14] = wall_clock64(); }
#endif
  MtgParams P;
  P.times = times; P.ts_b = 1; P.ts_k = B;
  P.dfix = dfix; P.fs_b = 1; P.fs_c = B; P.fs_d = (long long)C::offFEnd * B;
  P.coeffs = coeffs;
  P.dfree = nullptr; P.ps_b = P.ps_d = P.ps_c = 0;
  P.cost = nullptr; P.ws = nullptr; P.ws_stride = 0;
  P.status = status; P.tstatus = nullptr;
  P.vmask = nullptr; P.offF = nullptr; P.offP = nullptr;
  P.B = B; P.K = C::KT; P.Dtot = (int)gridDim.y * C::D; P.dim0 = (int)blockIdx.y * C::D;
  P.deriv = C::DV; P.h1off = C::H1OFF; P.ainvoff = C::AINVOFF;
  P.pert_on = 0; P.pert_seg = -1; P.pert_tpv = 1; P.pert_h = P.pert_corr = P.pert_lo = 0.0;

  const int nunits = (ntiles + 1) >> 1;
  MtgLane<C> ln;
  auto tile_of = [&](int it) { const int t = 2 * it + pair; return t < ntiles ? t : ntiles - 1; };
  auto fetch = [&](int tile_) {
    long long bb = (long long)tile_ * kWave + lane;
    if (bb >= B) bb = B - 1;
    if (dir == 0) mtg_preload_into<C, 1>(P, bb, ln.T, ln.fx);
    else mtg_preload_into<C, -1>(P, bb, ln.T, ln.fx);
  };
  if ((int)blockIdx.x < nunits) fetch(tile_of(blockIdx.x));
#if defined(MTG_LAB_TIMING)
  if (lane == 0) tdbg[1] = clock64();
#endif
  constexpr int mm = C::MI;                       // K >= 2 static configurations: the middle vertex is interior
  static_assert(C::KT >= 2, "small-launch form needs an interior middle vertex");
  constexpr int fmid = C::H - C::popc(mm);
  constexpr int nslots = fmid * (fmid + 1) / 2 + C::D * fmid;
  // LDS per pair: [staging A][staging B][exchange A][exchange B]
  constexpr size_t pair_doubles = 2 * mtg_stage_doubles<C>() + (size_t)2 * nslots * kWave;
  double* base = lds + (size_t)pair * pair_doubles;
  MtgLdsOut<C, (OUT & 4) != 0> io;
  io.init(P, base + (size_t)dir * mtg_stage_doubles<C>(), lane);
  double* xch = base + 2 * mtg_stage_doubles<C>();
  double* mine = xch + (size_t)dir * nslots * kWave + lane;
  const double* other = xch + (size_t)(1 - dir) * nslots * kWave + lane;
  for (int it = blockIdx.x; it < nunits; it += gridDim.x) {
    const int tile = tile_of(it);
    io.b0 = (long long)tile * kWave;
    const long long bl = io.b0 + lane;
    const bool active = bl < B;
    const long long b = active ? bl : B - 1;
    const bool first = it == (int)blockIdx.x;
    if (!first) fetch(tile);
    if (dir == 0) mtg_lane_forward<C, 1>(P, b, ln, nullptr, false);
    else mtg_lane_forward<C, -1>(P, b, ln, nullptr, false);
    mtg_pack_mid<C>(ln, mm, mine, kWave);
#if defined(MTG_LAB_TIMING)
    if (lane == 0 && first) tdbg[2] = clock64();
#endif
    __syncthreads();
#if defined(MTG_LAB_TIMING)
    if (lane == 0 && first) tdbg[3] = clock64();
#endif
    if (dir == 0) mtg_lane_finish<C, 1, OUT>(P, b, ln, nullptr, other, kWave, io, active);
    else mtg_lane_finish<C, -1, OUT>(P, b, ln, nullptr, other, kWave, io, active);
#if defined(MTG_LAB_TIMING)
    if (lane == 0 && first) tdbg[4] = clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && first) { tdbg[5] = clock64(); tdbg[15] = wall_clock64(); }
#endif
    __syncthreads();
  }
}

template <class C>
constexpr size_t mtg_small_lds_bytes() {
  constexpr int fmid = C::H - C::popc(C::MI);
  constexpr int nslots = fmid * (fmid + 1) / 2 + C::D * fmid;
  return 2 * (2 * mtg_stage_doubles<C>() + (size_t)2 * nslots * kWave) * sizeof(double);
}
#endif  // MTG_SMALL_H_
