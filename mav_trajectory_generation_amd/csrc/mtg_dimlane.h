// mtg_dimlane.h -- "dimension-in-lane" launch form of the solve kernel.
//
// Same per-lane algorithm as mtg_solve_kernel (mtg_lane.h: twisted block-LDL^T along the vertex chain, one lane per
// trajectory-half and DIMENSION, static configurations), but the DL dimensions of a trajectory sit in DL lanes of ONE
// wavefront (lane = dim * TPW + trajectory, TPW = 64 / DL trajectories per wave) instead of in DL different
// workgroups (the "dimension-split" form, grid.y).  The arithmetic per wave is identical to the split form (the
// factorisation is repeated per dimension lane; ~1450 VALU instructions per wave for BASELINE config 2), what changes
// is the memory side:
//   * a wave holds ALL coefficients of its trajectories' segments: the D*N*8-byte pieces of segment k (240 B for
//     N = 10, D = 3) of one trajectory are contiguous in the output, and the segments of a chain direction are
//     contiguous with each other (dir A: segments [0, K/2), dir B: the rest).  The wave assembles its half of every
//     trajectory in an LDS slab and streams it out in runs that start and end on 64-byte boundaries: EVERY store
//     instruction writes whole 64-byte sectors (4 consecutive lanes x 16 B).  The split form's 80-byte pieces completed
//     sectors from different workgroups at different times: 1.21x write amplification, and read-modify-write at the
//     memory side whenever the line was not already resident (B = 10k with rotating buffers: 14.5 us vs 9.9 us);
//   * segment times are loaded once per trajectory (3 lanes hit the same address), not once per dimension group.
// Workgroup = 4 wavefronts = two tiles x two chain directions: one workgroup per CU puts one wave on every SIMD.
// All kernel arguments fit the user SGPRs that the hardware preloads at wave launch
// (-mllvm -amdgpu-kernarg-preload-count=14): no kernarg round trip in front of the first input load.  Canonical SoA
// input layout only (times[K][B], d_fixed[D][n_fixed][B]).
#ifndef MTG_DIMLANE_H_
#define MTG_DIMLANE_H_
#include "mtg_kernels.h"   // (brings in mtg_slab.h: MtgSlabOut)

// (Built, measured and removed in round 3 / 4: requesting the NEXT tile's inputs before the current tile is solved -- 30 more
// registers, no effect: B = 125k 90.3 vs 91.4 us, N = 12 / K = 8 at 100k 93.7 vs 92.0 us.  The third of its lifetime a wave
// spends parked, profiles/r03f_pmc_stalls.txt, is not the input round trip.)
#ifndef MTG_DL_PEND
#define MTG_DL_PEND true   // MtgSlabOut::PEND: a drained range's chunks wait in registers across one back-substitution step
#endif


template <class C, int DL>
__host__ __device__ constexpr size_t mtg_dl_slab_bytes() {   // one wave's slab (the larger of the two directions')
  constexpr size_t a = (size_t)MtgSlabOut<C, DL, 1, 0>::TPW * MtgSlabOut<C, DL, 1, 0>::ROWB;
  constexpr size_t b = (size_t)MtgSlabOut<C, DL, -1, 0>::TPW * MtgSlabOut<C, DL, -1, 0>::ROWB;
  return a > b ? a : b;
}
template <class C, int DL>
__host__ __device__ constexpr size_t mtg_dl_half_bytes() {   // one direction's slab (the exchange buffer aliases the other's)
  constexpr int fmid = C::H - C::popc(C::MI);
  constexpr size_t xch = (size_t)(fmid * (fmid + 1) / 2 + fmid) * kWave * sizeof(double);
  constexpr size_t slab = (mtg_dl_slab_bytes<C, DL>() + 15) / 16 * 16;
  return slab > xch ? slab : xch;
}
template <class C>
__host__ __device__ constexpr size_t mtg_dl_steps_bytes() {  // one wave's LDS step area (MtgCfg::LSJ workspace steps)
  return (size_t)C::LSJ * C::WSE * kWave * sizeof(double);
}
template <class C, int DL>
__host__ __device__ constexpr size_t mtg_dl_pair_bytes() {   // [slab A][slab B][steps A][steps B]
  return 2 * mtg_dl_half_bytes<C, DL>() + 2 * mtg_dl_steps_bytes<C>();
}
template <class C, int DL, int NP>
constexpr size_t mtg_dl_lds_bytes() { return NP * mtg_dl_pair_bytes<C, DL>(); }

// Input loads of one lane's half-chain: 32-bit byte offsets from the two wave-uniform base pointers (global_load with an
// SGPR base), one add per load.  Needs 8 * B * (n_fixed * DL) < 4 GiB.  Two canonical layouts, chosen per launch
// (wave-uniform `aos`): SoA (times[K][B], d_fixed[DL][n_fixed][B]: a wave's loads are contiguous) and AoS (times[B][K],
// d_fixed[B][DL][n_fixed] -- the reference's natural order, one trajectory after the other: a lane's consecutive loads walk
// through its own rows, the sectors are shared by the loads that follow).
template <class C, int DIR>
__device__ __forceinline__ void mtg_dl_preload(const double* __restrict__ times, const double* __restrict__ dfix,
                                               unsigned B, unsigned b, unsigned d, unsigned dl, int aos,
                                               double (&T)[C::KCS], double (&fx)[1][C::NC]) {
  // B: row stride of the SoA layouts in elements (the batch size, or -- padded SoA -- its multiple of 16: mtg_solve_dl_body)
  constexpr int KC = DIR > 0 ? C::KA : C::KB;
  constexpr int c0 = DIR > 0 ? C::colBeginA : C::colBeginB;
  constexpr int nc = DIR > 0 ? C::NCA : C::NCB;
  constexpr unsigned k0 = DIR > 0 ? 0u : (unsigned)(C::KT - 1);
  const unsigned step = aos ? 8u : B * 8u;
  unsigned ot = aos ? (b * (unsigned)C::KT + k0) * 8u : (k0 * B + b) * 8u;
  // Both directions write EVERY element of T and fx (zeros beyond their own count): with different store counts in the two
  // instantiations (odd K: KA = KB + 1, NCA = NCB + 1) the optimiser sinks the common tail of the two branches into one
  // block that stores through a phi of element POINTERS, and the arrays can no longer be promoted to registers (odd-K
  // kernels ran with T / fx in scratch).
#pragma unroll
  for (int j = 0; j < C::KCS; ++j) {
    if (j < KC) {
      T[j] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(times) + ot);
      ot = DIR > 0 ? ot + step : ot - step;
    } else {
      T[j] = 0.0;
    }
  }
  unsigned of = aos ? ((b * dl + d) * (unsigned)C::offFEnd + (unsigned)c0) * 8u : ((d * (unsigned)C::offFEnd + (unsigned)c0) * B + b) * 8u;
#pragma unroll
  for (int c = 0; c < C::NC; ++c) {
    if (c < nc) {
      fx[0][c] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(dfix) + of);
      of += step;
    } else {
      fx[0][c] = 0.0;
    }
  }
}


// C: static configuration with C::D == 1 (one dimension per lane); DL: dimensions of the plan (lanes per trajectory);
// NP: (tile, direction-pair) units per workgroup (2: four waves, one per SIMD of a CU; 1 where two slabs pairs do not fit
// the LDS).  AUX: cache policy bits of the coefficient stores (0 write-back, 1 sc0, 2 nt, 16 sc1).
// QUEUE (mtg_solve_dl_queue_kernel, mtg_solve_linear_sequence): `ntiles` counts the tiles of ALL batches of the queue
// (batch-major, q->tiles_per_batch each); a wave's tile index -> (batch, tile inside it) is advanced incrementally
// (wave-uniform) and the batch's pointer triple comes from the kernel arguments.
// extra outputs of a launch (OUT bits 0 / 1: cost, d_P): destination pointers (either may be null) and the d_P strides
struct MtgDlExtra { double* dfree; double* cost; long long ps_b, ps_d, ps_c; };

template <class C, int DL, int NP, int OUT, int AUX, bool QUEUE>
__device__ __forceinline__ void mtg_solve_dl_body(const double* __restrict__ times, const double* __restrict__ dfix,
                                                  double* __restrict__ coeffs, int* status, int* traj_status, int B, int ntiles,
                                                  int nwg, double* ws, int aos, const MtgSeqQueue* q,
                                                  const MtgDlExtra* xo = nullptr) {
  static_assert(C::kStatic && C::D == 1 && C::KT >= 2, "dimension-in-lane form: static one-dimension configurations");
  static_assert(DL >= 1 && DL <= 4, "1..4 dimensions per trajectory");
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  constexpr int TPW = kWave / DL;
  // Layout word, bit 1: SoA with the row stride PADDED to a multiple of 16 trajectories (times[K][Bs], d_fixed[DL][n_fixed][Bs],
  // Bs = (B + 15) & ~15).  A tile's row pieces (16 or 21 trajectories x 8 bytes) then start on 128-byte boundaries whatever B is;
  // with the plain stride B the pieces of B = 12 500 (BASELINE config 5 per GPU: 100 000-byte rows) straddle two 128-byte lines
  // each -- measured 1.75x the algorithmic input bytes, 1.20x in total, on a kernel that runs at the HBM ceiling on ACTUAL traffic
  // (profiles/r04_config5_pmc_traffic.json).
  const unsigned Bs = (aos & 2) ? (((unsigned)B + 15u) & ~15u) : (unsigned)B;
  aos &= 1;
  const int lane = threadIdx.x & (kWave - 1);
  const int w = threadIdx.x >> 6;      // wave-uniform
  const int pair = w >> 1, dir = w & 1;
  int d = lane / TPW, t = lane - d * TPW;
  const bool dup = d >= DL;            // surplus lanes (64 % DL) duplicate the last lane's work, outputs suppressed
  if (dup) { d = DL - 1; t = TPW - 1; }
  MtgParams P;
  P.times = times; P.ts_b = aos ? C::KT : 1; P.ts_k = aos ? 1 : Bs;
  P.dfix = dfix; P.fs_b = aos ? DL * C::offFEnd : 1; P.fs_c = aos ? 1 : Bs; P.fs_d = aos ? C::offFEnd : (long long)C::offFEnd * Bs;
  P.coeffs = coeffs;
  P.dfree = nullptr; P.ps_b = P.ps_d = P.ps_c = 0;
  P.cost = nullptr; P.ws = ws; P.ws_stride = (long long)nwg * (NP * 2 * kWave);
  if constexpr ((OUT & 3) != 0) {     // cost: per-lane partial sums (dimension lane x direction) added atomically to cost[b] (zeroed by the host)
    P.dfree = xo->dfree; P.ps_b = xo->ps_b; P.ps_d = xo->ps_d; P.ps_c = xo->ps_c;
    P.cost = xo->cost;
  }
  double* wsl0 = C::WSJ > 0 ? ws + (size_t)blockIdx.x * (NP * 2 * kWave) + threadIdx.x : nullptr;   // long chains (C::WSJ): this lane's workspace column
  P.ws_share = (long long)t - lane;   // (dup lanes: their clamped trajectory's columns)
  P.status = status; P.tstatus = traj_status;
  P.vmask = nullptr; P.offF = nullptr; P.offP = nullptr;
  P.B = B; P.K = C::KT; P.Dtot = DL; P.dim0 = d;   // dim0 is a per-lane value here
  P.deriv = C::DV; P.h1off = C::H1OFF; P.ainvoff = C::AINVOFF;
  P.pert_on = 0; P.pert_seg = -1; P.pert_tpv = 1; P.pert_h = P.pert_corr = P.pert_lo = 0.0;
  P.rhs = nullptr; P.rh_b = P.rh_d = P.rh_c = 0;

  const int nunits = (ntiles + NP - 1) / NP;
  MtgLane<C> ln;
  // QUEUE: this wave's tile as (batch, tile inside the batch); the clamp of a surplus tile (odd tile count, NP == 2) lands
  // on the last tile of the last batch
  const int tpb = QUEUE ? q->tiles_per_batch : ntiles;
  struct Where { int batch, local; const double* t; const double* f; double* c; };
  auto tile_of = [&](int it) { const int tl = NP * it + pair; return tl < ntiles ? tl : ntiles - 1; };
  auto bind = [&](Where& w) {
    if constexpr (QUEUE) {
      const MtgSeqItem it = q->item[w.batch];
      w.t = it.times; w.f = it.dfix; w.c = it.coeffs;
    } else {
      w.t = times; w.f = dfix; w.c = coeffs;
    }
  };
  auto locate = [&](int tile_) {        // absolute (first unit) ...
    Where w;
    if constexpr (QUEUE) { w.batch = tile_ / tpb; w.local = tile_ - w.batch * tpb; } else { w.batch = 0; w.local = tile_; }
    bind(w);
    return w;
  };
  auto advance = [&](const Where& from, int delta) {   // ... then relative to the previous unit's tile
    Where w = from;
    w.local += delta;
    if constexpr (QUEUE) {
      while (w.local >= tpb) { w.local -= tpb; ++w.batch; }
      bind(w);
    }
    return w;
  };
  auto fetch = [&](const Where& w, double (&T_)[C::KCS], double (&fx_)[1][C::NC]) {
    unsigned bb = (unsigned)w.local * TPW + t;
    if (bb >= (unsigned)B) bb = B - 1;
    if (dir == 0) mtg_dl_preload<C, 1>(w.t, w.f, Bs, bb, (unsigned)d, (unsigned)DL, aos, T_, fx_);
    else mtg_dl_preload<C, -1>(w.t, w.f, Bs, bb, (unsigned)d, (unsigned)DL, aos, T_, fx_);
  };
  Where cur{0, 0, times, dfix, coeffs};
  int tile_prev = 0;
  if ((int)blockIdx.x < nunits) {
    tile_prev = tile_of(blockIdx.x);
    cur = locate(tile_prev);
    fetch(cur, ln.T, ln.fx);
  }
  const int lane_io = lane, t_io = t, d_io = d;
  constexpr int mm = C::MI;
  constexpr int fmid = C::H - C::popc(mm);
  constexpr int nslots = fmid * (fmid + 1) / 2 + fmid;
  // LDS per pair: [slab A][slab B]; the exchange buffer a direction publishes lives in the OTHER direction's slab
  // (read by that direction before it writes its first coefficient row; the end-of-tile barrier orders reuse)
  char* base = lds_raw + (size_t)pair * mtg_dl_pair_bytes<C, DL>();
  constexpr size_t half = mtg_dl_half_bytes<C, DL>();
  char* my_slab = base + (size_t)dir * half;
  P.lds_steps = (unsigned)(size_t)(base + 2 * half + (size_t)dir * mtg_dl_steps_bytes<C>()) + (unsigned)lane * 8u;
  double* mine = reinterpret_cast<double*>(base + (size_t)(1 - dir) * half) + lane_io;
  const double* other = reinterpret_cast<const double*>(my_slab) + lane_io;
  MtgSlabOut<C, DL, 1, AUX, MTG_DL_PEND> ioA;
  MtgSlabOut<C, DL, -1, AUX, MTG_DL_PEND> ioB;
  ioA.init(my_slab, lane_io, t_io, d_io);
  ioB.init(my_slab, lane_io, t_io, d_io);
  for (int it = blockIdx.x; it < nunits; it += nwg) {
    const bool has_next = it + nwg < nunits;
    Where nxt = cur;
    if (has_next) {
      const int tile_next = tile_of(it + nwg);
      nxt = advance(cur, tile_next - tile_prev);
      tile_prev = tile_next;
    }
    if constexpr (QUEUE) { P.times = cur.t; P.dfix = cur.f; P.coeffs = cur.c; }
    const long long b0 = (long long)cur.local * TPW;
    const long long bl = b0 + t;
    // (the surplus unit of an odd tile count, NP == 2, repeats the last tile: same coefficients, but no second cost term)
    const bool active = bl < B && !dup && NP * it + pair < ntiles;
    const long long b = bl < B ? bl : B - 1;
    // the workspace column pointer is re-defined opaquely per tile: otherwise every one of the ~(f*f + f) * WSJ store and
    // load addresses derived from it is loop-invariant, gets hoisted out of the tile loop and spills (measured: 233
    // scratch stores in the prologue of the K = 32 kernel)
    double* wsl = wsl0;
    if constexpr (C::WSJ > 0) asm volatile("" : "+v"(wsl));
    if (dir == 0) mtg_lane_forward<C, 1>(P, b, ln, wsl, false);
    else mtg_lane_forward<C, -1>(P, b, ln, wsl, false);
    mtg_pack_mid<C>(ln, mm, mine, kWave);
    __syncthreads();
    [[maybe_unused]] double part = 0.0;
    if (dir == 0) {
      ioA.begin_tile(cur.c, b0, B);
      mtg_lane_finish<C, 1, OUT>(P, b, ln, wsl, other, kWave, ioA, active, (OUT & 1) ? &part : nullptr);
    } else {
      ioB.begin_tile(cur.c, b0, B);
      mtg_lane_finish<C, -1, OUT>(P, b, ln, wsl, other, kWave, ioB, active, (OUT & 1) ? &part : nullptr);
    }
    if constexpr ((OUT & 1) != 0) {
      // cost of this half-chain: the DL dimension lanes of a trajectory summed in a fixed order by the dimension-0 lane,
      // then ONE atomic add per direction onto the zeroed cost[b] -- 0 + a + b in either order: bit-reproducible
#if defined(__HIP_DEVICE_COMPILE__)
      double sum = part;
#pragma unroll
      for (int k = 1; k < DL; ++k) sum += mtg_bperm(4 * ((lane_io + k * TPW) & 63), part);
      if (P.cost != nullptr && active && lane_io < TPW) atomicAdd(P.cost + b, sum);
#endif
    }
    if (has_next) {
      fetch(nxt, ln.T, ln.fx);
      cur = nxt;
    }
    // (This barrier comes out as s_waitcnt vmcnt(0) + s_barrier -- it sits through the loads just issued and the last coefficient
    // stores' acknowledgements.  Round 5 measured three hand-overs that avoid that -- an LDS-only barrier, an L2 warm-up of the
    // next tile's inputs and the first steps' inputs loaded early into registers of their own (commit 31baf6e) -- and none moves
    // the launch time: what the end of a tile waits for is the other direction's wave, whose back-substitution finishes up to
    // 20 % apart -- profiles/r05_long_timeline.txt.)
    __syncthreads();
  }
}

template <class C, int DL, int NP, int OUT, int AUX>
__global__ __launch_bounds__(NP * 2 * kWave, 1) void mtg_solve_dl_kernel(const double* __restrict__ times,
                                                                            const double* __restrict__ dfix,
                                                                            double* __restrict__ coeffs, int* status,
                                                                            int* traj_status, int B, int ntiles, int nwg,
                                                                            int aos, double* ws   // (aos: the 14th dword, still preloaded; ws is first needed late)
) {
  mtg_solve_dl_body<C, DL, NP, OUT, AUX, false>(times, dfix, coeffs, status, traj_status, B, ntiles, nwg, ws, aos, nullptr, nullptr);
}

// solves that also return the cost and / or d_P (OUT = 3; round 3): same body, the cost as per-lane partial sums (dimension
// lane x direction) added atomically to cost[b], d_P written by the lane that owns the vertex and the dimension
template <class C, int DL, int NP, int AUX>
__global__ __launch_bounds__(NP * 2 * kWave, 1) void mtg_solve_dl_extra_kernel(const double* __restrict__ times,
                                                                              const double* __restrict__ dfix,
                                                                              double* __restrict__ coeffs, int* status,
                                                                              int* traj_status, int B, int ntiles, int nwg,
                                                                              int aos, double* ws, MtgDlExtra xo) {
  mtg_solve_dl_body<C, DL, NP, 3, AUX, false>(times, dfix, coeffs, status, traj_status, B, ntiles, nwg, ws, aos, nullptr, &xo);
}

// the queue form: same body, the batches' pointer triples in the kernel arguments (mtg_solve_linear_sequence)
template <class C, int DL, int NP, int OUT, int AUX>
__global__ __launch_bounds__(NP * 2 * kWave, 1) void mtg_solve_dl_queue_kernel(int* status, int B, int ntiles, int nwg,
                                                                                  int aos, double* ws, MtgSeqQueue q) {
  mtg_solve_dl_body<C, DL, NP, OUT, AUX, true>(nullptr, nullptr, nullptr, status, nullptr, B, ntiles, nwg, ws, aos, &q);
}

// ---- cross-structure launch: the buckets of a mixed request (BASELINE config 4: N in {8, 10, 12} x K in {4, 8, 16, 32}) in ONE
// launch, each unit (one tile = 64 / DL trajectories, both chain directions) running its own static configuration.  The
// rolled (run-time K) merged launches stream every step's back-substitution data through the workspace and are bound by
// that traffic (30k trajectories: ~310 MB at ~3.5 TB/s); the static bodies keep it in registers.  Streams are no
// alternative on this runtime (tools/micro/stream_overlap.hip: two kernels overlap at best).
// one unit; workgroup = 2 waves (direction A / B) of one tile.  lds: the workgroup's dynamic LDS (>= mtg_dl_pair_bytes<C, DL>()).
template <class C, int DL, int AUX>
__device__ __forceinline__ void mtg_dl_any_unit(const MtgDlAnyItem& it, int tile, int* status, double* wsl0, long long ws_stride,
                                                char* lds_raw) {
  constexpr int TPW = kWave / DL;
  // the lane index is re-defined opaquely per unit: everything derived from it (lane -> (dimension, trajectory), the slab's
  // address maps) would otherwise be loop-invariant, hoisted in front of the unit loop for all twelve bodies at once and
  // spilled (measured: 855 spilled registers)
  int lane = threadIdx.x & (kWave - 1);
  asm volatile("" : "+v"(lane));
  const int dir = threadIdx.x >> 6;    // wave-uniform
  int d = lane / TPW, t = lane - d * TPW;
  const bool dup = d >= DL;
  if (dup) { d = DL - 1; t = TPW - 1; }
  const int B = it.B;
  MtgParams P;
  const int aos = it.aos;
  P.times = it.times; P.ts_b = aos ? C::KT : 1; P.ts_k = aos ? 1 : B;
  P.dfix = it.dfix; P.fs_b = aos ? DL * C::offFEnd : 1; P.fs_c = aos ? 1 : B; P.fs_d = aos ? C::offFEnd : (long long)C::offFEnd * B;
  P.coeffs = it.coeffs;
  P.dfree = nullptr; P.ps_b = P.ps_d = P.ps_c = 0;
  P.cost = nullptr; P.ws = nullptr; P.ws_stride = ws_stride;
  P.ws_share = (long long)t - lane;
  P.status = status; P.tstatus = nullptr;
  P.vmask = nullptr; P.offF = nullptr; P.offP = nullptr;
  P.B = B; P.K = C::KT; P.Dtot = DL; P.dim0 = d;
  P.deriv = C::DV; P.h1off = C::H1OFF; P.ainvoff = C::AINVOFF;
  P.pert_on = 0; P.pert_seg = -1; P.pert_tpv = 1; P.pert_h = P.pert_corr = P.pert_lo = 0.0;
  P.rhs = nullptr; P.rh_b = P.rh_d = P.rh_c = 0;
  MtgLane<C> ln;
  const long long b0 = (long long)tile * TPW;
  const long long bl = b0 + t;
  const bool active = bl < B && !dup;
  const long long b = bl < B ? bl : B - 1;
  if (dir == 0) mtg_dl_preload<C, 1>(it.times, it.dfix, (unsigned)B, (unsigned)b, (unsigned)d, (unsigned)DL, aos, ln.T, ln.fx);
  else mtg_dl_preload<C, -1>(it.times, it.dfix, (unsigned)B, (unsigned)b, (unsigned)d, (unsigned)DL, aos, ln.T, ln.fx);
  constexpr int mm = C::MI;
  constexpr size_t half = mtg_dl_half_bytes<C, DL>();
  char* my_slab = lds_raw + (size_t)dir * half;
  P.lds_steps = (unsigned)(size_t)(lds_raw + 2 * half + (size_t)dir * mtg_dl_steps_bytes<C>()) + (unsigned)lane * 8u;
  double* mine = reinterpret_cast<double*>(lds_raw + (size_t)(1 - dir) * half) + lane;
  const double* other = reinterpret_cast<const double*>(my_slab) + lane;
  double* wsl = wsl0;
  if constexpr (C::WSJ > 0) asm volatile("" : "+v"(wsl));
  if (dir == 0) mtg_lane_forward<C, 1>(P, b, ln, wsl, false);
  else mtg_lane_forward<C, -1>(P, b, ln, wsl, false);
  mtg_pack_mid<C>(ln, mm, mine, kWave);
  __syncthreads();
  if (dir == 0) {
    MtgSlabOut<C, DL, 1, AUX> io;
    io.init(my_slab, lane, t, d);
    io.begin_tile(it.coeffs, b0, B);
    mtg_lane_finish<C, 1, 0>(P, b, ln, wsl, other, kWave, io, active);
  } else {
    MtgSlabOut<C, DL, -1, AUX> io;
    io.init(my_slab, lane, t, d);
    io.begin_tile(it.coeffs, b0, B);
    mtg_lane_finish<C, -1, 0>(P, b, ln, wsl, other, kWave, io, active);
  }
  // (the caller's end-of-unit barrier frees the LDS)
}

// The configurations a cross-structure launch can hold (index = MtgDlAnyItem::cfg): X(index, H, K, MS, MI, ME, DV, WS, LS, RS),
// DL = 3; (WS, LS, RS) as in mtg_dimlane_variants.inc
#define MTG_DL_ANY_LIST(X)                \
  X(0, 4, 4, 15, 1, 15, 3, 0, 0, 0)       \
  X(1, 4, 8, 15, 1, 15, 3, 0, 0, 0)       \
  X(2, 4, 16, 15, 1, 15, 3, 0, 0, 0)      \
  X(3, 4, 32, 15, 1, 15, 3, 4, 4, 0)      \
  X(4, 5, 4, 31, 1, 31, 4, 0, 0, 0)       \
  X(5, 5, 8, 31, 1, 31, 4, 0, 0, 0)       \
  X(6, 5, 16, 31, 1, 31, 4, 0, 0, 0)      \
  X(7, 5, 32, 31, 1, 31, 4, 4, 4, 1)      \
  X(8, 6, 4, 63, 1, 63, 5, 0, 0, 0)       \
  X(9, 6, 8, 63, 1, 63, 5, 0, 0, 0)       \
  X(10, 6, 16, 63, 1, 63, 5, 3, 3, 0)     \
  X(11, 6, 32, 63, 1, 63, 5, 11, 4, 1)

// The host hands every persistent workgroup its own list of units (wg_begin[w] .. wg_begin[w + 1] of `units`), longest chains
// first: a greedy longest-processing-time assignment over the workgroups with a per-configuration cost estimate, so that no
// workgroup is left with a long chain at the end (round 2 took units w, w + grid, ... of the sorted list: the workgroups that
// started with the 37-us chains of N = 12 / K = 32 still got two more units each).  The first unit of neighbouring workgroups
// (same CU, same instruction cache) has the same configuration.  (Measured and rejected in round 2: a global unit counter,
// i.e. dynamic scheduling -- 81 us against 71 us for config 4 at 30k.)
template <int AUX>
__global__ __launch_bounds__(2 * kWave, 1) void mtg_solve_dl_any_kernel(const MtgDlAnyItem* __restrict__ items,
                                                                      const MtgDlAnyUnit* __restrict__ units,
                                                                      const int* __restrict__ wg_begin, int* status, double* ws) {
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  double* wsl0 = ws + (size_t)blockIdx.x * (2 * kWave) + threadIdx.x;
  const long long ws_stride = (long long)gridDim.x * (2 * kWave);
  const int u1 = wg_begin[blockIdx.x + 1];
  for (int u = wg_begin[blockIdx.x]; u < u1; ++u) {
    const MtgDlAnyUnit un = units[u];
    const MtgDlAnyItem it = items[un.item];
    switch (__builtin_amdgcn_readfirstlane(it.cfg)) {
#define MTG_X(I, H, K, MS, MI, ME, DV, WS, LS, RS) \
      case I: mtg_dl_any_unit<MtgCfg<H, 1, K, MS, MI, ME, DV, 0, WS, ((WS > 0 || RS) ? 3 : 0), LS, RS>, 3, AUX>(it, un.tile, status, wsl0, ws_stride, lds_raw); break;
      MTG_DL_ANY_LIST(MTG_X)
#undef MTG_X
      default: break;
    }
    __syncthreads();           // the unit's LDS is free again
  }
}
#endif  // MTG_DIMLANE_H_
