// mtg_kernels.h -- __global__ kernel templates shared by the per-variant translation units.
#ifndef MTG_KERNELS_H_
#define MTG_KERNELS_H_
#include <hip/hip_runtime.h>

#include "mtg_lane.h"
#include "mtg_slab.h"

#ifndef MTG_STORE_AUX
#define MTG_STORE_AUX 0   // cache-policy bits of the coefficient stores (16 = sc1 write-through; A/B knob)
#endif

constexpr int kWave = 64;
constexpr int kBlock = 2 * kWave;  // wave 0: direction A (forward), wave 1: direction B

// Coefficient output staged through LDS.  Each lane drops the D*N coefficients of the segment it
// just recovered into its row of the wave's staging buffer (ds_write_b128, row stride an odd number
// of 16-byte units => conflict-free), then the whole wave streams the buffer out: chunk o of 16 bytes
// belongs to trajectory t = o / Q at offset r = o % Q of that trajectory's contiguous D*N*8-byte piece
// coeffs[b0 + t][seg][dim0 .. dim0+D)[0..N), so a store instruction covers 64 consecutive chunks
// (~9 cache lines for the 240-byte pieces of N=10, D=3) instead of 64 lines with row-per-lane stores
// (measured on MI355X, B = 1M: 869 us -> 612 us; profiles/).
//
// The rows of a segment are written (commit) one back-substitution step before they are streamed out
// (drain), so the LDS write -> read latency overlaps the next step's arithmetic.  The stores are
// buffer_store_dwordx4 through a per-drain descriptor whose base is the tile's first piece and whose
// size ends at the last existing trajectory: per-lane byte offsets are computed once per kernel, and the
// hardware range check drops the chunks of the tail tile's non-existent trajectories (no predication).
// WRITE_THROUGH (kernel OUT bit 2): sc1 stores.  Small launches (one tile per workgroup) otherwise end with all
// their output dirty in L2 and pay the write-back as a serial tail (B = 10k: 13.4 -> 12.0 us); large launches
// are faster with write-back stores (B = 1M: 517 us vs 752 us), so the host picks by tile count.
template <class C, bool WRITE_THROUGH = false>
struct MtgLdsOut {
  static constexpr int Q = C::D * C::N / 2;   // 16-byte chunks per lane per segment
  static constexpr int QP = Q | 1;            // padded row stride (odd)
  typedef double d2 __attribute__((ext_vector_type(2)));
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  double* stage;        // this wave's staging buffer in LDS: 64 rows x QP chunks
  int lane;
  long long b0;         // first trajectory of the tile
  int pending;          // segment whose rows sit in the staging buffer (-1: none)
  unsigned goff[Q];     // byte offset of chunk i*64+lane inside the tile's output: t*piece_bytes + r*16
  unsigned loff[Q];     // byte offset of the same chunk inside the staging buffer
  unsigned piece_bytes; // K*Dtot*N*8: one trajectory's coefficients

  __device__ __forceinline__ void init(const MtgParams& P, double* stage_, int lane_) {
    stage = stage_;
    lane = lane_;
    pending = -1;
    piece_bytes = (unsigned)(mtg_nseg<C>(P) * P.Dtot * C::N) * 8u;
#pragma unroll
    for (int i = 0; i < Q; ++i) {
      const unsigned o = (unsigned)i * 64u + (unsigned)lane;
      const unsigned t = o / (unsigned)Q, r = o - t * (unsigned)Q;
      goff[i] = t * piece_bytes + r * 16u;
      loff[i] = (t * (unsigned)QP + r) * 16u;
    }
  }
  __device__ __forceinline__ double* row(int) { return stage + (size_t)lane * QP * 2; }
  __device__ __forceinline__ void flush(const MtgParams& P) { drain(P); }
  // Cross-lane hand-off through LDS inside one wave: LDS operations of a wave execute in order, so no hardware
  // barrier is needed, but the COMPILER must not move the staged writes / reads across each other (they use
  // different lanes' addresses, which alias analysis cannot see): compiler-level memory fences on both sides.
  __device__ __forceinline__ static void fence() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
  }
  __device__ __forceinline__ void commit(const MtgParams&, int seg) {
    pending = seg;
    fence();
  }
  __device__ __forceinline__ void drain(const MtgParams& P) {
    if (pending < 0) return;
    const int seg = pending;
    pending = -1;
    const long long segoff = ((long long)seg * P.Dtot + P.dim0) * C::N;          // doubles
    char* gbase = reinterpret_cast<char*>(P.coeffs + segoff) + b0 * (long long)piece_bytes;   // wave-uniform
    long long nvalid = P.B - b0;                                                 // trajectories of this tile that exist
    if (nvalid > 64) nvalid = 64;
    int nbytes = (int)(nvalid - 1) * (int)piece_bytes + Q * 16;
    // The descriptor must be provably wave-uniform or hipcc wraps every store in a waterfall loop
    // (cdna_hip_programming.md T20): pin its inputs to SGPRs explicitly.
    {
      const unsigned long long g = reinterpret_cast<unsigned long long>(gbase);
      const unsigned glo = __builtin_amdgcn_readfirstlane((unsigned)g);
      const unsigned ghi = __builtin_amdgcn_readfirstlane((unsigned)(g >> 32));
      gbase = reinterpret_cast<char*>(((unsigned long long)ghi << 32) | glo);
      nbytes = __builtin_amdgcn_readfirstlane(nbytes);
    }
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(gbase, 0, nbytes, 0x00020000);
    const char* sbase = reinterpret_cast<const char*>(stage);
    fence();
    // LDS reads in groups of G ahead of their stores (one lgkmcnt wait per group instead of per chunk)
    constexpr int G = 5;
#pragma unroll
    for (int i0 = 0; i0 < Q; i0 += G) {
      u4 v[G];
#pragma unroll
      for (int i = 0; i < G; ++i) {
        if (i0 + i < Q) v[i] = __builtin_bit_cast(u4, *reinterpret_cast<const d2*>(sbase + loff[i0 + i]));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < G; ++i) {
        if (i0 + i < Q) __builtin_amdgcn_raw_buffer_store_b128(v[i], rsrc, (int)goff[i0 + i], 0, WRITE_THROUGH ? 16 : MTG_STORE_AUX);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    fence();
  }
};

template <class C>
__host__ __device__ constexpr size_t mtg_stage_doubles() { return (size_t)64 * (C::D * C::N / 2 | 1) * 2; }

// Occupancy target (2nd launch-bound = waves per SIMD): the light variants (static, D <= 2 dimensions per
// workgroup) are held to 256 registers so two waves share a SIMD and hide each other's issue gaps.
template <class C>
constexpr int mtg_waves_per_simd() { return (C::kStatic && C::D <= 2) ? 2 : 1; }


template <class C, int OUT>
__global__ __launch_bounds__(kBlock, mtg_waves_per_simd<C>()) void mtg_solve_kernel(MtgParams P, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  // grid.y = dimension groups: workgroup (x, y) solves dimensions [dim0 + y*D, dim0 + (y+1)*D) of its tiles
  // (the factorisation is repeated per group; groups are independent).
  P.dim0 += (int)blockIdx.y * C::D;
  const int lane = threadIdx.x & (kWave - 1);
  const int dir = threadIdx.x >> 6;  // wave-uniform
  // Static variants: the first tile's inputs are requested before ANY other set-up (staging offsets, LDS pointers,
  // workspace slabs): at small batch every wave solves exactly one tile and the input round trip heads its serial
  // latency chain, so nothing may sit in front of the loads.
  MtgLane<C> ln;
  // Perturbed-time launches (mtg_mellinger_cost_gradient): tile -> (variant n, tile of the real batch); the variant's
  // perturbed segment and cost row are set in the tile's own copy of the parameters.  Everything else: identity.
  auto view = [&](int tile_, MtgParams& Pt) -> int {
    Pt = P;
    if constexpr (!C::kPert) return tile_;
    if (!P.pert_on) return tile_;
    const int n = tile_ / P.pert_tpv;
    Pt.pert_seg = n - 1;
    if (P.cost != nullptr) Pt.cost = P.cost + (long long)n * P.B;
    return tile_ - n * P.pert_tpv;
  };
  auto fetch = [&](int tile_, double (&T_)[C::KCS], double (&fx_)[C::D][C::NC]) {
    MtgParams Pt;
    const int tr = view(tile_, Pt);
    long long bb = (long long)tr * kWave + lane;
    if (bb >= P.B) bb = P.B - 1;
    if (dir == 0) mtg_preload_into<C, 1>(Pt, bb, T_, fx_);
    else mtg_preload_into<C, -1>(Pt, bb, T_, fx_);
  };
  if (C::kStatic && (int)blockIdx.x < ntiles) fetch(blockIdx.x, ln.T, ln.fx);
  const int K = mtg_nseg<C>(P);
  const int vm = (K + 1) / 2;
  const int mm = C::kRolled ? C::MI : mtg_mask<C>(P, vm);
  const int nslots = mtg_mid_slots<C>(mm);
  // LDS: [staging A][staging B][exchange A][exchange B]
  MtgLdsOut<C, (OUT & 4) != 0> io;
  io.init(P, lds + (size_t)dir * mtg_stage_doubles<C>(), lane);
  double* xch = lds + 2 * mtg_stage_doubles<C>();
  double* mine = xch + (size_t)dir * nslots * kWave + lane;
  const double* other = xch + (size_t)(1 - dir) * nslots * kWave + lane;
  double* wsl = (P.ws && !C::kStatic)
                    ? P.ws + (((long long)blockIdx.y * gridDim.x + blockIdx.x) * kBlock + threadIdx.x) : nullptr;
  // Software prefetch (register-rich static variants): the inputs of this workgroup's NEXT tile are requested
  // before the current tile is solved and land while it computes (measured: the exposed input latency was
  // ~10k of ~40k cycles per tile at B = 1M).  The light 2-waves-per-SIMD variants have no registers to spare
  // and rely on the co-resident wave instead.
  constexpr bool kPrefetch = C::kStatic && mtg_waves_per_simd<C>() == 1;
  double nT[C::KCS], nfx[C::D][C::NC];
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    MtgParams Pt;
    const int tile_r = view(tile, Pt);
    io.b0 = (long long)tile_r * kWave;
    const long long bl = io.b0 + lane;
    const bool active = bl < P.B;
    const long long b = active ? bl : P.B - 1;   // tail lanes duplicate the last trajectory, outputs suppressed
    const bool has_next = kPrefetch && tile + (int)gridDim.x < ntiles;
    if (has_next) fetch(tile + gridDim.x, nT, nfx);
    const bool need_preload = !kPrefetch && !(C::kStatic && tile == (int)blockIdx.x);   // first tile: fetched at the top
    if (dir == 0) mtg_lane_forward<C, 1>(Pt, b, ln, wsl, need_preload);
    else mtg_lane_forward<C, -1>(Pt, b, ln, wsl, need_preload);
    mtg_pack_mid<C>(ln, mm, mine, kWave);
    __syncthreads();
    if (dir == 0) mtg_lane_finish<C, 1, OUT>(Pt, b, ln, wsl, other, kWave, io, active);
    else mtg_lane_finish<C, -1, OUT>(Pt, b, ln, wsl, other, kWave, io, active);
    if (has_next) {
#pragma unroll
      for (int j = 0; j < C::KCS; ++j) ln.T[j] = nT[j];
#pragma unroll
      for (int dm = 0; dm < C::D; ++dm) {
#pragma unroll
        for (int c = 0; c < C::NC; ++c) ln.fx[dm][c] = nfx[dm][c];
      }
    }
    __syncthreads();
  }
}

// Slab-output form of the fused kernel (round 2): same lanes, same arithmetic as mtg_solve_kernel with all D dimensions
// per lane, but the coefficients leave through MtgSlabOut (mtg_slab.h): a ring of two segment slots per trajectory in LDS,
// and after every recovered segment the 64-byte-aligned range that has become complete is streamed out -- every store
// instruction writes whole sectors, every sector is written once.  mtg_solve_kernel's 240-byte pieces complete most
// sectors from two store instructions (read-modify-write at the memory side once the output is not cache-resident:
// B = 125k 63-66 us resident, 80-83 us with rotating buffers).  Static configurations with K >= 2, all dimensions in one
// workgroup (P.dim0 == 0, P.Dtot == C::D), coefficient output only.  Dynamic LDS = mtg_slab_lds_bytes<C>().
template <class C>
__host__ __device__ constexpr size_t mtg_slab_lds_bytes() {
  constexpr size_t a = (size_t)MtgSlabOut<C, 1, 1, 0, false>::TPW * MtgSlabOut<C, 1, 1, 0, false>::ROWB;
  constexpr size_t b = (size_t)MtgSlabOut<C, 1, -1, 0, false>::TPW * MtgSlabOut<C, 1, -1, 0, false>::ROWB;
  return 2 * (a > b ? a : b);
}

// A queue of batches in ONE launch (mtg_solve_linear_sequence): n independent batches of the same plan, batch size and
// layout; the persistent workgroups walk the tiles of all of them (tile -> (batch, tile inside the batch), batch-major), so
// a wave of batch i + 1 starts the moment a SIMD is free -- no drain / launch gap between the batches (1.0-1.3 us between
// dependent launches; ~1.8k cycles of store acknowledgement at the end of each) and the store tail of batch i runs under
// the forward chains of batch i + 1.  The pointer triples travel in the kernel arguments (no upload in front of the launch).
struct MtgSeqItem { const double* times; const double* dfix; double* coeffs; };
constexpr int kSeqMax = 96;    // batches per launch (kernel arguments <= 4 KB); longer queues are cut into several launches
struct MtgSeqQueue {
  int n, tiles_per_batch;
  MtgSeqItem item[kSeqMax];
};

template <class C, int AUX, bool QUEUE, int OUT = 0>
__device__ __forceinline__ void mtg_solve_slab_body(const MtgParams& P, int ntiles, const MtgSeqQueue* q) {
  static_assert(C::kStatic && C::KT >= 2 && !C::kPert, "slab-output form: static configurations, K >= 2");
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  const int lane = threadIdx.x & (kWave - 1);
  const int dir = threadIdx.x >> 6;  // wave-uniform
  MtgLane<C> ln;
  // QUEUE: (batch, tile inside the batch) of a global tile index, advanced incrementally (wave-uniform, scalar)
  const int tpb = QUEUE ? q->tiles_per_batch : ntiles;
  auto norm = [&](int& bt, int& lc) { if constexpr (QUEUE) { while (lc >= tpb) { lc -= tpb; ++bt; } } };
  auto params_of = [&](int bt) -> MtgParams {
    MtgParams Pt = P;
    if constexpr (QUEUE) {
      const MtgSeqItem it = q->item[bt];
      Pt.times = it.times; Pt.dfix = it.dfix; Pt.coeffs = it.coeffs;
    }
    return Pt;
  };
  auto fetch = [&](const MtgParams& Pt, int tile_, double (&T_)[C::KCS], double (&fx_)[C::D][C::NC]) {
    long long bb = (long long)tile_ * kWave + lane;
    if (bb >= P.B) bb = P.B - 1;
    if (dir == 0) mtg_preload_into<C, 1>(Pt, bb, T_, fx_);
    else mtg_preload_into<C, -1>(Pt, bb, T_, fx_);
  };
  int batch = 0, local = blockIdx.x;
  norm(batch, local);
  MtgParams Pc = params_of(batch);
  if ((int)blockIdx.x < ntiles) fetch(Pc, local, ln.T, ln.fx);
  constexpr int mm = C::MI;
  constexpr int fmid = C::H - C::popc(mm);
  constexpr int nslots = fmid * (fmid + 1) / 2 + C::D * fmid;
  constexpr size_t half = mtg_slab_lds_bytes<C>() / 2;
  static_assert((size_t)nslots * kWave * sizeof(double) <= half, "the exchange buffer lives in the other direction's slab");
  char* my_slab = lds_raw + (size_t)dir * half;
  double* mine = reinterpret_cast<double*>(lds_raw + (size_t)(1 - dir) * half) + lane;
  const double* other = reinterpret_cast<const double*>(my_slab) + lane;
  MtgSlabOut<C, 1, 1, AUX, false> ioA;
  MtgSlabOut<C, 1, -1, AUX, false> ioB;
  ioA.init(my_slab, lane, lane, 0);
  ioB.init(my_slab, lane, lane, 0);
  double nT[C::KCS], nfx[C::D][C::NC];
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long b0 = (long long)local * kWave;
    const long long bl = b0 + lane;
    const bool active = bl < P.B;
    const long long b = active ? bl : P.B - 1;   // tail lanes duplicate the last trajectory, outputs suppressed
    const bool has_next = tile + (int)gridDim.x < ntiles;
    int nbatch = batch, nlocal = local + (int)gridDim.x;
    norm(nbatch, nlocal);
    MtgParams Pn = P;
    if (has_next) {   // the next tile's inputs land while this one is solved
      Pn = params_of(nbatch);
      fetch(Pn, nlocal, nT, nfx);
    }
    if (dir == 0) mtg_lane_forward<C, 1>(Pc, b, ln, nullptr, false);
    else mtg_lane_forward<C, -1>(Pc, b, ln, nullptr, false);
    mtg_pack_mid<C>(ln, mm, mine, kWave);
    __syncthreads();
    if (dir == 0) {
      ioA.begin_tile(Pc.coeffs, b0, P.B);
      mtg_lane_finish<C, 1, OUT>(Pc, b, ln, nullptr, other, kWave, ioA, active);
    } else {
      ioB.begin_tile(Pc.coeffs, b0, P.B);
      mtg_lane_finish<C, -1, OUT>(Pc, b, ln, nullptr, other, kWave, ioB, active);
    }
    if (has_next) {
#pragma unroll
      for (int j = 0; j < C::KCS; ++j) ln.T[j] = nT[j];
#pragma unroll
      for (int dm = 0; dm < C::D; ++dm) {
#pragma unroll
        for (int c = 0; c < C::NC; ++c) ln.fx[dm][c] = nfx[dm][c];
      }
    }
    if constexpr (QUEUE) { batch = nbatch; local = nlocal; Pc = Pn; }
    else local = nlocal;
    __syncthreads();
  }
}

// OUT: as for mtg_solve_kernel (bit 0: cost, bit 1: d_P output)
template <class C, int AUX, int OUT = 0>
__global__ __launch_bounds__(kBlock, 1) void mtg_solve_slab_kernel(MtgParams P, int ntiles) {
  mtg_solve_slab_body<C, AUX, false, OUT>(P, ntiles, nullptr);
}

template <class C, int AUX>
__global__ __launch_bounds__(kBlock, 1) void mtg_solve_slab_queue_kernel(MtgParams P, int ntiles, MtgSeqQueue q) {
  mtg_solve_slab_body<C, AUX, true>(P, ntiles, &q);
}

// Several plans in ONE launch (BASELINE config 4: a mixed request whose buckets share N, D, masks and derivative but not
// K).  Rolled configurations only: K is a run-time field of MtgParams, so tiles of different buckets run the same code.
// `table[bucket]` holds the bucket's parameters (pointers, strides, B, K; all buckets share one workspace sized for the
// longest chain), `tiles[t]` names the bucket and the tile inside it; the host sorts tiles longest-chain-first so that
// the short ones fill in behind the long ones.  One launch instead of one per bucket: a 2500-trajectory bucket is 40
// tiles, far too few to fill 256 CUs, and back-to-back launches each pay their own latency chain.
struct MtgTileRef { int bucket, tile, cfg; };   // cfg: configuration index inside a cross-structure launch (mtg_solve_multi_any_kernel)

// one tile of a multi-plan launch (both waves of the workgroup; ends with a workgroup barrier: the LDS is free again)
template <class C, int OUT>
__device__ __forceinline__ void mtg_multi_tile(const MtgParams* __restrict__ table, const MtgTileRef ref, double* lds,
                                               int lane, int dir, int dimgroup, long long wg_linear) {
  static_assert(C::kRolled, "multi-plan launches use the rolled (run-time K) configurations");
  constexpr int kFreeMid = C::H - C::popc(C::MI);                     // the middle vertex is an interior one
  constexpr int nslots = kFreeMid * (kFreeMid + 1) / 2 + C::D * kFreeMid;
  double* xch = lds + 2 * mtg_stage_doubles<C>();
  double* mine = xch + (size_t)dir * nslots * kWave + lane;
  const double* other = xch + (size_t)(1 - dir) * nslots * kWave + lane;
  MtgLane<C> ln;
  MtgParams P = table[ref.bucket];
  P.dim0 += dimgroup * C::D;
  double* wsl = P.ws + (wg_linear * kBlock + threadIdx.x);
  MtgLdsOut<C, (OUT & 4) != 0> io;
  io.init(P, lds + (size_t)dir * mtg_stage_doubles<C>(), lane);
  io.b0 = (long long)ref.tile * kWave;
  const long long bl = io.b0 + lane;
  const bool active = bl < P.B;
  const long long b = active ? bl : P.B - 1;   // tail lanes duplicate the last trajectory, outputs suppressed
  if (dir == 0) mtg_lane_forward<C, 1>(P, b, ln, wsl, true);
  else mtg_lane_forward<C, -1>(P, b, ln, wsl, true);
  mtg_pack_mid<C>(ln, C::MI, mine, kWave);
  __syncthreads();
  if (dir == 0) mtg_lane_finish<C, 1, OUT>(P, b, ln, wsl, other, kWave, io, active);
  else mtg_lane_finish<C, -1, OUT>(P, b, ln, wsl, other, kWave, io, active);
  __syncthreads();
}

template <class C, int OUT>
__global__ __launch_bounds__(kBlock, mtg_waves_per_simd<C>()) void mtg_solve_multi_kernel(const MtgParams* __restrict__ table,
                                                                                         const MtgTileRef* __restrict__ tiles,
                                                                                         int ntiles) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x & (kWave - 1);
  const int dir = threadIdx.x >> 6;  // wave-uniform
  // grid.y = dimension groups, as in mtg_solve_kernel
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x)
    mtg_multi_tile<C, OUT>(table, tiles[t], lds, lane, dir, (int)blockIdx.y, (long long)blockIdx.y * gridDim.x + blockIdx.x);
}

// Cross-structure launch: the buckets of a mixed request differ in N as well (BASELINE config 4: N in {8, 10, 12}).  Streams
// do not help here -- measured on this runtime (tools/micro/stream_overlap.hip): kernels on different HIP streams overlap
// two at a time at best, a fork/join over events costs ~23 us of device time and ~10 us of host time per stream -- so
// the whole request is ONE launch: every tile carries the index of its (rolled) configuration and the workgroup
// branches (wave-uniformly) into that configuration's code.  DG = dimensions per workgroup (3: fused, 1: split form).
// One-dimensional grid of persistent workgroups over the (tile, dimension group) units, dimension fastest: the units are
// sorted longest-chain-first, so every workgroup's first unit is a long one and the short ones fill in behind (with
// dimension groups in grid.y the second round of workgroups would start with long chains again).
template <int DG, int OUT>
__global__ __launch_bounds__(kBlock, 1) void mtg_solve_multi_any_kernel(const MtgParams* __restrict__ table,
                                                                        const MtgTileRef* __restrict__ tiles, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x & (kWave - 1);
  const int dir = threadIdx.x >> 6;  // wave-uniform
  constexpr int NG = 3 / DG;         // dimension groups of a 3-dimensional plan
  for (int u = blockIdx.x; u < ntiles * NG; u += gridDim.x) {
    const MtgTileRef ref = tiles[u / NG];
    const int dg = u % NG;
    switch (__builtin_amdgcn_readfirstlane(ref.cfg)) {   // the configurations of mtg_any_cfg_index, same order
      case 0: mtg_multi_tile<MtgCfg<4, DG, -1, 15, 1, 15, 3>, OUT>(table, ref, lds, lane, dir, dg, blockIdx.x); break;
      case 1: mtg_multi_tile<MtgCfg<5, DG, -1, 31, 1, 31, 4>, OUT>(table, ref, lds, lane, dir, dg, blockIdx.x); break;
      default: mtg_multi_tile<MtgCfg<6, DG, -1, 63, 1, 63, 5>, OUT>(table, ref, lds, lane, dir, dg, blockIdx.x); break;
    }
  }
}

// setFreeConstraints path: one wave = 64 trajectories, one lane per trajectory, recovery only; coefficients leave
// through the same LDS-staged coalesced drain as the solve kernel.  Dynamic LDS = mtg_stage_doubles<C>() doubles.
template <class C, int OUT>
__global__ __launch_bounds__(kWave) void mtg_update_kernel(MtgParams P, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  P.dim0 += (int)blockIdx.y * C::D;
  const int lane = threadIdx.x;
  MtgLdsOut<C> io;
  io.init(P, lds, lane);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    io.b0 = (long long)tile * kWave;
    const long long bl = io.b0 + lane;
    const bool active = bl < P.B;
    mtg_lane_update<C, OUT>(P, active ? bl : P.B - 1, io, active);
  }
}


// setFreeConstraints path with whole-sector output (round 3): the same lanes and arithmetic as mtg_update_kernel, the
// coefficients leave through MtgSlabOutRt (run-time K, one lane per trajectory with all D dimensions, every segment in ascending
// order: "direction B with KA = 0") instead of the per-segment staging whose 240-byte pieces complete most sectors from two store
// instructions.  PHASE: instantiation for pieces of K * D * N * 8 bytes that are not a multiple of 64 bytes.
template <class C>
__host__ __device__ constexpr size_t mtg_update_slab_lds_bytes() {
  return ((size_t)64 * MtgSlabOutRt<C::N, 1, -1, 0, false, false, C::D>::ROWB + 15) / 16 * 16;
}
template <class C, int OUT, bool PHASE>
__global__ __launch_bounds__(kWave) void mtg_update_slab_kernel(MtgParams P, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  const int lane = threadIdx.x;
  MtgSlabOutRt<C::N, 1, -1, 18, PHASE, false, C::D> io;
  io.init(lds_raw, lane, lane, 0, P.K, 0);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long b0 = (long long)tile * kWave;
    const long long bl = b0 + lane;
    const bool active = bl < P.B;
    io.begin_tile(P.coeffs, b0, P.B);
    mtg_lane_update<C, OUT>(P, active ? bl : P.B - 1, io, active);
    __builtin_amdgcn_s_barrier();     // (one wave: orders the slab's reuse against the flush's LDS reads)
  }
}

using SolveFn = void (*)(MtgParams, int);
using UpdateFn = void (*)(MtgParams, int);
using SolveMultiFn = void (*)(const MtgParams*, const MtgTileRef*, int);
template <int H, int D> using GenericCfg = MtgCfg<H, D, 0, 0, 0, 0>;
template <int H, int D> using GenericCostCfg = MtgCfg<H, D, 0, 0, 0, 0, 0, 1>;   // cost-only launches: perturbed-time capable

// per-TU pickers (mtg_generic_hN.hip, mtg_static.hip)
SolveFn mtg_pick_generic_solve(int h, int d, int out_mode);   // out_mode: 0 plain, 1 extra outputs (OUT 3), 2 cost only (OUT 9)
UpdateFn mtg_pick_generic_update(int h, int d, bool with_cost);
struct MtgStaticEntry {
  int h, d, k, ms, mi, me, dv;
  int heavy;       // static variant that spills: prefer a rolled variant for large launches
  SolveFn fn[5];   // [extra outputs (cost / d_free)] + 2 * [write-through stores]; [4] = cost only (OUT 9)
  void (*upd[2])(MtgParams, int);   // rolled entries: setFreeConstraints kernel [with cost]; static entries: null
  void (*upd_slab[2][2])(MtgParams, int);   // rolled entries with all plan dimensions: the same with whole-sector output, [with cost][piece not a multiple of 64 bytes]
  size_t upd_slab_lds;
  SolveMultiFn multi[4];            // rolled entries: several plans in one launch, [extra outputs] + 2 * [write-through]
};
const MtgStaticEntry* mtg_find_static(int h, int d, int k, int deriv, const int* mask, bool rolled_only = false);
using SolveQueueFn = void (*)(MtgParams, int, MtgSeqQueue);
struct MtgSlabEntry {
  int h, d, k, ms, mi, me, dv;
  size_t lds;
  SolveFn fn[2];   // coefficient store policy: [0] write-back, [1] nt sc1
  SolveQueueFn queue;   // the same (nt sc1) over a queue of batches: mtg_solve_linear_sequence
  SolveFn extra;        // nt sc1 with the extra outputs (OUT = 3: cost and / or d_P), round 3
};
const MtgSlabEntry* mtg_find_slab(int h, int d, int k, int deriv, const int* mask);

// dimension-in-lane launch form (mtg_dimlane.h / mtg_dimlane.hip): canonical SoA inputs, coefficient output (+ status)
struct MtgDimlaneEntry {
  int h, k, ms, mi, me, dv, dl, np;
  int tpw;            // trajectories per wave (64 / dl)
  int lo_per_cu, hi_per_cu;   // default form while lo * CUs <= workgroups <= hi * CUs / 2 (hi = 0: no upper limit; hi counts HALF workgroups per CU)
  size_t lds;         // dynamic LDS per workgroup
  size_t ws_per_lane; // long-chain variants (MtgCfg::WSJ > 0): workspace bytes per resident lane (grid * np * 128 lanes), else 0
  // enqueues one launch on `stream` (a hipStream_t): grid workgroups of np * 128 threads (coefficient stores: nt sc1); aos: input
  // layout (0 canonical SoA, 1 canonical AoS, 2 padded SoA); returns 0 or -1 (attribute / launch set-up failed)
  int (*launch)(void* stream, int grid, const double* times, const double* dfix, double* coeffs, int* status,
                int* traj_status, int B, int ntiles, double* ws, int aos);
  // a queue of batches in one launch (mtg_solve_linear_sequence; main-table variants only, else null): ntiles = tiles of
  // all batches (q->n * q->tiles_per_batch)
  int (*launch_queue)(void* stream, int grid, const MtgSeqQueue* q, int* status, int B, int ntiles, double* ws, int aos);
  // solves that also return the cost and / or d_P (either pointer may be null; cost zeroed by the caller; ps_*: d_P strides
  // in doubles); main-table variants only, else null
  int (*launch_extra)(void* stream, int grid, const double* times, const double* dfix, double* coeffs, int* status,
                      int* traj_status, int B, int ntiles, double* ws, int aos, double* dfree, double* cost, long long ps_b,
                      long long ps_d, long long ps_c);
};
// MTG_FLAG_REFINE (mtg_refine.hip): r = -(R_PP x + R_PF d_F) in double-double for every trajectory and dimension ([B][D][n_free],
// contiguous), and x += delta over the free slots; 0 or -1 (launch failed)
int mtg_refine_residual_launch(void* stream, int H, int K, int D, int deriv, int h1off, const int* d_vmask, const int* d_offF,
                               const int* d_offP, long long B, const double* times, long long ts_b, long long ts_k,
                               const double* dfix, long long fs_b, long long fs_d, long long fs_c, const double* dfree,
                               long long ps_b, long long ps_d, long long ps_c, double* rhs, int n_free);
int mtg_refine_axpy_launch(void* stream, double* x, long long ps_b, long long ps_d, long long ps_c, const double* delta, long long B, int D, int np);
// cross-structure launches (mtg_solve_multi_any_kernel): index of a rolled entry's configuration, or -1; kernel for a
// dimension-group size (1 | 3) and output variant ([extra outputs] + 2 * [write-through])
int mtg_any_cfg_index(const MtgStaticEntry* e);
SolveMultiFn mtg_multi_any_fn(int dg, int variant);
const MtgDimlaneEntry* mtg_find_dimlane(int h, int dl, int k, int deriv, const int* mask);

// cross-structure dimension-in-lane launches (mtg_dimlane.h: mtg_solve_dl_any_kernel)
struct MtgDlAnyItem {     // one bucket
  const double* times;    // [K][B] (aos: [B][K])
  const double* dfix;     // [DL][n_fixed][B] (aos: [B][DL][n_fixed])
  double* coeffs;         // [B][K][DL][N]
  int B, cfg;
  int aos, pad_;          // input layout of this bucket: 0 canonical SoA, 1 canonical AoS
};
struct MtgDlAnyUnit { int item, tile; };
int mtg_dl_any_index(const MtgDimlaneEntry* e);     // configuration index of a 3-dimensional variant, or -1
size_t mtg_dl_any_lds_bytes();
size_t mtg_dl_any_ws_per_lane();
// wg_begin: [grid + 1] offsets into `units` -- workgroup w runs units[wg_begin[w] .. wg_begin[w + 1])
int mtg_dl_any_launch(void* stream, int grid, const MtgDlAnyItem* items, const MtgDlAnyUnit* units, const int* wg_begin,
                      int* status, double* ws);

#endif  // MTG_KERNELS_H_
