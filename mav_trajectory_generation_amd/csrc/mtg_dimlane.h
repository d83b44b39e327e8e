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
#include "mtg_kernels.h"

#ifndef MTG_DL_OCC
#define MTG_DL_OCC 1   // waves per SIMD the register allocation is held to
#endif


// Coefficient output of one wave (one chain direction of TPW trajectories, all DL dimensions) through an LDS slab:
// row t = the direction's contiguous half [half_lo, half_hi) of trajectory t's K*DL*N*8-byte output piece.
// commit(seg) marks the 64-byte-aligned byte range that the segment just recovered has completed; the next drain() (one
// back-substitution step later, so the LDS write -> read latency overlaps arithmetic) streams it out: 16-byte chunk o of
// the range belongs to trajectory o / nch at offset o % nch, so 4 consecutive lanes write one whole sector and a store
// instruction covers 64 consecutive chunks.
template <class C, int DL, int DIR, int AUX>
struct MtgSlabOut {
  static constexpr int N = C::N, K = C::KT, KA = C::KA;
  static constexpr int S = DL * N * 8;               // bytes of one segment (all dimensions)
  static constexpr int TPW = kWave / DL;
  static constexpr int PIECE = K * S;                // one trajectory's coefficients
  static constexpr int HALF_LO = DIR > 0 ? 0 : KA * S, HALF_HI = DIR > 0 ? KA * S : K * S;
  typedef double d2 __attribute__((ext_vector_type(2)));
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  char* slab;
  int lane, t, d;
  __amdgpu_buffer_rsrc_t rsrc;

  static constexpr int up64(int x) { return (x + 63) & ~63; }
  static constexpr int dn64(int x) { return x & ~63; }

  __device__ __forceinline__ void init(char* slab_, int lane_, int t_, int d_) {
    slab = slab_; lane = lane_; t = t_; d = d_;
    pn = 0;
    init_map();
  }
  // tile = TPW trajectories starting at b0; the descriptor ends at the last existing trajectory, the hardware range
  // check drops the chunks of the tail tile's missing ones
  __device__ __forceinline__ void begin_tile(double* coeffs, long long b0, long long B) {
    char* gbase = reinterpret_cast<char*>(coeffs) + b0 * (long long)PIECE;
    long long nvalid = B - b0;
    if (nvalid > TPW) nvalid = TPW;
    int nbytes = (int)nvalid * PIECE;
    const unsigned long long g = reinterpret_cast<unsigned long long>(gbase);
    const unsigned glo = __builtin_amdgcn_readfirstlane((unsigned)g);
    const unsigned ghi = __builtin_amdgcn_readfirstlane((unsigned)(g >> 32));
    gbase = reinterpret_cast<char*>(((unsigned long long)ghi << 32) | glo);
    nbytes = __builtin_amdgcn_readfirstlane(nbytes);
    rsrc = __builtin_amdgcn_make_buffer_rsrc(gbase, 0, nbytes, 0x00020000);
    pn = 0;
  }
  __device__ __forceinline__ static void fence() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
  }
  // where this lane puts the N coefficients of its dimension of segment `seg`
  __device__ __forceinline__ double* row(int seg) {
    const int off = kRing ? (seg & 1) * S : seg * S - HALF_LO;
    return reinterpret_cast<double*>(slab + t * ROWB + off + d * (N * 8));
  }
  // Chunk -> (trajectory, offset) mapping of a drained range.  A lone wave pays ~4 cycles for EVERY instruction, so the
  // index arithmetic matters: where the largest range has at most 16 chunks (256 bytes; BASELINE config 2: 192 / 256),
  // a range is laid out as CHP = 4 / 8 / 16 chunks per trajectory (padded; surplus chunks are sent out of range), so that
  // lane -> (trajectory, chunk) is a shift and a mask done ONCE per kernel (gl, ll below) and every chunk of every drain is
  // that plus a compile-time constant.  Other shapes divide by the run-time-free chunk count (a multiply-high).
  static constexpr int max_range_chunks() {
    int m = 0;
    for (int seg = (DIR > 0 ? 0 : KA); seg < (DIR > 0 ? KA : K); ++seg) {
      int lo, hi;
      range_of(seg, lo, hi);
      if ((hi - lo) / 16 > m) m = (hi - lo) / 16;
    }
    return m;
  }
  static constexpr void range_of(int seg, int& lo, int& hi) {
    if (DIR > 0) {   // segments arrive KA-1, ..., 0: the completed range grows downwards
      lo = seg == 0 ? 0 : up64(seg * S);
      hi = seg == KA - 1 ? KA * S : up64((seg + 1) * S);
    } else {         // segments arrive KA, ..., K-1: upwards
      lo = seg == KA ? KA * S : dn64(seg * S);
      hi = seg == K - 1 ? K * S : dn64((seg + 1) * S);
    }
    if (hi < lo) hi = lo;
  }
  static constexpr int MAXCH = max_range_chunks();
  static constexpr int CHP = MAXCH <= 4 ? 4 : (MAXCH <= 8 ? 8 : (MAXCH <= 16 ? 16 : 0));   // 0: generic mapping
  // LDS rows.  CHP mapping: a RING of two segment slots per trajectory (slot = segment & 1): a range is read out of the
  // slab when its segment is committed, the < 64-byte tail it leaves behind is read with the next range, i.e. before the
  // segment after that overwrites the slot (LDS operations of a wave execute in order) -- 2 * S bytes per trajectory
  // whatever the chain length.  Generic mapping: the direction's whole half.
  static constexpr bool kRing = CHP != 0;
  static_assert(!kRing || S >= 64, "ring slab: a range reaches into at most one neighbouring segment");
  static constexpr int ROWB = (((kRing ? 2 * S : HALF_HI - HALF_LO) / 16) | 1) * 16;   // odd number of 16-byte units: conflict-free b128 rows
  static constexpr int RPI = CHP ? kWave / CHP : 0;                                        // trajectories per store instruction
  static constexpr int MAXI = CHP ? (TPW + RPI - 1) / RPI : (TPW * MAXCH + 63) / 64;       // store instructions per range
  u4 pv[MAXI];          // chunks of the previously committed range, read from the slab, not yet stored
  unsigned pg[MAXI];    // their byte offsets in the tile's output
  int pn;               // how many of them are in use
  unsigned gl, ll;      // CHP mapping: this lane's (trajectory, chunk) part of the global / LDS byte offset
  __device__ __forceinline__ void init_map() {
    if constexpr (CHP != 0) {
      const unsigned tr = (unsigned)lane / (unsigned)CHP, rr = (unsigned)lane % (unsigned)CHP;
      gl = tr * (unsigned)PIECE + rr * 16u;
      ll = tr * (unsigned)ROWB + rr * 16u;
    }
  }
  __device__ __forceinline__ void store_pending() {
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
      if (i < pn) __builtin_amdgcn_raw_buffer_store_b128(pv[i], rsrc, (int)pg[i], 0, AUX);
    }
    pn = 0;
  }
  // The segment's rows are in the slab: issue the LDS reads of the 64-byte-aligned range this segment completed; the
  // next drain() (start of the next segment's recovery, after its back-substitution) stores them, so the LDS round trip
  // overlaps arithmetic.
  __device__ __forceinline__ void commit(const MtgParams&, int seg) {
    int lo = 0, hi = 0;
    range_of(seg, lo, hi);
    fence();
    __builtin_amdgcn_sched_barrier(0);
    if (hi > lo) {
      const int nch = (hi - lo) >> 4;      // 16-byte chunks per trajectory
      if constexpr (CHP != 0) {
        const unsigned rr = (unsigned)lane % (unsigned)CHP, tr = (unsigned)lane / (unsigned)CHP;
        // ring: the chunks of the range that belong to the neighbouring (earlier recovered) segment sit in the other slot
        const int nb = DIR > 0 ? seg + 1 : seg - 1;                       // that neighbour
        const int cut = DIR > 0 ? ((seg + 1) * S - lo) >> 4 : (seg * S - lo) >> 4;   // first chunk of the upper segment
        const unsigned in_cur = (unsigned)((seg & 1) * S + lo - seg * S), in_nb = (unsigned)((nb & 1) * S + lo - nb * S);
        const unsigned sel = (DIR > 0 ? (rr < (unsigned)cut) : (rr >= (unsigned)(cut > 0 ? cut : 0))) ? in_cur : in_nb;
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
          const bool all_rows = (i + 1) * RPI <= TPW, all_chunks = nch == CHP;
          unsigned g = gl + (unsigned)(i * RPI * PIECE + lo);
          if (!all_chunks || !all_rows) {
            bool ok = true;
            if (!all_chunks) ok = ok && rr < (unsigned)nch;
            if (!all_rows) ok = ok && tr < (unsigned)(TPW - i * RPI);
            g = ok ? g : 0x7ffffff0u;
          }
          pg[i] = g;
          pv[i] = __builtin_bit_cast(u4, *reinterpret_cast<const d2*>(slab + (unsigned)(ll + sel + (unsigned)(i * RPI * ROWB))));   // 32-bit sum first: sel may be a wrapped negative
          pn = i + 1;
        }
      } else {
        const int total = TPW * nch;
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
          if (i * 64 < total) {
            const unsigned o = (unsigned)(i * 64 + lane);
            const unsigned tt = o / (unsigned)nch, r = o - tt * (unsigned)nch;
            const bool ok = o < (unsigned)total;
            pg[i] = ok ? tt * (unsigned)PIECE + (unsigned)lo + r * 16u : 0x7ffffff0u;
            const unsigned loff = ok ? tt * (unsigned)ROWB + (unsigned)(lo - HALF_LO) + r * 16u : 0u;
            pv[i] = __builtin_bit_cast(u4, *reinterpret_cast<const d2*>(slab + loff));
            pn = i + 1;
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    fence();
  }
  __device__ __forceinline__ void drain(const MtgParams&) {
    __builtin_amdgcn_sched_barrier(0);
    store_pending();
    __builtin_amdgcn_sched_barrier(0);
  }
  __device__ __forceinline__ void flush(const MtgParams&) {
    __builtin_amdgcn_sched_barrier(0);
    store_pending();
    __builtin_amdgcn_sched_barrier(0);
  }
};

template <class C, int DL>
__host__ __device__ constexpr size_t mtg_dl_slab_bytes() {   // one wave's slab (the larger of the two directions')
  constexpr size_t a = (size_t)MtgSlabOut<C, DL, 1, 0>::TPW * MtgSlabOut<C, DL, 1, 0>::ROWB;
  constexpr size_t b = (size_t)MtgSlabOut<C, DL, -1, 0>::TPW * MtgSlabOut<C, DL, -1, 0>::ROWB;
  return a > b ? a : b;
}
template <class C, int DL>
__host__ __device__ constexpr size_t mtg_dl_pair_bytes() {
  constexpr int fmid = C::H - C::popc(C::MI);
  constexpr size_t xch = (size_t)(fmid * (fmid + 1) / 2 + fmid) * kWave * sizeof(double);
  constexpr size_t slab = (mtg_dl_slab_bytes<C, DL>() + 15) / 16 * 16;
  return 2 * (slab > xch ? slab : xch);
}
template <class C, int DL, int NP>
constexpr size_t mtg_dl_lds_bytes() { return NP * mtg_dl_pair_bytes<C, DL>(); }

// Input loads of one lane's half-chain, canonical SoA (times[K][B], d_fixed[DL][n_fixed][B]): 32-bit byte offsets from
// the two wave-uniform base pointers (global_load with an SGPR base), one add per load.  Needs 8 * B * (n_fixed * DL) < 4 GiB.
template <class C, int DIR>
__device__ __forceinline__ void mtg_dl_preload(const double* __restrict__ times, const double* __restrict__ dfix,
                                               unsigned B, unsigned b, unsigned d, double (&T)[C::KCS], double (&fx)[1][C::NC]) {
  constexpr int KC = DIR > 0 ? C::KA : C::KB;
  constexpr int c0 = DIR > 0 ? C::colBeginA : C::colBeginB;
  constexpr int nc = DIR > 0 ? C::NCA : C::NCB;
  const unsigned step = B * 8u;
  unsigned ot = ((DIR > 0 ? 0u : (unsigned)(C::KT - 1)) * B + b) * 8u;
#pragma unroll
  for (int j = 0; j < KC; ++j) {
    T[j] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(times) + ot);
    ot = DIR > 0 ? ot + step : ot - step;
  }
  unsigned of = ((d * (unsigned)C::offFEnd + (unsigned)c0) * B + b) * 8u;
#pragma unroll
  for (int c = 0; c < nc; ++c) {
    fx[0][c] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(dfix) + of);
    of += step;
  }
}

#if defined(MTG_LAB_TIMING)
#define MTG_DL_STAMP(slot) do { if (lane == 0 && first) tdbg[slot] = clock64(); } while (0)
#else
#define MTG_DL_STAMP(slot) do { } while (0)
#endif

// C: static configuration with C::D == 1 (one dimension per lane); DL: dimensions of the plan (lanes per trajectory);
// NP: (tile, direction-pair) units per workgroup (2: four waves, one per SIMD of a CU; 1 where two slabs pairs do not fit
// the LDS).  AUX: cache policy bits of the coefficient stores (0 write-back, 1 sc0, 2 nt, 16 sc1).
template <class C, int DL, int NP, int OUT, int AUX>
__global__ __launch_bounds__(NP * 2 * kWave, MTG_DL_OCC) void mtg_solve_dl_kernel(const double* __restrict__ times,
                                                                            const double* __restrict__ dfix,
                                                                            double* __restrict__ coeffs, int* status,
                                                                            int* traj_status, int B, int ntiles, int nwg
#if defined(MTG_LAB_TIMING)
                                                                            , long long* tdbg_base
#endif
) {
  static_assert(C::kStatic && C::D == 1 && C::KT >= 2, "dimension-in-lane form: static one-dimension configurations");
  static_assert(DL >= 1 && DL <= 4, "1..4 dimensions per trajectory");
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  constexpr int TPW = kWave / DL;
  const int lane = threadIdx.x & (kWave - 1);
  const int w = threadIdx.x >> 6;      // wave-uniform
  const int pair = w >> 1, dir = w & 1;
#if defined(MTG_LAB_TIMING)
  long long* tdbg = tdbg_base + ((long long)(blockIdx.x * (NP * 2) + w)) * 16;
  if (lane == 0) { tdbg[0] = clock64(); tdbg[14] = wall_clock64(); }
#endif
  int d = lane / TPW, t = lane - d * TPW;
  const bool dup = d >= DL;            // surplus lanes (64 % DL) duplicate the last lane's work, outputs suppressed
  if (dup) { d = DL - 1; t = TPW - 1; }
  MtgParams P;
  P.times = times; P.ts_b = 1; P.ts_k = B;
  P.dfix = dfix; P.fs_b = 1; P.fs_c = B; P.fs_d = (long long)C::offFEnd * B;
  P.coeffs = coeffs;
  P.dfree = nullptr; P.ps_b = P.ps_d = P.ps_c = 0;
  P.cost = nullptr; P.ws = nullptr; P.ws_stride = 0;
  P.status = status; P.tstatus = traj_status;
  P.vmask = nullptr; P.offF = nullptr; P.offP = nullptr;
  P.B = B; P.K = C::KT; P.Dtot = DL; P.dim0 = d;   // dim0 is a per-lane value here
  P.deriv = C::DV; P.h1off = C::H1OFF; P.ainvoff = C::AINVOFF;
  P.pert_on = 0; P.pert_seg = -1; P.pert_tpv = 1; P.pert_h = P.pert_corr = P.pert_lo = 0.0;

  const int nunits = (ntiles + NP - 1) / NP;
  MtgLane<C> ln;
  auto tile_of = [&](int it) { const int tl = NP * it + pair; return tl < ntiles ? tl : ntiles - 1; };
  auto fetch = [&](int tile_) {
    unsigned bb = (unsigned)tile_ * TPW + t;
    if (bb >= (unsigned)B) bb = B - 1;
    if (dir == 0) mtg_dl_preload<C, 1>(times, dfix, (unsigned)B, bb, (unsigned)d, ln.T, ln.fx);
    else mtg_dl_preload<C, -1>(times, dfix, (unsigned)B, bb, (unsigned)d, ln.T, ln.fx);
  };
  if ((int)blockIdx.x < nunits) fetch(tile_of(blockIdx.x));
  const int lane_io = lane, t_io = t, d_io = d;
#if defined(MTG_LAB_TIMING)
  if (lane == 0) tdbg[1] = clock64();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) tdbg[6] = clock64();
#endif
  constexpr int mm = C::MI;
  constexpr int fmid = C::H - C::popc(mm);
  constexpr int nslots = fmid * (fmid + 1) / 2 + fmid;
  // LDS per pair: [slab A][slab B]; the exchange buffer a direction publishes lives in the OTHER direction's slab
  // (read by that direction before it writes its first coefficient row; the end-of-tile barrier orders reuse)
  char* base = lds_raw + (size_t)pair * mtg_dl_pair_bytes<C, DL>();
  constexpr size_t half = mtg_dl_pair_bytes<C, DL>() / 2;
  char* my_slab = base + (size_t)dir * half;
  double* mine = reinterpret_cast<double*>(base + (size_t)(1 - dir) * half) + lane_io;
  const double* other = reinterpret_cast<const double*>(my_slab) + lane_io;
  MtgSlabOut<C, DL, 1, AUX> ioA;
  MtgSlabOut<C, DL, -1, AUX> ioB;
  ioA.init(my_slab, lane_io, t_io, d_io);
  ioB.init(my_slab, lane_io, t_io, d_io);
  for (int it = blockIdx.x; it < nunits; it += nwg) {
    const int tile = tile_of(it);
    const long long b0 = (long long)tile * TPW;
    const long long bl = b0 + t;
    const bool active = bl < B && !dup;
    const long long b = bl < B ? bl : B - 1;
    const bool first = it == (int)blockIdx.x;
    if (!first) fetch(tile);
    if (dir == 0) mtg_lane_forward<C, 1>(P, b, ln, nullptr, false);
    else mtg_lane_forward<C, -1>(P, b, ln, nullptr, false);
    mtg_pack_mid<C>(ln, mm, mine, kWave);
    MTG_DL_STAMP(2);
    __syncthreads();
    MTG_DL_STAMP(3);
    if (dir == 0) {
      ioA.begin_tile(coeffs, b0, B);
      mtg_lane_finish<C, 1, OUT>(P, b, ln, nullptr, other, kWave, ioA, active);
    } else {
      ioB.begin_tile(coeffs, b0, B);
      mtg_lane_finish<C, -1, OUT>(P, b, ln, nullptr, other, kWave, ioB, active);
    }
#if defined(MTG_LAB_TIMING)
    MTG_DL_STAMP(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && first) { tdbg[5] = clock64(); tdbg[15] = wall_clock64(); }
#endif
    __syncthreads();
  }
}
#endif  // MTG_DIMLANE_H_
