// mtg_slab.h -- whole-sector coefficient output through an LDS slab (shared by the dimension-in-lane kernel,
// mtg_dimlane.h, and the slab-output instantiations of the fused kernel, mtg_kernels.h).
#ifndef MTG_SLAB_H_
#define MTG_SLAB_H_
#include <type_traits>

#include "mtg_lane.h"

// Coefficient output of one wave (one chain direction of TPW trajectories, all dimensions) through an LDS slab.
// LPT = lanes per trajectory (the dimension-in-lane kernel: one lane per dimension, C::D == 1; the fused kernel: LPT == 1,
// a lane holds all C::D dimensions).  PEND: hold a range's chunks in registers across one back-substitution step (a lone
// wave per SIMD hides the LDS round trip that way); false: read and store within commit(), in groups of four (register-
// tight kernels).
// Layout:
// row t = the direction's contiguous half [half_lo, half_hi) of trajectory t's K*DL*N*8-byte output piece.
// commit(seg) marks the 64-byte-aligned byte range that the segment just recovered has completed; the next drain() (one
// back-substitution step later, so the LDS write -> read latency overlaps arithmetic) streams it out: 16-byte chunk o of
// the range belongs to trajectory o / nch at offset o % nch, so 4 consecutive lanes write one whole sector and a store
// instruction covers 64 consecutive chunks.
template <class C, int LPT, int DIR, int AUX, bool PEND = true>
struct MtgSlabOut {
  static constexpr int N = C::N, K = C::KT, KA = C::KA;
  static constexpr int LB = C::D * N * 8;            // bytes one lane contributes per segment (its C::D dimensions)
  static constexpr int S = LPT * LB;                 // bytes of one segment (all dimensions of the trajectory)
  static constexpr int TPW = 64 / LPT;
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
    ph0 = (unsigned)(b0 & 3);
    if constexpr (kPhaseChp) phi_l = (int)(((ph0 + (unsigned)lane / (unsigned)CHP) * (unsigned)PMOD) & 63u);
  }
  __device__ __forceinline__ static void fence() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
  }
  // where this lane puts the N coefficients of its dimension of segment `seg`
  __device__ __forceinline__ double* row(int seg) {
    const int off = kRing ? (seg & 1) * S : seg * S - HALF_LO;
    return reinterpret_cast<double*>(slab + t * ROWB + off + d * LB);
  }
  // Chunk -> (trajectory, offset) mapping of a drained range.  A lone wave pays ~4 cycles for EVERY instruction, so the
  // index arithmetic matters: where the largest range has at most 16 chunks (256 bytes; BASELINE config 2: 192 / 256),
  // a range is laid out as CHP = 4 / 8 / 16 chunks per trajectory (padded; surplus chunks are sent out of range), so that
  // lane -> (trajectory, chunk) is a shift and a mask done ONCE per kernel (gl, ll below) and every chunk of every drain is
  // that plus a compile-time constant.  Other shapes divide by the run-time-free chunk count (a multiply-high).
  static constexpr int max_range_chunks() {
    int m = 0;
    for (int seg = (DIR > 0 ? 0 : KA); seg < (DIR > 0 ? KA : K); ++seg) {
      int lo = 0, hi = 0;
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
  // PIECE not a multiple of 64 (N = 10 with K % 4 != 0, N = 12 with odd K): consecutive trajectories' pieces shift against
  // the 64-byte grid by PMOD bytes, so "64-byte aligned relative to the trajectory's start" is aligned in memory for every
  // 64 / gcd-th trajectory only (measured, N = 10 at B = 100k: 11.4 us per segment for K = 10 against 8.1 for K = 12 --
  // partial sectors again).  Phase mapping: every row's ranges are aligned in MEMORY, i.e. shifted by the row's phase
  // phi = ((b0 + row) * PMOD) mod 64 (a multiple of 16); rows have up to MAXCH_P chunks per range, addressed with a fixed
  // row stride of MAXCH_P chunk slots per store pass (division by a compile-time constant).
  static constexpr int PMOD = PIECE & 63;
  static constexpr bool kPhase = PMOD != 0 && S >= 64;
  static constexpr int MAXCH_P = (S + 63) / 16;
  static constexpr int MAXCH = max_range_chunks();
  // Row mapping: a store instruction covers RPI = 64 / CHP whole rows of CHP chunk slots each (CHP = the largest range, in
  // chunks); lane -> (row, chunk) is computed ONCE per kernel (gl, ll below), every chunk of every drain is that plus a
  // compile-time constant.  Used whenever at most 8 of the 64 lanes idle (CHP = 16: none; 12 or 20: four); otherwise the
  // generic mapping (chunk o of a range -> row o / nch: a multiply-high and a slot selection per chunk).
  static constexpr int row_width() { return kPhase ? ((S + 63) / 64) * 4 : MAXCH; }
  static constexpr bool row_ok(int w) { return w >= 4 && w <= 32 && 64 - (64 / w) * w <= 8; }
  // (rows that are not a power of two wide only up to K = 16 where the pieces are aligned: the register-tight K > 16 bodies of
  // N = 8 / 12 spill to scratch with them -- round 6 again: N = 12 / K = 32 418 -> 549 us, N = 8 / K = 32 176 -> 262 at 100k -- and keep the
  // generic mapping.  Phase-mapped pieces (N = 12 with an odd K) take the rows at every K: the generic phase mapping divides per lane
  // and per chunk -- N = 12 / K = 17 253 -> 203 us, K = 25 392 -> 333, K = 31 478 -> 403, fewer registers, same bits.)
  static constexpr bool pow2(int w) { return (w & (w - 1)) == 0; }
  static constexpr int CHP = (row_ok(row_width()) && (pow2(row_width()) || K <= 16 || kPhase)) ? row_width() : 0;
  static constexpr bool kPhaseChp = kPhase && CHP != 0;   // phase mapping with rows: the piece's misaligned head / tail (up to 48
                                                          // bytes each) go out in a pass of their own, ranges stay <= CHP chunks
  // LDS rows.  CHP mapping: a RING of two segment slots per trajectory (slot = segment & 1): a range is read out of the
  // slab when its segment is committed, the < 64-byte tail it leaves behind is read with the next range, i.e. before the
  // segment after that overwrites the slot (LDS operations of a wave execute in order) -- 2 * S bytes per trajectory
  // whatever the chain length.  Generic mapping: the direction's whole half.
  static constexpr bool kRing = S >= 64;   // (tiny shapes: the direction's whole half; a range could span several segments)
  static constexpr int ROWB = (((kRing ? 2 * S : HALF_HI - HALF_LO) / 16) | 1) * 16;   // odd number of 16-byte units: conflict-free b128 rows
  static constexpr int RPI = CHP ? 64 / CHP : 0;                                        // trajectories per store instruction
  static constexpr int MAXI_MAIN = CHP ? (TPW + RPI - 1) / RPI : (kPhase ? (TPW * MAXCH_P + 63) / 64 : (TPW * MAXCH + 63) / 64);
  static constexpr int MINI = kPhaseChp ? (TPW * 3 + 63) / 64 : 0;   // passes for the head / tail pieces (3 chunks per row)
  static constexpr int MAXI = MAXI_MAIN + MINI;                      // store instructions per range
  static constexpr int NPV = PEND ? MAXI : 1;
  u4 pv[NPV];           // PEND: chunks of the previously committed range, read from the slab, not yet stored
  unsigned pg[NPV];     // their byte offsets in the tile's output
  int pn;               // PEND: how many of them are in use; !PEND: the committed, not yet streamed segment + 1 (0: none)
  unsigned gl, ll;      // CHP mapping: this lane's (trajectory, chunk) part of the global / LDS byte offset
  unsigned ph0;         // phase mapping: (first trajectory of the tile) mod 4
  int phi_l;            // kPhaseChp: phase of this lane's rows (row = lane / 16 + 4 i)
  __device__ __forceinline__ void init_map() {
    if constexpr (CHP != 0) {
      const unsigned tr = (unsigned)lane / (unsigned)CHP, rr = (unsigned)lane % (unsigned)CHP;
      gl = tr * (unsigned)PIECE + rr * 16u;
      ll = tr * (unsigned)ROWB + rr * 16u;
    }
  }
  // ring (CHP mapping): LDS offset of this lane's chunk of range(seg) relative to (row, chunk) = ll -- the chunks that
  // belong to the neighbouring (earlier recovered) segment sit in the other slot
  __device__ __forceinline__ unsigned slot_select(int seg, int lo, unsigned rr) const {
    if constexpr (!kRing) return (unsigned)(lo - HALF_LO);
    const int nb = DIR > 0 ? seg + 1 : seg - 1;
    const int cut = DIR > 0 ? ((seg + 1) * S - lo) >> 4 : (seg * S - lo) >> 4;   // first chunk of the upper segment
    const unsigned in_cur = (unsigned)((seg & 1) * S + lo - seg * S), in_nb = (unsigned)((nb & 1) * S + lo - nb * S);
    return (DIR > 0 ? (rr < (unsigned)cut) : (rr >= (unsigned)(cut > 0 ? cut : 0))) ? in_cur : in_nb;
  }
  // chunk i of range(seg) = [lo, lo + 16 nch): global byte offset (out of range for surplus lanes) and LDS byte offset;
  // false if store instruction i does not exist for this range
  // this row's memory-aligned range of segment seg (phase mapping): [lo, hi) relative to the trajectory's piece
  // (kPhaseChp: the piece's misaligned head / tail are not part of the first / last range -- they have their own pass)
  __device__ __forceinline__ static void range_of_phase(int seg, int phi, int& lo, int& hi) {
    if (DIR > 0) {
      lo = (seg == 0 && !kPhaseChp) ? 0 : ((seg * S + phi + 63) & ~63) - phi;
      hi = seg == KA - 1 ? KA * S : (((seg + 1) * S + phi + 63) & ~63) - phi;
    } else {
      lo = seg == KA ? KA * S : ((seg * S + phi) & ~63) - phi;
      hi = (seg == K - 1 && !kPhaseChp) ? K * S : (((seg + 1) * S + phi) & ~63) - phi;
    }
    if (hi < lo) hi = lo;
  }
  __device__ __forceinline__ bool chunk(int seg, int lo, int nch, unsigned sel, int i, unsigned& g, unsigned& loff) const {
    if constexpr (kPhaseChp) {
      if (i >= MAXI) return false;
      if (i >= MAXI_MAIN) {   // head (direction A, segment 0) / tail (direction B, last segment) of every row's piece
        if (!(DIR > 0 ? seg == 0 : seg == K - 1)) return false;
        const unsigned o = (unsigned)((i - MAXI_MAIN) * 64 + lane);
        const unsigned tt = o / 3u, c = o - tt * 3u;
        const int ph = (int)(((ph0 + tt) * (unsigned)PMOD) & 63u);
        const int start = DIR > 0 ? 0 : ((K * S + ph) & ~63) - ph;
        const int end = DIR > 0 ? ((64 - ph) & 63) : K * S;
        const int at = start + (int)(c * 16u);
        const bool ok = tt < (unsigned)TPW && at < end;
        g = ok ? tt * (unsigned)PIECE + (unsigned)at : 0x7ffffff0u;
        loff = ok ? tt * (unsigned)ROWB + (unsigned)((seg & 1) * S + at - seg * S) : 0u;
        return true;
      }
      const unsigned rr = (unsigned)lane % (unsigned)CHP, tr = (unsigned)lane / (unsigned)CHP;
      // phase of row tr + i RPI (independent of the pass where RPI * PMOD is a multiple of 64, e.g. RPI = 4)
      const int phi = (RPI * PMOD) % 64 == 0 ? phi_l : (int)(((unsigned)phi_l + (unsigned)(i * RPI * PMOD)) & 63u);
      int lo_l, hi_l;
      range_of_phase(seg, phi, lo_l, hi_l);
      bool ok = (int)(rr * 16u) < hi_l - lo_l;
      if (RPI * CHP < 64) ok = ok && tr < (unsigned)RPI;
      if ((i + 1) * RPI > TPW) ok = ok && tr < (unsigned)(TPW - i * RPI);
      g = ok ? gl + (unsigned)(i * RPI * PIECE) + (unsigned)lo_l : 0x7ffffff0u;
      loff = ll + slot_select(seg, lo_l, rr) + (unsigned)(i * RPI * ROWB);
      return true;
    } else if constexpr (kPhase) {
      if (i >= MAXI) return false;
      const unsigned o = (unsigned)(i * 64 + lane);
      const unsigned tt = o / (unsigned)MAXCH_P, r = o - tt * (unsigned)MAXCH_P;
      const int phi = (int)(((ph0 + tt) * (unsigned)PMOD) & 63u);
      int lo_l, hi_l;
      range_of_phase(seg, phi, lo_l, hi_l);
      const bool ok = tt < (unsigned)TPW && (int)(r * 16u) < hi_l - lo_l;
      g = ok ? tt * (unsigned)PIECE + (unsigned)lo_l + r * 16u : 0x7ffffff0u;
      loff = ok ? (unsigned)(tt * (unsigned)ROWB + slot_select(seg, lo_l, r) + r * 16u) : 0u;
      return true;
    } else if constexpr (CHP != 0) {
      if (i >= MAXI) return false;
      const unsigned rr = (unsigned)lane % (unsigned)CHP, tr = (unsigned)lane / (unsigned)CHP;
      const bool all_rows = (i + 1) * RPI <= TPW, all_chunks = nch == CHP, all_lanes = RPI * CHP == 64;
      g = gl + (unsigned)(i * RPI * PIECE + lo);
      if (!all_chunks || !all_rows || !all_lanes) {
        bool ok = true;
        if (!all_chunks) ok = ok && rr < (unsigned)nch;
        if (!all_lanes) ok = ok && tr < (unsigned)RPI;
        if (!all_rows) ok = ok && tr < (unsigned)(TPW - i * RPI);
        g = ok ? g : 0x7ffffff0u;
      }
      loff = ll + sel + (unsigned)(i * RPI * ROWB);   // 32-bit sum: sel may be a wrapped negative
      return true;
    } else {
      const int total = TPW * nch;
      if (i * 64 >= total) return false;
      const unsigned o = (unsigned)(i * 64 + lane);
      const unsigned tt = o / (unsigned)nch, r = o - tt * (unsigned)nch;
      const bool ok = o < (unsigned)total;
      g = ok ? tt * (unsigned)PIECE + (unsigned)lo + r * 16u : 0x7ffffff0u;
      loff = ok ? (unsigned)(tt * (unsigned)ROWB + slot_select(seg, lo, r) + r * 16u) : 0u;
      return true;
    }
  }
  __device__ __forceinline__ u4 lds_chunk(unsigned loff) const {
    return __builtin_bit_cast(u4, *reinterpret_cast<const d2*>(slab + loff));
  }
  __device__ __forceinline__ void store_pending() {
    if constexpr (PEND) {
#pragma unroll
      for (int i = 0; i < MAXI; ++i) {
        if (i < pn) __builtin_amdgcn_raw_buffer_store_b128(pv[i], rsrc, (int)pg[i], 0, AUX);
      }
      pn = 0;
    } else {
      if (pn == 0) return;
      const int seg = pn - 1;
      pn = 0;
      int lo = 0, hi = 0;
      range_of(seg, lo, hi);
      if (!kPhase && hi <= lo) return;
      const int nch = (hi - lo) >> 4;
      const unsigned sel = CHP != 0 ? slot_select(seg, lo, (unsigned)lane % (unsigned)(CHP ? CHP : 1)) : 0u;
      fence();
      constexpr int G = 4;   // LDS reads in groups ahead of their stores: one lgkmcnt wait per group
#pragma unroll
      for (int i0 = 0; i0 < MAXI; i0 += G) {
        u4 v[G];
        unsigned g[G];
        bool on[G];
#pragma unroll
        for (int i = 0; i < G; ++i) {
          unsigned loff = 0;
          g[i] = 0;
          on[i] = chunk(seg, lo, nch, sel, i0 + i, g[i], loff);
          if (on[i]) v[i] = lds_chunk(loff);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < G; ++i) {
          if (on[i]) __builtin_amdgcn_raw_buffer_store_b128(v[i], rsrc, (int)g[i], 0, AUX);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      fence();
    }
  }
  // The segment's rows are in the slab.  PEND: issue the LDS reads of the 64-byte-aligned range this segment completed;
  // the next drain() (start of the next segment's recovery, after its back-substitution) stores them, so the LDS round
  // trip overlaps arithmetic.  !PEND: remember the segment; the next drain() reads and stores the range.
  __device__ __forceinline__ void commit(const MtgParams&, int seg) {
    fence();
    if constexpr (PEND) {
      int lo = 0, hi = 0;
      range_of(seg, lo, hi);
      __builtin_amdgcn_sched_barrier(0);
      if (kPhase || hi > lo) {
        const int nch = (hi - lo) >> 4;      // 16-byte chunks per trajectory
        const unsigned sel = CHP != 0 ? slot_select(seg, lo, (unsigned)lane % (unsigned)(CHP ? CHP : 1)) : 0u;
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
          unsigned loff = 0;
          if (chunk(seg, lo, nch, sel, i, pg[i], loff)) {
            pv[i] = lds_chunk(loff);
            pn = i + 1;
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      fence();
    } else {
      pn = seg + 1;
    }
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


// ---- run-time chain length ---------------------------------------------------------------------------------------------
// The same output scheme for kernels whose number of segments K is a RUN-TIME value (mtg_dimlane_rt.h: one body per N for
// every chain length).  Dimension-in-lane layout only (LPT lanes per trajectory, one dimension per lane), ring of two segment
// slots per trajectory, row mapping (a store instruction covers RPI rows of MAXCH 16-byte chunk slots; MAXCH = the largest
// range in chunks: 12 / 16 / 20 for N = 8 / 10 / 12 with three dimensions).
// When K * S is not a multiple of 64 bytes (N = 10: K mod 4 != 0; N = 12: odd K) consecutive trajectories' pieces shift against
// the 64-byte grid, and ranges aligned relative to the piece would be misaligned in memory for most rows (partial sectors:
// measured 1.2x -> 1.5x the static variants' time for those K).  As in MtgSlabOut's phase mapping every row's ranges are then
// aligned in MEMORY: row r of the tile has phase phi = ((b0 + r) * (K S mod 64)) mod 64 and its ranges are [up64(seg S + phi)
// - phi, ...); the piece's misaligned head (direction A, segment 0) / tail (direction B, last segment; also the < 64 bytes
// beyond the last boundary when the piece is aligned) go out in a pass of their own.  A wave-uniform branch: aligned pieces
// (phase 0 everywhere) keep the cheaper maps.
// PHASE: compiled for pieces that are NOT a multiple of 64 bytes (the per-row phase maps); false: aligned pieces only.  Two
// instantiations of the kernel instead of a run-time branch: with both maps in one body the aligned chain lengths lost 4-10 %.
// DPL: dimensions per lane (1: dimension-in-lane kernels; D with LPT = 1: the one-lane-per-trajectory update kernel, whose
// segments all arrive in ascending order -- direction B with KA = 0, init()'s last argument).
template <int N_, int LPT, int DIR, int AUX, bool PHASE, bool PEND = true, int DPL = 1>
struct MtgSlabOutRt {
  static constexpr int N = N_;
  static constexpr int LB = DPL * N * 8;             // bytes one lane contributes per segment
  static constexpr int S = LPT * LB;                 // bytes of one segment (all dimensions of the trajectory)
  static constexpr int TPW = 64 / LPT;
  static_assert(S >= 64, "ring layout needs segments of at least one sector");
  static constexpr int MAXCH = (S % 64 == 0) ? S / 16 : (S / 64 + 1) * 4;   // chunks of the largest range (a range never exceeds S rounded up to 64)
  static constexpr int RPI = 64 / MAXCH;             // trajectories per store instruction
  static constexpr int MAXI = (TPW + RPI - 1) / RPI; // store instructions per range
  static constexpr int ROWB = (((2 * S) / 16) | 1) * 16;   // two segment slots, odd number of 16-byte units
  static constexpr int EI = (3 * TPW + 63) / 64;     // store instructions of the head / tail pass (3 chunk slots per trajectory)
  static constexpr int NPV = PEND ? MAXI + EI : 1;
  typedef double d2 __attribute__((ext_vector_type(2)));
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  char* slab;
  int lane, t, d;
  int K, KA;            // run-time chain length; KA = (K + 1) / 2 segments belong to direction A
  unsigned piece;       // K * S
  int pmod;             // piece mod 64: 0 = every row aligned
  __amdgpu_buffer_rsrc_t rsrc;
  u4 pv[NPV];
  unsigned pg[NPV];
  int pn;
  unsigned gl_row, gl, ll;   // this lane's row of a store instruction; its (row, chunk) part of the global / LDS byte offset
  unsigned ph0;         // (first trajectory of the tile) mod 4
  int phi_l;            // phase of this lane's row of pass 0

  static __device__ __forceinline__ int up64(int x) { return (x + 63) & ~63; }
  static __device__ __forceinline__ int dn64(int x) { return x & ~63; }
  __device__ __forceinline__ void init(char* slab_, int lane_, int t_, int d_, int K_, int KA_ = -1) {
    slab = slab_; lane = lane_; t = t_; d = d_;
    K = K_; KA = KA_ >= 0 ? KA_ : (K_ + 1) / 2;
    piece = (unsigned)K_ * (unsigned)S;
    pmod = (int)(piece & 63u);
    pn = 0;
    ph0 = 0; phi_l = 0;
    const unsigned tr = (unsigned)lane / (unsigned)MAXCH, rr = (unsigned)lane % (unsigned)MAXCH;
    gl_row = tr;
    gl = tr * piece + rr * 16u;
    ll = tr * (unsigned)ROWB + rr * 16u;
  }
  __device__ __forceinline__ void begin_tile(double* coeffs, long long b0, long long B) {
    char* gbase = reinterpret_cast<char*>(coeffs) + b0 * (long long)piece;
    long long nvalid = B - b0;
    if (nvalid > TPW) nvalid = TPW;
    int nbytes = (int)nvalid * (int)piece;
    const unsigned long long g = reinterpret_cast<unsigned long long>(gbase);
    const unsigned glo = __builtin_amdgcn_readfirstlane((unsigned)g);
    const unsigned ghi = __builtin_amdgcn_readfirstlane((unsigned)(g >> 32));
    gbase = reinterpret_cast<char*>(((unsigned long long)ghi << 32) | glo);
    nbytes = __builtin_amdgcn_readfirstlane(nbytes);
    rsrc = __builtin_amdgcn_make_buffer_rsrc(gbase, 0, nbytes, 0x00020000);
    pn = 0;
    ph0 = (unsigned)(b0 & 3);
    phi_l = (int)(((ph0 + gl_row) * (unsigned)pmod) & 63u);
  }
  __device__ __forceinline__ static void fence() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
  }
  __device__ __forceinline__ double* row(int seg) {
    return reinterpret_cast<double*>(slab + t * ROWB + (seg & 1) * S + d * LB);
  }
  // the range that segment `seg` completes for a row of phase phi, relative to the row's piece: 64-byte aligned in memory.
  // Segments of direction A arrive KA-1, ..., 0 (the range grows downwards), of direction B KA, ..., K-1 (upwards).  The
  // piece's head [0, (64 - phi) mod 64) and tail [dn64(K S + phi) - phi, K S) are NOT part of the first / last range.
  __device__ __forceinline__ void range_of(int seg, int phi, int& lo, int& hi) const {
    if (DIR > 0) {
      lo = ((seg * S + phi + 63) & ~63) - phi;
      hi = seg == KA - 1 ? KA * S : (((seg + 1) * S + phi + 63) & ~63) - phi;
    } else {
      lo = seg == KA ? KA * S : ((seg * S + phi) & ~63) - phi;
      hi = (((seg + 1) * S + phi) & ~63) - phi;
    }
    if (hi < lo) hi = lo;
  }
  // LDS offset of chunk rr of a range starting at lo, relative to ll: the chunks that belong to the neighbouring (earlier
  // recovered) segment sit in the other ring slot
  __device__ __forceinline__ unsigned slot_select(int seg, int lo, unsigned rr) const {
    const int nb = DIR > 0 ? seg + 1 : seg - 1;
    const int cut = DIR > 0 ? ((seg + 1) * S - lo) >> 4 : (seg * S - lo) >> 4;   // first chunk of the upper segment
    const unsigned in_cur = (unsigned)((seg & 1) * S + lo - seg * S), in_nb = (unsigned)((nb & 1) * S + lo - nb * S);
    return (DIR > 0 ? (rr < (unsigned)cut) : (rr >= (unsigned)(cut > 0 ? cut : 0))) ? in_cur : in_nb;
  }
  // store instruction i of range(seg): global byte offset (out of range for lanes without a chunk) and LDS byte offset.
  // PHASE = false: aligned pieces, the range [lo, lo + 16 nch) and the slot selection `sel` are the same for every row;
  // PHASE = true: per-row ranges.
  __device__ __forceinline__ void chunk(int seg, int lo, int nch, unsigned sel, int i, unsigned& g, unsigned& loff) const {
    const unsigned rr = (unsigned)lane % (unsigned)MAXCH, tr = (unsigned)lane / (unsigned)MAXCH;
    bool ok = true;
    if (RPI * MAXCH < 64) ok = ok && tr < (unsigned)RPI;
    if ((i + 1) * RPI > TPW) ok = ok && tr < (unsigned)(TPW - i * RPI);
    if constexpr (!PHASE) {
      ok = ok && rr < (unsigned)nch;
      g = ok ? gl + (unsigned)(i * RPI) * piece + (unsigned)lo : 0x7ffffff0u;      // (pass offset and lo: scalar)
      loff = ll + sel + (unsigned)(i * RPI * ROWB);
    } else {
      const int phi = (int)(((unsigned)phi_l + (unsigned)(i * RPI) * (unsigned)pmod) & 63u);   // phase of row tr + i RPI
      int lo_l, hi_l;
      range_of(seg, phi, lo_l, hi_l);
      ok = ok && (int)(rr * 16u) < hi_l - lo_l;
      g = ok ? gl + (unsigned)(i * RPI) * piece + (unsigned)lo_l : 0x7ffffff0u;
      loff = ll + slot_select(seg, lo_l, rr) + (unsigned)(i * RPI * ROWB);
    }
  }
  // The head (direction A, segment 0) / tail (direction B, last segment) of every row's piece: up to 3 chunks per row,
  // lane -> (trajectory lane / 3, chunk lane % 3) -- one store instruction.  false: this segment has no such pass.
  __device__ __forceinline__ bool edge_chunk(int seg, int e, unsigned& g, unsigned& loff) const {
    if (!(DIR > 0 ? seg == 0 : seg == K - 1)) return false;            // (wave-uniform)
    if constexpr (!PHASE) return false;                                // aligned pieces have neither head nor tail
    const unsigned o = (unsigned)(e * 64 + lane);
    const unsigned tt = o / 3u, c = o - tt * 3u;
    const int ph = (int)(((ph0 + tt) * (unsigned)pmod) & 63u);
    const int start = DIR > 0 ? 0 : ((K * S + ph) & ~63) - ph;
    const int end = DIR > 0 ? ((64 - ph) & 63) : K * S;
    const int at = start + (int)(c * 16u);
    const bool ok = tt < (unsigned)TPW && at < end;
    g = ok ? tt * piece + (unsigned)at : 0x7ffffff0u;
    loff = ok ? tt * (unsigned)ROWB + (unsigned)((seg & 1) * S + at - seg * S) : 0u;
    return true;
  }
  __device__ __forceinline__ u4 lds_chunk(unsigned loff) const {
    return __builtin_bit_cast(u4, *reinterpret_cast<const d2*>(slab + loff));
  }
  __device__ __forceinline__ void store_pending() {
    if constexpr (PEND) {
#pragma unroll
      for (int i = 0; i < NPV; ++i) {
        if (i < pn) __builtin_amdgcn_raw_buffer_store_b128(pv[i], rsrc, (int)pg[i], 0, AUX);
      }
      pn = 0;
    } else {
      if (pn == 0) return;
      const int seg = pn - 1;
      pn = 0;
      int lo = 0, hi = 0;
      range_of(seg, 0, lo, hi);
      const int nch = (hi - lo) >> 4;
      const unsigned sel = slot_select(seg, lo, (unsigned)lane % (unsigned)MAXCH);
      fence();
      constexpr int G = 4;
      auto pass = [&](auto phase) {
        (void)phase;
#pragma unroll
        for (int i0 = 0; i0 < MAXI; i0 += G) {
          u4 v[G];
          unsigned g[G];
#pragma unroll
          for (int i = 0; i < G; ++i) {
            if (i0 + i < MAXI) {
              unsigned loff = 0;
              chunk(seg, lo, nch, sel, i0 + i, g[i], loff);
              v[i] = lds_chunk(loff);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < G; ++i) {
            if (i0 + i < MAXI) __builtin_amdgcn_raw_buffer_store_b128(v[i], rsrc, (int)g[i], 0, AUX);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      if (PHASE || hi > lo) pass(std::integral_constant<bool, PHASE>());      // (wave-uniform)
#pragma unroll
      for (int e = 0; e < EI; ++e) {
        unsigned g = 0, loff = 0;
        if (edge_chunk(seg, e, g, loff)) __builtin_amdgcn_raw_buffer_store_b128(lds_chunk(loff), rsrc, (int)g, 0, AUX);
      }
      fence();
    }
  }
  __device__ __forceinline__ void commit(const MtgParams&, int seg) {
    fence();
    if constexpr (PEND) {
      int lo = 0, hi = 0;
      range_of(seg, 0, lo, hi);
      __builtin_amdgcn_sched_barrier(0);
      pn = 0;
      if (PHASE || hi > lo) {       // (wave-uniform)
        const int nch = (hi - lo) >> 4;
        const unsigned sel = PHASE ? 0u : slot_select(seg, lo, (unsigned)lane % (unsigned)MAXCH);
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
          unsigned loff = 0;
          chunk(seg, lo, nch, sel, i, pg[i], loff);
          pv[i] = lds_chunk(loff);
        }
        pn = MAXI;
      }
#pragma unroll
      for (int e = 0; e < EI; ++e) {
        unsigned g = 0, loff = 0;
        if (edge_chunk(seg, e, g, loff)) {
          if (pn == 0) {     // (no main range: unused slots store out of range)
#pragma unroll
            for (int i = 0; i < MAXI; ++i) pg[i] = 0x7ffffff0u;
          }
          pg[MAXI + e] = g;
          pv[MAXI + e] = lds_chunk(loff);
          pn = MAXI + e + 1;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      fence();
    } else {
      pn = seg + 1;
    }
  }
  __device__ __forceinline__ void drain(const MtgParams&) {
    __builtin_amdgcn_sched_barrier(0);
    store_pending();
    __builtin_amdgcn_sched_barrier(0);
  }
  __device__ __forceinline__ void flush(const MtgParams& P) { drain(P); }
};

#endif  // MTG_SLAB_H_
