// mtg_kernels.h -- __global__ kernel templates shared by the per-variant translation units.
#ifndef MTG_KERNELS_H_
#define MTG_KERNELS_H_
#include <hip/hip_runtime.h>

#include "mtg_lane.h"

constexpr int kWave = 64;
constexpr int kBlock = 2 * kWave;  // wave 0: direction A (forward), wave 1: direction B

// Coefficient output staged through LDS.  Each lane drops the D*N coefficients of the segment it
// just recovered into its row of the wave's staging buffer (ds_write_b128, row stride an odd number
// of 16-byte units => conflict-free), then the whole wave streams the buffer out: chunk o of 16 bytes
// belongs to trajectory t = o / Q at offset r = o % Q of that trajectory's contiguous D*N*8-byte piece
// coeffs[b0 + t][seg][dim0 .. dim0+D)[0..N), so a store instruction covers 64 consecutive chunks
// (~9 cache lines for the 240-byte pieces of N=10, D=3) instead of 64 lines with row-per-lane stores.
// Measured on MI355X at B = 1M: 869 us -> see profiles/.
template <class C>
struct MtgLdsOut {
  static constexpr int Q = C::D * C::N / 2;   // 16-byte chunks per lane per segment
  static constexpr int QP = Q | 1;            // padded row stride (odd)
  typedef double d2 __attribute__((ext_vector_type(2)));
  double* stage;        // this wave's staging buffer in LDS: 64 rows x QP chunks
  int lane;
  long long b0;         // first trajectory of the tile
  __device__ __forceinline__ double* row() { return stage + (size_t)lane * QP * 2; }
  __device__ __forceinline__ void flush(const MtgParams& P, int seg) {
    const int K = mtg_nseg<C>(P);
    long long piece, segoff;
    if constexpr (C::kStatic) {
      piece = (long long)C::KT * C::D * C::N;
      segoff = (long long)seg * C::D * C::N;
    } else {
      piece = (long long)K * P.Dtot * C::N;
      segoff = ((long long)seg * P.Dtot + P.dim0) * C::N;
    }
    char* gbase = reinterpret_cast<char*>(P.coeffs + b0 * piece + segoff);   // wave-uniform
    const unsigned piece_bytes = (unsigned)piece * 8u;    // one tile spans < 4 GiB
    const long long nvalid = P.B - b0;                    // trajectories of this tile that exist
    // The per-chunk offsets depend only on the lane; recompute them per flush (a few integer ops)
    // instead of letting LICM park 2*Q of them in registers across the whole tile loop.
    unsigned l = (unsigned)lane;
    asm volatile("" : "+v"(l));
    __builtin_amdgcn_wave_barrier();
    const char* sbase = reinterpret_cast<const char*>(stage);
    if (nvalid >= 64) {
#pragma unroll
      for (int i = 0; i < Q; ++i) {
        const unsigned o = (unsigned)i * 64u + l;
        const unsigned t = o / (unsigned)Q, r = o - t * (unsigned)Q;
        const d2 v = *reinterpret_cast<const d2*>(sbase + (size_t)(t * (unsigned)QP + r) * 16u);
        *reinterpret_cast<d2*>(gbase + (size_t)(t * piece_bytes + r * 16u)) = v;
      }
    } else {
#pragma unroll
      for (int i = 0; i < Q; ++i) {
        const unsigned o = (unsigned)i * 64u + l;
        const unsigned t = o / (unsigned)Q, r = o - t * (unsigned)Q;
        const d2 v = *reinterpret_cast<const d2*>(sbase + (size_t)(t * (unsigned)QP + r) * 16u);
        if ((long long)t < nvalid) *reinterpret_cast<d2*>(gbase + (size_t)(t * piece_bytes + r * 16u)) = v;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
};

template <class C>
__host__ __device__ constexpr size_t mtg_stage_doubles() { return (size_t)64 * (C::D * C::N / 2 | 1) * 2; }

template <class C, int OUT>
__global__ __launch_bounds__(kBlock) void mtg_solve_kernel(MtgParams P, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x & (kWave - 1);
  const int dir = threadIdx.x >> 6;  // wave-uniform
  const int K = mtg_nseg<C>(P);
  const int vm = (K + 1) / 2;
  const int mm = mtg_mask<C>(P, vm);
  const int nslots = mtg_mid_slots<C>(mm);
  // LDS: [staging A][staging B][exchange A][exchange B]
  MtgLdsOut<C> io;
  io.stage = lds + (size_t)dir * mtg_stage_doubles<C>();
  io.lane = lane;
  double* xch = lds + 2 * mtg_stage_doubles<C>();
  double* mine = xch + (size_t)dir * nslots * kWave + lane;
  const double* other = xch + (size_t)(1 - dir) * nslots * kWave + lane;
  double* wsl = P.ws ? P.ws + ((long long)blockIdx.x * kBlock + threadIdx.x) : nullptr;
  MtgLane<C> ln;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    io.b0 = (long long)tile * kWave;
    const long long bl = io.b0 + lane;
    const bool active = bl < P.B;
    const long long b = active ? bl : P.B - 1;   // tail lanes duplicate the last trajectory, outputs suppressed
    if (dir == 0) mtg_lane_forward<C, 1>(P, b, ln, wsl);
    else mtg_lane_forward<C, -1>(P, b, ln, wsl);
    mtg_pack_mid<C>(ln, mm, mine, kWave);
    __syncthreads();
    if (dir == 0) mtg_lane_finish<C, 1, OUT>(P, b, ln, wsl, other, kWave, io, active);
    else mtg_lane_finish<C, -1, OUT>(P, b, ln, wsl, other, kWave, io, active);
    __syncthreads();
  }
}

template <class C, int OUT>
__global__ __launch_bounds__(256) void mtg_update_kernel(MtgParams P) {
  const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b < P.B) mtg_lane_update<C, OUT>(P, b);
}


using SolveFn = void (*)(MtgParams, int);
using UpdateFn = void (*)(MtgParams);
template <int H, int D> using GenericCfg = MtgCfg<H, D, 0, 0, 0, 0>;

// per-TU pickers (mtg_generic_hN.hip, mtg_static.hip)
SolveFn mtg_pick_generic_solve(int h, int d, bool extra_outputs);
UpdateFn mtg_pick_generic_update(int h, int d, bool with_cost);
struct MtgStaticEntry {
  int h, d, k, ms, mi, me, dv;
  SolveFn fn[2];
};
const MtgStaticEntry* mtg_find_static(int h, int d, int k, int deriv, const int* mask);

#endif  // MTG_KERNELS_H_
