// mtg_sample.hip -- batched trajectory sampling: the step every consumer runs right
// after the solve.  Replaces, for a batch, Trajectory::evaluate / evaluateRange (src/trajectory.cpp:48-141) +
// Polynomial::evaluate (polynomial.h:137-149) as used by sampleTrajectoryInRange (src/trajectory_sampling.cpp:45-110):
// derivatives 0..n_derivatives-1 of every dimension at t_i = t_start + i*dt.
//
// One lane per (trajectory, sample); lanes of a wave are consecutive samples of (mostly) one trajectory, so the
// coefficient reads are wave-broadcasts served by L1 and the kernel is bound by its output stream.  Each lane's
// n_derivatives*D results are transposed through LDS so the wave writes 512 contiguous bytes per store.
// Differences to the reference, by design: sample times are t_start + i*dt in closed form (the reference
// accumulates dt, a sequential loop); samples past the end of a trajectory are evaluated at its end time and
// reported through n_valid[b] instead of being dropped.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/mtg_hip.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxN = MTG_MAX_N;

struct SampleParams {
  const double* coeffs;   // [B][K][D][N]
  const double* times;    // times[b*ts_b + k*ts_k]
  long long ts_b, ts_k;
  double* out;            // [B][S][ND][D]
  int* n_valid;           // [B] or null
  long long B;
  int N, K, D, S, ND;
  double t_start, dt;
};

template <int ND>   // number of derivatives sampled (compile time: the ND running Horner sums stay in registers)
__global__ __launch_bounds__(kThreads) void mtg_sample_kernel(SampleParams P, long long total) {
  extern __shared__ double lds[];
  const int R = ND * P.D;          // results per lane
  const int RP = R | 1;            // odd row stride: conflict-free ds_write_b64
  // (trajectory, sample) of the block's first lane, advanced incrementally: one 64-bit division per thread per launch
  const long long step = (long long)gridDim.x * kThreads;
  const long long step_b = step / P.S;
  const int step_s = (int)(step - step_b * P.S);
  long long blk_b = ((long long)blockIdx.x * kThreads) / P.S;
  int blk_s = (int)((long long)blockIdx.x * kThreads - blk_b * P.S);
  for (long long base = (long long)blockIdx.x * kThreads; base < total; base += step) {
    if (base + threadIdx.x < total) {
      const unsigned off = (unsigned)blk_s + threadIdx.x;
      const unsigned q = off / (unsigned)P.S;
      const long long b = blk_b + q;
      const int s = (int)(off - q * (unsigned)P.S);
      const double t = P.t_start + P.dt * s;
      // segment lookup: the first segment whose accumulated end time exceeds t (src/trajectory.cpp:52-66);
      // t at or beyond the last vertex -> last segment, clamped to its end
      const double* tt = P.times + b * P.ts_b;
      // branch-free scan (no early exit: the loads of all K times are independent and stay in flight together)
      double acc = 0.0, seg_start = 0.0, seg_time = 0.0;
      int seg = 0;
      bool found = false;
#pragma unroll 4
      for (int i = 0; i < P.K; ++i) {
        const double ti = tt[(long long)i * P.ts_k];
        if (!found) { seg_time = ti; seg_start = acc; seg = i; }
        acc += ti;
        found = found || acc > t;
      }
      double local = t - seg_start;
      if (local > seg_time) local = seg_time;
      const double* c = P.coeffs + ((b * P.K + seg) * P.D) * (long long)P.N;
      // coefficients of one dimension are loaded as a batch (static register indices, uniform j < N guards) and the
      // next dimension's batch is issued before this one is consumed: one memory latency per lane, not N * D
      double cur[kMaxN], nxt[kMaxN];
#pragma unroll
      for (int j = 0; j < kMaxN; ++j) cur[j] = j < P.N ? c[j] : 0.0;
      for (int d = 0; d < P.D; ++d) {
        if (d + 1 < P.D) {
          const double* cn = c + (d + 1) * P.N;
#pragma unroll
          for (int j = 0; j < kMaxN; ++j) nxt[j] = j < P.N ? cn[j] : 0.0;
        }
        // all derivatives in one pass: a[m] accumulates p^(m)(t) / m!  (Horner with derivatives; polynomial.h:137-149
        // evaluates each derivative with its own Horner loop -- same values to rounding)
        double a[ND];
#pragma unroll
        for (int m = 0; m < ND; ++m) a[m] = 0.0;
#pragma unroll
        for (int j = kMaxN - 1; j >= 0; --j) {
          if (j < P.N) {
#pragma unroll
            for (int m = ND - 1; m >= 1; --m) a[m] = __builtin_fma(a[m], local, a[m - 1]);
            a[0] = __builtin_fma(a[0], local, cur[j]);
          }
        }
        double fact = 1.0;
#pragma unroll
        for (int m = 0; m < ND; ++m) {
          if (m > 1) fact *= (double)m;
          lds[threadIdx.x * RP + m * P.D + d] = a[m] * fact;
        }
#pragma unroll
        for (int j = 0; j < kMaxN; ++j) cur[j] = nxt[j];
      }
    }
    blk_b += step_b;
    blk_s += step_s;
    if (blk_s >= P.S) { blk_s -= P.S; ++blk_b; }
    __syncthreads();
    // coalesced write-out of this block's contiguous [kThreads][R] slab, two doubles (16 B) per lane per store;
    // element e belongs to lane e / R.  The slab starts 16-byte aligned (base is a multiple of kThreads).
    const long long slab = base * R;
    const int slab_len = (int)((total - base < kThreads ? total - base : kThreads) * R);
    const int e0 = 2 * (int)threadIdx.x;
    int owner = e0 / R, r = e0 - owner * R;
    const int qstep2 = (2 * kThreads) / R, rstep2 = 2 * kThreads - qstep2 * R;
    for (int e = e0; e < slab_len; e += 2 * kThreads) {
      const double v0 = lds[owner * RP + r];
      const int o1 = r + 1 == R ? owner + 1 : owner, r1 = r + 1 == R ? 0 : r + 1;
      if (e + 1 < slab_len) {
        const double v1 = lds[o1 * RP + r1];
        *reinterpret_cast<double2*>(P.out + slab + e) = make_double2(v0, v1);
      } else {
        P.out[slab + e] = v0;
      }
      owner += qstep2;
      r += rstep2;
      if (r >= R) { r -= R; ++owner; }
    }
    __syncthreads();
  }
}

__global__ void mtg_sample_valid_kernel(SampleParams P) {
  const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= P.B) return;
  double total_time = 0.0;
  for (int i = 0; i < P.K; ++i) total_time += P.times[b * P.ts_b + (long long)i * P.ts_k];
  // number of i in [0, S) with t_start + i*dt <= total_time: closed-form estimate, then exact with the same
  // expression the sampler uses for t_i (so the boundary sample is counted consistently)
  int nv = 0;
  if (total_time >= P.t_start) {
    const double q = (total_time - P.t_start) / P.dt;
    nv = q >= (double)(P.S - 1) ? P.S : (int)q + 1;
    while (nv < P.S && P.t_start + P.dt * nv <= total_time) ++nv;
    while (nv > 0 && P.t_start + P.dt * (nv - 1) > total_time) --nv;
  }
  P.n_valid[b] = nv;
}

}  // namespace

// C ABI (declared in include/mtg_hip.h).  The context type is opaque here: only its stream / device are needed.
extern "C" int mtg_context_stream_device(mtg_context* ctx, void** stream, int* device);

extern "C" int mtg_sample_range(mtg_context* ctx, int32_t n_coeffs, int32_t n_segments, int32_t dimension, int64_t batch,
                                const double* coeffs, const double* times, int64_t times_stride_b, int64_t times_stride_k,
                                double t_start, double dt, int32_t n_samples, int32_t n_derivatives, double* out,
                                int32_t* n_valid) {
  if (!ctx || !coeffs || !times || !out || batch < 0 || n_samples < 1 || n_derivatives < 1 || n_coeffs < 2 ||
      n_coeffs > MTG_MAX_N || n_segments < 1 || dimension < 1 || !(dt > 0.0))
    return MTG_ERR_INVALID_ARGUMENT;
  if (reinterpret_cast<uintptr_t>(out) & 15u) return MTG_ERR_INVALID_ARGUMENT;   // 16-byte stores
  if (n_derivatives > 5 || n_derivatives * dimension > 64) return MTG_ERR_UNSUPPORTED;
  if (batch == 0) return MTG_OK;
  void* stream = nullptr;
  int device = 0;
  int rc = mtg_context_stream_device(ctx, &stream, &device);
  if (rc != MTG_OK) return rc;
  if (hipSetDevice(device) != hipSuccess) return MTG_ERR_DEVICE;
  SampleParams P;
  P.coeffs = coeffs; P.times = times; P.ts_b = times_stride_b; P.ts_k = times_stride_k; P.out = out; P.n_valid = n_valid;
  P.B = batch; P.N = n_coeffs; P.K = n_segments; P.D = dimension; P.S = n_samples; P.ND = n_derivatives;
  P.t_start = t_start; P.dt = dt;
  const long long total = (long long)batch * n_samples;
  const int R = n_derivatives * dimension;
  const size_t lds = (size_t)kThreads * (R | 1) * sizeof(double);
  long long blocks = (total + kThreads - 1) / kThreads;
  if (blocks > 256 * 16) blocks = 256 * 16;
  void (*fn)(SampleParams, long long) = nullptr;
  switch (n_derivatives) {
    case 1: fn = mtg_sample_kernel<1>; break;
    case 2: fn = mtg_sample_kernel<2>; break;
    case 3: fn = mtg_sample_kernel<3>; break;
    case 4: fn = mtg_sample_kernel<4>; break;
    case 5: fn = mtg_sample_kernel<5>; break;
    default: return MTG_ERR_UNSUPPORTED;   // position .. snap (sampleTrajectoryInRange samples exactly these five)
  }
  hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(kThreads), lds, (hipStream_t)stream, P, total);
  if (n_valid)
    hipLaunchKernelGGL(mtg_sample_valid_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, (hipStream_t)stream, P);
  return hipGetLastError() == hipSuccess ? MTG_OK : MTG_ERR_DEVICE;
}
