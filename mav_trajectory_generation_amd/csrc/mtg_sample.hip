// mtg_sample.hip -- batched trajectory sampling: the step every consumer runs right
// after the solve.  Replaces, for a batch, Trajectory::evaluate / evaluateRange (src/trajectory.cpp:48-141) +
// Polynomial::evaluate (polynomial.h:137-149) as used by sampleTrajectoryInRange (src/trajectory_sampling.cpp:45-110):
// derivatives 0..n_derivatives-1 of every dimension at t_i = t_start + i*dt.
//
// One lane per (trajectory, sample); lanes of a wave are consecutive samples of (mostly) one trajectory, so the
// coefficient reads are wave-broadcasts served by L1 and the kernel is bound by its output stream.  Each lane's
// n_derivatives*D results are transposed through the wave's LDS slab so the wave writes 1 KiB contiguous per store.
// Differences to the reference, by design: sample times are t_start + i*dt in closed form (the reference
// accumulates dt, a sequential loop); samples past the end of a trajectory are evaluated at its end time and
// reported through n_valid[b] instead of being dropped.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "../../include/mtg_hip.h"

namespace {

constexpr int kThreads = 256;
#ifndef MTG_SAMPLE_CT_WAVES
#define MTG_SAMPLE_CT_WAVES 4   // waves per SIMD the compile-time-shape kernels are held to (A/B knob)
#endif
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

// Per-lane sample descriptor: where the sample lives (coefficient pointer of its segment) and its segment-local time.
struct SampleSite {
  const double* c;   // coeffs of (trajectory, segment), dimension 0
  double local;      // time inside the segment, clamped to the segment's duration
};

// (trajectory, sample) -> segment lookup: the first segment whose accumulated end time exceeds t
// (src/trajectory.cpp:52-66); t at or beyond the last vertex -> last segment, clamped to its end.
// Branch-free scan (no early exit: the loads of all K times are independent and stay in flight together).
__device__ __forceinline__ SampleSite mtg_sample_site(const SampleParams& P, long long b, int s) {
  const double t = P.t_start + P.dt * s;
  const double* tt = P.times + b * P.ts_b;
  double acc = 0.0, seg_start = 0.0, seg_time = 0.0;
  int seg = 0;
  bool found = false;
#pragma unroll 8
  for (int i = 0; i < P.K; ++i) {
    const double ti = tt[(long long)i * P.ts_k];
    if (!found) { seg_time = ti; seg_start = acc; seg = i; }
    acc += ti;
    found = found || acc > t;
  }
  SampleSite r;
  r.local = t - seg_start;
  if (r.local > seg_time) r.local = seg_time;
  r.c = P.coeffs + ((b * P.K + seg) * P.D) * (long long)P.N;
  return r;
}

// Run-time-shape kernel (any N <= 12, any D, any K).  Every WAVE is an independent worker: it owns 64 consecutive
// (trajectory, sample) slots per step, its own LDS slab for the transposition and its own grid-stride sequence -- no
// workgroup barrier anywhere.
template <int ND>   // number of derivatives sampled (compile time: the ND running Horner sums stay in registers)
__global__ __launch_bounds__(kThreads) void mtg_sample_kernel(SampleParams P, long long total) {
  extern __shared__ double lds_all[];
  const int R = ND * P.D;          // results per lane
  const int RP = R | 1;            // odd row stride: conflict-free ds_write_b64
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  double* lds = lds_all + (size_t)wave * 64 * RP;
  // (trajectory, sample) of the wave's first lane, advanced incrementally: one 64-bit division per wave per launch
  const long long nworkers = (long long)gridDim.x * (kThreads / 64);
  const long long step = nworkers * 64;
  const long long step_b = step / P.S;
  const int step_s = (int)(step - step_b * P.S);
  long long base = ((long long)blockIdx.x * (kThreads / 64) + wave) * 64;
  long long blk_b = base / P.S;
  int blk_s = (int)(base - blk_b * P.S);
  auto site_of = [&](long long base_, long long blk_b_, int blk_s_) {
    // lanes past the end of the launch re-evaluate the last sample (their results are never stored)
    long long idx = base_ + lane;
    unsigned off = (unsigned)blk_s_ + (unsigned)lane;
    if (idx >= total) off -= (unsigned)(idx - (total - 1));
    const unsigned q = off / (unsigned)P.S;
    return mtg_sample_site(P, blk_b_ + q, (int)(off - q * (unsigned)P.S));
  };
  double cur[kMaxN], nxt[kMaxN];
  SampleSite site;
  if (base < total) {
    site = site_of(base, blk_b, blk_s);
#pragma unroll
    for (int j = 0; j < kMaxN; ++j) cur[j] = j < P.N ? site.c[j] : 0.0;
  }
  while (base < total) {
    // ---- arithmetic of this chunk: coefficients of one dimension are a register batch (static indices, uniform
    // j < N guards); the next dimension's batch is requested before this one is consumed
    for (int d = 0; d < P.D; ++d) {
      if (d + 1 < P.D) {
        const double* cn = site.c + (d + 1) * P.N;
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
          for (int m = ND - 1; m >= 1; --m) a[m] = __builtin_fma(a[m], site.local, a[m - 1]);
          a[0] = __builtin_fma(a[0], site.local, cur[j]);
        }
      }
      double fact = 1.0;
#pragma unroll
      for (int m = 0; m < ND; ++m) {
        if (m > 1) fact *= (double)m;
        lds[lane * RP + m * P.D + d] = a[m] * fact;
      }
      if (d + 1 < P.D) {
#pragma unroll
        for (int j = 0; j < kMaxN; ++j) cur[j] = nxt[j];
      }
    }
    // ---- inputs of the next chunk, requested ahead of this chunk's stores
    const long long nbase = base + step;
    long long nblk_b = blk_b + step_b;
    int nblk_s = blk_s + step_s;
    if (nblk_s >= P.S) { nblk_s -= P.S; ++nblk_b; }
    if (nbase < total) {
      site = site_of(nbase, nblk_b, nblk_s);
#pragma unroll
      for (int j = 0; j < kMaxN; ++j) cur[j] = j < P.N ? site.c[j] : 0.0;
    }
    // ---- coalesced write-out of this chunk's contiguous [64][R] slab, two doubles (16 B) per lane per store;
    // element e belongs to lane e / R.  The slab starts 16-byte aligned (base is a multiple of 64).  Wave-level
    // hand-off through LDS: LDS operations of one wave execute in order; the compiler is fenced on both sides.
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    const long long slab = base * R;
    const int slab_len = (int)((total - base < 64 ? total - base : 64) * R);
    const int e0 = 2 * lane;
    int owner = e0 / R, r = e0 - owner * R;
    const int qstep2 = 128 / R, rstep2 = 128 - qstep2 * R;
    for (int e = e0; e < slab_len; e += 128) {
      const double v0 = lds[owner * RP + r];
      const int o1 = r + 1 == R ? owner + 1 : owner, r1 = r + 1 == R ? 0 : r + 1;
      if (e + 1 < slab_len) {
        const double v1 = lds[o1 * RP + r1];
        typedef double d2 __attribute__((ext_vector_type(2)));
        d2 v;
        v.x = v0;
        v.y = v1;
        __builtin_nontemporal_store(v, reinterpret_cast<d2*>(P.out + slab + e));   // see the note in the ct kernel
      } else {
        __builtin_nontemporal_store(v0, P.out + slab + e);
      }
      owner += qstep2;
      r += rstep2;
      if (r >= R) { r -= R; ++owner; }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    base = nbase;
    blk_b = nblk_b;
    blk_s = nblk_s;
  }
}

// Compile-time (ND, N, D) kernel for the common shapes with an ODD number of results per sample (R = ND*D: position..
// snap in 3-D is 15) and K <= 8 / K <= 16 segments (KMAX).  What the measurements on the run-time-shape kernel showed (100k x 1000
// samples, 12 GB out: 3.8 ms; same kernel with its stores removed 2.4 ms, with stores AND arithmetic AND coefficient
// loads removed still 1.6 ms; a bare store stream of the same shape 2.15 ms, tools/micro/write_pattern.hip):
//   * index arithmetic, register shuffles of the d-loop and the owner/remainder bookkeeping of the write-out were 40 %
//     of the issue slots.  Here every offset is an immediate: the D*N coefficients of a sample are one batch of loads
//     off one base pointer, a wave's LDS slab IS the output image of its 64 samples (row length R is odd, hence
//     conflict-free without padding) and the write-out is a straight ds_read_b128 -> global_store_dwordx4 copy;
//   * nothing overlapped.  On gfx9-family hardware loads and stores share the vmcnt counter and retire in issue order,
//     so a wave that stores chunk i and THEN requests inputs of chunk i+1 cannot consume them before every store of
//     chunk i is acknowledged: one store-drain latency per step, serialised with the arithmetic.  The loop is therefore
//     software-pipelined with a FIXED issue order per step -- segment times of chunk i+2, coefficients of chunk i+1,
//     stores of chunk i -- straight-line, so that the compiler's waitcnt counts stay exact: arithmetic of chunk i+1
//     waits for "all but the 8 youngest" (the stores), never for a store.
// Only full 64-sample chunks run through the pipeline; the (at most one) partial chunk of a launch takes the plain
// path at the end.
template <int ND, int N, int D, int KMAX>   // KMAX: segment times held in registers (K <= KMAX)
__global__ __launch_bounds__(kThreads, KMAX <= 8 ? MTG_SAMPLE_CT_WAVES : 3) void mtg_sample_kernel_ct(SampleParams P, long long total) {
  constexpr int R = ND * D;
  constexpr int RS = R | 1;        // LDS row stride: odd => the transposing ds_write_b64 are conflict-free
  typedef double d2 __attribute__((ext_vector_type(2)));
  extern __shared__ double lds_all[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  double* lds = lds_all + (size_t)wave * 64 * RS;
  double* row = lds + lane * RS;
  // write-out: pair p = lane + 64*i of the chunk's 64*R/2 output pairs.  Odd R: the slab is the output image, pair p
  // sits at doubles [2p, 2p+1].  Even R: rows are padded by one double; a pair never straddles a row (R/2 pairs each).
  constexpr int kPairs = 64 * R / 2, kIter = (kPairs + 63) / 64;
  int loff[kIter];
#pragma unroll
  for (int i = 0; i < kIter; ++i) {
    const int pr = lane + 64 * i;
    loff[i] = (R % 2) ? 2 * pr : (pr / (R / 2 > 0 ? R / 2 : 1)) * RS + 2 * (pr % (R / 2 > 0 ? R / 2 : 1));
  }
  const long long nworkers = (long long)gridDim.x * (kThreads / 64);
  const long long worker = (long long)blockIdx.x * (kThreads / 64) + wave;
  const long long nfull = total / 64;                 // chunks that exist completely
  const int K = P.K;

  // this lane's (trajectory, sample) in chunk `ch` (clamped to the last full chunk: the pipeline requests up to two
  // chunks past the end, harmlessly)
  auto locate = [&](long long ch, long long& b, int& s_idx) {
    if (ch >= nfull) ch = nfull - 1;
    const long long idx = ch * 64 + lane;
    b = idx / P.S;
    s_idx = (int)(idx - b * P.S);
  };
  auto request_times = [&](long long b, double (&T)[KMAX]) {
    const double* tt = P.times + b * P.ts_b;
#pragma unroll
    for (int i = 0; i < KMAX; ++i) T[i] = tt[(long long)(i < K ? i : K - 1) * P.ts_k];   // branch-free: K <= KMAX
  };
  auto scan = [&](const double (&T)[KMAX], long long b, int s_idx) {
    const double t = P.t_start + P.dt * s_idx;
    double acc = 0.0, seg_start = 0.0, seg_time = T[0];
    int seg = 0;
    bool found = false;
#pragma unroll
    for (int i = 0; i < KMAX; ++i) {
      if (i < K) {                                     // wave-uniform, arithmetic only
        if (!found) { seg_time = T[i]; seg_start = acc; seg = i; }
        acc += T[i];
        found = found || acc > t;
      }
    }
    SampleSite r;
    r.local = t - seg_start;
    if (r.local > seg_time) r.local = seg_time;
    r.c = P.coeffs + ((b * K + seg) * D) * (long long)N;
    return r;
  };
  auto evaluate_into_row = [&](const double (&c)[D * N], double t) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      double a[ND];
#pragma unroll
      for (int m = 0; m < ND; ++m) a[m] = 0.0;
#pragma unroll
      for (int j = N - 1; j >= 0; --j) {
#pragma unroll
        for (int m = ND - 1; m >= 1; --m) a[m] = __builtin_fma(a[m], t, a[m - 1]);
        a[0] = __builtin_fma(a[0], t, c[d * N + j]);
      }
      double fact = 1.0;
#pragma unroll
      for (int m = 0; m < ND; ++m) {
        if (m > 1) fact *= (double)m;
        row[m * D + d] = m > 1 ? a[m] * fact : a[m];
      }
    }
  };
  auto fence = [] {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
  };

  if (worker < nfull) {
    double c[D * N], T[KMAX];
    long long b;
    int s_idx;
    // prologue: site and coefficients of the first chunk, times of the second
    locate(worker, b, s_idx);
    request_times(b, T);
    SampleSite site = scan(T, b, s_idx);
#pragma unroll
    for (int i = 0; i < D * N; ++i) c[i] = site.c[i];
    locate(worker + nworkers, b, s_idx);
    request_times(b, T);
    for (long long ch = worker; ch < nfull; ch += nworkers) {
      evaluate_into_row(c, site.local);                          // needs c: every VMEM op younger than c is a store
      site = scan(T, b, s_idx);                                  // chunk ch + W (its times arrived a step ago)
      locate(ch + 2 * nworkers, b, s_idx);
      request_times(b, T);                                       // issue order: times(ch + 2W) ...
#pragma unroll
      for (int i = 0; i < D * N; ++i) c[i] = site.c[i];          // ... coefficients(ch + W) ...
      fence();
      double* dst = P.out + ch * (64 * R);                       // ... stores(ch).  16-byte aligned: 64 * R is even
      // non-temporal: a plain store stream fills the (write-back, write-allocate) L2 with dirty output lines and
      // evicts the trajectory data every wave is about to read -- read latency under that load is what bounded the
      // kernel (100k x 1000 samples: 3.2 ms with plain stores, 2.2 ms with nt).  The output is never re-read here.
      constexpr int G = 4;   // LDS reads in groups ahead of their stores
#pragma unroll
      for (int i0 = 0; i0 < kIter; i0 += G) {
        d2 v[G];
#pragma unroll
        for (int i = 0; i < G; ++i)
          if (i0 + i < kIter && lane + 64 * (i0 + i) < kPairs) {
            if constexpr (R % 2 == 1) {
              v[i] = *reinterpret_cast<const d2*>(lds + loff[i0 + i]);
            } else {
              v[i].x = lds[loff[i0 + i]];
              v[i].y = lds[loff[i0 + i] + 1];
            }
          }
#pragma unroll
        for (int i = 0; i < G; ++i)
          if (i0 + i < kIter && lane + 64 * (i0 + i) < kPairs)
            __builtin_nontemporal_store(v[i], reinterpret_cast<d2*>(dst + 2 * (lane + 64 * (i0 + i))));
      }
      fence();
    }
  }
  // the partial last chunk of the launch (total % 64 samples): plain path, one wave
  const int tail = (int)(total - nfull * 64);
  if (tail > 0 && worker == (nfull % nworkers)) {
    long long idx = nfull * 64 + (lane < tail ? lane : tail - 1);
    const long long b = idx / P.S;
    const SampleSite site = mtg_sample_site(P, b, (int)(idx - b * P.S));
    double c[D * N];
#pragma unroll
    for (int i = 0; i < D * N; ++i) c[i] = site.c[i];
    evaluate_into_row(c, site.local);
    fence();
    double* dst = P.out + nfull * (64 * R);
    for (int e = lane; e < tail * R; e += 64) dst[e] = lds[(e / R) * RS + e % R];
    fence();
  }
}

using SampleFn = void (*)(SampleParams, long long);
template <int N, int D, int KMAX>
SampleFn mtg_pick_sample_ct(int nd) {
  switch (nd) {
    case 1: return mtg_sample_kernel_ct<1, N, D, KMAX>;
    case 2: return mtg_sample_kernel_ct<2, N, D, KMAX>;
    case 3: return mtg_sample_kernel_ct<3, N, D, KMAX>;
    case 4: return mtg_sample_kernel_ct<4, N, D, KMAX>;
    case 5: return mtg_sample_kernel_ct<5, N, D, KMAX>;
    default: return nullptr;
  }
}
template <int N, int D>
SampleFn mtg_pick_sample_ct(int nd, int k) {
  return k <= 8 ? mtg_pick_sample_ct<N, D, 8>(nd) : mtg_pick_sample_ct<N, D, 16>(nd);
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
extern "C" bool mtg_context_sample_generic(const mtg_context* ctx);   // measurement knob "sample_generic" (mtg_hip_lab.h)

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
  void (*fn)(SampleParams, long long) = nullptr;
  switch (n_derivatives) {
    case 1: fn = mtg_sample_kernel<1>; break;
    case 2: fn = mtg_sample_kernel<2>; break;
    case 3: fn = mtg_sample_kernel<3>; break;
    case 4: fn = mtg_sample_kernel<4>; break;
    case 5: fn = mtg_sample_kernel<5>; break;
    default: return MTG_ERR_UNSUPPORTED;   // position .. snap (sampleTrajectoryInRange samples exactly these five)
  }
  // compile-time shapes (K <= 16): the reference's N = 10 / 12 / 8 in 3-D, N = 10 in 1-D (yaw) and 4-D (x, y, z, yaw)
  SampleFn fast = nullptr;
  if (n_segments <= 16 && total >= 64 && !mtg_context_sample_generic(ctx)) {
    if (n_coeffs == 10 && dimension == 3) fast = mtg_pick_sample_ct<10, 3>(n_derivatives, n_segments);
    else if (n_coeffs == 12 && dimension == 3) fast = mtg_pick_sample_ct<12, 3>(n_derivatives, n_segments);
    else if (n_coeffs == 8 && dimension == 3) fast = mtg_pick_sample_ct<8, 3>(n_derivatives, n_segments);
    else if (n_coeffs == 10 && dimension == 1) fast = mtg_pick_sample_ct<10, 1>(n_derivatives, n_segments);
    else if (n_coeffs == 10 && dimension == 4) fast = mtg_pick_sample_ct<10, 4>(n_derivatives, n_segments);
    else if (n_coeffs == 12 && dimension == 4) fast = mtg_pick_sample_ct<12, 4>(n_derivatives, n_segments);
  }
  // persistent waves: as many workgroups as the device holds at once (occupancy x CUs), several chunks each
  SampleFn launch_fn = fast ? fast : fn;
  const size_t launch_lds = fast ? (size_t)kThreads * (R | 1) * sizeof(double) : lds;
  int per_cu = 0, n_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, launch_fn, kThreads, launch_lds) != hipSuccess || per_cu < 1)
    per_cu = 4;
  if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || n_cu < 1) n_cu = 256;
  if (blocks > (long long)per_cu * n_cu) blocks = (long long)per_cu * n_cu;
  hipLaunchKernelGGL(launch_fn, dim3((unsigned)blocks), dim3(kThreads), launch_lds, (hipStream_t)stream, P, total);
  if (n_valid)
    hipLaunchKernelGGL(mtg_sample_valid_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, (hipStream_t)stream, P);
  return hipGetLastError() == hipSuccess ? MTG_OK : MTG_ERR_DEVICE;
}
