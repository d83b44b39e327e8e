// mtg_workload.hip -- device-side synthetic inputs and an on-device result check, for consumers of the C ABI that have no
// tensor library at hand (kernels K3 / K4): benchmarks and parity checks of large batches never cross
// PCIe.
//
// mtg_generate_waypoints: random-waypoint batches with the semantics of the reference's test generators
// (src/vertex.cpp:27-82 createRandomVertices: positions uniform in a box, start / goal at rest; :255-272
// estimateSegmentTimesNfabian: t = 2 d / v_max * (1 + magic * v_max / a_max * exp(-2 d / v_max))).  One thread per
// trajectory, counter-based random numbers (a hash of (seed, trajectory, vertex, dimension, draw) -- reproducible,
// independent of the launch geometry); distribution-equivalent, not bit-equal, to the mt19937 original (the bit-exact
// generator is test infrastructure).  The Python package's torch generator (workload.py) has the same semantics.
//
// mtg_compare_coefficients: max over polynomials of ||a - b||_inf / ||b||_inf and the max absolute element difference of two
// coefficient buffers -- the norm-wise measure the parity tests use.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/mtg_hip.h"

extern "C" int mtg_context_stream_device(mtg_context* ctx, void** stream, int* device);
extern "C" int mtg_plan_context_tables(const mtg_plan* plan, mtg_context** ctx, int* n_coeffs, int* dimension, int* n_segments,
                                       const int** device_masks);

namespace {

__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {   // splitmix64 finaliser
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
// uniform in [0, 1): 53 random bits
__device__ __forceinline__ double uniform01(unsigned long long seed, long long b, int v, int dim, int draw) {
  unsigned long long h = mix64(seed ^ mix64((unsigned long long)b));
  h = mix64(h ^ ((unsigned long long)(unsigned)v << 40) ^ ((unsigned long long)(unsigned)dim << 20) ^ (unsigned long long)(unsigned)draw);
  return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}

struct GenParams {
  double* times; long long ts_b, ts_k;
  double* dfix;  long long fs_b, fs_d, fs_c;
  const int* mask;         // [K + 1] fixed masks (device)
  long long B;
  int K, D, H;
  unsigned long long seed;
  double box, v_max, a_max, magic;
  int yaw_last;            // the last of 4 dimensions is a yaw angle in [-3 pi, 3 pi]
};

constexpr int kMaxD = 8;

__global__ __launch_bounds__(256) void mtg_gen_waypoints_kernel(GenParams P) {
  const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= P.B) return;
  double prev[kMaxD], cur[kMaxD];
  int col = 0;
  for (int v = 0; v <= P.K; ++v) {
    // position of vertex v: uniform in the box, re-drawn (up to 8 times) while closer than 0.2 to the previous vertex
    double dist = 0.0;
    for (int attempt = 0; attempt < 9; ++attempt) {
      double d2 = 0.0;
      for (int dm = 0; dm < P.D; ++dm) {
        const double u = uniform01(P.seed, b, v, dm, attempt) * 2.0 - 1.0;
        cur[dm] = (P.yaw_last && dm == P.D - 1) ? u * (3.0 * 3.14159265358979323846) : u * P.box;
        if (v > 0) d2 += (cur[dm] - prev[dm]) * (cur[dm] - prev[dm]);
      }
      dist = sqrt(d2);
      if (v == 0 || dist > 0.2) break;
    }
    if (v > 0)
      P.times[b * P.ts_b + (long long)(v - 1) * P.ts_k] =
          dist / P.v_max * 2.0 * (1.0 + P.magic * P.v_max / P.a_max * exp(-dist / P.v_max * 2.0));
    const int m = P.mask[v];
    for (int p = 0; p < P.H; ++p) {
      if (!((m >> p) & 1)) continue;
      // fixed derivative p of vertex v: position; zero at the end vertices (start / goal at rest); at interior vertices a
      // random direction times a random radius <= v_max (p = 1), a_max (p = 2), 1 (higher)
      double val[kMaxD];
      if (p == 0) {
        for (int dm = 0; dm < P.D; ++dm) val[dm] = cur[dm];
      } else if (v == 0 || v == P.K) {
        for (int dm = 0; dm < P.D; ++dm) val[dm] = 0.0;
      } else {
        double n2 = 0.0;
        for (int dm = 0; dm < P.D; ++dm) {   // Gaussian direction (Box-Muller)
          const double u1 = uniform01(P.seed, b, v, dm, 16 + 2 * p), u2 = uniform01(P.seed, b, v, dm, 17 + 2 * p);
          val[dm] = sqrt(-2.0 * log(1.0 - u1)) * cos(6.283185307179586 * u2);
          n2 += val[dm] * val[dm];
        }
        const double scale = p == 1 ? P.v_max : (p == 2 ? P.a_max : 1.0);
        const double r = uniform01(P.seed, b, v, P.D, 16 + 2 * p) * scale / sqrt(n2 > 0.0 ? n2 : 1.0);
        for (int dm = 0; dm < P.D; ++dm) val[dm] *= r;
      }
      for (int dm = 0; dm < P.D; ++dm) P.dfix[b * P.fs_b + (long long)dm * P.fs_d + (long long)col * P.fs_c] = val[dm];
      ++col;
    }
    for (int dm = 0; dm < P.D; ++dm) prev[dm] = cur[dm];
  }
}

// non-negative doubles order like their bit patterns: atomicMax on the 64-bit integer view
__device__ __forceinline__ void atomic_max_nonneg(double* addr, double v) {
  atomicMax(reinterpret_cast<unsigned long long*>(addr), (unsigned long long)__double_as_longlong(v));
}

__global__ __launch_bounds__(256) void mtg_compare_kernel(const double* __restrict__ a, const double* __restrict__ b, long long n_poly,
                                                          int N, double* out) {
  double rel = 0.0, ab = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_poly; i += (long long)gridDim.x * blockDim.x) {
    double num = 0.0, den = 0.0;
    bool bad = false;
    for (int j = 0; j < N; ++j) {
      const double x = a[i * N + j], y = b[i * N + j];
      bad = bad || !(x == x) || !(y == y);
      num = fmax(num, fabs(x - y));
      den = fmax(den, fabs(y));
    }
    if (bad) { num = 1.0e300; den = 1.0; }        // a NaN anywhere saturates BOTH outputs (relative and absolute)
    ab = fmax(ab, num);
    rel = fmax(rel, den > 0.0 ? fmin(num / den, 1.0e300) : (num > 0.0 ? 1.0e300 : 0.0));
  }
  for (int off = 32; off > 0; off >>= 1) {
    rel = fmax(rel, __shfl_xor(rel, off));
    ab = fmax(ab, __shfl_xor(ab, off));
  }
  if ((threadIdx.x & 63) == 0) {
    atomic_max_nonneg(out, rel);
    atomic_max_nonneg(out + 1, ab);
  }
}

}  // namespace

extern "C" int mtg_generate_waypoints(mtg_plan* plan, int64_t batch, const mtg_layout* layout, uint64_t seed, double box,
                                      double v_max, double a_max, int32_t yaw_dimension, double* times, double* d_fixed) {
  if (!plan || !layout || !times || batch < 0 || !(box > 0.0) || !(v_max > 0.0) || !(a_max > 0.0)) return MTG_ERR_INVALID_ARGUMENT;
  mtg_context* ctx = nullptr;
  int N = 0, D = 0, K = 0;
  const int* masks = nullptr;
  int rc = mtg_plan_context_tables(plan, &ctx, &N, &D, &K, &masks);
  if (rc != MTG_OK) return rc;
  if (D > kMaxD) return MTG_ERR_UNSUPPORTED;
  mtg_plan_info info;
  rc = mtg_plan_get_info(plan, &info);
  if (rc != MTG_OK) return rc;
  if (info.n_fixed > 0 && !d_fixed) return MTG_ERR_INVALID_ARGUMENT;
  if (batch == 0) return MTG_OK;
  void* stream = nullptr;
  int device = 0;
  rc = mtg_context_stream_device(ctx, &stream, &device);
  if (rc != MTG_OK) return rc;
  if (hipSetDevice(device) != hipSuccess) return MTG_ERR_DEVICE;
  GenParams P;
  P.times = times; P.ts_b = layout->times_stride_b; P.ts_k = layout->times_stride_k;
  P.dfix = d_fixed; P.fs_b = layout->fixed_stride_b; P.fs_d = layout->fixed_stride_d; P.fs_c = layout->fixed_stride_c;
  P.mask = masks; P.B = batch; P.K = K; P.D = D; P.H = N / 2; P.seed = seed;
  P.box = box; P.v_max = v_max; P.a_max = a_max; P.magic = 6.5;   // the reference's default magic_fabian_constant
  P.yaw_last = (yaw_dimension != 0 && D == 4) ? 1 : 0;
  hipLaunchKernelGGL(mtg_gen_waypoints_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, (hipStream_t)stream, P);
  return hipGetLastError() == hipSuccess ? MTG_OK : MTG_ERR_DEVICE;
}

extern "C" int mtg_compare_coefficients(mtg_context* ctx, const double* a, const double* b, int64_t n_polynomials,
                                        int32_t n_coeffs, double* max_normwise_rel, double* max_abs) {
  if (!ctx || !a || !b || n_polynomials < 0 || n_coeffs < 1 || (!max_normwise_rel && !max_abs)) return MTG_ERR_INVALID_ARGUMENT;
  void* stream = nullptr;
  int device = 0;
  int rc = mtg_context_stream_device(ctx, &stream, &device);
  if (rc != MTG_OK) return rc;
  if (hipSetDevice(device) != hipSuccess) return MTG_ERR_DEVICE;
  double h[2] = {0.0, 0.0};
  if (n_polynomials > 0) {
    double* d_out = nullptr;
    if (hipMalloc((void**)&d_out, 2 * sizeof(double)) != hipSuccess) return MTG_ERR_DEVICE;
    hipStream_t st = (hipStream_t)stream;
    bool ok = hipMemsetAsync(d_out, 0, 2 * sizeof(double), st) == hipSuccess;
    const long long blocks = (n_polynomials + 255) / 256;
    if (ok) hipLaunchKernelGGL(mtg_compare_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, a, b,
                               (long long)n_polynomials, (int)n_coeffs, d_out);
    ok = ok && hipMemcpyAsync(h, d_out, 2 * sizeof(double), hipMemcpyDeviceToHost, st) == hipSuccess &&
         hipStreamSynchronize(st) == hipSuccess;
    hipFree(d_out);
    if (!ok) return MTG_ERR_DEVICE;
  }
  if (max_normwise_rel) *max_normwise_rel = h[0];
  if (max_abs) *max_abs = h[1];
  return MTG_OK;
}
