// mtg_literal.hip -- EVIDENCE variant, reachable through include/mtg_hip_lab.h only (mtg_lab_segment_cost_matrices): the
// reference's per-segment contraction
//     H_k = A_k^-T Q_k A_k^-1                                (impl/polynomial_optimization_linear_impl.h:318)
// (a) done LITERALLY as two N x N x N products on the FP64 matrix cores (v_mfma_f64_16x16x4_f64; N <= 12 padded to one
//     16 x 16 tile, the contraction length padded to a multiple of 4), one wavefront per segment, A^-1 and Q assembled
//     per segment as the reference does (A^-1 from its closed form, Q from LIN:568-583) -- SURVEY.md section 7 "K1-alt" and
//     the MFMA clause of BASELINE.json's north_star; and
// (b) by the unit-time scaling identity  H(T) = T^(1-2d) S H(1) S  the solve kernels use (one multiply per entry of a
//     constant table; mtg_lane.h).
// DESIGN.md section 4 cites the measurement (profiles/r01_literal_h_mfma.txt: the literal form is 4.9x slower at 5.5 % of the
// FP64-MFMA peak -- a 10 x 10 x 10 product fills 24 % of the three 16 x 16 x 4 tiles it occupies, and the operands have to be
// built first); tests/test_gpu_literal_mfma.py checks both forms against the 50-digit oracle.  No solve kernel calls this.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <new>

#include "../../include/mtg_hip_lab.h"

#define MTG_TABLE_QUAL __constant__ const
#include "mtg_tables.inc"

extern "C" int mtg_context_stream_device(mtg_context* ctx, void** stream, int* device);

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

// One wave per segment (grid-stride).  LDS: powers of T and 1 / T, A^-1 (16 x 16, zero padded), P = Q A^-1.
template <int HH>
__global__ __launch_bounds__(64) void literal_kernel(const double* __restrict__ times, double* __restrict__ hout, long long nseg,
                                                      int deriv) {
  constexpr int N = 2 * HH, KP = (N + 3) / 4 * 4;
  __shared__ double sA[16][17], sP[16][17], tp[2 * N + 2], ti[N + 1];
  const int lane = threadIdx.x;
  const int aoff = kAinvLoOff[HH];
  for (long long seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
    const double T = times[seg], tinv = 1.0 / T;
    if (lane == 0) {
      tp[0] = 1.0; ti[0] = 1.0;
      for (int e = 1; e < 2 * N + 2; ++e) tp[e] = tp[e - 1] * T;
      for (int e = 1; e <= N; ++e) ti[e] = ti[e - 1] * tinv;
    }
    __syncthreads();
    // A(T)^-1 = diag(T^-r) A(1)^-1 diag(T^(c mod h)): upper half diag(1 / r!), lower half from the unit-time table
    for (int e = lane; e < 256; e += 64) {
      const int r = e >> 4, c = e & 15;
      double v = 0.0;
      if (r < N && c < N) {
        double a1;
        if (r < HH) {
          double f = 1.0;
          for (int i = 2; i <= r; ++i) f *= i;
          a1 = c == r ? 1.0 / f : 0.0;
        } else {
          a1 = kAinvLo[aoff + (r - HH) * N + c];
        }
        v = a1 * ti[r] * tp[c % HH];
      }
      sA[r][c] = v;
    }
    __syncthreads();
    const int i = lane & 15, kq = lane >> 4;
    // P = Q A^-1: A operand Q[i][k] (LIN:568-583), B operand A^-1[k][j]; lane >> 4 selects k inside a step of 4
    v4d acc = {0, 0, 0, 0};
    for (int k0 = 0; k0 < KP; k0 += 4) {
      const int r = i, c = k0 + kq;
      double q = 0.0;
      if (r < N && c < N && r >= deriv && c >= deriv) {
        double br = 1.0, bc = 1.0;
        for (int m = 0; m < deriv; ++m) { br *= (double)(r - m); bc *= (double)(c - m); }
        const int ex = r + c - 2 * deriv + 1;
        q = br * bc * tp[ex] * 2.0 / (double)ex;
      }
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(q, sA[k0 + kq][i], acc, 0, 0, 0);
    }
    // C / D layout of the f64 MFMA: column = lane & 15, row = (lane >> 4) + 4 * register
    for (int r = 0; r < 4; ++r) sP[kq + 4 * r][i] = acc[r];
    __syncthreads();
    // H = A^-T P: A operand (A^-T)[i][k] = A^-1[k][i]
    v4d h = {0, 0, 0, 0};
    for (int k0 = 0; k0 < KP; k0 += 4) h = __builtin_amdgcn_mfma_f64_16x16x4f64(sA[k0 + kq][i], sP[k0 + kq][i], h, 0, 0, 0);
    for (int r = 0; r < 4; ++r) {
      const int row = kq + 4 * r;
      if (row < N && i < N) hout[(size_t)seg * N * N + row * N + i] = h[r];
    }
    __syncthreads();
  }
}

// the solve kernels' formulation, one lane per segment
template <int HH>
__global__ __launch_bounds__(256) void scaled_kernel(const double* __restrict__ times, double* __restrict__ hout, long long nseg,
                                                      int deriv) {
  constexpr int N = 2 * HH;
  const long long seg = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (seg >= nseg) return;
  const double T = times[seg], tinv = 1.0 / T;
  double s[HH];
  s[0] = 1.0;
  for (int p = 1; p < HH; ++p) s[p] = s[p - 1] * T;
  double base = tinv;                      // T^(1 - 2 d)  (d = 0: T)
  if (deriv == 0) base = T;
  for (int i = 1; i < 2 * deriv - 1; ++i) base *= tinv;
  const int off = kH1Off[HH][deriv];
  for (int r = 0; r < N; ++r)
    for (int c = 0; c < N; ++c) hout[(size_t)seg * N * N + r * N + c] = base * s[r % HH] * s[c % HH] * kH1[off + r * N + c];
}

template <int HH>
void launch(int variant, hipStream_t st, const double* times, double* hout, long long nseg, int deriv) {
  if (variant == 1) {
    const long long g = nseg < 256 * 16 ? nseg : 256 * 16;
    hipLaunchKernelGGL(literal_kernel<HH>, dim3((unsigned)g), dim3(64), 0, st, times, hout, nseg, deriv);
  } else {
    hipLaunchKernelGGL(scaled_kernel<HH>, dim3((unsigned)((nseg + 255) / 256)), dim3(256), 0, st, times, hout, nseg, deriv);
  }
}

}  // namespace

extern "C" int mtg_lab_segment_cost_matrices(mtg_context* ctx, int32_t n_coeffs, int32_t derivative, int64_t n_segments,
                                             const double* times, double* h_out, int32_t variant) {
  if (ctx == nullptr || times == nullptr || h_out == nullptr || n_segments < 0 || (variant != 0 && variant != 1))
    return MTG_ERR_INVALID_ARGUMENT;
  if (n_coeffs < 2 || n_coeffs > 12 || (n_coeffs & 1) || derivative < 0 || derivative >= n_coeffs / 2) return MTG_ERR_INVALID_ARGUMENT;
  if (n_segments == 0) return MTG_OK;
  void* stream = nullptr;
  int device = 0;
  const int rc = mtg_context_stream_device(ctx, &stream, &device);
  if (rc != MTG_OK) return rc;
  if (hipSetDevice(device) != hipSuccess) return MTG_ERR_DEVICE;
  hipStream_t st = (hipStream_t)stream;
  switch (n_coeffs / 2) {
    case 1: launch<1>(variant, st, times, h_out, n_segments, derivative); break;
    case 2: launch<2>(variant, st, times, h_out, n_segments, derivative); break;
    case 3: launch<3>(variant, st, times, h_out, n_segments, derivative); break;
    case 4: launch<4>(variant, st, times, h_out, n_segments, derivative); break;
    case 5: launch<5>(variant, st, times, h_out, n_segments, derivative); break;
    default: launch<6>(variant, st, times, h_out, n_segments, derivative); break;
  }
  return hipGetLastError() == hipSuccess ? MTG_OK : MTG_ERR_DEVICE;
}

// ---- shader-clock probe (include/mtg_hip_lab.h: mtg_lab_clock_probe_*) --------------------------------------------------------
// One wavefront on a stream of its own reads the shader-clock counter (s_memtime) and the constant 100 MHz counter
// (s_memrealtime) when it starts, sleeps in s_sleep for `duration_us` of the constant clock, and reads both again: ticks of the
// first per tick of the second = the shader clock WHILE whatever else the caller has running on the device runs (the FP64-heavy
// solve kernels pull it from 2.4 GHz down to 1.75-1.96 GHz, profiles/r05_long_timeline.txt).  The wave occupies one SIMD slot
// and issues one instruction per ~64 cycles: it does not compete with the measured work.
namespace {
__global__ __launch_bounds__(64) void clock_probe_kernel(long long* out, long long duration_ticks) {
  if (threadIdx.x != 0) return;
  const long long r0 = wall_clock64(), c0 = clock64();
  long long r1 = r0;
  while (r1 - r0 < duration_ticks) {
    __builtin_amdgcn_s_sleep(16);
    r1 = wall_clock64();
  }
  const long long c1 = clock64();
  r1 = wall_clock64();
  out[0] = c1 - c0;
  out[1] = r1 - r0;
}
}  // namespace

struct mtg_lab_clock_probe {
  int device;
  hipStream_t stream;
  long long* host;     // pinned, mapped: [shader ticks, 100 MHz ticks]
};

extern "C" int mtg_lab_clock_probe_start(mtg_context* ctx, double duration_us, mtg_lab_clock_probe** out) {
  if (ctx == nullptr || out == nullptr || !(duration_us > 0.0) || duration_us > 60e6) return MTG_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  void* stream = nullptr;
  int device = 0;
  const int rc = mtg_context_stream_device(ctx, &stream, &device);
  if (rc != MTG_OK) return rc;
  if (hipSetDevice(device) != hipSuccess) return MTG_ERR_DEVICE;
  mtg_lab_clock_probe* p = new (std::nothrow) mtg_lab_clock_probe{device, nullptr, nullptr};
  if (!p) return MTG_ERR_DEVICE;
  if (hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) != hipSuccess ||
      hipHostMalloc((void**)&p->host, 2 * sizeof(long long), hipHostMallocDefault) != hipSuccess) {
    if (p->stream) hipStreamDestroy(p->stream);
    delete p;
    return MTG_ERR_DEVICE;
  }
  p->host[0] = p->host[1] = 0;
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, p->stream, p->host, (long long)(duration_us * 100.0));
  if (hipGetLastError() != hipSuccess) { hipHostFree(p->host); hipStreamDestroy(p->stream); delete p; return MTG_ERR_DEVICE; }
  *out = p;
  return MTG_OK;
}

extern "C" int mtg_lab_clock_probe_finish(mtg_lab_clock_probe* p, double* shader_mhz, double* measured_us) {
  if (p == nullptr) return MTG_ERR_INVALID_ARGUMENT;
  int rc = MTG_OK;
  if (hipSetDevice(p->device) != hipSuccess || hipStreamSynchronize(p->stream) != hipSuccess) rc = MTG_ERR_DEVICE;
  if (rc == MTG_OK && p->host[1] > 0) {
    if (shader_mhz) *shader_mhz = (double)p->host[0] / ((double)p->host[1] * 0.01);
    if (measured_us) *measured_us = (double)p->host[1] * 0.01;
  } else if (rc == MTG_OK) {
    rc = MTG_ERR_DEVICE;
  }
  hipHostFree(p->host);
  hipStreamDestroy(p->stream);
  delete p;
  return rc;
}
