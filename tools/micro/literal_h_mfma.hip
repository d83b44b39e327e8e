// Evidence microbenchmark (not part of the product): the reference's per-segment contraction
//   H = A^-T Q A^-1   (impl/polynomial_optimization_linear_impl.h:318, two N x N x N products, N = 10)
// done LITERALLY on the FP64 matrix cores (v_mfma_f64_16x16x4_f64, N padded to 16, K padded to 12), one wavefront
// per segment, versus the unit-time scaling identity the shipped kernels use
//   H(T) = T^(1-2d) * S * H(1) * S    (one multiply per entry).
// Prints time per segment, the implied FP64-MFMA utilisation, and the max relative difference of the two results.
// BASELINE.json's north-star reserves MFMA for exactly this contraction; DESIGN.md section 4 cites these numbers
// for why the shipped path does not use it.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

#define MTG_TABLE_QUAL __constant__ const
#include "../../mav_trajectory_generation_amd/csrc/mtg_tables.inc"

constexpr int N = 10, H = 5, DERIV = 4;
constexpr int H1OFF = 400 + DERIV * 100;   // kH1Off[5][4]
constexpr int AINVOFF = 60;                // kAinvLoOff[5]
typedef double v4d __attribute__((ext_vector_type(4)));

// A(T)^-1 entry (row r, col c) from the unit-time table: A(T)^-1 = diag(T^-r) A(1)^-1 S, S = diag(T^(c mod h))
__device__ double ainv_entry(int r, int c, double T, double tinv) {
  if (r >= N || c >= N) return 0.0;
  double a1;
  if (r < H) {
    double f = 1.0;
    for (int i = 2; i <= r; ++i) f *= i;
    a1 = (c == r) ? 1.0 / f : 0.0;
  } else {
    a1 = kAinvLo[AINVOFF + (r - H) * N + c];
  }
  return a1 * pow(tinv, (double)r) * pow(T, (double)(c % H));
}
__device__ double q_entry(int r, int c, double T) {   // impl/...:568-583
  if (r >= N || c >= N || r < DERIV || c < DERIV) return 0.0;
  double br = 1.0, bc = 1.0;
  for (int i = 0; i < DERIV; ++i) { br *= (r - i); bc *= (c - i); }
  const double e = r + c - 2 * DERIV + 1;
  return br * bc * pow(T, e) * 2.0 / e;
}

// one wave per segment; LDS holds Ainv (16x16, zero padded) and the intermediate P = Q * Ainv
__global__ __launch_bounds__(64) void literal_h(const double* times, double* hout, int nseg) {
  __shared__ double sA[16][17], sP[16][17];
  const int lane = threadIdx.x;
  for (int seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
    const double T = times[seg], tinv = 1.0 / T;
    for (int e = lane; e < 256; e += 64) sA[e >> 4][e & 15] = ainv_entry(e >> 4, e & 15, T, tinv);
    __syncthreads();
    const int i = lane & 15, kq = lane >> 4;
    // P = Q * Ainv : A operand = Q[i][k], B operand = Ainv[k][j]; k in steps of 4 (lane>>4 selects k within the step)
    v4d acc = {0, 0, 0, 0};
    for (int k0 = 0; k0 < 12; k0 += 4) {
      const double a = q_entry(i, k0 + kq, T);
      const double b = sA[k0 + kq][i];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    // C/D layout of the f64 MFMA: col = lane & 15, row = (lane >> 4) + 4 * reg
    for (int r = 0; r < 4; ++r) sP[kq + 4 * r][i] = acc[r];
    __syncthreads();
    // H = Ainv^T * P : A operand = Ainv^T[i][k] = Ainv[k][i]
    v4d h = {0, 0, 0, 0};
    for (int k0 = 0; k0 < 12; k0 += 4) {
      const double a = sA[k0 + kq][i];
      const double b = sP[k0 + kq][i];
      h = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, h, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) {
      const int row = kq + 4 * r, col = i;
      if (row < N && col < N) hout[(size_t)seg * N * N + row * N + col] = h[r];
    }
    __syncthreads();
  }
}

// shipped formulation: one LANE per segment, H(T) = T^(1-2d) S H(1) S
__global__ void scaled_h(const double* times, double* hout, int nseg) {
  const int seg = blockIdx.x * blockDim.x + threadIdx.x;
  if (seg >= nseg) return;
  const double T = times[seg], tinv = 1.0 / T;
  double s[H];
  s[0] = 1.0;
  for (int p = 1; p < H; ++p) s[p] = s[p - 1] * T;
  double base = tinv;
  for (int i = 1; i < 2 * DERIV - 1; ++i) base *= tinv;
  for (int r = 0; r < N; ++r)
    for (int c = 0; c < N; ++c) hout[(size_t)seg * N * N + r * N + c] = base * s[r % H] * s[c % H] * kH1[H1OFF + r * N + c];
}

int main() {
  const int nseg = 1 << 20;   // = 131k trajectories of 8 segments
  std::vector<double> t(nseg);
  for (int i = 0; i < nseg; ++i) t[i] = 1.0 + 19.0 * ((i * 2654435761u) % 100000) / 100000.0;
  double *dt, *h1, *h2;
  hipMalloc(&dt, nseg * 8); hipMalloc(&h1, (size_t)nseg * 800); hipMalloc(&h2, (size_t)nseg * 800);
  hipMemcpy(dt, t.data(), nseg * 8, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms_lit = 0, ms_sc = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(literal_h, dim3(256 * 16), dim3(64), 0, 0, dt, h1, nseg);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms_lit, e0, e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(scaled_h, dim3((nseg + 255) / 256), dim3(256), 0, 0, dt, h2, nseg);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms_sc, e0, e1);
  }
  std::vector<double> a(4096 * 100), b(4096 * 100);
  hipMemcpy(a.data(), h1, a.size() * 8, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), h2, b.size() * 8, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int s = 0; s < 4096; ++s) {
    double scale = 0;
    for (int e = 0; e < 100; ++e) scale = fmax(scale, fabs(b[s * 100 + e]));
    for (int e = 0; e < 100; ++e) worst = fmax(worst, fabs(a[s * 100 + e] - b[s * 100 + e]) / scale);
  }
  const double mfma_flops = 6.0 * 2 * 16 * 16 * 4;   // 6 MFMA 16x16x4 per segment
  printf("segments %d\n", nseg);
  printf("literal MFMA f64 : %8.1f us  (%.2f ns/segment; MFMA issue %.1f TFLOP/s = %.1f%% of the 78.6 TF FP64 peak; useful 2*2*N^3 = %.1f TF)\n",
         ms_lit * 1e3, ms_lit * 1e6 / nseg, mfma_flops * nseg / (ms_lit * 1e-3) * 1e-12,
         100 * mfma_flops * nseg / (ms_lit * 1e-3) / 78.6e12, 4000.0 * nseg / (ms_lit * 1e-3) * 1e-12);
  printf("scaling identity : %8.1f us  (%.2f ns/segment, writes the same 800 B/segment: %.2f TB/s)\n", ms_sc * 1e3,
         ms_sc * 1e6 / nseg, 800.0 * nseg / (ms_sc * 1e-3) * 1e-12);
  printf("ratio literal/scaled = %.1fx ; max |H_literal - H_scaled| / max|H| per segment = %.2e\n", ms_lit / ms_sc, worst);
  return 0;
}
