// Microbenchmark: FP64 VALU issue interval and dependent latency on gfx950, one wave per SIMD
// and several; plus v_rcp_f64, v_accvgpr moves, v_readlane.  Prints cycles per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CHAINS>
__global__ void k_fma(double* out, long long* cyc, int iters, double a, double b) {
  double x[CHAINS];
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) x[i] = threadIdx.x * 1e-3 + i;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < CHAINS; ++i) x[i] = __builtin_fma(x[i], a, b);
    }
  }
  long long t1 = clock64();
  double s = 0;
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int CHAINS>
__global__ void k_mul(double* out, long long* cyc, int iters, double a) {
  double x[CHAINS];
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) x[i] = threadIdx.x * 1e-3 + i + 1;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < CHAINS; ++i) x[i] = x[i] * a;
    }
  }
  long long t1 = clock64();
  double s = 0;
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int CHAINS>
__global__ void k_rcp(double* out, long long* cyc, int iters) {
  double x[CHAINS];
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) x[i] = threadIdx.x * 1e-3 + i + 1.5;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < CHAINS; ++i) x[i] = __builtin_amdgcn_rcp(x[i]);
    }
  }
  long long t1 = clock64();
  double s = 0;
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// fma with one SGPR (uniform) operand
template <int CHAINS>
__global__ void k_fma_sgpr(double* out, long long* cyc, int iters, double a, double b) {
  double x[CHAINS];
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) x[i] = threadIdx.x * 1e-3 + i;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < CHAINS; ++i) x[i] = __builtin_fma(x[i], x[(i + 1) % CHAINS], a);
    }
  }
  long long t1 = clock64();
  double s = 0;
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + b;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// half of the lanes masked off: does a wave64 FP64 op with EXEC = low 32 lanes take half the passes?
template <int CHAINS>
__global__ void k_fma_half(double* out, long long* cyc, int iters, double a, double b) {
  if (threadIdx.x >= 32) return;
  double x[CHAINS];
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) x[i] = threadIdx.x * 1e-3 + i;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < CHAINS; ++i) x[i] = __builtin_fma(x[i], a, b);
    }
  }
  long long t1 = clock64();
  double s = 0;
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// fma with THREE per-lane (VGPR) 64-bit operands: acc[i] += y[i] * z[j] -- the shape of nearly every FMA of the solve kernels
// (the kernels above have at most two VGPR operands; a and b are uniform)
template <int CHAINS>
__global__ void k_fma_vvv(double* out, long long* cyc, int iters, double a, double b) {
  double x[CHAINS], y[CHAINS], z[CHAINS];
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) { x[i] = threadIdx.x * 1e-3 + i; y[i] = 1.0 + threadIdx.x * 1e-9 * (i + 1); z[i] = a * (threadIdx.x + i + 1) * 1e-12 + b; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < CHAINS; ++i) x[i] = __builtin_fma(y[i], z[(i + r) % CHAINS], x[i]);
    }
  }
  long long t1 = clock64();
  double s = 0;
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) s += x[i] + y[i] + z[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// v_mul_f64 with two per-lane operands into a fresh destination (no accumulator read)
template <int CHAINS>
__global__ void k_mul_vv(double* out, long long* cyc, int iters, double a, double b) {
  double x[CHAINS], y[CHAINS], z[CHAINS];
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) { x[i] = 0.0; y[i] = 1.0 + threadIdx.x * 1e-9 * (i + 1); z[i] = a * (threadIdx.x + i + 1) * 1e-12 + b; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < CHAINS; ++i) { x[i] = y[i] * z[(i + r) % CHAINS]; asm volatile("" : "+v"(x[i])); }
    }
  }
  long long t1 = clock64();
  double s = 0;
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) s += x[i] + y[i] + z[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <class K, class... A>
double run(const char* name, K kern, int blocks, int threads, int per_iter, int iters, A... args) {
  double* out; long long* cyc;
  hipMalloc(&out, sizeof(double) * blocks * threads);
  hipMalloc(&cyc, sizeof(long long) * blocks);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, args...);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, args...);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks);
  hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
  double mean = 0; for (auto v : h) mean += v; mean /= blocks;
  double cpi = mean / ((double)iters * per_iter);
  printf("%-38s blocks=%5d thr=%4d  clock64-ticks/instr=%7.2f  wall=%8.1f us  instr/s/wave-lane... total Ginstr/s=%8.2f\n",
         name, blocks, threads, cpi, ms * 1e3, (double)blocks * threads / 64 * iters * per_iter / (ms * 1e-3) * 1e-9);
  hipFree(out); hipFree(cyc);
  return cpi;
}

int main() {
  const int iters = 2000;
  // one wave per SIMD (256 CUs * 4 SIMDs = 1024 waves as 1024 blocks of 64): does one wave reach the issue rate?
  run("fma dep chain x1 (1 wave/SIMD)", k_fma<1>, 1024, 64, 8 * 1, iters, 1.0000001, 1e-9);
  run("fma x2 indep", k_fma<2>, 1024, 64, 8 * 2, iters, 1.0000001, 1e-9);
  run("fma x4 indep", k_fma<4>, 1024, 64, 8 * 4, iters, 1.0000001, 1e-9);
  run("fma x8 indep", k_fma<8>, 1024, 64, 8 * 8, iters, 1.0000001, 1e-9);
  run("fma x8 indep (2 waves/SIMD)", k_fma<8>, 2048, 64, 8 * 8, iters, 1.0000001, 1e-9);
  run("fma x8 indep (4 waves/SIMD)", k_fma<8>, 4096, 64, 8 * 8, iters, 1.0000001, 1e-9);
  run("fma x1 dep (4 waves/SIMD)", k_fma<1>, 4096, 64, 8 * 1, iters, 1.0000001, 1e-9);
  run("fma x1 dep (8 waves/SIMD)", k_fma<1>, 8192, 64, 8 * 1, iters, 1.0000001, 1e-9);
  run("fma x8 indep, EXEC = low 32 lanes", k_fma_half<8>, 1024, 64, 8 * 8, iters, 1.0000001, 1e-9);
  run("fma x8, EXEC low32, 2 waves/SIMD", k_fma_half<8>, 2048, 64, 8 * 8, iters, 1.0000001, 1e-9);
  run("mul dep x1", k_mul<1>, 1024, 64, 8 * 1, iters, 1.0000001);
  run("mul x8 indep", k_mul<8>, 1024, 64, 8 * 8, iters, 1.0000001);
  run("fma vvs x8 (sgpr addend)", k_fma_sgpr<8>, 1024, 64, 8 * 8, iters, 1e-9, 0.0);
  run("fma vvv x8 (three VGPR operands)", k_fma_vvv<8>, 1024, 64, 8 * 8, iters, 1.0000001, 1e-9);
  run("fma vvv x16", k_fma_vvv<16>, 1024, 64, 8 * 16, iters, 1.0000001, 1e-9);
  run("fma vvv x8 (2 waves/SIMD)", k_fma_vvv<8>, 2048, 64, 8 * 8, iters, 1.0000001, 1e-9);
  run("mul vv x8 (two VGPR operands)", k_mul_vv<8>, 1024, 64, 8 * 8, iters, 1.0000001, 1e-9);
  run("rcp dep x1", k_rcp<1>, 1024, 64, 8 * 1, iters);
  run("rcp x8 indep", k_rcp<8>, 1024, 64, 8 * 8, iters);
  run("fma x8 indep, 1 wave total", k_fma<8>, 1, 64, 8 * 8, iters, 1.0000001, 1e-9);
  int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  int wclk = 0; hipDeviceGetAttribute(&wclk, hipDeviceAttributeWallClockRate, 0);
  printf("clockRate=%d kHz wallClockRate=%d kHz\n", clk, wclk);
  return 0;
}
